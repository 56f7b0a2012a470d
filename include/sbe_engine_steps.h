/* sbe_engine_steps.h -- the one-call MCMC step family of the sbe engine.  NO CALLER IN THE REFERENCE.
 *
 * The reference's sampler (sbayes/sampling/mcmc.py:237-241, mcmc_chain.py:128-172) asks for one quantity per Python call;
 * nothing in it hands a whole proposal to a likelihood object.  These entries fuse "apply the proposed delta, recount,
 * rebuild the tables, evaluate" into one engine call for a sampler WRITTEN AGAINST THE ENGINE (sbayes_amd/resident.py:
 * ResidentChain, ResidentChainBatch).  They are kept, tested (tests/test_gpu_steps.py) and exported by the same library,
 * but they are not part of the drop-in boundary (include/sbe_engine.h) and are FROZEN: not extended since round 4
 * (DESIGN.md section 10).  Conventions as in sbe_engine.h.
 */
#ifndef SBE_ENGINE_STEPS_H
#define SBE_ENGINE_STEPS_H

#include "sbe_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- one MCMC step in one call (resident flow) ---------------------------
 * Builds the candidate state in `cand_slot` from `cur_slot` plus the proposed delta and evaluates it:
 *   clusters        bool [K][N] of the candidate, or NULL if the clusters did not change
 *   changed_objects / source_rows  the objects whose source assignment changed and their bool
 *                   rows [n_changed][F][C]
 *   weights         float32 [F][C], or NULL if unchanged
 * On the device: slot copy, id / source-row update, delta update of the feature counts
 * (counts.py:55-95), probability tables of every component (conditionals.py:175-204), collapsed
 * per-group log-likelihoods (likelihood.py:65-101) and the fused mixture log-likelihood
 * (SURVEY.md 8(d)).  One PCIe round trip, one stream synchronisation.
 *   group_logliks_out  float64 [G_total] (Likelihood.__call__ = their sum)
 *   mixture_out        float64 scalar
 *   changed_groups_out bool [G_total] (may be NULL): groups whose counts changed
 * The caller accepts by swapping the roles of the two slots, rejects by doing nothing. */
int sbe_step(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters,
             const int32_t* changed_objects, int n_changed, const uint8_t* source_rows,
             const float* weights, double* group_logliks_out, double* mixture_out,
             uint8_t* changed_groups_out);

/* ---- batched multi-chain step: one sbe_step for each of n_chains independent chains, in ONE call ------------------
 * The reference steps its chains one after the other in one Python loop (MCMC.generate_samples,
 * sbayes/sampling/mcmc.py:237-241; MC3 workers, sbayes/mcmc_setup.py:528-534); chains are independent, so their
 * candidates are built by one launch (chain <-> blockIdx.y), evaluated by one launch of the fused mixture kernel over
 * the candidate slots and finished by one reduction launch: one synchronisation per batch, the per-chain host work
 * spread over a few worker threads.  Chain i: current slot cur_slots[i], candidate slot cand_slots[i] (all distinct).
 *   clusters       bool [n_chains][K][N] (chain i's block is read iff clusters_mask == NULL or clusters_mask[i] != 0),
 *                  or NULL: no chain changes its clusters
 *   rows_ptr       [n_chains + 1], rows_ptr[0] = 0: chain i's changed objects are changed_objects[rows_ptr[i] ..
 *                  rows_ptr[i+1]) and its rows source_rows[rows_ptr[i] ..) (bool [.][F][C]); at most 256 per chain
 *   weights        float32 [n_chains][F][C] (read iff weights_mask == NULL or weights_mask[i] != 0), or NULL
 *   out            group_logliks_out float64 [n_chains][G_total], mixture_out float64 [n_chains],
 *                  changed_groups_out bool [n_chains][G_total] (may be NULL)
 * Same numbers as n_chains calls of sbe_step (bit for bit for counts, tables and per-group values; the mixture scalar to
 * rounding: its block geometry depends on the launch's batch size).  Accept = swap a chain's two slots, reject = nothing.
 * Host side: the chains' payloads are packed into one pinned block by a pool of worker threads (8 including the caller;
 * environment SBE_STEP_THREADS) and sent with one copy; from 128 chains on the batch runs as two pipelined parts
 * (SBE_STEP_PARTS).  The workers poll for ~400 us after a call before they block, so consecutive sweeps find them awake:
 * keep the thread count below the number of free cores.  SBE_STEP_TIMING=1 prints the call's phase times to stderr. */
int sbe_step_batch(sbe_engine* e, int n_chains, const int32_t* cur_slots, const int32_t* cand_slots,
                   const uint8_t* clusters, const uint8_t* clusters_mask, const int32_t* rows_ptr,
                   const int32_t* changed_objects, const uint8_t* source_rows, const float* weights,
                   const uint8_t* weights_mask, double* group_logliks_out, double* mixture_out,
                   uint8_t* changed_groups_out);

/* The single-chain step with the proposal in delta form (see sbe_step_batch_delta below): moved objects + their new
 * cluster (-1: none), changed source rows.  Falls back to sbe_step internally when the two slots' records do not allow
 * patching, and when an object is listed more than once in either list (the last entry of a repeated object wins, as in
 * the matrix form).  Outputs as sbe_step. */
int sbe_step_delta(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* moved_objects, const int32_t* moved_cluster,
                   int n_moved, const int32_t* changed_objects, int n_changed, const uint8_t* source_rows /* [n_changed][F][C] bool */,
                   const float* weights /* [F][C] or NULL */, double* group_logliks_out /* [G_total] */, double* mixture_out,
                   uint8_t* changed_groups_out /* [G_total] or NULL */);

/* The same batched step with the proposals in DELTA form (round 3; what an MCMC operator actually produces): per chain
 * the objects that change cluster with their new cluster index (-1: leaves every cluster; CSR by moved_ptr) and the
 * objects whose source rows change (CSR by rows_ptr).  Within a chain every object may be listed ONCE in moved_objects
 * and ONCE in changed_objects: a repeated entry fails with SBE_ERR_ARG ("chain i: object n listed twice in ...") -- a
 * patch applied twice is not the last-wins result of the matrix form.  A chain's two slots differ only in what its last
 * step changed, so the candidate is built by patching -- host mirror, device id arrays, source rows -- in O(delta): no
 * [K][N] matrix is scanned, no slot state copied, no [N]-sized array packed or sent.  Chains whose slots were touched
 * by another call since their last step (or that step for the first time), and steps that change the SET of
 * has_components patterns or overflow the tuple table, run through sbe_step_batch internally; results are the same
 * (counts, tables, per-group values, flags bit for bit; the mixture scalar to rounding).  Outputs as sbe_step_batch. */
int sbe_step_batch_delta(sbe_engine* e, int n_chains, const int32_t* cur_slots, const int32_t* cand_slots,
                         const int32_t* moved_ptr /* [n_chains + 1] */, const int32_t* moved_objects,
                         const int32_t* moved_cluster, const int32_t* rows_ptr /* [n_chains + 1] */,
                         const int32_t* changed_objects, const uint8_t* source_rows /* [total][F][C] bool */,
                         const float* weights /* [n_chains][F][C] or NULL */, const uint8_t* weights_mask /* [n_chains] or NULL */,
                         double* group_logliks_out /* [n_chains][G_total] */, double* mixture_out /* [n_chains] */,
                         uint8_t* changed_groups_out /* [n_chains][G_total] or NULL */);

/* One MCMC step of the Gibbs source operator on the resident state (GibbsSampleSource._propose,
   sbayes/sampling/operators.py:495-552, + the likelihoods the MH ratio needs): candidate slot = current slot with the
   source of the listed objects redrawn from its posterior on the device (z: the caller's uniforms [n_sub][F], drawn
   where sample_categorical, preprocessing.py:248, draws them; NULL: the engine's Philox stream), count delta and
   tables follow on the device.  Out: log_q, log_q_back (fp64 sums of the logs of the float32 probabilities), the
   candidate's collapsed per-group log-likelihoods [G_total], its mixture log-likelihood, changed-group flags
   [G_total] (may be NULL).  One synchronisation; nothing of the sample state crosses PCIe. */
int sbe_gibbs_step(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                   double prior_temperature, int from_prior, const double* z, double* log_q_out,
                   double* log_q_back_out, double* group_logliks_out, double* mixture_out,
                   uint8_t* changed_groups_out);

#ifdef __cplusplus
}
#endif
#endif /* SBE_ENGINE_STEPS_H */
