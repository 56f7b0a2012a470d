/* sbe_engine.h -- C ABI of the MI355X-native sBayes likelihood engine ("sbe").
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json `north_star`
 * (SURVEY.md section 8): plain C, opaque handle, plain pointers and sizes, no torch / numpy
 * types.  The reference (NicoNeureiter/sBayes, pure Python) has no FFI of its own; each
 * entry point below names the reference function (file:line under /root/reference) whose
 * arithmetic it replaces.  The ctypes binding a reference maintainer would add is shown in
 * INTEGRATION.md and shipped as sbayes_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is available
 *     through sbe_last_error().  No exception crosses the ABI.
 *   - the caller owns every host buffer; the engine copies what it needs during the call
 *     and never retains host pointers.  The engine owns all device memory.
 *   - the one-hot feature block is uploaded once (sbe_create) and stays resident in HBM.
 *   - sample state lives in *slots* (0 .. n_slots-1) so that several MCMC states (chains,
 *     or current + candidate) are resident at once; every state call names its slot.
 *   - one engine per process per GPU; calls on one engine must be serialised by the caller.
 *     Calls are synchronous unless the name ends in _async.
 *   - how a synchronous call waits: latency-bound result calls (one eval, the one-call steps, the
 *     resident operator forms) write their results into host-mapped memory, and the last block of
 *     the call's final kernel then stores a sequence number into a host-mapped word; the host spins
 *     on that word for up to 300 us and falls back to the HIP stream wait (long launches, faults).
 *     Measured 3-4 us per call below hipStreamSynchronize.  Environment SBE_POLL_DONE=0 uses the
 *     stream wait everywhere (same results: tests/test_gpu_poll_done.py).  A caller thread inside
 *     such a call therefore busy-waits on one core for the duration of the launch.
 *   - how a large result reaches the caller: sbe_component_lh and sbe_likelihood_per_component hand back
 *     [N][F] / [N][F][C] float64 arrays (1.6 / 3.2 MB at the headline shape).  From 512 KB on, their kernels
 *     store the result straight into a host-mapped staging buffer (16-byte coalesced stores over PCIe), chunk
 *     after chunk in order (8 chunks of >= 128 KB; the grid is one chunk's worth of blocks), and raise one
 *     host-mapped flag per completed chunk; the engine's host worker threads (the pool of the
 *     batched steps: SBE_STEP_THREADS, default 8 including the caller) copy / scatter each chunk into the
 *     caller's array as soon as its flag shows the call's sequence number, while the later chunks are still
 *     in flight.  No copy-engine operation and no event is involved.  A flag that stays away for 2 ms + 10 GB/s
 *     sends the calling thread to the HIP stream wait.  SBE_D2H_THREADS=1 keeps the copy on the calling thread,
 *     SBE_STREAM_RESULTS=0 uses the copy engine, SBE_STREAM_ORDERED=0 the one-block-per-tile form whose chunks all
 *     complete at the end of the kernel (same bits: tests/test_gpu_streamed_results.py).  The worker
 *     threads poll for ~400 us after a call before they block.
 *   - bool arrays are one byte per element (NumPy bool layout), C order.
 */
#ifndef SBE_ENGINE_H
#define SBE_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbe_engine sbe_engine;

#define SBE_ABI_VERSION 6   /* 4 (round 4): + sbe_given_unchanged_gibbs, sbe_host_*; sbe_set_groups rejects overlap;
                               5: + SBE_OPT_FUSE_TABLES, sbe_host_subset_ids, sbe_host_diff_rows, sbe_set_counts_rows_probs,
                               sbe_gibbs_propose, sbe_given_unchanged_gibbs_counts, sbe_test_roundtrip,
                               sbe_collapsed_and_source_prior, sbe_counts_delta_apply,
                               sbe_given_unchanged_gibbs_apply, sbe_gibbs_propose_apply,
                               sbe_set_slot_delta, sbe_get_counts_all;
                               6 (round 6): + sbe_get_group_ids, sbe_get_weights, sbe_timer_mark, sbe_timer_elapsed;
                               sbe_set_groups takes overlapping groups (last group = the id, slot marked); the
                               header is split in three: THIS file holds the production surface -- every entry cites
                               the reference function it serves; sbe_engine_steps.h the one-call MCMC step family (no
                               caller in the reference, frozen); sbe_engine_diag.h self-tests and measurement hooks.
                               One library exports all three. */

/* error codes */
#define SBE_OK 0
#define SBE_ERR_ARG 1        /* invalid argument / shape                                   */
#define SBE_ERR_HIP 2        /* HIP runtime error (message holds hipGetErrorString)        */
#define SBE_ERR_STATE 3      /* call sequence error (e.g. probs not set)                   */
#define SBE_ERR_DATA 4       /* data violates a precondition the reference asserts         */
#define SBE_ERR_NODEVICE 5   /* no usable GPU                                              */

/* fused-kernel variants (sbe_set_option(SBE_OPT_MIXTURE_KERNEL, ...)) */
#define SBE_OPT_MIXTURE_KERNEL 1
#define SBE_MIXTURE_PACKED 0   /* reads the packed state-index block  (N*F bytes); uses the
                                  group-tuple form (one log per (tuple, state, feature)
                                  instead of per observation) whenever it applies: the
                                  scalar-unit kernel k_mixture_tuple64 at tile width 64 and
                                  S <= 127, k_mixture_combo otherwise                       */
#define SBE_MIXTURE_ONEHOT 1   /* reads the one-hot block as handed over (N*F*S bytes);
                                  group-tuple form whenever it applies                      */
#define SBE_MIXTURE_ONEHOT_GENERAL 4  /* one-hot stream, never the group-tuple form          */
#define SBE_MIXTURE_PACKED_GENERAL 2  /* packed, but never the group-tuple form: the rows kernel k_mixture_rows
                                         (1024-thread blocks over 32-feature tiles) when its LDS image fits and
                                         C <= 4, else k_mixture_v2 */
#define SBE_MIXTURE_PACKED_V2 6       /* packed, never the group-tuple form, never the rows kernel: k_mixture_v2
                                         (testing / A-B) */
#define SBE_MIXTURE_PACKED_TUPLE 3    /* packed, group-tuple form forced (error if not applicable) */
#define SBE_MIXTURE_PACKED_TUPLE_LDS 5  /* same, but never the scalar-unit 64-feature-tile variant
                                          (k_mixture_tuple64): the LDS-metadata kernel (testing / A-B) */
#define SBE_MIXTURE_PACKED_TUPLE_MFMA 7  /* packed, group-tuple form with the per-observation gather done as an exact
                                          0/1 byte contraction on the matrix pipe (k_mixture_tuple_mfma: counts per
                                          (slot, tuple, feature, state) by v_mfma_f32_32x32x64_f8f6f4 on FP4 operands --
                                          0 and 1 are exact in e2m1, the counts in the f32 accumulator -- then one log per
                                          table entry), forced (error if not applicable: more than 64 tuples, C > 4, LDS).
                                          SBE_MIXTURE_PACKED picks it by itself for launches of >= 320 slots -- >= 32 when the launch holds
                                          6.4 M observations: four slots per block -- with <= 8 group tuples per slot, and -- 9..64 tuples: 4 / 2 slots per block -- from 16 objects per
                                          padded tuple on (tools/diag/mfma_threshold.py: 24.5 / 24.6 / 24.8 / 25.8 us against
                                          22.2 / 32.1 / 34.9 / 58.3 us of k_mixture_tuple64 at 256 / 384 / 512 / 1024 headline
                                          states; profiles/r6/wide_forms.log)                                   */
#define SBE_OPT_LOG_MODE 2
#define SBE_LOG_PER_OBS 0      /* fp64 log per observation, fp64 sum                        */
#define SBE_LOG_PRODUCT 1      /* fp64 mantissa product + integer exponent, one log/thread  */
#define SBE_OPT_DEFERRED_CHECKS 3   /* 1: data checks raised by kernels (normalize's positive-sum
                                       assert, one-hot source) are reported by the next call that
                                       synchronizes (sbe_sync, any result fetch) instead of
                                       immediately, so state-setting calls never stall the stream */
#define SBE_OPT_STEP_FORM 4         /* sbe_step: 0 = few-launch form whenever the step fits its payload
                                       (default), 1 = always the call-by-call form (testing / A-B)   */

#define SBE_OPT_STEP_DERIVE 5       /* one-call steps: 0 = the has_components patterns / group tuples of a candidate are updated
                                       from the current slot's for the moved objects only (default), 1 = always derived
                                       from all objects (testing / A-B) */

#define SBE_OPT_FUSE_TABLES 6       /* resident operator calls whose kernel reads a tempered effect table built from the slot's
                                       counts (sbe_cluster_posterior_marginals, sbe_jump_lh_resident, sbe_given_unchanged_lh,
                                       sbe_given_unchanged_gibbs): 1 = the consuming kernel builds the tables itself -- builder waves of
                                       every block, into LDS, while the other waves run their load chains -- one launch per call
                                       (default; same operations in the same order, same bits);
                                       0 = table kernels in front (testing / A-B; environment SBE_FUSE_TABLES=0 sets the
                                       default of new engines) */

typedef struct sbe_info {
    int32_t abi_version;
    int32_t device;
    int32_t n_objects, n_features, n_states, n_components, n_slots;
    int32_t n_groups_total;
    int64_t n_na;              /* observations with an all-False state row (load_data.py:105) */
    int64_t hbm_bytes;         /* device memory held by the engine                          */
    int32_t compute_units;
    char device_name[64];
} sbe_info;

int sbe_abi_version(void);
int sbe_device_count(int* out_count);

/* Global (handle-less) error text for failures of sbe_create itself. */
const char* sbe_last_error(const sbe_engine* e);

/* ---- lifetime ------------------------------------------------------------------------
 * Likelihood.__init__ (sbayes/model/likelihood.py:36-45): stores features, derives
 * na_features = (sum over states == 0).  Here: uploads the one-hot block, validates that
 * every (object, feature) row has at most one set state, derives the packed state-index
 * block [N][F] (0xFF = NA) on the device.
 *   n_groups[c]: number of groups of mixture component c (c = 0: clusters, K;
 *                c >= 1: confounders, load_data.py:138-184).  0 is allowed (n_clusters == 0, the
 *                confounders-only baseline, sbayes/sampling/initializers.py:357): that component has
 *                empty tables and contributes nothing; at least one component must have a group.
 *   n_slots: 1 .. 16384 resident sample states.  */
int sbe_create(sbe_engine** out, int device, int n_objects, int n_features, int n_states,
               int n_components, const int32_t* n_groups, int n_slots,
               const uint8_t* features_onehot /* [N][F][S] bool */);
int sbe_destroy(sbe_engine* e);
int sbe_get_info(const sbe_engine* e, sbe_info* out);
int sbe_get_na(const sbe_engine* e, uint8_t* out_na /* [N][F] bool */);
int sbe_set_option(sbe_engine* e, int option, int value);
int sbe_sync(sbe_engine* e);

/* ---- a1: compute_component_likelihood (sbayes/model/likelihood.py:104-133) -------------
 * out[g_i, :] = sum_s features[g_i, :, s] * probs[i, :, s] for i in changed_groups; rows of
 * objects in no group <- 0; rows of members of unchanged groups are left untouched; later
 * groups overwrite earlier ones.  `out` is a strided host view (the reference passes
 * component_likelihood[..., i], element stride C*8 bytes along F).  Stateless: touches no slot.
 *   probs_f64: 0 = float32 tables (what normalize() returns, util.py:1007), 1 = float64.
 *   na_value : value written for NA observations in the rows this call writes: 0.0 is the literal
 *              a1 result (an all-False one-hot row sums to 0); likelihood_per_component passes 1.0,
 *              which is what its final `component_likelihood[na_values] = 1.` leaves there
 *              (conditionals.py:216).  */
int sbe_component_lh(sbe_engine* e, const void* probs /* [G][F][S] */, int probs_f64, int n_groups,
                     const uint8_t* groups /* [G][N] bool */, const int64_t* changed_groups,
                     int n_changed, double* out, int64_t out_stride_n_bytes, int64_t out_stride_f_bytes,
                     double na_value);

/* ---- a2: compute_component_likelihood_exact (likelihood.py:136-150) via
 * likelihood_per_component_exact (sbayes/sampling/conditionals.py:300-367): leave-one-out
 * tables built from the slot's counts + concentration and the slot's source; all components;
 * NA observations <- 1.  out: dense [N][F][C] float64. */
int sbe_likelihood_per_component_exact(sbe_engine* e, int slot, double* out);

/* ---- slot state: groups (state.py Clusters / load_data.py Confounder.group_assignment) --
 * Resident state keeps ONE group id per object and component.  For a matrix with an object in
 * several rows of one component the reference has two readings: a1 lets the last WRITTEN group
 * win (likelihood.py:126-130; an uncached evaluation writes the groups in index order), while
 * compute_effect_counts counts the object once PER group (counts.py:28-30).  ABI 6: sbe_set_groups
 * TAKES such a matrix -- the id is the LAST group containing the object, which is what every
 * likelihood evaluation on resident state needs (sbe_mixture_loglik*, sbe_likelihood_per_component,
 * sbe_observation_lh, the has_components patterns; the collapsed likelihood reads the counts the
 * caller set) -- and marks the slot; calls that would DERIVE counts from the ids (sbe_recount,
 * sbe_update_counts, sbe_accumulate_counts, the one-call steps, sbe_gibbs_propose*,
 * sbe_given_unchanged_gibbs*, sbe_observation_lh_exact) refuse a marked slot with SBE_ERR_DATA
 * ("object n is in groups g1 and g2 of component c ..."): there the caller's counts (sbe_set_counts) or
 * the stateless sbe_effect_counts are the reference's.  sbe_set_group_ids clears the mark.  The
 * cluster matrices handed to sbe_step / sbe_step_batch stay strict (a batch reports the chain).
 * Also refreshes has_components (state.py:353-376): the ids, the has_components pattern of every object, the group-tuple
 * tables and -- when the slot has weights -- the per-pattern normalised weights (likelihood.py:171-190) go up and are
 * computed in ONE asynchronous launch; after a cluster move the host-side tables follow the moved objects.  More
 * distinct patterns than the engine holds (min(2^C, 64)) is reported by the next call that reads them. */
int sbe_set_groups(sbe_engine* e, int slot, int component, const uint8_t* groups /* [G_c][N] bool */);
int sbe_set_group_ids(sbe_engine* e, int slot, int component, const int32_t* ids /* [N], -1 = none */);
/* The slot's resident ids of one component read back from the device (a pickled / resumed sample is re-bound from the host
 * arrays; this is the inverse of sbe_set_group_ids for checkers: bench.py's parity gate rebuilds a slot's state
 * from what the device holds).  ids_out [N]: group index inside the component, -1 = in no group. */
int sbe_get_group_ids(sbe_engine* e, int slot, int component, int32_t* ids_out /* [N] */);

/* sbe_set_slot_delta: several state-setting calls of one bind (conditionals._bind_slot: after a rejected step "the old group
 * ids and the old count rows", after a proposal "the new group ids and the new source rows") as ONE launch.  Exactly
 * sbe_set_groups(slot, groups_component, groups) when groups != NULL, then sbe_set_counts_rows (update_probs = 0) or
 * sbe_set_counts_rows_probs (update_probs != 0) when n_count_rows > 0, then sbe_set_source_rows when n_src_rows > 0 -- the
 * same checks and results; the kernels are issued together when the inputs went through the mapped staging ring (the three
 * touch disjoint resident arrays), one after the other otherwise or with SBE_OPT_DEFERRED_CHECKS off. */
int sbe_set_slot_delta(sbe_engine* e, int slot, int groups_component, const uint8_t* groups /* [G_c][N] bool or NULL */,
                       const int32_t* count_idx, int n_count_rows, const float* count_rows /* [n][F][S] */, int update_probs,
                       const int32_t* src_objects, int n_src_rows, const uint8_t* src_rows /* [n][F][C] bool */);

/* ---- slot state: source (state.py:510, bool [N][F][C] one-hot over components) --------- */
int sbe_set_source(sbe_engine* e, int slot, const uint8_t* source /* [N][F][C] bool */);
int sbe_set_source_rows(sbe_engine* e, int slot, const int32_t* objects, int n_rows,
                        const uint8_t* rows /* [n_rows][F][C] bool */);
/* read the listed objects' rows back (after sbe_sample_source wrote them on the device) */
int sbe_get_source_rows(sbe_engine* e, int slot, const int32_t* objects, int n_rows,
                        uint8_t* rows_out /* [n_rows][F][C] bool */);

/* ---- a9: feature counts (sbayes/sampling/counts.py:10-95) ------------------------------
 * sbe_recount: recalculate_feature_counts (counts.py:35-52) for one component (or all: -1)
 *              from the slot's groups + source.
 * sbe_accumulate_counts: one half of update_feature_counts (counts.py:55-95): adds
 *              sign * compute_effect_counts(groups, source, object_subset) using the slot's
 *              CURRENT groups/source; call with sign=-1 before editing the slot and sign=+1
 *              after.  changed_groups_out[G_total] (may be NULL) receives 1 for every group
 *              whose counts changed in this call (FeatureCounts.add_changes, state.py:340-350).
 * Counts are integers; they are exchanged as float32 (FLOAT_TYPE, counts.py:20).          */
int sbe_recount(sbe_engine* e, int slot, int component);
/* update_feature_counts(sample_old, sample_new, features, object_subset) (counts.py:55-95):
 * counts[slot_new] += compute_effect_counts(new state, subset) - compute_effect_counts(old
 * state, subset), for every component, in one launch.  Precondition: counts[slot_new] hold
 * the old state's counts (sbe_copy_slot, then edit groups / source rows of slot_new).
 * changed_groups_out[G_total] = np.any(diff != 0, axis=(1,2)) per group (state.py:349-350). */
int sbe_update_counts(sbe_engine* e, int slot_new, int slot_old, const int32_t* objects,
                      int n_objects_subset, uint8_t* changed_groups_out);
int sbe_accumulate_counts(sbe_engine* e, int slot, const int32_t* objects, int n_objects_subset,
                          int sign, uint8_t* changed_groups_out);
int sbe_set_counts(sbe_engine* e, int slot, int component, const float* counts /* [G_c][F][S] */);
int sbe_get_counts(sbe_engine* e, int slot, int component, float* out /* [G_c][F][S] */);
/* every component's table in one call, component after component (recalculate_feature_counts, counts.py:35-52, reads them all) */
int sbe_get_counts_all(sbe_engine* e, int slot, float* out /* [G_total][F][S] */);

/* ---- Dirichlet concentration tables (sbayes/model/prior.py:325-354, 453-455) -----------
 * per_group = 0: [F][S] broadcast over groups (cluster effect prior); 1: [G_c][F][S]. Shared
 * by all slots. */
int sbe_set_concentration(sbe_engine* e, int component, const double* conc, int per_group);

/* ---- a4 + a3/a10: probability tables ----------------------------------------------------
 * sbe_update_probs: probs = normalize(counts/T + prior') -> float32 (util.py:990-1007 called
 *   from conditionals.py:175-179, 200-204; tempered form conditionals.py:105-122 with
 *   prior' = unif + (prior - unif)/T_prior when prior_temperature > 0).  temperature <= 0
 *   and prior_temperature <= 0 mean "not given".  Fails with SBE_ERR_DATA when a row sum
 *   is not positive (the reference's assert, util.py:1006).
 * sbe_set_probs: explicit tables instead. */
int sbe_update_probs(sbe_engine* e, int slot, int component, double temperature,
                     double prior_temperature, const double* unif_counts /* [F][S] or NULL */);
/* The same for every component whose bit is set in `component_mask` (bit c = component c): components with adjacent
 * group ranges share a launch.  likelihood_per_component (conditionals.py:175-179, 200-204) refreshes the tables of
 * all components whose counts changed since the last evaluation; the drop-in layer asks for them together. */
int sbe_update_probs_mask(sbe_engine* e, int slot, unsigned component_mask, double temperature,
                          double prior_temperature, const double* unif_counts /* [F][S] or NULL */);
int sbe_set_probs(sbe_engine* e, int slot, int component, const float* probs /* [G_c][F][S] */);
int sbe_get_probs(sbe_engine* e, int slot, int component, float* out /* [G_c][F][S] */);

/* ---- a5: normalize_weights / update_weights (likelihood.py:153-190) ---------------------
 * Stores the mixture weights [F][C] (float32, state.py:546) and builds the per-pattern
 * normalised tables on the device from the slot's has_components. */
int sbe_set_weights(sbe_engine* e, int slot, const float* weights /* [F][C] */);
int sbe_get_weights_normalized(sbe_engine* e, int slot, float* out /* [N][F][C] */);
/* the slot's resident mixture weights as set (Sample.weights.value, sbayes/sampling/state.py:546): float32 [F][C] */
int sbe_get_weights(sbe_engine* e, int slot, float* out /* [F][C] */);

/* ---- a3: likelihood_per_component (sbayes/sampling/conditionals.py:152-223) -------------
 * Dense float64 [N][F][C] from the slot's groups + probs; NA observations <- 1, objects in
 * no group of a component <- 0. */
int sbe_likelihood_per_component(sbe_engine* e, int slot, double* out /* [N][F][C] */);

/* ---- a6: per-observation mixture likelihood sum_c w*lh (loggers.py:355-357,
 * operators.py:568-574, 1060-1061): float64 [N][F]; NA observations get sum_c w. */
int sbe_observation_lh(sbe_engine* e, int slot, double* out /* [N][F] */);

/* ---- north-star kernel: one mixture log-likelihood eval ----------------
 * LL = sum_{n,f not NA} log sum_c w[n,f,c] * p_c[g_c(n), f, x(n,f)]: fused gather + weighted
 * sum + log + wavefront/block reduction; scalar double out.
 * _batch evaluates slots first_slot .. first_slot+n-1 in one launch sequence.
 * _async variants only enqueue; results land in the engine's result buffer and are fetched
 * with sbe_fetch_results after sbe_sync. */
int sbe_mixture_loglik(sbe_engine* e, int slot, double* out);
int sbe_mixture_loglik_batch(sbe_engine* e, int first_slot, int n, double* out /* [n] */);
int sbe_mixture_loglik_batch_async(sbe_engine* e, int first_slot, int n);
int sbe_fetch_results(sbe_engine* e, int first_slot, int n, double* out /* [n] */);

/* ---- a7 + a8: collapsed Dirichlet-categorical likelihood (likelihood.py:47-101,
 * util.py:1373-1394): per group float32-sum over features of the per-feature float32 log-pdf,
 * from the slot's counts and the component's concentration.  per_group_out: float64 [G_c];
 * per_feature_out (may be NULL): float32 [G_c][F]. */
int sbe_collapsed_loglik(sbe_engine* e, int slot, int component, double* per_group_out,
                         float* per_feature_out);
/* The same for every component of the slot in one call and one synchronisation (Likelihood.__call__ with
 * caching=False sums all of them, likelihood.py:47-63): per_group_out float64 [G_total], components in order. */
int sbe_collapsed_loglik_all(sbe_engine* e, int slot, double* per_group_out);

/* ---- stateless forms of the reference's free functions (no slot involved) ----------------
 * sbe_normalize_tables : normalize(counts [/T] + prior['], axis=-1) -> float32
 *                        (sbayes/util.py:990-1007; conditionals.py:105-122, 175-179)
 * sbe_dirichlet_logpdf : dirichlet_categorical_logpdf per (group, feature) -> float32 [G][F]
 *                        and its float32 NumPy-order sum per group -> float64 [G]
 *                        (sbayes/util.py:1373-1394, likelihood.py:74-77)
 * sbe_effect_counts    : compute_effect_counts(features, group_assignment,
 *                        source_is_component, object_subset) (counts.py:10-32);
 *                        n_subset = -1 means all objects
 * sbe_normalize_weights: normalize_weights(weights, has_components) (likelihood.py:171-190); has_components has
 *                        `n_rows` rows -- all objects (update_weights, likelihood.py:153-168) or any subset of them
 *                        (compute_feature_weights_with_and_without passes has_components[available],
 *                        operators.py:1075-1095); only F is taken from the engine */
int sbe_normalize_tables(sbe_engine* e, const float* counts /* [G][F][S] */, int n_groups,
                         const double* conc, int conc_per_group, double temperature,
                         double prior_temperature, const double* unif_counts, float* out);
int sbe_dirichlet_logpdf(sbe_engine* e, const float* counts /* [G][F][S] */, int n_groups,
                         const double* conc, int conc_per_group, float* per_feature_out,
                         double* per_group_out);
int sbe_effect_counts(sbe_engine* e, const uint8_t* groups /* [G][N] bool */, int n_groups,
                      const uint8_t* source_is_component /* [N][F] bool */, const int32_t* objects,
                      int n_subset, float* out /* [G][F][S] */);
int sbe_normalize_weights(sbe_engine* e, const float* weights /* [F][C] */, int n_comp,
                          const uint8_t* has_components /* [n_rows][C] bool */, int n_rows,
                          float* out /* [n_rows][F][C] */);

/* ---- cluster-membership marginals --------------------------------------
 * AlterCluster.compute_cluster_posterior (sbayes/sampling/operators.py:1035-1073) and
 * AlterClusterWide.compute_raw_cluster_probs (:1420-1472), the data-parallel part: for every
 * listed (available) object and z in {0,1}
 *     out[z][i] = log prod_f sum_c lh_c(n_i, f) * w_z(n_i)[f][c]
 * where lh_0 comes from `table` (candidate cluster effect, float32 [F][S]), lh_{c>=1} from the
 * slot's probability tables, NA observations count 1, and w_0 / w_1 are the weights of
 * compute_feature_weights_with_and_without (:1075-1095) built from the slot's weights, its
 * has_components and `prior_temperature`.  Log space: no underflow at large F.
 * The caller applies ** (1/temperature), the geo-prior factor and the normalisation. */
int sbe_cluster_marginals(sbe_engine* e, int slot, const float* table /* [F][S] */, const int32_t* objects,
                          int n_objects_av, double prior_temperature, double* out /* [2][n_objects_av] */);

/* ---- next-heaviest operator expressions (missing #2 / #3) -----------------------------------------
 * sbe_jump_lh: ClusterJump.get_jump_lh (sbayes/sampling/operators.py:1679-1722) with
 *     ClusterEffectProposals.expected_confounder_features (:1342-1379), the data-parallel part: for every listed
 *     member n of the source cluster
 *         out[0][i] = sum_{f not NA} log( p_conf(n,f) + wh(n)[f][0] * p_source[f][x(n,f)] )     "stay"
 *         out[1][i] = sum_{f not NA} log( p_conf(n,f) + wh(n)[f][0] * p_target[f][x(n,f)] )     "jump"
 *     p_conf(n,f) = sum_{c>=1} wh(n)[f][c] * pconf[g_c(n)][f][x] over the confounder groups the object is in,
 *     wh = normalize(update_weights(sample) ** (1/prior_temperature)) from the slot's weights and patterns; all of it
 *     in the reference's float32 arithmetic, the product over features as a sum of fp64 logs (the reference's
 *     float32 np.prod underflows beyond F ~ 75).  pconf: float32 [G_total - n_clusters][F][S], the
 *     tempered tables of every confounder group (posterior_counts + normalize, operators.py:1254-1259, 1364-1371);
 *     p_source / p_target: float32 [F][S] (conditional_effect_mean, conditionals.py:105-122).  The caller applies
 *     ** (1/temperature), + EPS and the ratio (operators.py:1712-1722).
 * sbe_source_lh_by_feature: GibbsSampleWeights.source_lh_by_feature (operators.py:677-685): per feature
 *     float32( sum_n log sum_c source[n,f,c] * w[n,f,c] ), NA observations count 1, from the slot's source, patterns
 *     and weights -- the [N, F, C] normalised-weight array never crosses PCIe.  float32 logs like the reference's; the
 *     sum over the objects is taken in float64 (fixed order) and rounded to float32 once, where the reference adds the N
 *     float32 logs in float32 one after the other (np.sum(axis=0)): the two agree within the reference's own
 *     accumulated rounding, at most N * 2^-25 relative (observed 2e-4 at N = 30 000, 1e-6 at N = 1000). */
int sbe_jump_lh(sbe_engine* e, int slot, const float* pconf /* [G_total - K][F][S] */, const float* p_source /* [F][S] */,
                const float* p_target /* [F][S] */, const int32_t* objects, int n_members, double prior_temperature,
                double* out /* [2][n_members] */);
int sbe_source_lh_by_feature(sbe_engine* e, int slot, float* out /* [F] */);

/* ---- round 3: delta / resident forms for the UNCHANGED sampler on the drop-in layer -----------
 * What the reference's own operators ask per MCMC step crosses PCIe as object lists and a few changed rows, never as
 * [N][F] masks or whole [G][F][S] tables ("What crosses PCIe per step").
 *
 * sbe_set_uniform_counts: DirichletPrior.uniform_concentration_array (sbayes/model/prior.py:184-186), float64 [F][S]:
 *     1 on applicable states, 0 elsewhere.  Resident operand of the tempered tables below (conditional_effect_mean,
 *     sbayes/sampling/conditionals.py:105-122: normalize(counts/T + unif + (prior - unif)/T_prior)).
 * sbe_counts_delta: update_feature_counts(sample_old, sample_new, features, object_subset)
 *     (sbayes/sampling/counts.py:55-95), stateless.  For the n_subset listed objects the caller passes both states:
 *     gid_old / gid_new int32 [C][n_subset] = global group index of the object in each component (-1: in no group),
 *     src_old / src_new uint8 [n_subset][F] = source component of each observation (0xFF: none), and `touched` =
 *     the global indices of the groups any listed object is in, in either state.  out_diff float32
 *     [n_touched][F][S] = rows `touched` of the reference's `new_counts - old_counts`, all components at once
 *     (every other row of that difference is zero).  The caller adds them with FeatureCounts.add_changes
 *     (sbayes/sampling/state.py:340-350).
 * sbe_set_counts_rows: rows `group_idx` (global) of the slot's resident counts <- rows float32 [n_rows][F][S]: the bind
 *     cache of the host layer sends only the groups whose counts differ from what the slot holds.  The rows PATCH a
 *     table: the counts of every component a listed group belongs to must be resident already (sbe_set_counts,
 *     sbe_recount or a step), else SBE_ERR_STATE -- a component is never marked set through rows alone.
 * sbe_given_unchanged_lh: component_likelihood_given_unchanged(model, sample, object_subset, i_cluster, T, T_prior)
 *     (sbayes/sampling/operators.py:863-928) for static priors, from RESIDENT data of the slot the candidate is bound
 *     to (new clusters, source not yet resampled, counts still the old state's -- exactly the reference's inputs at
 *     that point): kept cluster counts (:876-883), unchangeable confounder counts (:896-901), their tempered tables,
 *     the gather, NA -> 1, ** (1/T).  out float32 [n_sub][F][C].  Only the object list goes up.
 * sbe_cluster_posterior_marginals: sbe_cluster_marginals with the candidate table built on the device from the
 *     slot's resident counts of cluster `i_cluster`: conditional_effect_mean(prior, counts[[i_cluster]], unif, T_prior,
 *     T) (AlterCluster.compute_cluster_posterior, operators.py:1046-1052).  out float64 [2][n].
 * sbe_jump_lh_resident: sbe_jump_lh with the tempered tables of the source / target cluster and of every confounder
 *     group built on the device from the slot's resident counts (ClusterJump.get_jump_lh, operators.py:1679-1722;
 *     expected_confounder_features :1342-1379).  out float64 [2][n_members]. */
/* Host helpers of the marshalling for sbe_counts_delta (no device, no engine; return 0 = ok, 1 = a listed object is in
 * several groups: no single id, -1 = bad argument):
 * sbe_host_group_ids: ids_out[i] = offset + g for the one row g of `groups` ([n_groups][n_objects] bool, C order) that
 *     has objects[i] set, -1 if none (group_assignment[:, object_subset], sbayes/sampling/counts.py:21-24).
 * sbe_host_source_ids: ids_out[i][f] = the component c with source[objects[i]][f][c] set, 0xFF if none
 *     (source[object_subset, :, c], counts.py:25-27); `source` is [n_objects][F][C] bool, C order.
 * sbe_host_touched_groups: the sorted distinct global group indices >= 0 among gid_old / gid_new (`count` entries each):
 *     sbe_counts_delta's `touched` argument (np.union1d of the two id arrays without the -1s); touched_out has room for
 *     n_groups_total entries.
 */
int sbe_host_group_ids(const uint8_t* groups, int n_groups, int64_t n_objects, const int32_t* objects, int n, int offset,
                       int32_t* ids_out /* [n] */);
int sbe_host_touched_groups(const int32_t* gid_old, const int32_t* gid_new, int64_t count, int n_groups_total,
                            int32_t* touched_out /* [n_groups_total] */, int32_t* n_touched_out);
int sbe_host_source_ids(const uint8_t* source, int64_t n_objects, int n_features, int n_components, const int32_t* objects,
                        int n, uint8_t* ids_out /* [n][F] */);
/* sbe_host_subset_ids: everything sbe_counts_delta needs about the listed objects in ONE pass over the reference's own
 *     arrays (drop-in update_feature_counts, counts.py:55-95): group ids of every component in both samples
 *     (groups_new[c] / groups_old[c]: [n_groups[c]][n_objects] bool; the same pointer in both = the ids are copied) and source
 *     ids of both (source_*: [n_objects][F][C] bool; the same pointer = copied).  Returns 1 also when an object is listed
 *     twice (the reference's fancy index counts it twice: the caller takes the two-count form).
 * sbe_host_diff_rows: the bind cache's content compare (sbayes_amd/binding.py): rows of `rows` ([n_rows][row_bytes]) that
 *     differ from `mirror` bytewise are copied into `mirror` and their indices written to changed_out (ascending);
 *     returns how many, -1 on a bad argument. */
int sbe_host_subset_ids(const int32_t* objects, int n, int64_t n_objects, int n_features, int n_components,
                        const int32_t* n_groups /* [C] */, const uint8_t* const* groups_new /* [C] */,
                        const uint8_t* const* groups_old /* [C] */, const uint8_t* source_new, const uint8_t* source_old,
                        int32_t* gid_new_out /* [C][n] */, int32_t* gid_old_out /* [C][n] */, uint8_t* sid_new_out /* [n][F] */,
                        uint8_t* sid_old_out /* [n][F] */);
int64_t sbe_host_diff_rows(const void* rows, void* mirror, int64_t n_rows, int64_t row_bytes, int32_t* changed_out /* [n_rows] */);

int sbe_set_uniform_counts(sbe_engine* e, const double* unif_counts /* [F][S] */);
int sbe_counts_delta(sbe_engine* e, const int32_t* objects, int n_subset, const int32_t* gid_old /* [C][n_subset] */,
                     const int32_t* gid_new, const uint8_t* src_old /* [n_subset][F] */, const uint8_t* src_new,
                     const int32_t* touched, int n_touched, float* out_diff /* [n_touched][F][S] */);
/* sbe_counts_delta_apply: the same difference, and `slot` FOLLOWS it.  update_feature_counts (counts.py:55-95) adds the
 * difference to the new sample's counts on the host; a slot that holds the counts the difference is added to -- the
 * usual state inside an MCMC step -- takes it on the device (counts[touched rows] += difference; with update_probs != 0
 * the probability rows of those groups are rebuilt, exactly sbe_set_counts_rows_probs' result), so the rows need not be
 * sent back: inside the same launch for subsets of up to 256 objects, by one more kernel otherwise.  With
 * update_source != 0 the slot's source rows of the listed objects become src_new (the rows the operator has just drawn:
 * they are in the call anyway).  The touched components' counts (and, for update_probs, tables; for update_source, a
 * source) must be resident (SBE_ERR_STATE); the caller vouches that the slot's rows of the touched groups are the
 * counts the difference belongs to. */
int sbe_counts_delta_apply(sbe_engine* e, int slot, int update_probs, int update_source, const int32_t* objects, int n_subset,
                           const int32_t* gid_old /* [C][n_subset] */, const int32_t* gid_new,
                           const uint8_t* src_old /* [n_subset][F] */, const uint8_t* src_new, const int32_t* touched,
                           int n_touched, float* out_diff /* [n_touched][F][S] */);
int sbe_set_counts_rows(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows /* [n_rows][F][S] */);
/* sbe_set_counts_rows that also rebuilds the probability rows of the patched groups (normalize(counts + concentration),
 * sbe_update_probs' untempered arithmetic) in the same launch: after it the slot's tables of those components are current
 * again, provided they were before.  Requires the components' tables (sbe_update_probs) and concentrations to be set;
 * normalize's positive-sum assert is reported like sbe_update_probs reports it. */
int sbe_set_counts_rows_probs(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows /* [n_rows][F][S] */);
int sbe_given_unchanged_lh(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                           double prior_temperature, float* out /* [n_sub][F][C] */);
/* ClusterOperator.gibbs_sample_source (sbayes/sampling/operators.py:796-851: the source resampling inside every
 * AlterCluster / AlterClusterWide / ClusterJump proposal) from RESIDENT data of the slot the NEW sample is bound to
 * (clusters already changed, source not yet resampled, counts still the old state's -- the reference's own inputs):
 *   lh     = component_likelihood_given_unchanged(model, sample_new, object_subset, i_cluster, T, T_prior)   (:808-811)
 *   p      = normalize(update_weights(sample_new)[subset] ** (1/T_prior) * lh)   (sample_from_prior: the weights alone)
 *   x      = sample_categorical(p) with the caller's uniforms z [n][F] (np.random.random((n, F, 1)) where the reference draws)
 *   p_back = normalize(update_weights(sample_old)[subset] ** (1/T_prior) * lh)                                (:838-844)
 * hc_new / hc_old: has_components rows [n][C] of the two samples for the listed objects (bool); src_old: the old sample's
 * source component per observation [n][F] (0xFF: none).  Out: src_new_out [n][F] drawn component (0xFF: NA observation),
 * sel_new_out = p[x] and sel_back_out = p_back[old source] float32 [n][F] (1 where nothing is selected): log_q and
 * log_q_back are the float32 sums of their logs over the valid observations, taken by the caller as the reference takes
 * them (:832, :847).  Static priors only (like sbe_given_unchanged_lh). */
int sbe_given_unchanged_gibbs(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                              double prior_temperature, int from_prior, const uint8_t* hc_new /* [n_sub][C] */,
                              const uint8_t* hc_old, const uint8_t* src_old /* [n_sub][F] */, const double* z /* [n_sub][F] */,
                              uint8_t* src_new_out /* [n_sub][F] */, float* sel_new_out, float* sel_back_out);
/* The same with the count delta of the proposal (update_feature_counts(sample_old, sample_new, features, subset),
 * sbayes/sampling/counts.py:55-95, which the reference calls right after): gid_old / gid_new [C][n] = the listed objects'
 * GLOBAL group index per component in the old / new sample (-1: none; sbe_host_subset_ids).  touched_out [<= G_total]
 * (ascending) + *n_touched_out: the groups any listed object is in, in either sample; diff_rows_out [n_touched][F][S] =
 * new counts - old counts of those groups (every other row of the difference is zero).  The kernel that draws the new
 * source also bins the delta -- one launch, one synchronisation for what was this call followed by sbe_counts_delta. */
int sbe_given_unchanged_gibbs_counts(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                                     double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                     const uint8_t* src_old, const double* z, const int32_t* gid_old, const int32_t* gid_new,
                                     uint8_t* src_new_out, float* sel_new_out, float* sel_back_out, int32_t* touched_out,
                                     int32_t* n_touched_out, float* diff_rows_out);
/* ... and the slot FOLLOWS the proposal (cf. sbe_counts_delta_apply): when the call touches any group, the slot's counts
 * take the delta, the subset's source rows become the drawn ids and, with update_probs != 0, the probability rows of the
 * touched groups are rebuilt -- behind the completion flag of the same launch, so that the bind that follows the
 * reference's own bookkeeping (source.edit(), update_feature_counts) has nothing to send.  The caller vouches that the
 * slot's counts are the ones the delta belongs to (they are: the call reads them). */
int sbe_given_unchanged_gibbs_apply(sbe_engine* e, int slot, int update_probs, int i_cluster, const int32_t* objects, int n_sub,
                                    double temperature, double prior_temperature, int from_prior, const uint8_t* hc_new,
                                    const uint8_t* hc_old, const uint8_t* src_old, const double* z, const int32_t* gid_old,
                                    const int32_t* gid_new, uint8_t* src_new_out, float* sel_new_out, float* sel_back_out,
                                    int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out);
int sbe_cluster_posterior_marginals(sbe_engine* e, int slot, int i_cluster, double temperature, double prior_temperature,
                                    const int32_t* objects, int n_objects_av, double* out /* [2][n_objects_av] */);
int sbe_jump_lh_resident(sbe_engine* e, int slot, int i_source, int i_target, double temperature, double prior_temperature,
                         const int32_t* objects, int n_members, double* out /* [2][n_members] */);

/* ---- data-parallel cores of Gibbs source resampling ----------------------
 * sbe_source_posterior: GibbsSampleSource.calculate_source_posterior (operators.py:554-574):
 *     out[i][f][:] = normalize(lh[n_i][f][:] ** (1/T) * w[n_i][f][:] ** (1/T_prior)) (float32)
 *     for the listed objects, from the slot's tables, groups and weights.
 * sbe_subset_lh: the gather of component_likelihood_given_unchanged (operators.py:863-928):
 *     float32 likelihoods of the listed objects' observations under caller-built tables
 *     (`tables`: n_tables_total x [F][S]; component c uses rows table_offsets[c] + group_idx[c][i],
 *     -1 = object in no group -> 0), NA -> 1, then ** (1/T).  Stateless.
 * sbe_sample_source: the draw of GibbsSampleSource._propose (operators.py:518-528) on the device:
 *     p = the posterior above (or, from_prior != 0, normalize(w ** (1/T_prior)) in float32,
 *     operators.py:520-522); sample_categorical (sbayes/preprocessing.py:224-256) with the CALLER's
 *     uniforms z[i][f] (the reference draws np.random.random([n_sub, F, 1]): pass those numbers and
 *     the assignments are the reference's, draw for draw): cdf = cumsum(p) / cdf[-1] in float32,
 *     first component with z < cdf; NA observations get no source (operators.py:527).  The rows of
 *     the listed objects are written into `dst_slot`'s source (rows of other objects keep what
 *     dst_slot held: copy the slot first, sbe_copy_slot); counts are NOT touched -- follow with
 *     sbe_update_counts(dst_slot, slot, objects).  *log_q_out = sum log p[new source] (operators.py:539;
 *     fp64 logs and sum, the reference's are float32); p_selected_out (nullable) receives the selected
 *     probabilities [n_sub][F] (1 for NA) so a caller can redo the sum in the reference's precision.
 * sbe_source_logprob: log_q_back (operators.py:544-550): sum log p_slot[ source of src_slot ] over the
 *     listed objects, p from `slot`'s current tables.
 * z == NULL: the uniforms come from the engine's own counter-based stream instead (Philox4x32-10:
 *     uniform i of draw d = 53 bits of philox(counter = (i, d), key = seed); sbe_set_rng sets
 *     (seed, d), every z == NULL call uses draw d and then increments it) -- nothing but the object
 *     ids crosses PCIe.  Statistically equivalent to, not the same numbers as, np.random. */
int sbe_source_posterior(sbe_engine* e, int slot, const int32_t* objects, int n_sub, double temperature,
                         double prior_temperature, float* out /* [n_sub][F][C] */);
int sbe_sample_source(sbe_engine* e, int slot, int dst_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z /* [n_sub][F] */,
                      double* log_q_out, float* p_selected_out /* [n_sub][F] or NULL */);
int sbe_source_logprob(sbe_engine* e, int slot, int src_slot, const int32_t* objects, int n_sub, double temperature,
                       double prior_temperature, int from_prior, double* log_q_out,
                       float* p_selected_out /* [n_sub][F] or NULL */);
int sbe_set_rng(sbe_engine* e, uint64_t seed, uint64_t draw);
int sbe_get_rng(sbe_engine* e, uint64_t* seed, uint64_t* draw);
int sbe_subset_lh(sbe_engine* e, const int32_t* objects, int n_sub, int n_comp, const float* tables,
                  const int32_t* table_offsets /* [n_comp] */, int n_tables_total,
                  const int32_t* group_idx /* [n_comp][n_sub] */, double temperature,
                  float* out /* [n_sub][F][n_comp] */);

/* ---- SURVEY.md 8(f) rank 4 ------------------------------------------------------------------------
 * sbe_source_prior: SourcePrior.__call__ (sbayes/model/prior.py:573-611), the per-object values
 *     sp[n] = float32(sum_{f not NA} log w[n][f][source(n,f)]) stored as float64 [N]; the caller
 *     keeps the reference's cache logic and sums.
 * sbe_observation_lh_exact: the row LikelihoodLogger._write_sample stores (loggers.py:354-359):
 *     sum_c w[n][f][c] * lh_exact[n][f][c] with the leave-one-out tables of a2; float64 [N][F]. */
int sbe_source_prior(sbe_engine* e, int slot, double* per_object_out /* [N] */);
/* Model.__call__ = likelihood + prior (sbayes/model/model.py:47-51; mcmc.py:273-328 asks one after the other for the
 * same candidate): sbe_collapsed_loglik_all's and sbe_source_prior's results for the same slot state, ONE launch and
 * one synchronisation.  Same values as the two calls, bit for bit. */
int sbe_collapsed_and_source_prior(sbe_engine* e, int slot, double* per_group_out /* [G_total] */,
                                   double* per_object_out /* [N] */);
int sbe_observation_lh_exact(sbe_engine* e, int slot, double* out /* [N][F] */);

/* GibbsSampleSource._propose (sbayes/sampling/operators.py:495-552) for the drop-in layer, in one call: the device chain
   of sbe_gibbs_step -- draw into the candidate slot with the caller's uniforms z [n_sub][F], the rest of the slot, count
   delta, tables, backward probabilities -- and what the reference's sample bookkeeping needs of it: src_new_out [n_sub][F]
   the drawn component of every observation (0xFF: an NA observation), sel_out / sel_back_out [n_sub][F] float32 = p[drawn]
   and p_back[old source] (1 where there is none), touched_out [<= G_total] (ascending) + *n_touched_out = the groups the
   listed objects are in, diff_rows_out [n_touched][F][S] = candidate counts - current counts of those groups (every other
   row of that difference is zero: counts.py:55-95).  One synchronisation.  Two forms, same results: ONE kernel, a block per
   16-feature tile doing draw, count delta, the touched groups' new tables and the backward probabilities in LDS -- the
   candidate slot is then NOT written -- or, when that kernel's LDS image does not fit (or SBE_OPT_FUSE_TABLES is 0),
   sbe_gibbs_step's chain, which builds the candidate in cand_slot: treat cand_slot as scratch.
   sbe_gibbs_propose_supported: 1 if the engine's tables fit the chain form (else the call can fail with SBE_ERR_ARG for
   large subsets; the call-by-call forms remain). */
int sbe_gibbs_propose_supported(sbe_engine* e);
int sbe_gibbs_propose(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                      float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out);
/* ... and the CURRENT slot takes the proposal (cf. sbe_counts_delta_apply): when it touches any group, cur_slot's counts,
 * the touched groups' probability tables and the subset's source rows become the proposal's -- inside the same launch
 * (tables and ids behind its completion flag), or as a copy of the candidate slot behind the chain form -- so the bind
 * of the sample the caller builds from these results has nothing to send. */
int sbe_gibbs_propose_apply(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                      float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out);

/* ---- slot management -------------------------------------------------------------------- */
int sbe_copy_slot(sbe_engine* e, int dst_slot, int src_slot);

#ifdef __cplusplus
}
#endif
#endif /* SBE_ENGINE_H */
