/* sbe_engine_diag.h -- self-test hooks and measurement support of the sbe engine.  NOT part of the drop-in boundary.
 *
 * Exported by the same library as include/sbe_engine.h, for this repository's tests (device routines checked in isolation:
 * the table-driven logs, lgamma, Philox, the completion-flag round trip) and for bench.py / tools/ (HIP-event timing on the
 * engine's own stream, the name of the kernel form a launch ran).  None of these has a counterpart in the reference; a
 * reference maintainer binding the engine does not need this header.
 */
#ifndef SBE_ENGINE_DIAG_H
#define SBE_ENGINE_DIAG_H

#include "sbe_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook: out[i][0..4) = philox4x32_10(counter = ctr_key[i][0..4), key = ctr_key[i][4..6)) */
int sbe_test_philox(sbe_engine* e, const uint32_t* ctr_key /* [n][6] */, int n, uint32_t* out /* [n][4] */);

/* ---- self-test hook: fp64 log used by the group-tuple table build vs the device library's log ---- */
int sbe_test_fast_log(sbe_engine* e, const double* in, int n, double* out_fast, double* out_lib);
/* table-driven fp64 log of k_mixture_tuple64's table build (error <= 1 ulp + 2^-53 absolute) */
int sbe_test_tab_log(sbe_engine* e, const double* in, int n, double* out);
/* the floor of a host-synchronous call: an empty kernel of n_blocks blocks that (mode bit 0) reads one word of the
   host-mapped input block and (bit 1) stores one double per block to the host-mapped result block, completion by flag */
int sbe_test_roundtrip(sbe_engine* e, int n_blocks, int mode);
/* lgamma of the Dirichlet-categorical terms (recurrence + Stirling series; a8, util.py:39-45, 1373-1394) */
int sbe_test_lgamma(sbe_engine* e, const double* in, int n, double* out);

/* ---- measurement support (bench.py): HIP events on the engine's own stream -------------- */
int sbe_timer_start(sbe_engine* e);
int sbe_timer_stop(sbe_engine* e, float* elapsed_ms);
/* One event pair around a whole loop of launches without a host wait in between: sbe_timer_start records the first event,
   sbe_timer_mark records the second one behind whatever has been enqueued since (no synchronisation), sbe_timer_elapsed waits
   for it and returns the span.  bench.py: span of the K back-to-back launches of the timed loop / K = the per-launch kernel
   time INCLUDING the dispatch gap between consecutive kernels -- a figure that fits inside ms_per_step by construction. */
int sbe_timer_mark(sbe_engine* e);
int sbe_timer_elapsed(sbe_engine* e, float* elapsed_ms);
/* Times `iters` back-to-back launches of the fused mixture kernel sequence on slots
 * [first_slot, first_slot+n) with one HIP event pair per launch sequence; returns the sum and
 * the per-launch average of the dominant kernel's duration in milliseconds. */
/* Event timing of the dominant kernel INSIDE the caller's own loop: enable = 1 starts recording one HIP event pair
   (on the engine's stream) around the fused kernel of every sbe_mixture_loglik[_batch[_async]] call; 2 pauses and
   3 resumes without forgetting the recorded pairs (so that only some launches of a loop are bracketed); enable = 0
   stops, synchronises and returns the number of recorded launches and their average duration. */
int sbe_kernel_timing(sbe_engine* e, int enable, int* n_launches, float* main_kernel_avg_ms);
/* name and form of the kernel the most recent fused-kernel launch ran (static string owned by the engine) */
const char* sbe_last_mixture_kernel(const sbe_engine* e);
int sbe_profile_mixture(sbe_engine* e, int first_slot, int n, int iters, float* total_ms,
                        float* main_kernel_avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* SBE_ENGINE_DIAG_H */
