#pragma once
// sbe_engine_internal.hip.h -- what the translation units of the engine's host side share (sbe_engine.hip: lifetime, state setters
// and getters, resident evaluation; sbe_engine_stateless.hip; sbe_engine_resident.hip; sbe_engine_steps.hip): the engine
// object, the error macros and the host helpers behind the C ABI declared in include/sbe_engine.h -- staging and result
// transport, completion by flag, data-check reporting, pattern / tuple derivation, the launch of the fused mixture kernels.
// The helpers live in an unnamed namespace: every unit compiles its own copy of the ones it uses (host code only; the
// kernels they launch are `static` for the same reason) -- a header, not a library, so that a change to one area rebuilds
// that area's unit and the units build in parallel.
//
// Plain HIP runtime (own stream, own events, own device memory); no torch, no compatibility
// layer.  One engine = one process' view of one GPU.  The one-hot feature block and every
// slot's state stay resident in HBM; only small tables / id vectors cross PCIe per call.
#include "sbe_kernels.hip.h"
#include "../../include/sbe_engine.h"
#include "../../include/sbe_engine_steps.h"
#include "../../include/sbe_engine_diag.h"
#include "sbe_host_helpers.h"   // marshalling helpers shared with the CPython extension (plain C)
#include "sbe_pool.h"          // host worker threads of sbe_step_batch (plain C++: also built under ThreadSanitizer)

#include <sched.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace sbe;

// the message of the last failed call without an engine (sbe_last_error(NULL)): ONE object for all units
inline thread_local std::string g_last_error;

namespace {

struct Slot {
    std::vector<uint16_t> h_gid;          // [C][N] host mirror (pattern derivation)
    std::vector<uint8_t> h_pid;           // [N]
    std::vector<uint32_t> patterns;       // distinct has_components bit patterns, sorted like np.unique
    int n_tuples = 0;                     // distinct group tuples of the objects (0 = more than kMaxTuples)
    std::vector<uint8_t> h_tid;           // [Np] tuple index per object            } host mirrors of the group-tuple
    std::vector<uint32_t> h_toff;         // [Np] tid * (S+1) * 512                  } tables (valid when n_tuples > 0;
    std::vector<uint16_t> h_tuple_g;      // [kMaxTuples][kMaxComponents]            } the one-call step ships them in
    std::vector<uint8_t> h_tuple_p;       // [kMaxTuples]                            } its payload)
    bool groups_set = false, weights_set = false, source_set = false;
    std::vector<uint8_t> probs_set, counts_set;   // per component
    bool patterns_dirty = true;
    uint32_t gid_pending = 0;             // components whose new ids (h_gid) are not resident yet: they travel with the pattern tables
    bool tables_follow = false;           // the host pattern / tuple tables already match h_gid (updated in O(moved objects))
    // round 3: object counts behind the pattern / tuple tables, so that a step which moves a few objects updates the
    // tables in O(moved objects) instead of re-deriving them from all N (prepare_step); valid while inc_ok
    std::vector<int32_t> pat_cnt;         // [256] objects per has_components bit pattern
    std::vector<int32_t> tup_cnt;         // [kMaxTuples] objects per group tuple
    bool inc_ok = false;
    uint64_t group_epoch = 0;             // identifies the content of the slot's group / pattern ids (k_rowoff's inputs)
    // round 6: components whose bool [G][N] matrix had an object in SEVERAL groups (sbe_set_groups keeps the LAST one as the
    // object's id: the group whose table an uncached evaluation ends up with, likelihood.py:126-130).  Calls that would DERIVE
    // counts from one id per object refuse such a slot (reject_overlap); ov_* remember the first example for their message.
    uint32_t overlap_mask = 0;
    int32_t ov_obj = -1, ov_g1 = 0, ov_g2 = 0, ov_comp = 0;
};

}  // namespace

// SBE_OPT_FUSE_TABLES' default for new engines (environment SBE_FUSE_TABLES=0: table kernels in front, for A/B runs)
static int fuse_tables_default() {
    static const int on = [] { const char* v = getenv("SBE_FUSE_TABLES"); return (v && atoi(v) == 0) ? 0 : 1; }();
    return on;
}

struct sbe_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> ev_pool;
    bool ev_timing = false;  int ev_used = 0;      // sbe_kernel_timing: event pairs recorded around the dominant kernel
    int N = 0, F = 0, S = 0, C = 0, n_slots = 0;
    int Fp = 0, rs_pitch = 0, Gtot = 0, Pmax = 0;
    int Np = 0, NQ = 0;            // objects padded to a multiple of 4; object quads
    int ft = 64, n_ftiles = 0, Fq = 0;   // v2 fused-kernel feature tile width, tiles, padded features
    bool direct = false;           // tables of a 16-feature tile exceed LDS: gather from the global tiled tables
    uint64_t rng_seed = 0, rng_draw = 0;   // Philox key / draw counter of sbe_sample_source(z = NULL)
    int compute_units = 256;
    std::vector<int> G, goff;
    int64_t n_na = 0;
    int64_t hbm_bytes = 0;
    std::string last_error;
    char device_name[64] = {0};
    char last_kernel[128] = "none";     // kernel form of the most recent fused-kernel launch (sbe_last_mixture_kernel)

    // options
    int opt_kernel = SBE_MIXTURE_PACKED;
    int opt_log = SBE_LOG_PRODUCT;
    int opt_fuse_tables = fuse_tables_default();   // SBE_OPT_FUSE_TABLES

    // resident data
    uint8_t* d_onehot = nullptr;   // [N][rs_pitch]
    uint8_t* d_state = nullptr;    // [N][Fp]
    uint8_t* d_state_q = nullptr;  // [NQ][Fq][4]  object-quad interleaved (v2 fused kernel)
    float* d_probs_t = nullptr;    // [slots][n_ftiles][Gtot+1][S][ft]
    double* d_wpat_t = nullptr;    // [slots][n_ftiles][Pmax][C][ft]
    // slot-strided state
    uint16_t* d_gid = nullptr;     // [slots][C][N]
    uint8_t* d_pid = nullptr;      // [slots][N]
    uint8_t* d_src = nullptr;      // [slots][N][Fp]
    int32_t* d_counts = nullptr;   // [slots][Gtot][F][S]
    float* d_probs = nullptr;      // [slots][Gtot][F][S]
    float* d_weights = nullptr;    // [slots][F][C]
    float* d_wpat = nullptr;       // [slots][Pmax][F][C]
    uint32_t* d_patbits = nullptr; // [slots][Pmax]
    uint16_t* d_state_h = nullptr; // [NQ][Fq][4] prepared LDS offsets of k_mixture_tuple64 (ft == 64, S <= 127) or null
    double2* d_logtab = nullptr;   // [128] {1/c, log c}: table of tab_log_pos (k_mixture_tuple64's table build)
    uint32_t* d_toff = nullptr;    // [slots][Np] byte offset of the object's tuple block, tid*(S+1)*512 (k_mixture_tuple64)
    uint32_t* d_rowoff = nullptr;  // [slots][C+1][Np] LDS byte offsets of k_mixture_rows (k_rowoff), or null
    int rows_ft = 0;               // tile width of k_mixture_rows (32 / 16; 0: its LDS image does not fit, or C > 4)
    std::vector<uint64_t> rowoff_epoch;   // per slot: Slot::group_epoch the device array was built from
    // pattern-sorted form of the rows kernel (built at its first launch): the slot's objects by has_components pattern
    uint32_t* d_rowoff_s = nullptr;  int rs_nq_max = 0;   // [slots][rs_nq_max][C+1][4] (k_rowsort)
    int32_t* d_rs_nq = nullptr;           // [slots] quads of the slot's padded order
    uint8_t* d_state_s = nullptr;         // [N + 1][Fp] state index, NA = S; row N all NA (the null object)
    std::vector<uint64_t> rowsort_epoch;
    int opt_rows_sorted = 1;              // SBE_ROWS_SORTED: 0 never, 1 launches of >= 16 slots at 32-feature tiles (default), 2 whenever it applies (tests)
    std::atomic<uint64_t> epoch_counter{0};
    double2* d_logtab_fine = nullptr;    // the matrix-pipe kernel's 1024-interval log table (built with d_xt)
    unsigned* d_arrive = nullptr;  // [slots + 1] tickets of the north-star kernels' in-kernel final reduction (finish_partial: per slot; matrix-pipe
                                   // form: per group of 16 slots); every launch leaves them at 0
    uint8_t* d_xt = nullptr;       // one-hot block in MFMA fragment order (k_mixture_tuple_mfma), built at the first batched launch
    int xt_NT = 0, xt_KBp = 0;  size_t xt_bytes = 0;
    int mfma_small_sl4 = 1;        // four slots per block for launches whose blocks fit two rounds on the CUs (SBE_MFMA_SMALL_SL4=0: A/B)
    int64_t mfma_min_obs = 6400000;  // ... chosen from 32 states per launch on when n x N x F reaches this (SBE_MFMA_MIN_OBS)
    int mfma_wide_min_share = 16;  // wide matrix-pipe forms (> 8 tuples) by default only from this many objects per padded tuple on (SBE_MFMA_WIDE_MIN_SHARE)
    int mfma_min_batch = 320;      // smallest launch the matrix-pipe form is chosen for under SBE_MIXTURE_PACKED (SBE_MFMA_MIN_BATCH);
                                   // round 6, FP4 operands: 24.5 / 24.6 / 24.8 us against 22.2 / 32.1 / 34.9 us of k_mixture_tuple64 at
                                   // 256 / 384 / 512 headline states (tools/diag/mfma_threshold.py, profiles/r6/mfma_threshold.log)
    uint8_t* d_tid = nullptr;      // [slots][Np] group-tuple index per object (k_mixture_combo)
    uint16_t* d_tuple_g = nullptr; // [slots][kMaxTuples][kMaxComponents]
    uint8_t* d_tuple_p = nullptr;  // [slots][kMaxTuples]
    double* d_conc = nullptr;      // [Gtot][F][S]
    double* d_unif = nullptr;      // [F][S]  staging of the per-call unif_counts argument
    double* d_lg_conc = nullptr; double* d_sum_a = nullptr; double* d_lg_sum_a = nullptr;   // k_conc_lgamma: [Gtot][F][S], [Gtot][F] x 2
    double* d_unif_res = nullptr;  bool unif_set = false;   // [F][S] resident (sbe_set_uniform_counts): the resident operator forms
    int32_t* d_comp_of_group = nullptr;                     // [Gtot] mixture component of every global group index
    std::vector<uint8_t> conc_set;
    // scratch
    double* d_partials = nullptr;  int64_t partials_stride = 0;   // [slots][max_blocks]
    double* d_results = nullptr;   // [slots] device view of h_results (host-mapped)
    double* h_results = nullptr;   // pinned + mapped [slots]: k_reduce_partials writes straight to the host
    int* d_status = nullptr;       // [ST_WORDS]
    int* h_status = nullptr;       // pinned
    int* h_flag = nullptr;  int* d_flag = nullptr;   // host-mapped [ST_WORDS]: "a kernel raised this word" (raise_status)
    unsigned long long* h_done = nullptr;  unsigned long long* d_done = nullptr;   // host-mapped: sequence number of the last call
    unsigned* d_ticket = nullptr;  unsigned long long done_seq = 0;                // finished by flag (signal_done / wait_done)
    // results streamed by the kernel into host-mapped staging, chunk by chunk (signal_chunk / stream_result)
    static constexpr int kMaxChunks = 16;
    unsigned long long* h_chunk_flags = nullptr;  unsigned long long* d_chunk_flags = nullptr;   // host-mapped [kMaxChunks]
    unsigned* d_chunk_tickets = nullptr;  unsigned long long chunk_seq = 0;
    uint8_t* h_stream = nullptr;  uint8_t* d_stream = nullptr;  size_t stream_bytes = 0;        // host-mapped staging + its device view
    uint8_t* d_changed = nullptr;  // [Gtot]
    uint32_t* d_step_stamp = nullptr;  uint32_t step_id = 0;   // [Gtot] group changed in step `step_id` (k_step_core)
    float* d_step_pf = nullptr;    // [Gtot][F]  per-feature collapsed log-pdf of the fused step call
    double* d_step_pg = nullptr;   // [Gtot]     per-group collapsed log-likelihood of the fused step call
    // one-call step (sbe_step): payload sections (byte offsets into d_step_payload / its pinned staging copy) and
    // the host-mapped result block (per-group values | data-check words | changed-group flags)
    struct StepLayout { size_t ids, pid, tid, toff, tuple_g, tuple_p, patbits, weights, row_of, subset, stale, objects, rows, total; } sl{};
    int step_max_rows = 0;
    uint8_t* h_step_payload = nullptr; uint8_t* d_step_payload = nullptr;   // host-mapped pinned: the kernels read it over PCIe
    uint8_t* h_io = nullptr; uint8_t* d_io = nullptr; size_t io_bytes = 0;  // host-mapped pinned: small inputs / outputs of
                                                                            // latency-bound calls, read / written in place
    uint8_t* h_step = nullptr;     uint8_t* d_step_host = nullptr;   // mapped: [Gtot] f64 | [ST_WORDS] i32 | [Gtot] u8
    // batched steps (sbe_step_batch): one lane per chain of the batch = its own payload block, result block,
    // per-feature buffer and change stamps (lane 0 of the single-step calls is the set of members above)
    struct Lane { uint8_t* h_payload; uint8_t* d_payload; uint8_t* h_step; uint8_t* d_step_host; float* d_pf;
                  uint32_t* d_stamp; uint32_t step_id;
                  int* d_status; };     // data-check words of THIS lane's kernels: a malformed proposal of one chain of a
                                        // batch is reported for that chain only (lane 0 of the single steps: e->d_status)
    std::vector<Lane> lanes;
    uint8_t* d_batch_meta = nullptr; size_t batch_meta_bytes = 0;     // device copy of the batch's StepCore / StepFinish / slot lists
    // batched steps: the chains' payloads packed back to back in ONE pinned block and sent with ONE copy into device
    // memory (64 chains reading ~25 KB each in place over PCIe made k_step_core_batch PCIe-bound: 160 us)
    uint8_t* h_batch_payload = nullptr; uint8_t* d_batch_payload = nullptr; size_t batch_payload_bytes = 0;
    std::vector<Slot> batch_cands;                                    // candidates' host state, storage reused across calls
    std::vector<std::vector<int32_t>> batch_moved;                    // per chain of a batch: the objects its step moved
    std::vector<int32_t> step_moved;                                  // ... of the single step
    struct Pool;                                                      // host worker threads of sbe_step_batch (lazily started)
    Pool* pool = nullptr;
    uint8_t* d_scratch = nullptr;  size_t scratch_bytes = 0;     // general staging
    uint8_t* h_pinned = nullptr;   size_t pinned_bytes = 0;      // pinned D2H staging
    uint8_t* h_arena = nullptr;    uint8_t* d_arena = nullptr;   size_t arena_bytes = 0, arena_off = 0;   // pinned, host-mapped H2D staging ring
    SetterJobs* batch = nullptr;   // sbe_set_slot_delta: the setters' launches are collected here and issued as ONE kernel
    int opt_step_form = 0;         // SBE_OPT_STEP_FORM
    int opt_step_derive = 0;       // SBE_OPT_STEP_DERIVE: 1 = always re-derive patterns / tuples from all objects
    int opt_deferred = 0;          // SBE_OPT_DEFERRED_CHECKS: data checks reported at the next sync
    bool status_pending = false;
    // deferred data checks: which entry points enqueued a kernel that may have raised one since the last report
    // (the report is delivered by a LATER call: its message names where the data came in)
    const char* pending_origin[2] = {nullptr, nullptr};      // [0] normalize (tables), [1] one-hot source; most recent caller
    int pending_calls = 0;
    std::vector<Slot> slots;
    // One-call steps: which rows of a slot's source array differ from its partner slot's (round 3).  A chain's two
    // slots hold the same source except for the rows the LAST step changed (accepted: the old current slot lacks them;
    // rejected: the candidate slot carries them), so the next step copies those rows instead of the whole [N][Fp]
    // array (256 KB per chain and step at the headline shape).  `version` counts every write to the slot's source;
    // a record is valid only while both versions are the ones it was made at -- any other writer (sbe_set_source[_rows],
    // sbe_sample_source, sbe_copy_slot, the call-by-call step, sbe_gibbs_step) bumps the version and the next step
    // falls back to the full copy.  Kept outside `Slot` (slots are assigned wholesale: candidate = copy of current).
    struct SrcSync { uint64_t version = 1; int peer = -1; uint64_t peer_version = 0, own_version = 0; std::vector<int32_t> diff; };
    std::vector<SrcSync> src_sync;
    // the same bookkeeping for the per-object id arrays (group id of component 0, pattern id, tuple id / offset) on the
    // device AND in the host mirror `Slot`: `diff` = the objects whose entries differ between the two slots of a chain
    // (sbe_step_batch_delta patches those entries instead of re-deriving / copying whole arrays)
    std::vector<SrcSync> ids_sync;

    int64_t table_elems() const { return (int64_t)Gtot * F * S; }
    int64_t tile_tab_elems() const { return (int64_t)(Gtot + 1) * S * ft; }
    int64_t probs_t_elems() const { return (int64_t)n_ftiles * tile_tab_elems(); }
    int64_t wpat_tile_elems() const { return (int64_t)Pmax * C * ft; }
    int64_t wpat_t_elems() const { return (int64_t)n_ftiles * wpat_tile_elems(); }
};

namespace {
inline void bump_src(sbe_engine* e, int slot) { ++e->src_sync[slot].version; }
inline void bump_ids(sbe_engine* e, int slot) { ++e->ids_sync[slot].version; }
}

struct sbe_engine::Pool : sbe_host::StepPool { using sbe_host::StepPool::StepPool; };

namespace {

int fail(sbe_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (e) e->last_error = buf;
    return code;
}

#define HIPCHK(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _err = (call);                                                              \
        if (_err != hipSuccess)                                                                \
            return fail(e, SBE_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_err), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

#define CHECK_ENGINE(e) \
    if (!(e)) return fail(nullptr, SBE_ERR_ARG, "null engine handle")
#define CHECK_SLOT(e, s) \
    if ((s) < 0 || (s) >= (e)->n_slots) return fail(e, SBE_ERR_ARG, "slot %d out of range [0,%d)", (s), (e)->n_slots)
#define CHECK_COMP(e, c) \
    if ((c) < 0 || (c) >= (e)->C) return fail(e, SBE_ERR_ARG, "component %d out of range [0,%d)", (c), (e)->C)
#define CHECK_PTR(e, p) \
    if (!(p)) return fail(e, SBE_ERR_ARG, "null pointer argument: %s", #p)

inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

template <class T>
int dmalloc(sbe_engine* e, T** p, int64_t n) {
    const size_t bytes = std::max<int64_t>(n, 1) * sizeof(T);
    HIPCHK(e, hipMalloc((void**)p, bytes));
    e->hbm_bytes += (int64_t)bytes;
    return SBE_OK;
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

int ensure_scratch(sbe_engine* e, size_t bytes) {
    if (bytes <= e->scratch_bytes) return SBE_OK;
    if (e->d_scratch) { HIPCHK(e, hipStreamSynchronize(e->stream)); HIPCHK(e, hipFree(e->d_scratch)); }
    e->scratch_bytes = bytes + bytes / 4 + 4096;
    HIPCHK(e, hipMalloc((void**)&e->d_scratch, e->scratch_bytes));
    return SBE_OK;
}

// host-mapped I/O block for latency-bound calls (the kernels read their small inputs and write their small results
// over PCIe in place: no copy-engine hop in the dependency chain); one call at a time, each ends with a stream sync
int ensure_io(sbe_engine* e, size_t bytes) {
    if (bytes <= e->io_bytes) return SBE_OK;
    if (e->h_io) { HIPCHK(e, hipStreamSynchronize(e->stream)); HIPCHK(e, hipHostFree(e->h_io)); e->h_io = nullptr; e->io_bytes = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(e, hipHostMalloc((void**)&e->h_io, want, hipHostMallocMapped));
    HIPCHK(e, hipHostGetDevicePointer((void**)&e->d_io, e->h_io, 0));
    e->io_bytes = want;
    return SBE_OK;
}

int ensure_pinned(sbe_engine* e, size_t bytes) {
    if (bytes <= e->pinned_bytes) return SBE_OK;
    if (e->h_pinned) { HIPCHK(e, hipStreamSynchronize(e->stream)); HIPCHK(e, hipHostFree(e->h_pinned)); }
    e->pinned_bytes = bytes + bytes / 4 + 4096;
    HIPCHK(e, hipHostMalloc((void**)&e->h_pinned, e->pinned_bytes, hipHostMallocDefault));
    return SBE_OK;
}

int upload(sbe_engine* e, void* dst_dev, const void* src, size_t bytes);
int synced(sbe_engine* e);

int ensure_step_pool(sbe_engine* e);

// ---- large results streamed by the kernel (the literal a1 / a3 surfaces) ---------------------------
// tools/d2h_probe.hip on an MI355X box: 3.2 MB cross PCIe in 67 us by one hipMemcpyAsync and in 69 us when a kernel
// stores them straight into host-mapped memory; every further copy operation on the stream costs ~7 us and every event
// behind one more (four pieces with events: the round-2 / round-3 form, ~40 us over the plain copy); one host thread
// copies out of pinned memory at 27 GB/s -- 120 us for 3.2 MB, twice the transfer.  So: the kernel writes its result
// into the host-mapped staging buffer `h_stream` and reports completion chunk by chunk (signal_chunk); the host jobs --
// work(j) copies / scatters staging bytes [.., job_end(j)) to the caller -- run on the engine's pool (sbe_pool.h:
// run_as_chunks_land) as the chunks land, every thread reading the chunk flags.  No copy engine, no event, one launch.
// The kernel's grid is ONE chunk's worth of blocks walking the chunks in order (plan_stream), so chunk k is complete and
// being copied out while chunk k+1 crosses PCIe.
int ensure_stream(sbe_engine* e, size_t bytes) {
    if (bytes <= e->stream_bytes) return SBE_OK;
    if (e->h_stream) { HIPCHK(e, hipStreamSynchronize(e->stream)); HIPCHK(e, hipHostFree(e->h_stream)); e->h_stream = nullptr; e->stream_bytes = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(e, hipHostMalloc((void**)&e->h_stream, want, hipHostMallocMapped));
    HIPCHK(e, hipHostGetDevicePointer((void**)&e->d_stream, e->h_stream, 0));
    e->stream_bytes = want;
    return SBE_OK;
}

struct StreamPlan { ChunkSig sig; int n_chunks; size_t chunk_bytes; size_t bytes; unsigned grid; };

// chunks of whole blocks: `bytes_per_block` result bytes per block, n_blocks blocks, at most kMaxChunks chunks of >= 128 KB
StreamPlan plan_stream(sbe_engine* e, unsigned n_blocks, size_t bytes_per_block, size_t bytes) {
    // 8 chunks, walked IN ORDER by a grid of one chunk's worth of blocks (signal_chunk_ordered): same-box A/B
    // (profiles/r4/ab_d2h_4_ordered_chunks.log) a1 11.6-12.1 -> 13.1-14.9 k calls/s, a3 9.3-10.3 -> 11.2-12.6 k against one block
    // per 2 x 256 elements with every block resident at once (all chunks then complete together, at the end of the kernel,
    // and the host copy overlaps nothing); 4 / 8 / 16 chunks within noise of each other.
    static const int max_chunks = [] { const char* v = getenv("SBE_STREAM_CHUNKS"); const int n = v ? atoi(v) : 0;      // (experiments)
                                       return n >= 1 && n <= sbe_engine::kMaxChunks ? n : 8; }();
    unsigned per = std::max<unsigned>(1, (unsigned)div_up((int64_t)n_blocks, max_chunks));
    per = std::max<unsigned>(per, (unsigned)div_up((int64_t)128 << 10, (int64_t)bytes_per_block));
    const int n_chunks = (int)div_up((int64_t)n_blocks, (int64_t)per);
    // SBE_STREAM_ORDERED=0 (A/B): one block per 2 x 256 elements, every block signalling its own chunk
    static const bool ordered = [] { const char* v = getenv("SBE_STREAM_ORDERED"); return !(v && atoi(v) == 0); }();
    ChunkSig sig{e->d_chunk_tickets, e->d_chunk_flags, ++e->chunk_seq, per, n_blocks, (unsigned)n_chunks, 0};
    unsigned grid = n_blocks;
    if (ordered && n_chunks > 1) {
        sig.chunk_elems = (long long)per * (long long)(bytes_per_block / sizeof(double));
        grid = per;
    }
    return StreamPlan{sig, n_chunks, (size_t)per * bytes_per_block, bytes, grid};
}

template <class JobBegin, class JobEnd, class Work>
int stream_result(sbe_engine* e, const StreamPlan& plan, int n_jobs, JobBegin job_begin, JobEnd job_end, Work work) {
    static const bool single_thread = [] { const char* v = getenv("SBE_D2H_THREADS"); return v && atoi(v) == 1; }();   // (A/B)
    if (!single_thread) { int rc = ensure_step_pool(e); if (rc) return rc; }
    std::atomic<bool> all_landed{false};
    hipError_t sync_err = hipSuccess;
    const volatile unsigned long long* flags = e->h_chunk_flags;
    const unsigned long long seq = plan.sig.seq;
    const size_t chunk_bytes = plan.chunk_bytes;
    const auto t_limit = std::chrono::steady_clock::now() + std::chrono::microseconds(2000 + (int64_t)(plan.bytes / 10000));   // 2 ms + 10 GB/s
    unsigned spins = 0;
    sbe_host::run_as_chunks_land(
        single_thread ? nullptr : e->pool, n_jobs,
        [=](int j) { return (int)(job_begin(j) / chunk_bytes); },
        [=](int j) { return (int)((job_end(j) - 1) / chunk_bytes); },
        [&, flags, seq](int k) { return flags[k] == seq || all_landed.load(std::memory_order_acquire); },
        [&] {                                     // calling thread only: a kernel that never reports -> the runtime's wait ends the call
            if ((++spins & 1023u) == 0 && !all_landed.load(std::memory_order_relaxed) && std::chrono::steady_clock::now() > t_limit) {
                sync_err = hipStreamSynchronize(e->stream);
                all_landed.store(true, std::memory_order_release);
            }
        },
        work);
    if (sync_err != hipSuccess) return fail(e, SBE_ERR_HIP, "hipStreamSynchronize (streamed result): %s", hipGetErrorString(sync_err));
    return SBE_OK;                                // every chunk flag seen: the kernel's blocks have finished their stores
}

// D2H through the pinned staging buffer (pageable destinations would be staged by the
// runtime anyway, in smaller pieces)
int d2h(sbe_engine* e, void* dst, const void* src_dev, size_t bytes) {
    int rc = ensure_pinned(e, bytes);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(e->h_pinned, src_dev, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    // large results (observation likelihoods, exact forms, normalised weights; the [N, F] / [N, F, C] arrays of the
    // literal a1 / a3 surfaces are streamed by their kernels: stream_result): ONE copy operation -- every further piece
    // costs ~7 us of stream time and every event behind one more (tools/d2h_probe.hip), which is what the piecewise form
    // of rounds 2-3 paid for its overlap -- then the copy out of the staging buffer (27 GB/s per thread: as long as the
    // transfer itself on one thread) spread over the host pool in 64 KB jobs
    static const bool single_copy = [] { const char* v = getenv("SBE_D2H_THREADS"); return v && atoi(v) == 1; }();   // (A/B: tools/ab_d2h.py)
    if (bytes >= ((size_t)1 << 20) && !single_copy) {
        rc = ensure_step_pool(e);
        if (rc) return rc;
        constexpr size_t kJob = (size_t)64 << 10;
        uint8_t* out = (uint8_t*)dst;
        const uint8_t* stage = e->h_pinned;
        e->pool->run((int)((bytes + kJob - 1) / kJob),
                     [=](int j) { const size_t o = (size_t)j * kJob; memcpy(out + o, stage + o, std::min(kJob, bytes - o)); });
        return synced(e);
    }
    memcpy(dst, e->h_pinned, bytes);
    return synced(e);
}

// Where a result kernel writes: small results go straight into the host-mapped I/O block (posted PCIe writes, no copy
// operation behind the kernel), large ones into `dev_fallback` and back through the staging copy.  For calls that do
// not use the I/O block for anything else.  out_fetch ends the call: synchronise, data checks, result to the caller.
constexpr size_t kMappedOutMax = (size_t)1 << 18;
int out_target(sbe_engine* e, size_t bytes, void* dev_fallback, void** target) {
    *target = dev_fallback;
    if (bytes > kMappedOutMax) return SBE_OK;
    int rc = ensure_io(e, bytes);
    if (rc) return rc;
    *target = e->d_io;
    return SBE_OK;
}
int wait_done(sbe_engine* e, const DoneSig& d);
DoneSig next_done(sbe_engine* e, unsigned n_blocks);
// (`done`: the descriptor the result kernel was launched with -- out_done() -- when the result is in mapped memory)
DoneSig out_done(sbe_engine* e, const void* target, unsigned n_blocks) {
    return target == (const void*)e->d_io ? next_done(e, n_blocks) : DoneSig{};
}
int out_fetch(sbe_engine* e, void* host_out, const void* target, size_t bytes, const DoneSig& done = DoneSig{}) {
    if (target != (const void*)e->d_io) return d2h(e, host_out, target, bytes);
    int rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(host_out, e->h_io, bytes);
    return synced(e);
}

int h2d(sbe_engine* e, void* dst_dev, const void* src, size_t bytes) { return upload(e, dst_dev, src, bytes); }

// H2D of caller-owned (pageable) memory without a stream synchronize: small payloads are copied into a
// pinned staging ring and sent with a truly asynchronous hipMemcpyAsync, so the caller's buffer is
// free when the call returns and state-setting calls do not stall the stream.  The ring wraps after a
// stream synchronize (single in-order stream: everything staged before it has been consumed).
int upload(sbe_engine* e, void* dst_dev, const void* src, size_t bytes) {
    if (bytes == 0) return SBE_OK;
    if (e->h_arena && bytes <= e->arena_bytes / 4) {
        const size_t need = (bytes + 63) / 64 * 64;
        if (e->arena_off + need > e->arena_bytes) {
            HIPCHK(e, hipStreamSynchronize(e->stream));
            e->arena_off = 0;
        }
        uint8_t* stage = e->h_arena + e->arena_off;
        e->arena_off += need;
        memcpy(stage, src, bytes);
        HIPCHK(e, hipMemcpyAsync(dst_dev, stage, bytes, hipMemcpyHostToDevice, e->stream));
        return SBE_OK;
    }
    HIPCHK(e, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return SBE_OK;
}

// Host data that ONE kernel reads once, element-parallel (rows to ingest, index lists): staged in the mapped ring and
// read by that kernel in place over PCIe -- no copy operation in the stream, one enqueue per setter instead of one per
// array.  (Not for kernels that WALK their input: every dependent step would be a PCIe round trip -- sbe_counts_delta.)
// Payloads above 64 KB go to `dev_fallback` with an ordinary upload.  *out = what the kernel reads.
int stage(sbe_engine* e, const void* src, size_t bytes, void* dev_fallback, const void** out) {
    constexpr size_t kDirect = (size_t)64 << 10;
    *out = dev_fallback;
    if (bytes == 0) return SBE_OK;
    if (!e->d_arena || bytes > kDirect) return upload(e, dev_fallback, src, bytes);
    const size_t need = (bytes + 63) / 64 * 64;
    if (e->arena_off + need > e->arena_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->arena_off = 0;
    }
    memcpy(e->h_arena + e->arena_off, src, bytes);
    *out = e->d_arena + e->arena_off;
    e->arena_off += need;
    return SBE_OK;
}

// Up to eight small host arrays to their resident places with ONE enqueue (k_scatter_bytes out of the mapped ring);
// ordinary uploads, one per array, when the arrays do not fit the direct path.  `wp` (optional): the same launch also
// computes a slot's per-pattern normalised weights (k_scatter_weight_patterns) -- pattern bits = segment wp->bits_seg as
// staged, weights = the resident copy or `wp->new_weights` staged in the same reservation; wp->done says whether it did.
struct UploadSeg { void* dst; const void* src; size_t bytes; };
struct FusedWeightPatterns { WeightPatternArgs args; int bits_seg; const float* new_weights; bool done; };
int upload_segments(sbe_engine* e, const UploadSeg* segs, int n, FusedWeightPatterns* wp = nullptr) {
    size_t total = 0, largest = 0;
    for (int i = 0; i < n; ++i) { total += (segs[i].bytes + 63) / 64 * 64; largest = std::max(largest, segs[i].bytes); }
    if (wp) wp->done = false;
    if (total == 0) return SBE_OK;
    if (!e->d_arena || n > 8 || total > ((size_t)64 << 10)) {
        for (int i = 0; i < n; ++i) { int rc = upload(e, segs[i].dst, segs[i].src, segs[i].bytes); if (rc) return rc; }
        return SBE_OK;
    }
    const size_t w_bytes = wp && wp->new_weights ? (size_t)wp->args.F * wp->args.C * sizeof(float) : 0;
    const bool fuse = wp && segs[wp->bits_seg].bytes > 0 && w_bytes <= ((size_t)64 << 10);
    const size_t reserve = total + (fuse ? (w_bytes + 63) / 64 * 64 : 0);
    if (e->arena_off + reserve > e->arena_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->arena_off = 0;
    }
    ScatterSegs sg{};
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (segs[i].bytes == 0) continue;
        memcpy(e->h_arena + e->arena_off + off, segs[i].src, segs[i].bytes);
        sg.dst[sg.n] = (uint8_t*)segs[i].dst; sg.off[sg.n] = (uint32_t)off; sg.bytes[sg.n] = (uint32_t)segs[i].bytes;
        ++sg.n;
        if (fuse && i == wp->bits_seg) wp->args.pattern_bits = (const uint32_t*)(e->d_arena + e->arena_off + off);
        off += (segs[i].bytes + 63) / 64 * 64;
    }
    const unsigned sx = (unsigned)std::min<size_t>(div_up((int64_t)largest, 1024), 16);
    unsigned wx = 0;
    if (fuse) {
        if (w_bytes) {
            memcpy(e->h_arena + e->arena_off + off, wp->new_weights, w_bytes);
            wp->args.weights = (const float*)(e->d_arena + e->arena_off + off);
        }
        wx = (unsigned)div_up((int64_t)wp->args.P * wp->args.F, 256);
        wp->done = true;
    }
    if (e->batch && e->batch->n_group_blocks == 0 && (fuse || !wp)) {   // (sbe_set_slot_delta: launched with the call's other setters;
                                                                        //  never when a separate weight kernel would follow the scatter)
        SetterJobs& j = *e->batch;
        j.group_base = e->d_arena + e->arena_off; j.sg = sg; j.has_wp = fuse ? 1 : 0;
        if (fuse) j.wp = wp->args;
        j.group_gx = std::max(sx, wx); j.n_group_blocks = j.group_gx * (unsigned)(sg.n + (fuse ? 1 : 0));
    } else if (fuse) {
        k_scatter_weight_patterns<<<dim3(std::max(sx, wx), sg.n + 1), 256, 0, e->stream>>>(e->d_arena + e->arena_off, sg, wp->args);
    } else {
        k_scatter_bytes<<<dim3(sx, sg.n), 256, 0, e->stream>>>(e->d_arena + e->arena_off, sg);
    }
    HIPCHK(e, hipGetLastError());
    e->arena_off += reserve;
    return SBE_OK;
}

// Completion by flag (signal_done in the kernels): the call's last kernel carries next_done()'s descriptor, the host
// spins on the mapped word in wait_done() -- a few hundred microseconds at most, then the runtime's wait (a long launch, or
// a fault, which that wait reports).  After wait_done() the results in host-mapped memory are complete and every earlier
// operation of the in-order stream has finished; what may still be pending is the kernel's own retirement.
// SBE_POLL_DONE=0 switches the mechanism off (every wait is hipStreamSynchronize: A/B and fallback).
bool poll_done_enabled() {
    static const bool on = [] { const char* v = getenv("SBE_POLL_DONE"); return !(v && atoi(v) == 0); }();
    return on;
}
DoneSig next_done(sbe_engine* e, unsigned n_blocks) {
    if (!poll_done_enabled()) return DoneSig{};
    return DoneSig{e->d_ticket, e->d_done, ++e->done_seq, n_blocks};
}
int wait_done(sbe_engine* e, const DoneSig& d) {
    if (d.flag) {
        const volatile unsigned long long* f = e->h_done;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1;; ++spins) {
            if (*f == d.seq) { std::atomic_thread_fence(std::memory_order_acquire); return SBE_OK; }
            // (every 128th turn the core is offered to whoever else is runnable: several single-chain processes share a host,
            //  and a spinner that never yields holds back the thread that would feed the GPU; free on an idle host)
            if ((spins & 127u) == 0u) sched_yield(); else __builtin_ia32_pause();
            if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
        }
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return SBE_OK;
}

// The device words hold the counts; the host-mapped flag words (raise_status) say whether anything was raised, so a
// clean call costs no read-back.  Flags are read after a stream synchronisation (kernel stores are visible then).
bool status_raised(const sbe_engine* e) {
    const volatile int* f = e->h_flag;
    return (f[ST_BAD_NORMALIZE] | f[ST_MULTI_SOURCE]) != 0;
}

// slow path (something was raised): the counts into h_status, device words and flags back to zero.  Stream idle on return.
int fetch_and_clear_status(sbe_engine* e) {
    HIPCHK(e, hipMemcpyAsync(e->h_status, e->d_status, ST_FLAG_PTR * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_status + ST_BAD_NORMALIZE, 0, 2 * sizeof(int), e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->h_flag[ST_BAD_NORMALIZE] = e->h_flag[ST_MULTI_SOURCE] = 0;
    return SBE_OK;
}

int report_status(sbe_engine* e, bool deferred = false) {        // after a stream synchronisation
    const int n_calls = e->pending_calls;
    const char* origin[2] = {e->pending_origin[0], e->pending_origin[1]};
    e->pending_calls = 0;
    e->pending_origin[0] = e->pending_origin[1] = nullptr;
    if (!status_raised(e)) return SBE_OK;
    int rc = fetch_and_clear_status(e);
    if (rc) return rc;
    const int bad_norm = e->h_status[ST_BAD_NORMALIZE], multi_src = e->h_status[ST_MULTI_SOURCE];
    e->h_status[ST_BAD_NORMALIZE] = e->h_status[ST_MULTI_SOURCE] = 0;
    // a deferred report surfaces in a later call than the one that supplied the data: say so, and say which
    char where[200] = "";
    const char* who = origin[bad_norm ? 0 : 1];
    if (deferred && who)
        snprintf(where, sizeof where, " [deferred data check: raised by %s (%d state-setting call%s queued since the last report), "
                 "reported by the first call that waited for the device]", who, n_calls, n_calls == 1 ? "" : "s");
    if (bad_norm)
        return fail(e, SBE_ERR_DATA, "normalize: %d rows have a non-positive sum (sbayes/util.py:1006 assert)%s", bad_norm, where);
    return fail(e, SBE_ERR_DATA, "source is not one-hot over components in %d observations%s", multi_src, where);
}

// Data checks raised by kernels (normalize's positive-sum assert, one-hot source).  Immediate mode: synchronize and
// report now.  Deferred mode (SBE_OPT_DEFERRED_CHECKS): nothing is enqueued; the next call that synchronizes anyway
// looks at the flag words and reports.
int check_after(sbe_engine* e, int word, const char* who = __builtin_FUNCTION()) {   // after enqueuing a kernel that may raise a data check
    if (e->opt_deferred) {
        e->status_pending = true;
        e->pending_origin[word == ST_BAD_NORMALIZE ? 0 : 1] = who;
        ++e->pending_calls;
        return SBE_OK;
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->status_pending = false;
    return report_status(e);
}

int synced(sbe_engine* e) {               // call right after any hipStreamSynchronize in a result path
    if (!e->status_pending) return SBE_OK;
    e->status_pending = false;
    return report_status(e, true);
}

// Synchronize; deliver a deferred report; then leave THIS call's counts in h_status (zeros when nothing was raised)
// for callers with their own wording.  Device words and flags are cleared either way.
int read_status(sbe_engine* e) {
    HIPCHK(e, hipStreamSynchronize(e->stream));
    int rc = synced(e);
    if (rc) return rc;
    e->h_status[ST_BAD_NORMALIZE] = e->h_status[ST_MULTI_SOURCE] = 0;
    if (status_raised(e)) return fetch_and_clear_status(e);
    return SBE_OK;
}

// read_status for a call that has already waited (wait_done / out_fetch): this call's counts into h_status.
int take_status(sbe_engine* e) {
    e->h_status[ST_BAD_NORMALIZE] = e->h_status[ST_MULTI_SOURCE] = 0;
    if (status_raised(e)) return fetch_and_clear_status(e);
    return SBE_OK;
}

int clear_status_word(sbe_engine* e, int word) {
    if (e->status_pending) return SBE_OK;      // sticky until the deferred report has been delivered
    if (!e->h_flag[word]) return SBE_OK;       // never raised since the last report: the device word is zero
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_status + word, 0, sizeof(int), e->stream));
    e->h_flag[word] = 0;
    return SBE_OK;
}

// has_components patterns in np.unique(axis=0) order: rows compared lexicographically over
// components 0..C-1 with False < True (likelihood.py:183).
void derive_patterns(sbe_engine* e, Slot& s) {
    const int N = e->N, C = e->C;          // C <= 8: a pattern is an 8-bit mask
    static thread_local std::vector<uint8_t> bits;     // (called per chain and step by the batched step's pool threads)
    bits.resize(N);
    bool seen[256] = {false};
    s.pat_cnt.assign(256, 0);
    for (int n = 0; n < N; ++n) {
        uint32_t b = 0;
        for (int c = 0; c < C; ++c)
            if (s.h_gid[(size_t)c * N + n] != kNoGroup) b |= 1u << c;
        bits[n] = (uint8_t)b;
        seen[b] = true;
        ++s.pat_cnt[b];
    }
    auto key = [C](uint32_t b) {   // component 0 most significant => lexicographic row order of np.unique
        uint32_t k = 0;
        for (int c = 0; c < C; ++c) k |= ((b >> c) & 1u) << (C - 1 - c);
        return k;
    };
    std::vector<uint32_t> uniq;
    for (uint32_t b = 0; b < 256; ++b) if (seen[b]) uniq.push_back(b);
    std::sort(uniq.begin(), uniq.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
    uint8_t index_of[256] = {0};
    for (size_t i = 0; i < uniq.size(); ++i) index_of[uniq[i]] = (uint8_t)i;
    s.patterns = uniq;
    s.h_pid.resize(N);
    for (int n = 0; n < N; ++n) s.h_pid[n] = index_of[bits[n]];
}

// distinct group tuples (g_0..g_{C-1}) of the objects, for the group-tuple kernels (host mirrors only)
void derive_tuples(sbe_engine* e, Slot& s) {
    const int N = e->N, C = e->C;
    s.h_tid.assign(e->Np, 0);
    s.h_toff.assign(e->Np, 0);
    s.h_tuple_g.assign((size_t)kMaxTuples * kMaxComponents, (uint16_t)e->Gtot);
    s.h_tuple_p.assign(kMaxTuples, 0xFF);            // 0xFF = tuple not present in this slot
    uint16_t tuples[kMaxTuples][kMaxComponents];
    int n_tup = 0;
    bool ok = true;
    s.tup_cnt.assign(kMaxTuples, 0);
    uint64_t packed[kMaxTuples];                      // C <= 4: a tuple is one 64-bit key (integer compares, no memcmp)
    int last = 0;                                     // neighbouring objects often share their tuple
    for (int n = 0; n < N && ok; ++n) {
        uint16_t key[kMaxComponents];
        for (int c = 0; c < C; ++c) key[c] = s.h_gid[(size_t)c * N + n];
        int t = 0;
        if (C <= 4) {
            uint64_t k64 = 0;
            for (int c = 0; c < C; ++c) k64 |= (uint64_t)key[c] << (16 * c);
            if (n_tup && packed[last] == k64) t = last;
            else for (; t < n_tup; ++t) if (packed[t] == k64) break;
            if (t == n_tup && n_tup < kMaxTuples) packed[n_tup] = k64;
        } else {
            for (; t < n_tup; ++t) if (memcmp(tuples[t], key, (size_t)C * sizeof(uint16_t)) == 0) break;
        }
        last = t;
        if (t == n_tup) {
            if (n_tup == kMaxTuples) { ok = false; break; }
            memcpy(tuples[n_tup++], key, (size_t)C * sizeof(uint16_t));
            for (int c = 0; c < C; ++c) s.h_tuple_g[(size_t)t * kMaxComponents + c] = key[c] == kNoGroup ? (uint16_t)e->Gtot : key[c];
            s.h_tuple_p[t] = s.h_pid[n];
        }
        s.h_tid[n] = (uint8_t)t;
        s.h_toff[n] = (uint32_t)t * (uint32_t)(e->S + 1) * 512u;
        ++s.tup_cnt[t];
    }
    s.n_tuples = ok ? n_tup : 0;
    s.inc_ok = ok;                                    // (derive_patterns ran just before: both count tables are current)
}

// The same tables after a few objects changed their component-0 group (a cluster move), in O(moved): `s` holds the
// OLD tables and counts and already the NEW ids in h_gid; `moved` lists the objects, `old_gid0` their previous ids.
// Returns false when the update needs the full derivation (the SET of patterns changes, or no tuple index is free);
// `s` is then only partly updated and the caller re-derives everything.  Tuple numbering is history-dependent (a
// vacated index is reused by the next new tuple); the kernels only look tuples up, so results do not depend on it.
bool update_patterns_and_tuples(sbe_engine* e, Slot& s, const int32_t* moved, const uint16_t* old_gid0, int n_moved) {
    const int N = e->N, C = e->C;
    if (!s.inc_ok || s.n_tuples == 0 || (int)s.pat_cnt.size() != 256 || (int)s.tup_cnt.size() != kMaxTuples) return false;
    auto bits_rest = [&](int n) { uint32_t b = 0; for (int c = 1; c < C; ++c) if (s.h_gid[(size_t)c * N + n] != kNoGroup) b |= 1u << c; return b; };
    // pass 1: the set of patterns must stay what it is (ranks of the other patterns would shift otherwise)
    for (int i = 0; i < n_moved; ++i) {
        const int n = moved[i];
        const uint32_t rest = bits_rest(n);
        const uint32_t b0 = rest | (old_gid0[i] != kNoGroup ? 1u : 0u), b1 = rest | (s.h_gid[n] != kNoGroup ? 1u : 0u);
        if (b0 == b1) continue;
        --s.pat_cnt[b0]; ++s.pat_cnt[b1];
    }
    {
        size_t live = 0;
        for (uint32_t b = 0; b < 256; ++b) if (s.pat_cnt[b] > 0) ++live;
        bool same = live == s.patterns.size();
        for (size_t i = 0; same && i < s.patterns.size(); ++i) same = s.pat_cnt[s.patterns[i]] > 0;
        if (!same) return false;
    }
    uint8_t rank_of[256];
    for (size_t i = 0; i < s.patterns.size(); ++i) rank_of[s.patterns[i]] = (uint8_t)i;
    // pass 2: pattern id and tuple of every moved object
    for (int i = 0; i < n_moved; ++i) {
        const int n = moved[i];
        const uint32_t b1 = bits_rest(n) | (s.h_gid[n] != kNoGroup ? 1u : 0u);
        s.h_pid[n] = rank_of[b1];
        uint16_t key[kMaxComponents];
        for (int c = 0; c < C; ++c) { const uint16_t g = s.h_gid[(size_t)c * N + n]; key[c] = g == kNoGroup ? (uint16_t)e->Gtot : g; }
        const int t0 = s.h_tid[n];
        int t1 = -1, free_t = -1;
        for (int t = 0; t < s.n_tuples; ++t) {
            if (s.tup_cnt[t] == 0) { if (free_t < 0 && t != t0) free_t = t; continue; }
            if (memcmp(&s.h_tuple_g[(size_t)t * kMaxComponents], key, (size_t)C * sizeof(uint16_t)) == 0) { t1 = t; break; }
        }
        if (t1 < 0) {                                     // a tuple no object had: a vacated index, else a new one
            if (s.tup_cnt[t0] == 1) t1 = t0;              // (the object was alone in its tuple: the index moves with it)
            else if (free_t >= 0) t1 = free_t;
            else if (s.n_tuples < kMaxTuples) t1 = s.n_tuples++;
            else return false;
            for (int c = 0; c < C; ++c) s.h_tuple_g[(size_t)t1 * kMaxComponents + c] = key[c];
        }
        if (t1 != t0) {
            if (--s.tup_cnt[t0] == 0) s.h_tuple_p[t0] = 0xFF;          // no object left: "not present", like the full derivation
            ++s.tup_cnt[t1];
        }
        s.h_tuple_p[t1] = s.h_pid[n];
        s.h_tid[n] = (uint8_t)t1;
        s.h_toff[n] = (uint32_t)t1 * (uint32_t)(e->S + 1) * 512u;
    }
    // the table must stay DENSE: the number of tuples decides which fused kernel evaluates the slot and with which
    // geometry, and that must not depend on the slot's history (found by tools/fuzz_gpu.py: a vacated index inside the
    // table made the two step forms pick different kernels at N = 18).  Vacated indices at the end are dropped; one in
    // the middle sends the step to the full derivation.
    while (s.n_tuples > 0 && s.tup_cnt[s.n_tuples - 1] == 0) --s.n_tuples;
    for (int t = 0; t < s.n_tuples; ++t)
        if (s.tup_cnt[t] == 0) return false;
    return s.n_tuples > 0;
}

// The slot's pending state to the device: new group ids (gid_pending), the pattern / tuple tables derived from them and
// the per-pattern normalised weights -- ONE launch when they fit the mapped ring (k_scatter_weight_patterns).  `eager`
// (sbe_set_groups: the call comes straight from the setter): more patterns than the engine holds is not an error yet --
// another component's ids may still follow -- the ids go up alone and the next consumer reports it.
int upload_patterns_and_weights(sbe_engine* e, int slot, const float* new_weights = nullptr, bool eager = false) {
    Slot& s = e->slots[slot];
    float* d_w = e->d_weights + (int64_t)slot * e->F * e->C;
    auto gid_seg = [&](int c) { return UploadSeg{e->d_gid + ((int64_t)slot * e->C + c) * e->Np, s.h_gid.data() + (size_t)c * e->N, (size_t)e->N * sizeof(uint16_t)}; };
    bool patterns_done = false;
    if (s.patterns_dirty) {
        const bool follow = s.tables_follow;
        s.tables_follow = false;
        if (!follow) derive_patterns(e, s);
        if ((int)s.patterns.size() > e->Pmax) {
            if (eager) {
                for (int c = 0; c < e->C; ++c)
                    if (s.gid_pending >> c & 1u) { const UploadSeg g = gid_seg(c); int rc = upload(e, g.dst, g.src, g.bytes); if (rc) return rc; }
                s.gid_pending = 0;
                return SBE_OK;
            }
            return fail(e, SBE_ERR_ARG, "%zu distinct has_components patterns exceed capacity %d",
                        s.patterns.size(), e->Pmax);
        }
        if (!follow) derive_tuples(e, s);
        UploadSeg segs[6 + kMaxComponents] = {{e->d_pid + (int64_t)slot * e->Np, s.h_pid.data(), (size_t)e->N},
                                              {e->d_patbits + (int64_t)slot * e->Pmax, s.patterns.data(), s.patterns.size() * sizeof(uint32_t)}};
        int n_segs = 2;
        if (s.n_tuples) {
            segs[n_segs++] = {e->d_tid + (int64_t)slot * e->Np, s.h_tid.data(), (size_t)e->Np};
            segs[n_segs++] = {e->d_toff + (int64_t)slot * e->Np, s.h_toff.data(), (size_t)e->Np * sizeof(uint32_t)};
            segs[n_segs++] = {e->d_tuple_g + (int64_t)slot * kMaxTuples * kMaxComponents, s.h_tuple_g.data(), s.h_tuple_g.size() * sizeof(uint16_t)};
            segs[n_segs++] = {e->d_tuple_p + (int64_t)slot * kMaxTuples, s.h_tuple_p.data(), s.h_tuple_p.size()};
        }
        for (int c = 0; c < e->C; ++c) if (s.gid_pending >> c & 1u) segs[n_segs++] = gid_seg(c);
        const int P = (int)s.patterns.size();
        FusedWeightPatterns wp{};
        const bool want_wp = (s.weights_set || new_weights) && P > 0;
        if (want_wp) {
            wp.args = WeightPatternArgs{d_w, nullptr, e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, new_weights ? d_w : nullptr,
                                        e->d_wpat_t + (int64_t)slot * e->wpat_t_elems(), P, e->F, e->C, e->Pmax, e->ft};
            wp.bits_seg = 1;
            wp.new_weights = new_weights;
        }
        { int _urc = upload_segments(e, segs, n_segs, want_wp ? &wp : nullptr); if (_urc) return _urc; }
        patterns_done = wp.done;
        s.patterns_dirty = false;
        s.gid_pending = 0;
    }
    const int P = (int)s.patterns.size();
    if (new_weights && P == 0) {                   // nothing to normalise for: just keep the weights
        int rc = upload(e, d_w, new_weights, (size_t)e->F * e->C * sizeof(float));
        if (rc) return rc;
    }
    if ((s.weights_set || new_weights) && P > 0 && !patterns_done) {
        // one launch: per-pattern normalised weights, their tile-transposed copy and -- sbe_set_weights -- the slot's
        // resident copy of the new weights, read out of the mapped staging ring
        const void* w_in = d_w;
        if (new_weights) { int rc = stage(e, new_weights, (size_t)e->F * e->C * sizeof(float), d_w, &w_in); if (rc) return rc; }
        k_weight_patterns<<<div_up((int64_t)P * e->F, 256), 256, 0, e->stream>>>(
            (const float*)w_in, e->d_patbits + (int64_t)slot * e->Pmax, e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, P, e->F, e->C,
            w_in != d_w ? d_w : nullptr, e->d_wpat_t + (int64_t)slot * e->wpat_t_elems(), e->Pmax, e->ft);
        HIPCHK(e, hipGetLastError());
    }
    return SBE_OK;
}

// refresh the tile-transposed copy of one component's probability tables (v2 fused kernel)
int retile_probs(sbe_engine* e, int slot, int component) {
    const int g_lo = e->goff[component], g_hi = g_lo + e->G[component];
    const int64_t n = (int64_t)(g_hi - g_lo) * e->S * e->ft * e->n_ftiles;
    if (n == 0) return SBE_OK;                     // component without groups
    k_tile_probs<<<div_up(n, 256), 256, 0, e->stream>>>(
        e->d_probs + (int64_t)slot * e->table_elems(), e->d_probs_t + (int64_t)slot * e->probs_t_elems(),
        g_lo, g_hi, e->Gtot, e->F, e->S, e->ft, e->n_ftiles);
    HIPCHK(e, hipGetLastError());
    return SBE_OK;
}

// ---- geometry + launch of the fused kernel ---------------------------------------------------
struct MixGeom {
    int ft, ft_shift, n_ftiles, objs_per_chunk, n_chunks, n_blocks;
    size_t lds_bytes;
};

// v2 geometry: chunks of object quads; one wave step = 64/ft quads.  The chunk's ids are staged
// in LDS (8*C + 4 bytes per quad), which caps the chunk length.
MixGeom mix_geometry_v2(const sbe_engine* e, int P, int n_batch, int blocks_per_cu = 4) {
    MixGeom g{};
    g.ft = e->ft;
    g.n_ftiles = e->n_ftiles;
    // no more workgroups than the CUs hold at once when the tile image is large: every workgroup stages the whole
    // image, so extra generations only multiply the staging traffic (stress shape, single eval: 15.9 -> 13 us)
    if (!e->direct) {
        const size_t image = (size_t)e->tile_tab_elems() * sizeof(float) + (size_t)P * e->C * e->ft * sizeof(double);
        blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)blocks_per_cu, (160 * 1024) / (image + 4096)));
    }
    if (const char* env = getenv("SBE_BLOCKS_PER_CU")) { if (atoi(env) > 0) blocks_per_cu = atoi(env); }   // experiments
    const int64_t target_blocks = (int64_t)blocks_per_cu * e->compute_units;
    int64_t chunks = std::max<int64_t>(1, target_blocks / ((int64_t)g.n_ftiles * std::max(1, n_batch)));
    const int min_quads = 4 * (kWave / e->ft);            // one step for each of the 4 waves
    const int max_quads = std::max(min_quads, (8 * 1024) / (8 * e->C + 4));
    g.objs_per_chunk = std::min<int>(max_quads, std::max<int>(min_quads, div_up(e->NQ, chunks)));   // in quads
    g.n_chunks = div_up(e->NQ, g.objs_per_chunk);
    g.n_blocks = g.n_chunks * g.n_ftiles;
    g.lds_bytes = (size_t)g.objs_per_chunk * (8 * e->C + 4);
    if (!e->direct) g.lds_bytes += (size_t)e->tile_tab_elems() * sizeof(float) + (size_t)P * e->C * e->ft * sizeof(double);
    return g;
}

int max_patterns(sbe_engine* e, int first_slot, int n) {
    int P = 1;
    for (int s = first_slot; s < first_slot + n; ++s) P = std::max<int>(P, (int)e->slots[s].patterns.size());
    return P;
}

int check_slot_ready(sbe_engine* e, int slot, bool need_weights) {
    Slot& s = e->slots[slot];
    if (!s.groups_set) return fail(e, SBE_ERR_STATE, "slot %d: groups not set for every component", slot);
    for (int c = 0; c < e->C; ++c)
        if (!s.probs_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: probability tables of component %d not set", slot, c);
    if (need_weights && !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: weights not set", slot);
    return SBE_OK;
}

// ---- matrix-pipe form of the group-tuple kernel (sbe_mixture_mfma.hip) ---------------------------------------------
// The one-hot block in MFMA fragment order, built once, at the first launch that wants it: [NT + 1][KBp] fragments of
// 1 KB (tile NT and the PF fragments behind it are zero: what the kernel reads instead of branching on bounds).
int ensure_xt(sbe_engine* e) {
    if (e->d_xt) return SBE_OK;
    const int NT = div_up((int64_t)e->F * e->S, 32), KBp = round_up(div_up(e->N, tuple_mfma_kblock_objects()), 4);
    const size_t bytes = ((size_t)(NT + 1) * KBp + 4) * 1024;
    // built into locals and published only when every step has succeeded: a caller that catches the first
    // error and retries gets the same error again, never a launch with half of this in place
    uint8_t* xt = nullptr;
    double2* logtab = nullptr;
    auto build = [&]() -> int {
        HIPCHK(e, hipMalloc((void**)&xt, bytes));
        HIPCHK(e, hipMemsetAsync(xt, 0, bytes, e->stream));
        std::vector<double> tab(2 * 1024);               // the kernel's own log table (tab_log4_n)
        fine_log_table(tab.data());
        // (the per-column object counts of the exponent bias sit behind the table in the same allocation)
        HIPCHK(e, hipMalloc((void**)&logtab, tab.size() * sizeof(double) + column_tables_bytes(NT)));
        HIPCHK(e, hipMemcpy(logtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        launch_column_tables(e->d_state, reinterpret_cast<int32_t*>(logtab + 1024), e->N, e->F, e->S, e->Fp, NT, e->stream);
        HIPCHK(e, hipGetLastError());
        launch_xt_frags(e->d_state, xt, e->N, e->F, e->S, e->Fp, NT, KBp, tuple_mfma_fp4(), e->stream);
        HIPCHK(e, hipGetLastError());
        return SBE_OK;
    };
    const int rc = build();
    if (rc) {
        (void)hipStreamSynchronize(e->stream);
        if (xt) (void)hipFree(xt);
        if (logtab) (void)hipFree(logtab);
        return rc;
    }
    e->d_xt = xt; e->d_logtab_fine = logtab;
    e->hbm_bytes += (int64_t)bytes;
    e->xt_NT = NT; e->xt_KBp = KBp; e->xt_bytes = bytes;
    return SBE_OK;
}

// geometry of a matrix-pipe launch over n slots with at most KT tuples each; n_split = 0: the form does not apply
struct MfmaGeom { int n_split, nt_per_split, MT, SL; size_t lds; bool ws; };      // SL: slots per block (16 / 4 / 2)
// the wave-specialised kernel (sbe_mixture_mfma_ws.hip): OPT-IN (SBE_MFMA_WS=1).  Measured on hardware it loses to the
// unspecialised kernel at the headline shape (72.0 against 64.5 us per 4096 states, profiles/r6/ws_experiment.log); kept, tested
// (tests/test_gpu_shapes.py::test_mfma_wave_specialised_form) and selectable for same-box comparisons.
static bool mfma_ws_wanted() {
    static const bool on = [] { const char* v = getenv("SBE_MFMA_WS"); return v && atoi(v) == 1; }();
    return on;
}
MfmaGeom mfma_geometry(const sbe_engine* e, int n, int KT) {
    MfmaGeom g{};
    g.SL = tuple_mfma_slots_per_block(KT);           // 16 slots x <= 8 tuples, 4 x <= 32, 2 x <= 64 per block: the most KT allows
    if (g.SL == 0 || e->C > 4) return g;
    const int NT = div_up((int64_t)e->F * e->S, 32), KBp = round_up(div_up(e->N, tuple_mfma_kblock_objects()), 4);
    // Few states per launch: with 16 slots per block a launch of n states is ceil(n / 16) x (at most 4 column splits) blocks -- at
    // 512 states half the CUs idle and every block still walks a whole pass of MT M tiles (24.6 us at 256..1024 headline states).
    // Four slots per block (ONE M tile of <= 8 tuples x 4 slots) make four times the blocks of a third of the work per pass.  The
    // two geometries are compared by rounds of blocks x passes per block x M tiles (a pass costs ~4.3 us per M tile at the headline
    // shape): 14.6 / 16.1 / 17.9 / 21.4 us at 32 / 256 / 320 / 512 states (1 / 1 / 2 / 2 units against 3) where k_mixture_tuple64
    // takes 16.0 / 21.1 / 26.5 / 35.1 us; at 576 states the four-slot form needs 4 units (30.6 us) and the 16-slot form stays
    // (24.6 us): profiles/r6/mfma_threshold.log
    if (g.SL == 16 && tuple_mfma_fp4() && e->mfma_small_sl4) {
        auto units = [&](int SL, int MT) -> int64_t {
            const int groups = div_up(n, SL);
            const int split = std::max(1, std::min(div_up(NT, 16), e->compute_units / std::max(1, groups)));
            const int passes = div_up(round_up(div_up(NT, split), 2), 16);
            return (int64_t)div_up(groups * split, e->compute_units) * passes * MT;
        };
        if (units(4, 1) < units(16, div_up(KT, 2))) g.SL = 4;
    }
    if (const char* env = getenv("SBE_MFMA_SL")) { const int v = atoi(env); if ((v == 4 || v == 2) && v < g.SL && tuple_mfma_fp4()) g.SL = v; }   // experiments
    // ... and fewer slots per block (FP4 operands) while the A image -- MT x KBp KB, MT = tuples x slots / 32 -- does not fit a
    // CU's LDS: many objects with few tuples (5000 objects x 6 tuples: 16 slots need 3 x 80 KB, 4 slots 1 x 80 KB)
    for (;;) {
        g.MT = div_up(KT, 32 / g.SL);
        g.lds = tuple_mfma_lds_bytes(g.MT, e->C, KBp);
        if (g.lds <= 160 * 1024 || g.SL == 2 || !tuple_mfma_fp4()) break;
        g.SL = g.SL == 16 ? 4 : 2;
    }
    if (g.SL == 16 && tuple_mfma_fp4() && mfma_ws_wanted() && tuple_mfma_ws_lds_bytes(g.MT, e->C, KBp) <= 160 * 1024) {
        g.ws = true;
        g.lds = tuple_mfma_ws_lds_bytes(g.MT, e->C, KBp);
    }
    if (g.lds > 160 * 1024) return g;
    // the tables are addressed through 32-bit buffer offsets
    const int64_t probs_bytes = ((int64_t)e->n_slots * e->table_elems() + (int64_t)e->F * e->S) * 4;
    const int64_t wpat_bytes = ((int64_t)e->n_slots * e->Pmax * e->F * e->C + (int64_t)e->F * e->C) * 4;
    if (probs_bytes >= ((int64_t)1 << 32) || wpat_bytes >= ((int64_t)1 << 32) || ((int64_t)(NT + 1) * KBp + 4) * 1024 >= ((int64_t)1 << 31)) return g;
    // one block = SL slots x a range of column tiles; its 8 waves take the tiles in pairs, so a split of fewer than
    // 16 tiles leaves waves idle: as many splits as fill the CUs, no finer
    const int groups = div_up(n, g.SL);
    int n_split = std::max(1, std::min(div_up(NT, 16), e->compute_units / std::max(1, groups)));
    if (const char* env = getenv("SBE_MFMA_SPLIT")) { if (atoi(env) > 0) n_split = std::min(atoi(env), NT); }   // experiments
    g.nt_per_split = round_up(div_up(NT, n_split), 2);
    // (the kernel sums count * binary exponent in 32-bit integers, one accumulator per lane and slot: in a pass of 16 tiles a
    //  lane adds the entries of its kTupleMfmaColsPerPass columns -- over every M tile and both tuples of a tile -- into the
    //  same accumulator, and a slot's counts of ONE column add up to at most N over its tuples; the exponents are summed
    //  BIASED (0 .. 2046; the bias leaves as 1023 x the columns' object count at the end), 2100 covers every double)
    if ((int64_t)div_up(g.nt_per_split, 16) * kTupleMfmaColsPerPass * e->N * 2100 >= ((int64_t)1 << 31)) return g;
    g.n_split = div_up(NT, g.nt_per_split);
    return g;
}

int launch_mfma_form(sbe_engine* e, int first_slot, int n, int KT, const MfmaGeom& mg, const int32_t* d_slots,
                     bool reduce_in_kernel = false, const DoneSig& done = DoneSig{}) {
    int rc = ensure_xt(e);
    if (rc) return rc;
    MfmaMixParams p{};
    p.F = e->F; p.S = e->S; p.FS = e->F * e->S; p.Gtot = e->Gtot; p.Np = e->Np;
    p.NT = e->xt_NT; p.KBp = e->xt_KBp; p.KT = KT; p.SL = mg.SL;
    p.n_batch = n; p.n_split = mg.n_split; p.nt_per_split = mg.nt_per_split;
    p.first_slot = first_slot; p.slot_list = d_slots;
    p.xt = e->d_xt; p.xt_bytes = (uint32_t)e->xt_bytes;
    p.tid = e->d_tid; p.tid_stride = e->Np;
    p.tuple_g = e->d_tuple_g; p.tuple_g_stride = (int64_t)kMaxTuples * kMaxComponents;
    p.tuple_p = e->d_tuple_p; p.tuple_p_stride = kMaxTuples;
    p.probs = e->d_probs; p.probs_stride = e->table_elems();
    p.probs_ones_off = (uint32_t)((int64_t)e->n_slots * e->table_elems() * 4);
    p.probs_bytes = p.probs_ones_off + (uint32_t)(e->F * e->S * 4);
    p.wpat = e->d_wpat; p.wpat_stride = (int64_t)e->Pmax * e->F * e->C;
    p.wpat_ones_off = (uint32_t)((int64_t)e->n_slots * e->Pmax * e->F * e->C * 4);
    p.wpat_bytes = p.wpat_ones_off + (uint32_t)(e->F * e->C * 4);
    p.logtab = e->d_logtab_fine;
    p.colcount = reinterpret_cast<const int32_t*>(e->d_logtab_fine + 1024);
    p.colfeat = p.colcount + (size_t)(e->xt_NT + 1) * 32;
    p.tile_prefix = p.colcount + (size_t)(e->xt_NT + 1) * 64;
    p.partials = e->d_partials; p.partials_stride = e->partials_stride;
    if (reduce_in_kernel) { p.results = e->d_results; p.arrive = e->d_arrive; p.done = done; }
    const dim3 grid((unsigned)(div_up(n, mg.SL) * mg.n_split));
    if (!(mg.ws ? launch_tuple_mfma_ws(e->C, p, grid, mg.lds, e->stream) : launch_tuple_mfma(e->C, p, grid, mg.lds, e->stream)))
        return fail(e, SBE_ERR_STATE, "k_mixture_tuple_mfma was built with static LDS: its log table must sit at LDS address 0 "
                                      "(toolchain change; rebuild without static __shared__ in sbe_mixture_mfma.hip)");
    return SBE_OK;
}

// sbe_kernel_timing: the next event pair of the pool while recording is on (one pair per fused-kernel launch, on the engine's
// own stream), else two nulls
int next_timing_events(sbe_engine* e, hipEvent_t* ev_a, hipEvent_t* ev_b) {
    *ev_a = *ev_b = nullptr;
    if (!e->ev_timing) return SBE_OK;
    while ((int)e->ev_pool.size() < 2 * (e->ev_used + 1)) {
        hipEvent_t ev;
        HIPCHK(e, hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    *ev_a = e->ev_pool[2 * e->ev_used]; *ev_b = e->ev_pool[2 * e->ev_used + 1];
    ++e->ev_used;
    return SBE_OK;
}

// Enqueue the dominant kernel (optionally bracketed by an event pair) and the fixed-order
// partial reduction.  mode: LOG_PER_OBS / LOG_PRODUCT.
// Slots: first_slot .. first_slot+n-1, or (batched steps) the n slots listed in `slots` (host) / `d_slots` (the same
// list, device-visible); then `d_fins` holds one step epilogue per listed slot.
int launch_mixture(sbe_engine* e, int first_slot, int n, int mode, hipEvent_t ev_a, hipEvent_t ev_b,
                   const StepFinish* fin = nullptr, const int32_t* slots = nullptr, const int32_t* d_slots = nullptr,
                   const StepFinish* d_fins = nullptr, DoneSig* done_out = nullptr) {
    auto slot_at = [&](int i) { return slots ? (int)slots[i] : first_slot + i; };
    int P = 1;
    for (int i = 0; i < n; ++i) P = std::max<int>(P, (int)e->slots[slot_at(i)].patterns.size());
    const bool onehot = e->opt_kernel == SBE_MIXTURE_ONEHOT || e->opt_kernel == SBE_MIXTURE_ONEHOT_GENERAL;
    MixGeom g = mix_geometry_v2(e, P, n);
    if (!g.ft) return fail(e, SBE_ERR_ARG, "probability tables too large for LDS staging (G_total=%d, S=%d)", e->Gtot, e->S);
    // group-tuple form: eligible when every slot of the launch has few distinct tuples, the log table fits
    // LDS and a block sees enough observations to amortise building it.  It prefers long chunks (one block
    // per CU is enough: the table build is per block), so it gets its own geometry.
    int KT = 0;
    const bool force_mfma = e->opt_kernel == SBE_MIXTURE_PACKED_TUPLE_MFMA;
    const bool force_combo = e->opt_kernel == SBE_MIXTURE_PACKED_TUPLE || e->opt_kernel == SBE_MIXTURE_PACKED_TUPLE_LDS;
    bool combo = e->opt_kernel == SBE_MIXTURE_PACKED || e->opt_kernel == SBE_MIXTURE_ONEHOT || force_combo || force_mfma;
    for (int i = 0; i < n && combo; ++i) {
        const int sl = slot_at(i);
        if (e->slots[sl].n_tuples == 0) combo = false;
        KT = std::max(KT, e->slots[sl].n_tuples);
    }
    // large batches: the per-observation gather as an integer contraction on the matrix pipe (k_mixture_tuple_mfma)
    MfmaGeom mg{};
    // (default: from mfma_min_batch states per launch on whatever the shape, and -- FP4 operands, four slots per block -- from 32
    //  states on when the launch holds enough observations for the kernel's fixed costs: 32 headline states = 6.4 M)
    const bool mfma_default = e->opt_kernel == SBE_MIXTURE_PACKED &&
                              (n >= e->mfma_min_batch || (tuple_mfma_fp4() && e->mfma_small_sl4 && n >= 32 && (int64_t)n * e->N * e->F >= e->mfma_min_obs));
    if (combo && (force_mfma || mfma_default)) {
        mg = mfma_geometry(e, n, KT);
        // The wide forms (more than 8 tuples: 4 / 2 slots per block) pay one log per (padded tuple, feature, state) where the
        // vector-pipe form pays one gather per observation: by default only where a table entry is shared by enough objects
        // (SBE_MFMA_WIDE_MIN_SHARE, objects per padded tuple; measured crossover: profiles/r6/wide_forms.log)
        if (!force_mfma && mg.n_split > 0 && KT > 8 && e->N < e->mfma_wide_min_share * mg.MT * (32 / mg.SL)) mg = MfmaGeom{};
    }
    const bool mfma = mg.n_split > 0;
    if (force_mfma && !mfma)
        return fail(e, SBE_ERR_ARG, "matrix-pipe group-tuple kernel forced but not applicable (tuples=%d, C=%d, LDS %zu bytes)", KT, e->C, mg.lds);
    if (mfma) combo = false;
    size_t combo_lds = 0;
    int combo_w_off = 0, combo_tab_off = 0;
    bool tuple64 = false;
    if (combo) {
        const MixGeom gc = mix_geometry_v2(e, P, n, 2);
        // 64-feature tiles, packed stream: the scalar-unit form (tuple metadata in VGPRs, no id staging)
        tuple64 = !onehot && gc.ft == 64 && e->d_state_h && e->opt_kernel != SBE_MIXTURE_PACKED_TUPLE_LDS;
        // LDS image: T[KT][S+1][ft] f64 | tq[quads] u32 | tuple rows u16 | tuple patterns u32 | weights f64 [| byte table]
        //   (tuple64: T | weights)
        const int cu = e->C <= 4 ? e->C : kMaxComponents;
        combo_lds = (size_t)KT * (e->S + 1) * gc.ft * sizeof(double);
        if (!tuple64) {
            combo_lds += (size_t)gc.objs_per_chunk * 4;
            combo_lds += ((size_t)KT * cu + ((KT * cu) & 1)) * sizeof(uint16_t) + (size_t)KT * sizeof(uint32_t);
        }
        combo_lds = (combo_lds + 15) / 16 * 16;
        combo_w_off = (int)combo_lds;
        combo_lds += (size_t)P * e->C * gc.ft * sizeof(double);
        if (tuple64) combo_lds += tuple64_waves() * sizeof(double) + kLogTabEntries * sizeof(double2);   // reduction scratch (the kernel has no static LDS) + log table
        if (onehot) {      // byte-position lookup table [seg16][32] u16; a tile row segment must fit one step
            const int seg16 = gc.ft * e->S / 16;
            if (seg16 > kBlock) combo = false;
            combo_tab_off = (int)combo_lds;
            combo_lds += (size_t)seg16 * 32 * sizeof(uint16_t);
        }
        const int64_t obs_per_block = (int64_t)gc.objs_per_chunk * 4 * gc.ft;
        if (combo && !force_combo && (combo_lds > 40 * 1024 || obs_per_block < (int64_t)3 * KT * e->S * gc.ft)) combo = false;
        if (force_combo && (!combo || combo_lds > 150 * 1024))
            return fail(e, SBE_ERR_ARG, "group-tuple kernel forced but not applicable (tuples=%d, LDS %zu bytes)", KT, combo_lds);
        if (combo) g = gc;
    } else if (force_combo) {
        return fail(e, SBE_ERR_ARG, "group-tuple kernel forced but not applicable (tuples=%d, LDS %zu bytes)", KT, combo_lds);
    }
    // rows form (k_mixture_rows): the general packed kernel whenever its LDS image fits -- 1024-thread blocks over
    // 32-feature (or 16-feature) tiles; SBE_MIXTURE_PACKED_V2 keeps the older k_mixture_v2 (A/B, tests)
    bool rows = !mfma && !combo && !onehot && e->rows_ft != 0 && e->opt_kernel != SBE_MIXTURE_PACKED_V2;
    const size_t rows_image = rows ? (size_t)(e->Gtot + 1) * (e->S + 1) * e->rows_ft * 4 + (size_t)P * ((e->C + 1) / 2) * e->rows_ft * 16
                                         + (size_t)kRowsWaves * (kWave / e->rows_ft) * (e->C + 1) * 16 : 0;      // tables | weights | offset slots
    if (rows && rows_image > 160 * 1024 - 512) rows = false;      // more patterns than the tile width was sized for
    // pattern-sorted objects (weights in registers): 32-feature tiles, a second offsets slot per wave in LDS, state-row
    // offsets of 24 bits
    const size_t sorted_image = rows_image + (size_t)kRowsWaves * (kWave / std::max(1, e->rows_ft)) * (e->C + 1) * 16;
    const bool sorted = rows && (e->opt_rows_sorted == 2 || (e->opt_rows_sorted == 1 && n >= 16)) && e->rows_ft == 32 && sorted_image <= 160 * 1024 - 512 &&
                        (int64_t)(e->N + 1) * e->Fq < ((int64_t)1 << 24) && e->Pmax <= 64;
    // a single eval with a large image (stress shape: 153 KB per block) is staging-bound in the rows form (measured
    // 13.8 us against 12.5 us for k_mixture_v2's many small blocks); from two evals per launch on the rows form wins
    if (rows && n == 1 && rows_image > 72 * 1024 && e->opt_kernel != SBE_MIXTURE_PACKED_GENERAL) rows = false;
    // The rows form needs long object ranges (a 1024-thread block covers 32 quads per step) and enough observations
    // per launch to fill one block per CU; below that k_mixture_v2's 256-thread blocks win.  Thresholds from
    // tools/rows_crossover.py on an MI355X (kernel time of both forms over N = 500..5000, B = 8..256, C = 2 / 4,
    // and the stress shape itself): they depend on the tile width k_mixture_v2 would run at (64: efficient, 16: not).
    if (rows && e->opt_kernel != SBE_MIXTURE_PACKED_GENERAL) {
        const int64_t obs = (int64_t)n * e->N * e->F;
        const int64_t min_obs = g.ft >= 64 ? 64000000 : g.ft >= 32 ? 24000000 : 10000000;
        const int min_quads = g.ft >= 64 ? 500 : g.ft >= 32 ? 375 : 250;
        if (obs < min_obs || e->NQ < min_quads) rows = false;
    }
    if (rows) {
        const int rft = e->rows_ft, gran = kRowsWaves * (kWave / rft);         // quads per block step
        const int n_t = div_up(e->F, rft);
        const size_t image = sorted ? sorted_image : rows_image;
        if (sorted && !e->d_rowoff_s) {                                        // one-time: the sorted form's arrays
            const int step_objs = 4 * (kWave / rft);
            e->rs_nq_max = round_up(e->N + e->Pmax * (step_objs - 1), step_objs) / 4;
            int rc = dmalloc(e, &e->d_rowoff_s, (int64_t)e->n_slots * e->rs_nq_max * (e->C + 1) * 4); if (rc) return rc;
            rc = dmalloc(e, &e->d_rs_nq, e->n_slots); if (rc) return rc;
            rc = dmalloc(e, &e->d_state_s, (int64_t)(e->N + 1) * e->Fq); if (rc) return rc;
            launch_state_s(e->d_state, e->d_state_s, e->N, e->F, e->Fp, e->Fq, e->S, e->stream);
            HIPCHK(e, hipGetLastError());
            e->rowsort_epoch.assign(e->n_slots, ~0ull);
        }
        const int NQ_geo = sorted ? e->rs_nq_max : e->NQ;                      // (sorted: the longest padded order a slot can have)
        // every block stages the whole image: with a large image one block per CU and as few object chunks as fill
        // the chip; small images take two generations of blocks
        // (a block that stages a large image wants at least ~8 block steps of work behind it)
        const int64_t target = (int64_t)e->compute_units * (image > 72 * 1024 ? 1 : 2);
        int min_steps = image > 72 * 1024 ? 8 : image > 24 * 1024 ? 4 : 1;
        if (const char* env = getenv("SBE_ROWS_MIN_STEPS")) { if (atoi(env) > 0) min_steps = atoi(env); }   // experiments
        int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(div_up(NQ_geo, (int64_t)gran * min_steps), div_up(target, (int64_t)n_t * n)));
        const int qpc = round_up(div_up(NQ_geo, chunks), gran);
        g.ft = rft; g.n_ftiles = n_t; g.objs_per_chunk = qpc; g.n_chunks = div_up(NQ_geo, qpc);
        g.n_blocks = g.n_chunks * n_t; g.lds_bytes = image;
        // per-object row offsets of the slots whose group ids changed since their offsets were built
        bool stale = false;
        if (sorted) {
            for (int i = 0; i < n; ++i) stale |= e->rowsort_epoch[slot_at(i)] != e->slots[slot_at(i)].group_epoch;
            if (stale) {
                launch_rowsort(e->d_gid, e->d_pid, e->d_rowoff_s, e->d_rs_nq, (int64_t)e->C * e->Np, e->Np,
                               (int64_t)e->rs_nq_max * (e->C + 1) * 4, first_slot, d_slots, n, e->N, e->Np, e->C, e->Gtot, e->Pmax,
                               (uint32_t)((e->S + 1) * rft * 4), (uint32_t)e->Fq, 4 * (kWave / rft), e->stream);
                HIPCHK(e, hipGetLastError());
                for (int i = 0; i < n; ++i) e->rowsort_epoch[slot_at(i)] = e->slots[slot_at(i)].group_epoch;
            }
            stale = false;
        } else
        for (int i = 0; i < n; ++i) stale |= e->rowoff_epoch[slot_at(i)] != e->slots[slot_at(i)].group_epoch;
        if (stale) {
            const int cells = (e->C + 1) * e->Np;
            k_rowoff<<<dim3(div_up(cells, 256), n), 256, 0, e->stream>>>(
                e->d_gid, e->d_pid, e->d_rowoff, (int64_t)e->C * e->Np, e->Np, (int64_t)cells, first_slot, d_slots, e->N, e->Np,
                e->C, e->Gtot, (uint32_t)((e->S + 1) * rft * 4), (uint32_t)(((e->C + 1) / 2) * rft * 16));
            HIPCHK(e, hipGetLastError());
            for (int i = 0; i < n; ++i) e->rowoff_epoch[slot_at(i)] = e->slots[slot_at(i)].group_epoch;
        }
    }
    if (mfma) g.n_blocks = mg.n_split;                // partial sums per slot: one per column split
    if (g.n_blocks > e->partials_stride) return fail(e, SBE_ERR_STATE, "internal: partials buffer too small (%d > %lld)", g.n_blocks, (long long)e->partials_stride);
    if (!mfma && !combo && !rows && g.lds_bytes > 159 * 1024)
        return fail(e, SBE_ERR_ARG, "probability / weight tables too large for LDS staging at tile width %d (%zu bytes; G_total=%d, S=%d, P=%d)",
                    g.ft, g.lds_bytes, e->Gtot, e->S, P);
    dim3 grid(g.n_blocks, n);
    if (mfma) { int rc = ensure_xt(e); if (rc) return rc; }      // (one-time build: outside the event pair)
    if (ev_a) HIPCHK(e, hipEventRecord(ev_a, e->stream));
    // Without a step epilogue the kernel finishes the reduction itself (the last block of a slot -- of a group of 16 slots in
    // the matrix-pipe form -- adds the partial sums): one launch per eval (batch) instead of two.  SBE_REDUCE_IN_KERNEL=0
    // keeps k_reduce_partials (A/B runs).
    static const bool in_kernel_opt = !(getenv("SBE_REDUCE_IN_KERNEL") && atoi(getenv("SBE_REDUCE_IN_KERNEL")) == 0);
    // Where it pays (measured, tools/probe/single_eval_latency.py and bench.py): always in the matrix-pipe form (two to four
    // blocks per 16 slots) and when a slot is ONE block (no tickets at all); for a few blocks per slot in the asynchronous
    // calls (throughput: one launch less per eval, cfg1 138 -> 164 k evals/s).  A host-synchronous call waits for the last
    // block's store -> ticket -> loads, three dependent trips to the coherence point, which is 1.2-2 us MORE than the second
    // launch; and with hundreds of blocks per slot the tickets at one address serialise (headline, one eval: 8.1 -> 11.2 us).
    const bool in_kernel = !fin && !d_fins && in_kernel_opt && (mfma || g.n_blocks == 1 || (!done_out && g.n_blocks <= 16));
    const bool mfma_reduce = mfma && in_kernel;
    DoneSig done_k{};
    if (in_kernel) {
        done_k = done_out ? next_done(e, (unsigned)(mfma ? div_up(n, mg.SL) : n)) : DoneSig{};
        if (done_out) *done_out = done_k;
    }
    if (mfma) {
        snprintf(e->last_kernel, sizeof e->last_kernel, "k_mixture_tuple_mfma%s<packed stream, group-tuple form, matrix pipe %s, %d slots x M tiles %d, C=%d>", mg.ws ? "_ws" : "", tuple_mfma_fp4() ? "fp4" : "i8", mg.SL, mg.MT, e->C);
        int rc = launch_mfma_form(e, first_slot, n, KT, mg, d_slots, mfma_reduce, done_k);
        if (rc) return rc;
    } else {
        // XCD-aware 1-D grid (see k_mixture_v2): units = work items x slot groups, unit u on XCD u % 8
        int gcd8 = 8;
        while (g.n_blocks % gcd8) gcd8 >>= 1;
        const int slot_groups = std::max(1, std::min(8 / gcd8, n));
        const int slots_per_group = div_up(n, slot_groups);
        const int n_units = g.n_blocks * slot_groups;
        grid = dim3(8 * div_up(n_units, 8) * slots_per_group, 1);
        Mix2Params p{};
        p.N = e->N; p.NQ = e->NQ; p.Np = e->Np; p.F = e->F; p.Fq = e->Fq; p.S = e->S; p.C = e->C;
        p.Gtot = e->Gtot; p.P = P; p.n_ftiles = g.n_ftiles; p.quads_per_chunk = g.objs_per_chunk;
        p.state_q = reinterpret_cast<const uint32_t*>(e->d_state_q);
        p.onehot = e->d_onehot; p.rs_pitch = e->rs_pitch;
        p.gid = e->d_gid; p.gid_stride = (int64_t)e->C * e->Np;
        p.pid = e->d_pid; p.pid_stride = e->Np;
        p.probs_t = e->d_probs_t; p.probs_t_stride = e->probs_t_elems();
        p.wpat_t = e->d_wpat_t; p.wpat_t_stride = e->wpat_t_elems(); p.wpat_tile_stride = (int)e->wpat_tile_elems();
        p.partials = e->d_partials; p.partials_stride = e->partials_stride; p.first_slot = first_slot;
        if (in_kernel) { p.results = e->d_results; p.arrive = e->d_arrive; p.done = done_k; }
        p.slot_list = d_slots;
        p.n_work = g.n_blocks; p.n_batch = n;
        p.slot_groups = slot_groups; p.slots_per_group = slots_per_group;
        p.tid = e->d_tid; p.tid_stride = e->Np;
        p.state_h = reinterpret_cast<const uint2*>(e->d_state_h);
        p.toff = e->d_toff; p.toff_stride = e->Np;
        p.logtab = e->d_logtab;
        p.ragged_w = (tuple64 && e->F % 64 != 0 && e->F % 64 <= 32) ? e->F % 64 : 0;
        if (combo && tuple64) {   // own block order (slots dealt to XCDs, generations, heavy work items first; see the kernel)
            p.gen_slots = std::max(1, (4 * e->compute_units / 8) / g.n_blocks);
            if (const char* env = getenv("SBE_T64_GEN_SLOTS")) { if (atoi(env) > 0) p.gen_slots = atoi(env); }      // experiments: block order
            const int gens = div_up(div_up(n, 8), p.gen_slots);
            grid = n >= 8 ? dim3(8 * gens * p.gen_slots * g.n_blocks, 1) : dim3(n * g.n_blocks, 1);
        }
        p.tuple_g = e->d_tuple_g; p.tuple_g_stride = (int64_t)kMaxTuples * kMaxComponents;
        p.tuple_p = e->d_tuple_p; p.tuple_p_stride = kMaxTuples;
        p.combo_w_off = combo_w_off; p.combo_tab_off = combo_tab_off;
        p.KT = KT;
        p.eft = e->ft;
        p.wpat = e->d_wpat; p.wpat_stride = (int64_t)e->Pmax * e->F * e->C;
        p.rowoff = e->d_rowoff; p.rowoff_stride = (int64_t)(e->C + 1) * e->Np;
        p.rowoff_s = e->d_rowoff_s; p.rowoff_s_stride = (int64_t)e->rs_nq_max * (e->C + 1) * 4;
        p.rs_nq = e->d_rs_nq; p.state_s = e->d_state_s; p.state_s_pitch = e->Fq;
        {   // shares of a rows block's steps by wave age class (see k_mixture_rows); SBE_ROWS_SPLIT="a,b,c,d" per mille
            static int split[4] = {450, 270, 170, 110};
            static bool parsed = false;
            if (!parsed) {
                parsed = true;
                if (const char* env = getenv("SBE_ROWS_SPLIT")) {
                    int v[4];
                    if (sscanf(env, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] + v[1] + v[2] + v[3] == 1000 &&
                        v[0] >= 0 && v[1] >= 0 && v[2] >= 0 && v[3] >= 0)
                        for (int i = 0; i < 4; ++i) split[i] = v[i];
                }
            }
            p.rows_cum[0] = 0;
            for (int i = 0; i < 4; ++i) p.rows_cum[i + 1] = p.rows_cum[i] + split[i];
        }
        snprintf(e->last_kernel, sizeof e->last_kernel, "%s<%s%s, tile %d, C=%d>",
                 combo ? (tuple64 ? "k_mixture_tuple64" : "k_mixture_combo") : rows ? "k_mixture_rows" : (onehot ? "k_mixture_onehot_v2" : "k_mixture_v2"),
                 onehot ? "one-hot stream" : "packed stream", combo ? ", group-tuple form" : (rows && sorted ? ", pattern-sorted objects" : (e->direct ? ", direct tables" : "")), g.ft, e->C);
        // (the kernels live in their own translation unit: sbe_mixture.hip)
        if (combo && tuple64) launch_tuple64(e->C, p, grid, combo_lds, e->stream);
        else if (combo) launch_combo(onehot, g.ft, e->C, p, grid, combo_lds, e->stream);
        else if (rows) launch_rows(mode, g.ft, e->C, p, grid, g.lds_bytes, e->stream, sorted);
        else if (onehot) launch_oh2(mode, g.ft, e->C, p, grid, g.lds_bytes, e->stream, e->direct);
        else launch_v2(mode, g.ft, e->C, p, grid, g.lds_bytes, e->stream, e->direct);
    }
    if (ev_b) HIPCHK(e, hipEventRecord(ev_b, e->stream));
    HIPCHK(e, hipGetLastError());
    if (in_kernel) return SBE_OK;
    const unsigned n_red = (unsigned)(n + (d_fins ? n : (fin ? 1 : 0)));
    const DoneSig done = done_out ? next_done(e, n_red) : DoneSig{};      // (the caller waits with wait_done)
    if (done_out) *done_out = done;
    k_reduce_partials<<<n_red, kBlock, 0, e->stream>>>(e->d_partials, e->partials_stride, g.n_blocks,
                                                       e->d_results, first_slot, n, fin ? *fin : StepFinish{},
                                                       d_slots, d_fins, done);
    HIPCHK(e, hipGetLastError());
    return SBE_OK;
}

int enqueue_mixture(sbe_engine* e, int first_slot, int n, int mode, DoneSig* done_out = nullptr) {
    for (int s = first_slot; s < first_slot + n; ++s) {
        int rc = check_slot_ready(e, s, true);
        if (rc) return rc;
        if (e->slots[s].patterns_dirty) { rc = upload_patterns_and_weights(e, s); if (rc) return rc; }
    }
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    { int rc = next_timing_events(e, &ev_a, &ev_b); if (rc) return rc; }
    return launch_mixture(e, first_slot, n, mode, ev_a, ev_b, nullptr, nullptr, nullptr, nullptr, done_out);
}

int counts_launch(sbe_engine* e, int slot_a, int sign_a, int slot_b, int sign_b, const int32_t* d_objects,
                  int n_listed, int dst_slot, bool single_chunk, uint8_t* d_changed) {
    CountSide A{e->d_gid + (int64_t)slot_a * e->C * e->Np, e->d_src + (int64_t)slot_a * e->N * e->Fp, sign_a};
    CountSide B{e->d_gid + (int64_t)slot_b * e->C * e->Np, e->d_src + (int64_t)slot_b * e->N * e->Fp, sign_b};
    int32_t* counts = e->d_counts + (int64_t)dst_slot * e->table_elems();
    // feature tile: as wide as fits the LDS budget
    int ft = 32;
    auto lds_for = [&](int t) { return (size_t)e->Gtot * t * e->S * sizeof(int32_t); };
    while (ft > 4 && lds_for(ft) > 96 * 1024) ft >>= 1;
    if (lds_for(ft) > 150 * 1024) {
        k_counts_global<<<div_up((int64_t)n_listed * e->F, 256), 256, 0, e->stream>>>(
            e->d_state, A, B, d_objects, n_listed, e->Np, e->F, e->S, e->C, e->Fp, counts, d_changed);
        HIPCHK(e, hipGetLastError());
        return SBE_OK;
    }
    const int n_ftiles = div_up(e->F, ft);
    int chunks = 1;
    if (!single_chunk) chunks = std::max(1, std::min(div_up(n_listed, 32), div_up(2 * e->compute_units, n_ftiles)));
    const int opc = div_up(n_listed, chunks);
    chunks = div_up(n_listed, opc);
    k_counts<<<dim3(n_ftiles, chunks), kBlock, lds_for(ft), e->stream>>>(
        e->d_state, A, B, d_objects, n_listed, opc, e->Np, e->F, e->S, e->C, e->Fp, e->Gtot, ft, counts, d_changed);
    HIPCHK(e, hipGetLastError());
    return SBE_OK;
}

}  // namespace

// ---- helpers more than one unit of the C ABI uses ---------------------------------------------------------------------------
namespace {

// ---- groups -----------------------------------------------------------------------------------
int set_gid_common(sbe_engine* e, int slot, int component, const std::vector<uint16_t>& ids) {
    Slot& s = e->slots[slot];
    // a cluster move changes the ids of a few objects of component 0: the host pattern / tuple tables follow in
    // O(moved objects) (update_patterns_and_tuples, what the one-call steps use) instead of being derived from all N
    s.tables_follow = false;
    static const bool follow_on = [] { const char* v = getenv("SBE_FOLLOW_TABLES"); return !(v && atoi(v) == 0); }();   // (A/B: tools/ab_env.sh)
    if (follow_on && component == 0 && !s.patterns_dirty && s.inc_ok) {
        static thread_local std::vector<int32_t> moved;
        static thread_local std::vector<uint16_t> old_ids;
        moved.clear(); old_ids.clear();
        const int N = e->N;
        for (int n = 0; n < N; ++n)
            if (ids[n] != s.h_gid[n]) { moved.push_back(n); old_ids.push_back(s.h_gid[n]); }
        std::copy(ids.begin(), ids.end(), s.h_gid.begin());
        s.tables_follow = (int)moved.size() * 8 <= N &&
                          update_patterns_and_tuples(e, s, moved.data(), old_ids.data(), (int)moved.size());
    } else {
        std::copy(ids.begin(), ids.end(), s.h_gid.begin() + (size_t)component * e->N);
    }
    s.gid_pending |= 1u << component;
    s.patterns_dirty = true;
    s.group_epoch = ++e->epoch_counter;
    bump_ids(e, slot);
    s.groups_set = true;   // components never set keep "no group" ids
    // the ids, the pattern / tuple tables they imply and the per-pattern weights go up together, now: one launch, and no
    // reader of the resident ids ever sees the slot between the two
    return upload_patterns_and_weights(e, slot, nullptr, true);
}

// Resident slot state keeps ONE group id per object and component (u16).  A bool [G][N] matrix with an object in two rows has two
// readings in the reference: its a1 lets the LAST WRITTEN group win (likelihood.py:126-130; an uncached evaluation writes the
// groups in index order, so the highest group index containing the object), while compute_effect_counts counts the object once
// PER GROUP it is in (sbayes/sampling/counts.py:28-30).  Round 6: sbe_set_groups takes such a matrix -- id = the last group,
// which is what every likelihood evaluation on resident state needs (mixture, per-component and per-observation likelihoods,
// has_components patterns, the collapsed likelihood of the caller's counts) -- and marks the slot (Slot::overlap_mask); the
// calls that would derive COUNTS from the ids (recount, count deltas, the one-call steps, the Gibbs forms) refuse a marked
// slot with SBE_ERR_DATA (reject_overlap): the caller's counts, or the stateless sbe_effect_counts, are the reference's there.
// The one-call steps' cluster matrices stay strict.  (sBayes itself never produces overlap: operators.py:724-725, :1099-1101
// for clusters, load_data.py:174-178 for confounders.)
void overlap_message(char* buf, size_t len, int n, int g1, int g2, int component) {
    snprintf(buf, len, "object %d is in groups %d and %d of component %d: resident slot state keeps one group id per object "
             "and component (counts.py:28-30 would count it in both); counts of overlapping groups come from the caller or "
             "the stateless sbe_effect_counts", n, g1, g2, component);
}

struct OverlapNote { bool found = false; int obj = -1, g1 = 0, g2 = 0; };

// bool [G][N] -> one id per object (off + g, kNoGroup: in no group).  Strict form (note == nullptr): false + message on
// overlap.  Permissive form: the LAST group containing the object wins and the first overlapping object is noted.
bool matrix_to_ids(const uint8_t* groups, int G, int N, int off, int component, uint16_t* ids, char* msg, size_t msg_len,
                   OverlapNote* note = nullptr) {
    std::fill(ids, ids + N, kNoGroup);
    auto hit = [&](int n, int g) -> bool {
        if (ids[n] != kNoGroup) {
            if (!note) { overlap_message(msg, msg_len, n, (int)ids[n] - off, g, component); return false; }
            if (!note->found) { note->found = true; note->obj = n; note->g1 = (int)ids[n] - off; note->g2 = g; }
        }
        ids[n] = (uint16_t)(off + g);
        return true;
    };
    for (int g = 0; g < G; ++g) {                     // (mostly zeros: eight objects per test)
        const uint8_t* row = groups + (size_t)g * N;
        int n = 0;
        for (; n + 8 <= N; n += 8) {
            uint64_t w8;
            memcpy(&w8, row + n, 8);
            if (!w8) continue;
            for (int k = 0; k < 8; ++k) if (row[n + k] && !hit(n + k, g)) return false;
        }
        for (; n < N; ++n) if (row[n] && !hit(n, g)) return false;
    }
    return true;
}

// SBE_ERR_DATA for a call that would derive counts from the ids of a slot holding overlapping groups
int reject_overlap(sbe_engine* e, int slot, const char* what) {
    const Slot& s = e->slots[slot];
    if (!s.overlap_mask) return SBE_OK;
    char msg[320];
    overlap_message(msg, sizeof msg, s.ov_obj, s.ov_g1, s.ov_g2, s.ov_comp);
    return fail(e, SBE_ERR_DATA, "%s (slot %d, %s)", msg, slot, what);
}


// the call's final synchronisation, then the data checks its kernels may have raised (flag words: no read-back)
int sync_and_report(sbe_engine* e, const DoneSig& done = DoneSig{}) {
    { int rc = wait_done(e, done); if (rc) return rc; }          // (no flag asked for: the runtime's stream wait)
    const bool was_pending = e->status_pending;
    e->status_pending = false;
    return report_status(e, was_pending);
}

int check_objects(sbe_engine* e, const int32_t* objects, int n) {
    for (int i = 0; i < n; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    return SBE_OK;
}



// k_given_unchanged_fused: LDS image of a 16-feature tile and the arguments both forms share
constexpr size_t kGuFusedLdsMax = (size_t)96 << 10;

}  // namespace

// the engine's host worker pool (sbe_pool.h): the batched steps' chain jobs and the chunk-wise copy-out of streamed results
namespace {
int ensure_step_pool(sbe_engine* e) {
    if (!e->pool) {
        int nt = 7;                                                  // + the calling thread.  (Measured on a 16-CPU share:
        // 64 chains 747 / 416 / 304 / 246 / 225 us per sweep with 1 / 2 / 4 / 8 / 16 threads; the workers poll while sweeps
        // follow each other, so more threads than free cores is far worse than too few: 32 threads 3.5 ms.)
        if (const char* env = getenv("SBE_STEP_THREADS")) nt = std::max(0, atoi(env) - 1);
        // this process' share of the host: the CPUs it may run on, divided by the ranks of the node (one process per
        // GPU, torch.distributed.run exports LOCAL_WORLD_SIZE) -- eight ranks x eight polling threads on one host is
        // exactly the oversubscribed regime above
        int cpus = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min(cpus > 0 ? cpus : CPU_COUNT(&set), CPU_COUNT(&set));
        // ... and the container's CPU quota, which the affinity mask does not show (a GPU box of this pool: 256 CPUs
        // visible, 16 granted): cgroup v2 cpu.max = "<quota> <period>" or "max"
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                cpus = std::min<int>(cpus, (int)std::max<long long>(1, quota / period));
            fclose(f);
        }
        int local_world = 1;
        if (const char* env = getenv("LOCAL_WORLD_SIZE")) local_world = std::max(1, atoi(env));
        nt = std::min<int>(nt, std::max(0, cpus / local_world - 1));
        e->pool = new sbe_engine::Pool(nt);
    }
    return SBE_OK;
}
}  // namespace
