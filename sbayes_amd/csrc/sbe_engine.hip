// sbe_engine.hip -- host side of the MI355X sBayes likelihood engine: the C ABI declared in
// include/sbe_engine.h over the gfx950 kernels in sbe_kernels.hip.h.
//
// Plain HIP runtime (own stream, own events, own device memory); no torch, no compatibility
// layer.  One engine = one process' view of one GPU.  The one-hot feature block and every
// slot's state stay resident in HBM; only small tables / id vectors cross PCIe per call.
#include "sbe_kernels.hip.h"
#include "../../include/sbe_engine.h"
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

namespace {

thread_local std::string g_last_error;

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
    char last_kernel[96] = "none";     // kernel form of the most recent fused-kernel launch (sbe_last_mixture_kernel)

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
    uint8_t* d_xt = nullptr;       // one-hot block in MFMA fragment order (k_mixture_tuple_mfma), built at the first batched launch
    int xt_NT = 0, xt_KBp = 0;  size_t xt_bytes = 0;
    int mfma_min_batch = 512;      // smallest launch the matrix-pipe form is chosen for under SBE_MIXTURE_PACKED (SBE_MFMA_MIN_BATCH)
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
    const int NT = div_up((int64_t)e->F * e->S, 32), KBp = round_up(div_up(e->N, 32), 4);
    const size_t bytes = ((size_t)(NT + 1) * KBp + 4) * 1024;
    HIPCHK(e, hipMalloc((void**)&e->d_xt, bytes));
    e->hbm_bytes += (int64_t)bytes;
    HIPCHK(e, hipMemsetAsync(e->d_xt, 0, bytes, e->stream));
    launch_xt_frags(e->d_state, e->d_xt, e->N, e->F, e->S, e->Fp, NT, KBp, e->stream);
    HIPCHK(e, hipGetLastError());
    e->xt_NT = NT; e->xt_KBp = KBp; e->xt_bytes = bytes;
    return SBE_OK;
}

// geometry of a matrix-pipe launch over n slots with at most KT tuples each; n_split = 0: the form does not apply
struct MfmaGeom { int n_split, nt_per_split, MT; size_t lds; };
MfmaGeom mfma_geometry(const sbe_engine* e, int n, int KT) {
    MfmaGeom g{};
    if (KT < 1 || KT > 8 || e->C > 4) return g;
    const int NT = div_up((int64_t)e->F * e->S, 32), KBp = round_up(div_up(e->N, 32), 4);
    g.MT = (KT + 1) / 2;
    g.lds = tuple_mfma_lds_bytes(g.MT, e->C, KBp);
    if (g.lds > 160 * 1024) return g;
    // the tables are addressed through 32-bit buffer offsets
    const int64_t probs_bytes = ((int64_t)e->n_slots * e->table_elems() + (int64_t)e->F * e->S) * 4;
    const int64_t wpat_bytes = ((int64_t)e->n_slots * e->Pmax * e->F * e->C + (int64_t)e->F * e->C) * 4;
    if (probs_bytes >= ((int64_t)1 << 32) || wpat_bytes >= ((int64_t)1 << 32) || ((int64_t)(NT + 1) * KBp + 4) * 1024 >= ((int64_t)1 << 31)) return g;
    // one block = 16 slots x a range of column tiles; its 8 waves take the tiles in pairs, so a split of fewer than
    // 16 tiles leaves waves idle: as many splits as fill the CUs, no finer
    const int groups = div_up(n, 16);
    int n_split = std::max(1, std::min(div_up(NT, 16), e->compute_units / std::max(1, groups)));
    if (const char* env = getenv("SBE_MFMA_SPLIT")) { if (atoi(env) > 0) n_split = std::min(atoi(env), NT); }   // experiments
    g.nt_per_split = round_up(div_up(NT, n_split), 2);
    g.n_split = div_up(NT, g.nt_per_split);
    return g;
}

int launch_mfma_form(sbe_engine* e, int first_slot, int n, int KT, const MfmaGeom& mg, const int32_t* d_slots) {
    int rc = ensure_xt(e);
    if (rc) return rc;
    MfmaMixParams p{};
    p.F = e->F; p.S = e->S; p.FS = e->F * e->S; p.Gtot = e->Gtot; p.Np = e->Np;
    p.NT = e->xt_NT; p.KBp = e->xt_KBp; p.KT = KT;
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
    p.logtab = e->d_logtab;
    p.partials = e->d_partials; p.partials_stride = e->partials_stride;
    launch_tuple_mfma(e->C, p, dim3((unsigned)(div_up(n, 16) * mg.n_split)), mg.lds, e->stream);
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
    if (combo && (force_mfma || (e->opt_kernel == SBE_MIXTURE_PACKED && n >= e->mfma_min_batch))) mg = mfma_geometry(e, n, KT);
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
    if (mfma) {
        snprintf(e->last_kernel, sizeof e->last_kernel, "k_mixture_tuple_mfma<packed stream, group-tuple form, matrix pipe, M tiles %d, C=%d>", mg.MT, e->C);
        int rc = launch_mfma_form(e, first_slot, n, KT, mg, d_slots);
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
#ifdef SBE_STAMPS
        static uint64_t* d_stamps = nullptr;
        if (((combo && tuple64) || rows) && getenv("SBE_STAMPS_FILE")) {
            static size_t stamps_cap = 0;
            if (grid.x > stamps_cap) {
                if (d_stamps) { (void)hipStreamSynchronize(e->stream); (void)hipFree(d_stamps); }
                stamps_cap = (size_t)grid.x;
                (void)hipMalloc((void**)&d_stamps, stamps_cap * 48 * sizeof(uint64_t));
            }
            (void)hipMemsetAsync(d_stamps, 0, (size_t)grid.x * 48 * sizeof(uint64_t), e->stream);
            p.stamps = d_stamps;
        }
#endif
        snprintf(e->last_kernel, sizeof e->last_kernel, "%s<%s%s, tile %d, C=%d>",
                 combo ? (tuple64 ? "k_mixture_tuple64" : "k_mixture_combo") : rows ? "k_mixture_rows" : (onehot ? "k_mixture_onehot_v2" : "k_mixture_v2"),
                 onehot ? "one-hot stream" : "packed stream", combo ? ", group-tuple form" : (rows && sorted ? ", pattern-sorted objects" : (e->direct ? ", direct tables" : "")), g.ft, e->C);
        // (the kernels live in their own translation unit: sbe_mixture.hip)
        if (combo && tuple64) launch_tuple64(e->C, p, grid, combo_lds, e->stream);
        else if (combo) launch_combo(onehot, g.ft, e->C, p, grid, combo_lds, e->stream);
        else if (rows) launch_rows(mode, g.ft, e->C, p, grid, g.lds_bytes, e->stream, sorted);
        else if (onehot) launch_oh2(mode, g.ft, e->C, p, grid, g.lds_bytes, e->stream, e->direct);
        else launch_v2(mode, g.ft, e->C, p, grid, g.lds_bytes, e->stream, e->direct);
#ifdef SBE_STAMPS
        if (p.stamps) {                             // diagnostic build: dump the in-kernel stamps of this launch
            std::vector<uint64_t> h((size_t)grid.x * 48);
            (void)hipStreamSynchronize(e->stream);
            (void)hipMemcpy(h.data(), d_stamps, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
            FILE* f = fopen(getenv("SBE_STAMPS_FILE"), "wb");
            if (f) { fwrite(h.data(), sizeof(uint64_t), h.size(), f); fclose(f); }
        }
#endif
    }
    if (ev_b) HIPCHK(e, hipEventRecord(ev_b, e->stream));
    HIPCHK(e, hipGetLastError());
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
    if (e->ev_timing) {                             // one event pair per launch, on the engine's own stream
        while ((int)e->ev_pool.size() < 2 * (e->ev_used + 1)) {
            hipEvent_t ev;
            HIPCHK(e, hipEventCreate(&ev));
            e->ev_pool.push_back(ev);
        }
        ev_a = e->ev_pool[2 * e->ev_used]; ev_b = e->ev_pool[2 * e->ev_used + 1];
        ++e->ev_used;
    }
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

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int sbe_abi_version(void) { return SBE_ABI_VERSION; }

int sbe_device_count(int* out_count) {
    if (!out_count) return fail(nullptr, SBE_ERR_ARG, "null out_count");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) { *out_count = 0; return fail(nullptr, SBE_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(err)); }
    *out_count = n;
    return SBE_OK;
}

const char* sbe_last_error(const sbe_engine* e) { return e ? e->last_error.c_str() : g_last_error.c_str(); }

// ---- host helpers of the drop-in layer's marshalling (no device, no engine) -----------------------------------------
// What update_feature_counts hands to sbe_counts_delta is derived from the samples' own arrays: the listed objects'
// group id per component (group_assignment[:, object_subset], counts.py:21-24) and source component per observation
// (source[object_subset], counts.py:25-27).  In NumPy that derivation is a dozen small array operations per MCMC step
// (tools/host_residual.py: the host layer, not the device, bounds the patched sampler); here it is one pass each.
int sbe_host_group_ids(const uint8_t* groups, int n_groups, int64_t n_objects, const int32_t* objects, int n, int offset,
                       int32_t* ids_out) {
    return sbeh_group_ids(groups, n_groups, n_objects, objects, n, offset, ids_out);
}

int sbe_host_source_ids(const uint8_t* source, int64_t n_objects, int n_features, int n_components, const int32_t* objects, int n,
                        uint8_t* ids_out) {
    return sbeh_source_ids(source, n_objects, n_features, n_components, objects, n, ids_out);
}

int sbe_host_touched_groups(const int32_t* gid_old, const int32_t* gid_new, int64_t count, int n_groups_total,
                            int32_t* touched_out, int32_t* n_touched_out) {
    return sbeh_touched_groups(gid_old, gid_new, count, n_groups_total, touched_out, n_touched_out);
}

int sbe_host_subset_ids(const int32_t* objects, int n, int64_t n_objects, int n_features, int n_components,
                        const int32_t* n_groups, const uint8_t* const* groups_new, const uint8_t* const* groups_old,
                        const uint8_t* source_new, const uint8_t* source_old,
                        int32_t* gid_new_out, int32_t* gid_old_out, uint8_t* sid_new_out, uint8_t* sid_old_out) {
    return sbeh_subset_ids(objects, n, n_objects, n_features, n_components, n_groups, groups_new, groups_old, source_new, source_old,
                           gid_new_out, gid_old_out, sid_new_out, sid_old_out);
}

int64_t sbe_host_diff_rows(const void* rows, void* mirror, int64_t n_rows, int64_t row_bytes, int32_t* changed_out) {
    return sbeh_diff_rows(rows, mirror, n_rows, row_bytes, changed_out);
}

int sbe_destroy(sbe_engine* e) {
    if (!e) return SBE_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    delete e->pool;
    for (auto& ln : e->lanes) {
        if (ln.h_payload) (void)hipHostFree(ln.h_payload);
        if (ln.h_step) (void)hipHostFree(ln.h_step);
        if (ln.d_pf) (void)hipFree(ln.d_pf);
        if (ln.d_stamp) (void)hipFree(ln.d_stamp);
        if (ln.d_status) (void)hipFree(ln.d_status);
    }
    if (e->d_batch_meta) (void)hipFree(e->d_batch_meta);
    if (e->h_batch_payload) (void)hipHostFree(e->h_batch_payload);
    if (e->d_batch_payload) (void)hipFree(e->d_batch_payload);
    if (e->h_step) (void)hipHostFree(e->h_step);
    if (e->h_step_payload) (void)hipHostFree(e->h_step_payload);
    if (e->h_io) (void)hipHostFree(e->h_io);
    void* dev_ptrs[] = {e->d_step_pf, e->d_step_pg, e->d_logtab, e->d_state_h, e->d_toff, e->d_tid, e->d_tuple_g, e->d_tuple_p, e->d_state_q, e->d_probs_t, e->d_wpat_t, e->d_onehot, e->d_state, e->d_gid, e->d_pid, e->d_src, e->d_counts, e->d_probs,
                        e->d_weights, e->d_wpat, e->d_patbits, e->d_conc, e->d_lg_conc, e->d_sum_a, e->d_lg_sum_a, e->d_unif, e->d_unif_res, e->d_comp_of_group, e->d_partials, e->d_rowoff,
                        e->d_status, e->d_changed, e->d_step_stamp, e->d_scratch, e->d_xt, e->d_rowoff_s, e->d_rs_nq, e->d_state_s};
    for (void* p : dev_ptrs) if (p) (void)hipFree(p);
    if (e->h_results) (void)hipHostFree(e->h_results);
    if (e->h_status) (void)hipHostFree(e->h_status);
    if (e->h_flag) (void)hipHostFree(e->h_flag);
    if (e->h_done) (void)hipHostFree(e->h_done);
    if (e->h_chunk_flags) (void)hipHostFree(e->h_chunk_flags);
    if (e->d_chunk_tickets) (void)hipFree(e->d_chunk_tickets);
    if (e->h_stream) (void)hipHostFree(e->h_stream);
    if (e->d_ticket) (void)hipFree(e->d_ticket);
    if (e->h_pinned) (void)hipHostFree(e->h_pinned);
    if (e->h_arena) (void)hipHostFree(e->h_arena);
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return SBE_OK;
}

int sbe_create(sbe_engine** out, int device, int n_objects, int n_features, int n_states,
               int n_components, const int32_t* n_groups, int n_slots, const uint8_t* features_onehot) {
    if (!out) return fail(nullptr, SBE_ERR_ARG, "null out handle");
    *out = nullptr;
    if (!features_onehot || !n_groups) return fail(nullptr, SBE_ERR_ARG, "null features / n_groups");
    if (n_objects < 1 || n_features < 1) return fail(nullptr, SBE_ERR_ARG, "empty feature block (%d objects x %d features)", n_objects, n_features);
    if (n_states < 1 || n_states > 254) return fail(nullptr, SBE_ERR_ARG, "n_states=%d unsupported (1..254; state index is one byte, 0xFF = NA)", n_states);
    if (n_components < 1 || n_components > kMaxComponents) return fail(nullptr, SBE_ERR_ARG, "n_components=%d unsupported (1..%d)", n_components, kMaxComponents);
    if (n_slots < 1 || n_slots > 4096) return fail(nullptr, SBE_ERR_ARG, "n_slots=%d unsupported (1..4096)", n_slots);
    int64_t gtot = 0;
    for (int c = 0; c < n_components; ++c) {
        // a component may have no group at all (n_clusters == 0, the confounders-only baseline: the reference's
        // initializer returns an empty cluster matrix, sbayes/sampling/initializers.py:357): its tables are empty and
        // every object is in "no group" of it
        if (n_groups[c] < 0) return fail(nullptr, SBE_ERR_ARG, "component %d has %d groups", c, n_groups[c]);
        gtot += n_groups[c];
    }
    if (gtot < 1) return fail(nullptr, SBE_ERR_ARG, "no component has any group");
    if (gtot >= 0xFFFF) return fail(nullptr, SBE_ERR_ARG, "%lld groups in total exceed the 16-bit group index", (long long)gtot);

    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev == 0)
        return fail(nullptr, SBE_ERR_NODEVICE, "no HIP device available (%s); the engine has no CPU fallback",
                    err != hipSuccess ? hipGetErrorString(err) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, SBE_ERR_ARG, "device %d out of range [0,%d)", device, ndev);

    sbe_engine* e = new sbe_engine();
    e->device = device;
    e->N = n_objects; e->F = n_features; e->S = n_states; e->C = n_components; e->n_slots = n_slots;
    e->G.assign(n_groups, n_groups + n_components);
    e->goff.resize(n_components);
    for (int c = 0, o = 0; c < n_components; ++c) { e->goff[c] = o; o += n_groups[c]; }
    e->Gtot = (int)gtot;
    e->Fp = round_up(n_features, 64);
    e->rs_pitch = round_up(n_features * n_states, 16);
    e->Pmax = std::min(1 << n_components, 64);
    e->Np = round_up(n_objects, 4);
    e->NQ = e->Np / 4;
    {   // v2 feature-tile width: widest of 64/32/16 whose LDS image leaves two blocks per CU
        const char* env = getenv("SBE_FT");
        int ft = 64;
        auto lds_for = [&](int t) { return ((size_t)(gtot + 1) * t * n_states) * sizeof(float) + (size_t)e->Pmax * n_components * t * sizeof(double) + 8 * 1024; };
        while (ft > 16 && lds_for(ft) > 78 * 1024) ft >>= 1;
        if (env && (atoi(env) == 64 || atoi(env) == 32 || atoi(env) == 16)) ft = atoi(env);
        if (lds_for(ft) > 156 * 1024) {
            if (env) { delete e; return fail(nullptr, SBE_ERR_ARG, "probability tables too large for LDS staging at the forced tile width (G_total=%lld, S=%d)", (long long)gtot, n_states); }
            ft = 16;                 // very many groups x states: no LDS staging of tables (L2-served gathers)
            e->direct = true;
        }
        if (getenv("SBE_DIRECT") && atoi(getenv("SBE_DIRECT")) == 1) { ft = 16; e->direct = true; }   // experiments / tests
        e->ft = ft;
        e->n_ftiles = div_up(n_features, ft);
        e->Fq = round_up(n_features, 64);       // row pitch of the quad-interleaved state streams (>= any tiling of F)
    }
    e->conc_set.assign(n_components, 0);
    e->slots.resize(n_slots);
    e->src_sync.resize(n_slots);
    e->ids_sync.resize(n_slots);
    for (Slot& s : e->slots) {
        s.h_gid.assign((size_t)n_components * n_objects, kNoGroup);
        s.probs_set.assign(n_components, 0);
        s.counts_set.assign(n_components, 0);
    }

#define CREATE_CHK(call)                                                                        \
    do {                                                                                        \
        hipError_t _e2 = (call);                                                                \
        if (_e2 != hipSuccess) {                                                                \
            int _rc = fail(nullptr, SBE_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e2)); \
            sbe_destroy(e);                                                                     \
            return _rc;                                                                         \
        }                                                                                       \
    } while (0)
#define CREATE_RC(expr)                                        \
    do {                                                       \
        int _rc = (expr);                                      \
        if (_rc) { g_last_error = e->last_error; sbe_destroy(e); return _rc; } \
    } while (0)

    CREATE_CHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, device));
    e->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* env = getenv("SBE_MFMA_MIN_BATCH")) { if (atoi(env) > 0) e->mfma_min_batch = atoi(env); }      // (A/B runs, tests)
    if (const char* env = getenv("SBE_ROWS_SORTED")) e->opt_rows_sorted = atoi(env);
    snprintf(e->device_name, sizeof e->device_name, "%s%s%s", prop.name, prop.name[0] ? " " : "", prop.gcnArchName);
    CREATE_CHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreate(&e->ev0));
    CREATE_CHK(hipEventCreate(&e->ev1));
    for (int i = 0; i < 64; ++i) {                  // event pool of sbe_kernel_timing (grown on demand beyond this)
        hipEvent_t ev;
        CREATE_CHK(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }

    const int64_t N = e->N, F = e->F, S = e->S, C = e->C, NS = n_slots;
    CREATE_RC(dmalloc(e, &e->d_onehot, N * e->rs_pitch));
    CREATE_RC(dmalloc(e, &e->d_state, N * e->Fp));
    CREATE_RC(dmalloc(e, &e->d_state_q, (int64_t)e->NQ * e->Fq * 4));
    CREATE_RC(dmalloc(e, &e->d_probs_t, NS * e->probs_t_elems()));
    CREATE_RC(dmalloc(e, &e->d_wpat_t, NS * e->wpat_t_elems()));
    CREATE_RC(dmalloc(e, &e->d_gid, NS * C * e->Np));
    CREATE_RC(dmalloc(e, &e->d_pid, NS * e->Np));
    CREATE_RC(dmalloc(e, &e->d_src, NS * N * e->Fp));
    CREATE_RC(dmalloc(e, &e->d_counts, NS * e->table_elems()));
    CREATE_RC(dmalloc(e, &e->d_probs, NS * e->table_elems() + F * S));      // + a row of ones [F][S] } what k_mixture_tuple_mfma reads for
    CREATE_RC(dmalloc(e, &e->d_weights, NS * F * C));
    CREATE_RC(dmalloc(e, &e->d_wpat, NS * e->Pmax * F * C + F * C));        // + a row of ones [F][C] } tuples / groups that are not there
    {
        const std::vector<float> ones((size_t)(F * std::max(S, C)), 1.0f);
        CREATE_CHK(hipMemcpy(e->d_probs + NS * e->table_elems(), ones.data(), F * S * sizeof(float), hipMemcpyHostToDevice));
        CREATE_CHK(hipMemcpy(e->d_wpat + NS * e->Pmax * F * C, ones.data(), F * C * sizeof(float), hipMemcpyHostToDevice));
    }
    CREATE_RC(dmalloc(e, &e->d_patbits, NS * e->Pmax));
    CREATE_RC(dmalloc(e, &e->d_tid, NS * e->Np));
    CREATE_RC(dmalloc(e, &e->d_toff, NS * e->Np + 64));       // + padding: the kernel prefetches 16 entries ahead
    CREATE_CHK(hipMemsetAsync(e->d_toff, 0, (NS * e->Np + 64) * sizeof(uint32_t), e->stream));
    if (C <= 4) {   // k_mixture_rows: widest tile whose LDS image (tables f32 [(Gtot+1)][S+1][ft] + f64 weight planes) fits
        // (sized for half the possible has_components patterns: a component every object has -- `universal` -- halves
        //  them; a launch whose slots really have more falls back to k_mixture_v2) + the waves' offset slots
        const int p_assumed = std::max(1, e->Pmax / 2);
        auto rows_lds = [&](int t) { return (size_t)(gtot + 1) * (S + 1) * t * 4 + (size_t)p_assumed * ((C + 1) / 2) * t * 16 + (size_t)kRowsWaves * (kWave / t) * (C + 1) * 16; };
        e->rows_ft = rows_lds(32) <= 160 * 1024 - 512 ? 32 : rows_lds(16) <= 160 * 1024 - 512 ? 16 : 0;
        if (const char* env = getenv("SBE_ROWS_FT")) { const int v = atoi(env); if (v == 0 || ((v == 16 || v == 32) && rows_lds(v) <= 160 * 1024 - 512)) e->rows_ft = v; }
        if (e->rows_ft) {
            CREATE_RC(dmalloc(e, &e->d_rowoff, NS * (C + 1) * e->Np));
            e->rowoff_epoch.assign(n_slots, ~0ull);
        }
    }
    {   // table of tab_log_pos: interval centres c_i = 1 + (i + 1/2)/128 (c_0 = 1), {RN(1/c), RN(-log(RN(1/c)))}
        std::vector<double> tab(2 * kLogTabEntries);
        for (int i = 0; i < kLogTabEntries; ++i) {
            const double c = i == 0 ? 1.0 : 1.0 + (i + 0.5) / kLogTabEntries;
            const double inv_c = 1.0 / c;
            tab[2 * i] = inv_c;
            tab[2 * i + 1] = i == 0 ? 0.0 : (double)(-logl((long double)inv_c));
        }
        CREATE_RC(dmalloc(e, &e->d_logtab, (int64_t)kLogTabEntries));
        CREATE_CHK(hipMemcpy(e->d_logtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (e->ft == 64 && S <= 127 && (int64_t)e->NQ * e->Fq * 8 < ((int64_t)1 << 31)) {
        CREATE_RC(dmalloc(e, &e->d_state_h, (int64_t)e->NQ * e->Fq * 4));
        k_init_state_h<<<div_up((int64_t)e->NQ * e->Fq, 256), 256, 0, e->stream>>>(e->d_state_h, (int64_t)e->NQ * e->Fq, e->Fq, e->S);
        CREATE_CHK(hipGetLastError());
    }
    CREATE_RC(dmalloc(e, &e->d_tuple_g, NS * kMaxTuples * kMaxComponents));
    CREATE_RC(dmalloc(e, &e->d_tuple_p, NS * (int64_t)kMaxTuples));
    CREATE_CHK(hipMemsetAsync(e->d_tid, 0, NS * e->Np, e->stream));
    CREATE_RC(dmalloc(e, &e->d_conc, e->table_elems()));
    CREATE_RC(dmalloc(e, &e->d_unif, F * S));
    CREATE_RC(dmalloc(e, &e->d_unif_res, F * S));
    CREATE_RC(dmalloc(e, &e->d_lg_conc, e->table_elems()));
    CREATE_RC(dmalloc(e, &e->d_sum_a, (int64_t)e->Gtot * F));
    CREATE_RC(dmalloc(e, &e->d_lg_sum_a, (int64_t)e->Gtot * F));
    CREATE_RC(dmalloc(e, &e->d_comp_of_group, e->Gtot));
    {
        std::vector<int32_t> cog(std::max(e->Gtot, 1), 0);
        for (int c = 0; c < e->C; ++c)
            for (int g = 0; g < e->G[c]; ++g) cog[e->goff[c] + g] = c;
        CREATE_CHK(hipMemcpy(e->d_comp_of_group, cog.data(), (size_t)std::max(e->Gtot, 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // partials: worst-case block count of the fused kernel (ft = 16, one packed step per thread)
    {
        const int64_t min_objs = kBlock / (16 / 4);
        e->partials_stride = std::max<int64_t>(div_up(F, 16) * std::max<int64_t>(div_up(N, min_objs), 4 * e->compute_units), 1024);
    }
    CREATE_RC(dmalloc(e, &e->d_partials, NS * e->partials_stride));
    CREATE_RC(dmalloc(e, &e->d_status, (int64_t)ST_WORDS));
    CREATE_RC(dmalloc(e, &e->d_changed, (int64_t)e->Gtot));
    CREATE_RC(dmalloc(e, &e->d_step_stamp, (int64_t)e->Gtot));
    CREATE_CHK(hipMemsetAsync(e->d_step_stamp, 0, e->Gtot * sizeof(uint32_t), e->stream));
    CREATE_RC(dmalloc(e, &e->d_step_pf, (int64_t)e->Gtot * F));
    CREATE_RC(dmalloc(e, &e->d_step_pg, (int64_t)e->Gtot));
    {   // one-call step: payload layout (every section 16-byte aligned) and the mapped result block
        auto al = [](size_t v) { return (v + 15) / 16 * 16; };
        e->step_max_rows = (int)std::min<int64_t>(N, 256);
        size_t o = 0;
        e->sl.ids = o;      o = al(o + (size_t)e->Np * 2);
        e->sl.pid = o;      o = al(o + (size_t)e->Np);
        e->sl.tid = o;      o = al(o + (size_t)e->Np);
        e->sl.toff = o;     o = al(o + (size_t)e->Np * 4);
        e->sl.tuple_g = o;  o = al(o + (size_t)kMaxTuples * kMaxComponents * 2);
        e->sl.tuple_p = o;  o = al(o + (size_t)kMaxTuples);
        e->sl.patbits = o;  o = al(o + (size_t)e->Pmax * 4);
        e->sl.weights = o;  o = al(o + (size_t)(F * C) * 4);
        e->sl.row_of = o;   o = al(o + (size_t)e->Np * 2);
        e->sl.subset = o;   o = al(o + (size_t)e->Np * 4);
        e->sl.stale = o;    o = al(o + (size_t)e->step_max_rows * 4);
        e->sl.objects = o;  o = al(o + (size_t)e->step_max_rows * 4);
        e->sl.rows = o;     o = al(o + (size_t)e->step_max_rows * (size_t)(F * C));
        e->sl.total = o;
        CREATE_CHK(hipHostMalloc((void**)&e->h_step_payload, e->sl.total, hipHostMallocMapped));
        CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_step_payload, e->h_step_payload, 0));
        const size_t hb = ((size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int) + (size_t)e->Gtot + 7) / 8 * 8 + 2 * sizeof(double);   // (step_host_lq_offset + log_q, log_q_back)
        CREATE_CHK(hipHostMalloc((void**)&e->h_step, hb, hipHostMallocMapped));
        memset(e->h_step, 0, hb);
        CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_step_host, e->h_step, 0));
    }
    CREATE_CHK(hipHostMalloc((void**)&e->h_results, NS * sizeof(double), hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_results, e->h_results, 0));
    CREATE_CHK(hipHostMalloc((void**)&e->h_status, ST_WORDS * sizeof(int), hipHostMallocDefault));
    memset(e->h_status, 0, ST_WORDS * sizeof(int));
    CREATE_CHK(hipHostMalloc((void**)&e->h_flag, ST_WORDS * sizeof(int), hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_flag, e->h_flag, 0));
    memset(e->h_flag, 0, ST_WORDS * sizeof(int));
    CREATE_CHK(hipHostMalloc((void**)&e->h_done, 64, hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_done, e->h_done, 0));
    memset(e->h_done, 0, 64);
    CREATE_CHK(hipMalloc((void**)&e->d_ticket, 64));
    CREATE_CHK(hipMemsetAsync(e->d_ticket, 0, 64, e->stream));
    CREATE_CHK(hipHostMalloc((void**)&e->h_chunk_flags, sbe_engine::kMaxChunks * sizeof(unsigned long long), hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_chunk_flags, e->h_chunk_flags, 0));
    memset(e->h_chunk_flags, 0, sbe_engine::kMaxChunks * sizeof(unsigned long long));
    CREATE_CHK(hipMalloc((void**)&e->d_chunk_tickets, sbe_engine::kMaxChunks * sizeof(unsigned)));
    CREATE_CHK(hipMemsetAsync(e->d_chunk_tickets, 0, sbe_engine::kMaxChunks * sizeof(unsigned), e->stream));
    e->arena_bytes = (size_t)16 << 20;
    CREATE_CHK(hipHostMalloc((void**)&e->h_arena, e->arena_bytes, hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_arena, e->h_arena, 0));

    CREATE_CHK(hipMemsetAsync(e->d_status, 0, ST_WORDS * sizeof(int), e->stream));
    CREATE_CHK(hipMemcpyAsync(e->d_status + ST_FLAG_PTR, &e->d_flag, sizeof(int*), hipMemcpyHostToDevice, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_onehot, 0, N * e->rs_pitch, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_state, 0xFF, N * e->Fp, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_src, 0xFF, NS * N * e->Fp, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_gid, 0xFF, NS * C * e->Np * sizeof(uint16_t), e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_pid, 0, NS * e->Np, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_state_q, S, (int64_t)e->NQ * e->Fq * 4, e->stream));   // NA / padding byte = S
    CREATE_CHK(hipMemsetAsync(e->d_probs_t, 0, NS * e->probs_t_elems() * sizeof(float), e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_wpat_t, 0, NS * e->wpat_t_elems() * sizeof(double), e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_counts, 0, NS * e->table_elems() * sizeof(int32_t), e->stream));

    // ingest: raw one-hot -> normalised padded copy + packed state index + validation
    CREATE_RC(ensure_scratch(e, (size_t)(N * F * S)));
    CREATE_CHK(hipMemcpyAsync(e->d_scratch, features_onehot, (size_t)(N * F * S), hipMemcpyHostToDevice, e->stream));
    k_ingest_onehot<<<div_up(N * F, 256), 256, 0, e->stream>>>(e->d_scratch, e->d_onehot, e->d_state, e->d_state_q,
                                                              e->d_state_h, e->N, e->F, e->S, e->rs_pitch, e->Fp, e->Fq, e->d_status);
    CREATE_CHK(hipGetLastError());
    CREATE_CHK(hipMemcpyAsync(e->h_status, e->d_status, ST_FLAG_PTR * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    CREATE_CHK(hipStreamSynchronize(e->stream));
    if (e->h_status[ST_MULTI_STATE] != 0) {
        int rc = fail(nullptr, SBE_ERR_DATA, "features are not one-hot: %d (object, feature) rows have more than one state set",
                      e->h_status[ST_MULTI_STATE]);
        sbe_destroy(e);
        return rc;
    }
    e->n_na = e->h_status[ST_NA_COUNT];
#undef CREATE_CHK
#undef CREATE_RC
    *out = e;
    return SBE_OK;
}

int sbe_get_info(const sbe_engine* e, sbe_info* out) {
    CHECK_ENGINE(e);
    if (!out) return fail(nullptr, SBE_ERR_ARG, "null out");
    memset(out, 0, sizeof *out);
    out->abi_version = SBE_ABI_VERSION;
    out->device = e->device;
    out->n_objects = e->N; out->n_features = e->F; out->n_states = e->S;
    out->n_components = e->C; out->n_slots = e->n_slots; out->n_groups_total = e->Gtot;
    out->n_na = e->n_na; out->hbm_bytes = e->hbm_bytes; out->compute_units = e->compute_units;
    memcpy(out->device_name, e->device_name, sizeof out->device_name);
    return SBE_OK;
}

int sbe_get_na(const sbe_engine* ce, uint8_t* out_na) {
    sbe_engine* e = const_cast<sbe_engine*>(ce);
    CHECK_ENGINE(e); CHECK_PTR(e, out_na);
    HIPCHK(e, hipSetDevice(e->device));
    std::vector<uint8_t> st((size_t)e->N * e->Fp);
    int rc = d2h(e, st.data(), e->d_state, st.size());
    if (rc) return rc;
    for (int n = 0; n < e->N; ++n)
        for (int f = 0; f < e->F; ++f) out_na[(size_t)n * e->F + f] = st[(size_t)n * e->Fp + f] == kNA;
    return SBE_OK;
}

int sbe_set_option(sbe_engine* e, int option, int value) {
    CHECK_ENGINE(e);
    if (option == SBE_OPT_MIXTURE_KERNEL && (value == SBE_MIXTURE_PACKED || value == SBE_MIXTURE_ONEHOT || value == SBE_MIXTURE_PACKED_GENERAL || value == SBE_MIXTURE_PACKED_TUPLE || value == SBE_MIXTURE_PACKED_TUPLE_LDS || value == SBE_MIXTURE_ONEHOT_GENERAL || value == SBE_MIXTURE_PACKED_V2 || value == SBE_MIXTURE_PACKED_TUPLE_MFMA)) { e->opt_kernel = value; return SBE_OK; }
    if (option == SBE_OPT_LOG_MODE && (value == SBE_LOG_PER_OBS || value == SBE_LOG_PRODUCT)) { e->opt_log = value; return SBE_OK; }
    if (option == SBE_OPT_STEP_FORM && (value == 0 || value == 1)) { e->opt_step_form = value; return SBE_OK; }
    if (option == SBE_OPT_STEP_DERIVE && (value == 0 || value == 1)) { e->opt_step_derive = value; return SBE_OK; }
    if (option == SBE_OPT_FUSE_TABLES && (value == 0 || value == 1)) { e->opt_fuse_tables = value; return SBE_OK; }
    if (option == SBE_OPT_DEFERRED_CHECKS && (value == 0 || value == 1)) {
        if (!value && e->status_pending) { HIPCHK(e, hipStreamSynchronize(e->stream)); int rc = synced(e); e->opt_deferred = 0; return rc; }
        e->opt_deferred = value;
        return SBE_OK;
    }
    return fail(e, SBE_ERR_ARG, "unknown option %d / value %d", option, value);
}

int sbe_sync(sbe_engine* e) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return synced(e);
}

// ---- a1 -------------------------------------------------------------------------------------
int sbe_component_lh(sbe_engine* e, const void* probs, int probs_f64, int n_groups, const uint8_t* groups,
                     const int64_t* changed_groups, int n_changed, double* out, int64_t out_stride_n_bytes,
                     int64_t out_stride_f_bytes, double na_value) {
    CHECK_ENGINE(e); CHECK_PTR(e, probs); CHECK_PTR(e, groups); CHECK_PTR(e, out);
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    if (n_changed < 0 || (n_changed > 0 && !changed_groups)) return fail(e, SBE_ERR_ARG, "bad changed_groups");
    for (int k = 0; k < n_changed; ++k)
        if (changed_groups[k] < 0 || changed_groups[k] >= n_groups)
            return fail(e, SBE_ERR_ARG, "changed_groups[%d]=%lld out of range [0,%d)", k, (long long)changed_groups[k], n_groups);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, F = e->F, S = e->S;
    // row selector: -2 untouched, -1 zero, >=0 table index (last changed group wins)
    std::vector<int32_t> sel(N, -2);
    for (int n = 0; n < N; ++n) {
        bool any = false;
        for (int g = 0; g < n_groups && !any; ++g) any = groups[(size_t)g * N + n] != 0;
        if (!any) sel[n] = -1;
    }
    for (int k = 0; k < n_changed; ++k) {
        const int64_t g = changed_groups[k];
        const uint8_t* row = groups + (size_t)g * N;
        for (int n = 0; n < N; ++n) if (row[n]) sel[n] = (int32_t)g;
    }
    const size_t esz = probs_f64 ? sizeof(double) : sizeof(float);
    const size_t tab_bytes = (size_t)n_groups * F * S * esz;
    const size_t tab_pad = (tab_bytes + 255) / 256 * 256;
    const size_t sel_bytes = ((size_t)N * sizeof(int32_t) + 255) / 256 * 256;
    const size_t out_bytes = (size_t)N * F * sizeof(double);
    int rc = ensure_scratch(e, tab_pad + sel_bytes + out_bytes);
    if (rc) return rc;
    uint8_t* d_tab = e->d_scratch;
    int32_t* d_sel = (int32_t*)(e->d_scratch + tab_pad);
    double* d_out = (double*)(e->d_scratch + tab_pad + sel_bytes);
    {   // table + row selector: one enqueue when they fit the mapped ring (upload_segments), two copies otherwise
        const UploadSeg segs[2] = {{d_tab, probs, tab_bytes}, {d_sel, sel.data(), (size_t)N * sizeof(int32_t)}};
        int _urc = upload_segments(e, segs, 2);
        if (_urc) return _urc;
    }
    const int blocks = div_up((int64_t)N * F, 512);       // (k_component_lh: two output elements per thread)
    const int32_t* selp = sel.data();
    char* base = (char*)out;
    auto scatter_rows = [=](const double* dense, int n0, int n1) {   // rows the call writes: the caller's (strided) view <- staging rows
        for (int n = n0; n < n1; ++n) {
            if (selp[n] == -2) continue;
            char* row = base + (int64_t)n * out_stride_n_bytes;
            const double* srow = dense + (size_t)n * F;
            if (out_stride_f_bytes == (int64_t)sizeof(double)) memcpy(row, srow, (size_t)F * sizeof(double));
            else for (int f = 0; f < F; ++f) *(double*)(row + (int64_t)f * out_stride_f_bytes) = srow[f];
        }
    };
    static const bool no_stream = [] { const char* v = getenv("SBE_STREAM_RESULTS"); return v && atoi(v) == 0; }();   // (A/B)
    if (out_bytes >= ((size_t)1 << 19) && !no_stream) {
        // large result: the kernel stores it into host-mapped staging and reports chunk by chunk; the scatter into the
        // caller's view runs on the host pool while the later chunks cross PCIe (stream_result)
        rc = ensure_stream(e, out_bytes);
        if (rc) return rc;
        const StreamPlan plan = plan_stream(e, (unsigned)blocks, (size_t)512 * sizeof(double), out_bytes);
        double* s_out = (double*)e->d_stream;
        if (probs_f64) k_component_lh<double><<<plan.grid, 256, 0, e->stream>>>(e->d_state, (const double*)d_tab, d_sel, s_out, N, F, S, e->Fp, na_value, plan.sig);
        else k_component_lh<float><<<plan.grid, 256, 0, e->stream>>>(e->d_state, (const float*)d_tab, d_sel, s_out, N, F, S, e->Fp, na_value, plan.sig);
        HIPCHK(e, hipGetLastError());
        const double* dense = (const double*)e->h_stream;
        const int rows_per_job = std::max(1, (int)(((size_t)64 << 10) / ((size_t)F * sizeof(double))));
        const int n_jobs = div_up(N, rows_per_job);
        return stream_result(e, plan, n_jobs,
                             [=](int j) { return (size_t)j * rows_per_job * F * sizeof(double); },
                             [=](int j) { return (size_t)std::min(N, (j + 1) * rows_per_job) * F * sizeof(double); },
                             [=](int j) { scatter_rows(dense, j * rows_per_job, std::min(N, (j + 1) * rows_per_job)); });
    }
    if (probs_f64) k_component_lh<double><<<blocks, 256, 0, e->stream>>>(e->d_state, (const double*)d_tab, d_sel, d_out, N, F, S, e->Fp, na_value);
    else k_component_lh<float><<<blocks, 256, 0, e->stream>>>(e->d_state, (const float*)d_tab, d_sel, d_out, N, F, S, e->Fp, na_value);
    HIPCHK(e, hipGetLastError());
    rc = ensure_pinned(e, out_bytes);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(e->h_pinned, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    scatter_rows((const double*)e->h_pinned, 0, N);
    return SBE_OK;
}

// ---- groups -----------------------------------------------------------------------------------
static int set_gid_common(sbe_engine* e, int slot, int component, const std::vector<uint16_t>& ids) {
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

// Resident slot state keeps ONE group per object and component (u16 ids).  A bool [G][N] matrix with an object in two
// rows has no such form: the reference counts the object once per group it is in (compute_effect_counts,
// sbayes/sampling/counts.py:28-30) while its a1 lets the last written group win (likelihood.py:126-130) -- collapsing the
// matrix to ids would silently follow only one of the two.  Overlap is therefore REJECTED here (SBE_ERR_DATA); the
// stateless sbe_effect_counts / sbe_component_lh take such matrices and follow the reference.  (sBayes itself never
// produces overlap: operators.py:724-725, :1099-1101 for clusters, load_data.py:174-178 for confounders.)
static void overlap_message(char* buf, size_t len, int n, int g1, int g2, int component) {
    snprintf(buf, len, "object %d is in groups %d and %d of component %d: resident slot state keeps one group per object "
             "and component (counts.py:28-30 would count it in both); overlapping groups are served by the stateless "
             "sbe_effect_counts / sbe_component_lh only", n, g1, g2, component);
}

// bool [G][N] -> one id per object (off + g, kNoGroup: in no group).  false + message on overlap.
static bool matrix_to_ids(const uint8_t* groups, int G, int N, int off, int component, uint16_t* ids, char* msg, size_t msg_len) {
    std::fill(ids, ids + N, kNoGroup);
    for (int g = 0; g < G; ++g) {                     // (mostly zeros: eight objects per test)
        const uint8_t* row = groups + (size_t)g * N;
        int n = 0;
        for (; n + 8 <= N; n += 8) {
            uint64_t w8;
            memcpy(&w8, row + n, 8);
            if (!w8) continue;
            for (int k = 0; k < 8; ++k) if (row[n + k]) {
                if (ids[n + k] != kNoGroup) { overlap_message(msg, msg_len, n + k, (int)ids[n + k] - off, g, component); return false; }
                ids[n + k] = (uint16_t)(off + g);
            }
        }
        for (; n < N; ++n) if (row[n]) {
            if (ids[n] != kNoGroup) { overlap_message(msg, msg_len, n, (int)ids[n] - off, g, component); return false; }
            ids[n] = (uint16_t)(off + g);
        }
    }
    return true;
}

int sbe_set_groups(sbe_engine* e, int slot, int component, const uint8_t* groups) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] > 0) CHECK_PTR(e, groups);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, G = e->G[component], off = e->goff[component];
    std::vector<uint16_t> ids(N, kNoGroup);
    char msg[320];
    if (!matrix_to_ids(groups, G, N, off, component, ids.data(), msg, sizeof msg)) return fail(e, SBE_ERR_DATA, "%s", msg);
    return set_gid_common(e, slot, component, ids);
}

int sbe_set_group_ids(sbe_engine* e, int slot, int component, const int32_t* ids_in) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component); CHECK_PTR(e, ids_in);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, G = e->G[component], off = e->goff[component];
    std::vector<uint16_t> ids(N, kNoGroup);
    for (int n = 0; n < N; ++n) {
        if (ids_in[n] >= G) return fail(e, SBE_ERR_ARG, "group id %d of object %d out of range [0,%d)", ids_in[n], n, G);
        if (ids_in[n] >= 0) ids[n] = (uint16_t)(off + ids_in[n]);
    }
    return set_gid_common(e, slot, component, ids);
}

// ---- source -----------------------------------------------------------------------------------
int sbe_set_source(sbe_engine* e, int slot, const uint8_t* source) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, source);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bytes = (size_t)e->N * e->F * e->C;
    int rc = ensure_scratch(e, bytes);
    if (rc) return rc;
    rc = clear_status_word(e, ST_MULTI_SOURCE);
    if (rc) return rc;
    { int _urc = upload(e, e->d_scratch, source, bytes); if (_urc) return _urc; }
    k_ingest_source<<<div_up((int64_t)e->N * e->F, 256), 256, 0, e->stream>>>(
        e->d_scratch, nullptr, e->d_src + (int64_t)slot * e->N * e->Fp, e->N, e->F, e->C, e->Fp, e->d_status);
    HIPCHK(e, hipGetLastError());
    bump_src(e, slot);
    e->slots[slot].source_set = true;
    return check_after(e, ST_MULTI_SOURCE);
}

int sbe_set_source_rows(sbe_engine* e, int slot, const int32_t* objects, int n_rows, const uint8_t* rows) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, rows);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    for (int i = 0; i < n_rows; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t row_bytes = (size_t)n_rows * e->F * e->C;
    const size_t row_pad = (row_bytes + 255) / 256 * 256;
    int rc = ensure_scratch(e, row_pad + (size_t)n_rows * sizeof(int32_t));
    if (rc) return rc;
    rc = clear_status_word(e, ST_MULTI_SOURCE);
    if (rc) return rc;
    const void *v_rows, *v_obj;
    rc = stage(e, rows, row_bytes, e->d_scratch, &v_rows);
    if (rc) return rc;
    rc = stage(e, objects, (size_t)n_rows * sizeof(int32_t), e->d_scratch + row_pad, &v_obj);
    if (rc) return rc;
    if (e->batch && e->batch->n_src_blocks == 0 && v_rows != (const void*)e->d_scratch) {     // (staged in the ring: sbe_set_slot_delta launches it)
        SetterJobs& j = *e->batch;
        j.src_rows = (const uint8_t*)v_rows; j.src_objects = (const int32_t*)v_obj; j.src_id = e->d_src + (int64_t)slot * e->N * e->Fp;
        j.src_n = n_rows; j.src_F = e->F; j.src_C = e->C; j.src_Fp = e->Fp; j.src_status = e->d_status;
        j.n_src_blocks = (unsigned)div_up((int64_t)n_rows * e->F, 256);
    } else {
        k_ingest_source<<<div_up((int64_t)n_rows * e->F, 256), 256, 0, e->stream>>>(
            (const uint8_t*)v_rows, (const int32_t*)v_obj, e->d_src + (int64_t)slot * e->N * e->Fp, n_rows, e->F, e->C, e->Fp, e->d_status);
        HIPCHK(e, hipGetLastError());
    }
    bump_src(e, slot);
    return check_after(e, ST_MULTI_SOURCE);
}

int sbe_get_source_rows(sbe_engine* e, int slot, const int32_t* objects, int n_rows, uint8_t* rows_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, rows_out);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    if (!e->slots[slot].source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set", slot);
    for (int i = 0; i < n_rows; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t row_bytes = (size_t)n_rows * e->F * e->C;
    const size_t row_pad = (row_bytes + 255) / 256 * 256;
    int rc = ensure_scratch(e, row_pad + (size_t)n_rows * sizeof(int32_t));
    if (rc) return rc;
    int32_t* d_obj = (int32_t*)(e->d_scratch + row_pad);
    { int _urc = upload(e, d_obj, objects, (size_t)n_rows * sizeof(int32_t)); if (_urc) return _urc; }
    k_expand_source<<<div_up((int64_t)n_rows * e->F, 256), 256, 0, e->stream>>>(
        e->d_src + (int64_t)slot * e->N * e->Fp, d_obj, e->d_scratch, n_rows, e->F, e->C, e->Fp);
    HIPCHK(e, hipGetLastError());
    return d2h(e, rows_out, e->d_scratch, row_bytes);
}

// ---- counts -----------------------------------------------------------------------------------
int sbe_recount(sbe_engine* e, int slot, int component) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (component != -1) CHECK_COMP(e, component);
    Slot& s = e->slots[slot];
    if (!s.source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set", slot);
    if (!s.groups_set) return fail(e, SBE_ERR_STATE, "slot %d: groups not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    // the kernel counts every component in one pass (each observation has one source
    // component); a single-component request recounts all and is still exact.
    HIPCHK(e, hipMemsetAsync(e->d_counts + (int64_t)slot * e->table_elems(), 0, e->table_elems() * sizeof(int32_t), e->stream));
    int rc = counts_launch(e, slot, +1, slot, 0, nullptr, e->N, slot, false, nullptr);
    if (rc) return rc;
    std::fill(s.counts_set.begin(), s.counts_set.end(), 1);
    return SBE_OK;
}

int sbe_update_counts(sbe_engine* e, int slot_new, int slot_old, const int32_t* objects, int n_subset,
                      uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot_new); CHECK_SLOT(e, slot_old);
    if (n_subset < 0) return fail(e, SBE_ERR_ARG, "n_subset=%d", n_subset);
    if (n_subset > 0) CHECK_PTR(e, objects);
    for (int i = 0; i < n_subset; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemsetAsync(e->d_changed, 0, e->Gtot, e->stream));
    if (n_subset > 0) {
        int rc = ensure_scratch(e, (size_t)n_subset * sizeof(int32_t));
        if (rc) return rc;
        { int _urc = upload(e, e->d_scratch, objects, (size_t)n_subset * sizeof(int32_t)); if (_urc) return _urc; }
        rc = counts_launch(e, slot_new, +1, slot_old, -1, (const int32_t*)e->d_scratch, n_subset, slot_new, true, e->d_changed);
        if (rc) return rc;
    }
    if (changed_groups_out) {
        int rc = d2h(e, changed_groups_out, e->d_changed, e->Gtot);
        if (rc) return rc;
    }
    return SBE_OK;
}

int sbe_accumulate_counts(sbe_engine* e, int slot, const int32_t* objects, int n_subset, int sign,
                          uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (sign != 1 && sign != -1) return fail(e, SBE_ERR_ARG, "sign must be +1 or -1");
    if (n_subset < 0) return fail(e, SBE_ERR_ARG, "n_subset=%d", n_subset);
    if (n_subset > 0) CHECK_PTR(e, objects);
    for (int i = 0; i < n_subset; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemsetAsync(e->d_changed, 0, e->Gtot, e->stream));
    if (n_subset > 0) {
        int rc = ensure_scratch(e, (size_t)n_subset * sizeof(int32_t));
        if (rc) return rc;
        { int _urc = upload(e, e->d_scratch, objects, (size_t)n_subset * sizeof(int32_t)); if (_urc) return _urc; }
        rc = counts_launch(e, slot, sign, slot, 0, (const int32_t*)e->d_scratch, n_subset, slot, true, e->d_changed);
        if (rc) return rc;
    }
    if (changed_groups_out) return d2h(e, changed_groups_out, e->d_changed, e->Gtot);
    return SBE_OK;
}

int sbe_set_counts(sbe_engine* e, int slot, int component, const float* counts) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) { e->slots[slot].counts_set[component] = 1; return SBE_OK; }
    CHECK_PTR(e, counts);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    const void* v_counts;
    rc = stage(e, counts, n * sizeof(float), e->d_scratch, &v_counts);
    if (rc) return rc;
    int32_t* dst = e->d_counts + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    k_f32_to_i32<<<div_up(n, 256), 256, 0, e->stream>>>((const float*)v_counts, dst, n);
    HIPCHK(e, hipGetLastError());
    e->slots[slot].counts_set[component] = 1;
    return SBE_OK;
}

int sbe_get_counts(sbe_engine* e, int slot, int component, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) return SBE_OK;
    CHECK_PTR(e, out);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    const int32_t* src = e->d_counts + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    void* d_out;
    rc = out_target(e, n * sizeof(float), e->d_scratch, &d_out);
    if (rc) return rc;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, 1024), 64);
    const DoneSig done = out_done(e, d_out, blocks);
    if (done.flag) k_i32_to_f32_done<<<blocks, 1024, 0, e->stream>>>(src, (float*)d_out, n, done);
    else k_i32_to_f32<<<div_up(n, 256), 256, 0, e->stream>>>(src, (float*)d_out, n);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, out, d_out, n * sizeof(float), done);
}

// ---- concentration / probs ----------------------------------------------------------------------
int sbe_set_concentration(sbe_engine* e, int component, const double* conc, int per_group) {
    CHECK_ENGINE(e); CHECK_COMP(e, component);
    if (e->G[component] == 0) { e->conc_set[component] = 1; return SBE_OK; }
    CHECK_PTR(e, conc);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S;
    double* dst = e->d_conc + (int64_t)e->goff[component] * fs;
    if (per_group) {
        { int _urc = upload(e, dst, conc, e->G[component] * fs * sizeof(double)); if (_urc) return _urc; }
    } else {
        for (int g = 0; g < e->G[component]; ++g)
            { int _urc = upload(e, dst + g * fs, conc, fs * sizeof(double)); if (_urc) return _urc; }
    }
    // the count-independent lgamma terms of this component's tables, for the one-call steps
    const int g_lo = e->goff[component], g_hi = g_lo + e->G[component];
    k_conc_lgamma<<<div_up((int64_t)(g_hi - g_lo) * e->F, 256), 256, 0, e->stream>>>(e->d_conc, e->d_lg_conc, e->d_sum_a, e->d_lg_sum_a,
                                                                              g_lo, g_hi, e->F, e->S);
    HIPCHK(e, hipGetLastError());
    e->conc_set[component] = 1;
    return SBE_OK;
}

int sbe_update_probs_mask(sbe_engine* e, int slot, unsigned component_mask, double temperature, double prior_temperature,
                          const double* unif_counts) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (e->C < 32 && (component_mask >> e->C) != 0) return fail(e, SBE_ERR_ARG, "component mask 0x%x names components >= %d", component_mask, e->C);
    Slot& s = e->slots[slot];
    bool any = false;
    for (int c = 0; c < e->C; ++c) {
        if (!((component_mask >> c) & 1u)) continue;
        if (e->G[c] == 0) { s.probs_set[c] = 1; continue; }                        // component without groups: empty tables
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
        if (!s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
        any = true;
    }
    if (!any) return SBE_OK;
    if (prior_temperature > 0.0 && !unif_counts) return fail(e, SBE_ERR_ARG, "prior_temperature given without unif_counts (conditionals.py:114)");
    HIPCHK(e, hipSetDevice(e->device));
    const double* d_unif = nullptr;
    if (prior_temperature > 0.0) {
        { int _urc = upload(e, e->d_unif, unif_counts, (size_t)e->F * e->S * sizeof(double)); if (_urc) return _urc; }
        d_unif = e->d_unif;
    }
    int rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    for (int c = 0; c < e->C; ++c) {                       // runs of selected components: adjacent group ranges, one launch
        if (!((component_mask >> c) & 1u) || e->G[c] == 0) continue;
        const int g_lo = e->goff[c];
        int g_hi = g_lo + e->G[c];
        s.probs_set[c] = 1;
        while (c + 1 < e->C && ((component_mask >> (c + 1)) & 1u)) { ++c; g_hi = e->goff[c] + e->G[c]; s.probs_set[c] = 1; }
        k_probs<int32_t><<<div_up((int64_t)(g_hi - g_lo) * e->F, 256), 256, 0, e->stream>>>(
            e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, d_unif,
            e->d_probs + (int64_t)slot * e->table_elems(), g_lo, g_hi, e->F, e->S, temperature, prior_temperature, 1, e->d_status, 0,
            e->d_probs_t + (int64_t)slot * e->probs_t_elems(), e->Gtot, e->ft);   // (+ the tile-transposed copy: one launch)
        HIPCHK(e, hipGetLastError());
    }
    return check_after(e, ST_BAD_NORMALIZE);
}

int sbe_update_probs(sbe_engine* e, int slot, int component, double temperature, double prior_temperature,
                     const double* unif_counts) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    return sbe_update_probs_mask(e, slot, 1u << component, temperature, prior_temperature, unif_counts);
}

int sbe_set_probs(sbe_engine* e, int slot, int component, const float* probs) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) { e->slots[slot].probs_set[component] = 1; return SBE_OK; }
    CHECK_PTR(e, probs);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    float* dst = e->d_probs + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    int rc = h2d(e, dst, probs, n * sizeof(float));
    if (rc) return rc;
    rc = retile_probs(e, slot, component);
    if (rc) return rc;
    e->slots[slot].probs_set[component] = 1;
    return SBE_OK;
}

// every component's table in one call (recalculate_feature_counts reads them all back: counts.py:35-52)
int sbe_get_counts_all(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (e->Gtot == 0) return SBE_OK;
    CHECK_PTR(e, out);
    for (int c = 0; c < e->C; ++c)
        if (e->G[c] > 0 && !e->slots[slot].counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->Gtot * e->F * e->S;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    const int32_t* src = e->d_counts + (int64_t)slot * e->table_elems();
    void* d_out;
    rc = out_target(e, n * sizeof(float), e->d_scratch, &d_out);
    if (rc) return rc;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, 1024), 64);
    const DoneSig done = out_done(e, d_out, blocks);
    if (done.flag) k_i32_to_f32_done<<<blocks, 1024, 0, e->stream>>>(src, (float*)d_out, n, done);
    else k_i32_to_f32<<<div_up(n, 256), 256, 0, e->stream>>>(src, (float*)d_out, n);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, out, d_out, n * sizeof(float), done);
}

int sbe_get_probs(sbe_engine* e, int slot, int component, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) return SBE_OK;
    CHECK_PTR(e, out);
    if (!e->slots[slot].probs_set[component]) return fail(e, SBE_ERR_STATE, "slot %d: probs of component %d not set", slot, component);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    const float* src = e->d_probs + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    return d2h(e, out, src, n * sizeof(float));
}

// ---- weights ------------------------------------------------------------------------------------
int sbe_set_weights(sbe_engine* e, int slot, const float* weights) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, weights);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_patterns_and_weights(e, slot, weights);
    if (rc) return rc;
    e->slots[slot].weights_set = true;
    return SBE_OK;
}

int sbe_get_weights_normalized(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    if (s.patterns_dirty) { int rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t n = (int64_t)e->N * e->F * e->C;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    k_expand_weights<<<div_up(n, 256), 256, 0, e->stream>>>(e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C,
                                                           e->d_pid + (int64_t)slot * e->Np, (float*)e->d_scratch, e->N, e->F, e->C);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch, n * sizeof(float));
}

// ---- a3 / a2 / a6 dense outputs ------------------------------------------------------------------
int sbe_likelihood_per_component(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    int rc = check_slot_ready(e, slot, false);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->N * e->F * e->C;
    const int blocks = div_up(n, 512);                    // (k_lh_dense: two output elements per thread)
    static const bool no_stream = [] { const char* v = getenv("SBE_STREAM_RESULTS"); return v && atoi(v) == 0; }();   // (A/B)
    if ((size_t)n * sizeof(double) >= ((size_t)1 << 19) && !no_stream) {
        // large result: stored by the kernel into host-mapped staging, copied to the caller chunk by chunk on the host
        // pool while the rest crosses PCIe (stream_result)
        const size_t bytes = (size_t)n * sizeof(double);
        rc = ensure_stream(e, bytes);
        if (rc) return rc;
        const StreamPlan plan = plan_stream(e, (unsigned)blocks, (size_t)512 * sizeof(double), bytes);
        k_lh_dense<<<plan.grid, 256, 0, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_probs + (int64_t)slot * e->table_elems(),
            (double*)e->d_stream, e->N, e->Np, e->F, e->S, e->C, e->Fp, plan.sig);
        HIPCHK(e, hipGetLastError());
        constexpr size_t kJob = (size_t)64 << 10;
        const int n_jobs = (int)((bytes + kJob - 1) / kJob);
        uint8_t* dst = (uint8_t*)out;
        const uint8_t* stage = e->h_stream;
        rc = stream_result(e, plan, n_jobs,
                           [=](int j) { return (size_t)j * kJob; },
                           [=](int j) { return std::min(bytes, (size_t)(j + 1) * kJob); },
                           [=](int j) { const size_t o = (size_t)j * kJob; memcpy(dst + o, stage + o, std::min(kJob, bytes - o)); });
        return rc ? rc : synced(e);               // (the call waited for the device: a deferred data check is delivered here)
    }
    rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    k_lh_dense<<<blocks, 256, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_probs + (int64_t)slot * e->table_elems(),
        (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C, e->Fp);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch, n * sizeof(double));
}

int sbe_likelihood_per_component_exact(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source not set", slot);
    for (int c = 0; c < e->C; ++c)
        if (!s.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->N * e->F * e->C;
    int rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    k_lh_exact<<<div_up((int64_t)e->N * e->F, 256), 256, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_gid + (int64_t)slot * e->C * e->Np,
        e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C, e->Fp, e->d_status);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out, e->d_scratch, n * sizeof(double));
    if (rc) return rc;
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE]) return fail(e, SBE_ERR_DATA, "normalize: non-positive row sum in leave-one-out tables (sbayes/util.py:1006 assert)");
    return SBE_OK;
}

int sbe_observation_lh(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    int rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (e->slots[slot].patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t n = (int64_t)e->N * e->F;
    rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    k_observation_lh<<<div_up(n, 256), 256, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_pid + (int64_t)slot * e->Np,
        e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C,
        (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C, e->Fp);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch, n * sizeof(double));
}

// ---- north-star scalar ----------------------------------------------------------------------------
int sbe_mixture_loglik_batch_async(sbe_engine* e, int first_slot, int n) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range [%d,%d) out of range", first_slot, first_slot + n);
    HIPCHK(e, hipSetDevice(e->device));
    return enqueue_mixture(e, first_slot, n, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS);
}

int sbe_fetch_results(sbe_engine* e, int first_slot, int n, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot); CHECK_PTR(e, out);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range out of range");
    HIPCHK(e, hipStreamSynchronize(e->stream));     // results were written straight into mapped host memory
    memcpy(out, e->h_results + first_slot, (size_t)n * sizeof(double));
    return synced(e);
}

int sbe_mixture_loglik_batch(sbe_engine* e, int first_slot, int n, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot); CHECK_PTR(e, out);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range out of range");
    HIPCHK(e, hipSetDevice(e->device));
    DoneSig done;
    int rc = enqueue_mixture(e, first_slot, n, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, &done);
    if (rc) return rc;
    rc = wait_done(e, done);                        // results were written straight into mapped host memory
    if (rc) return rc;
    memcpy(out, e->h_results + first_slot, (size_t)n * sizeof(double));
    return synced(e);
}

int sbe_mixture_loglik(sbe_engine* e, int slot, double* out) { return sbe_mixture_loglik_batch(e, slot, 1, out); }

// ---- collapsed likelihood -------------------------------------------------------------------------
// a7/a8 of the groups [g_lo, g_lo + G) of a slot: per-group float64 into host-mapped memory (and, optionally, the
// float32 per-feature rows into d_pf).  One launch (k_collapsed_groups) when a group's F*S terms fit LDS, the k_dcl /
// k_group_sum_f32 pair otherwise.  Ends with the stream synchronised; results at e->h_io.
static int collapsed_groups(sbe_engine* e, int slot, int g_lo, int G, float* d_pf) {
    int rc = ensure_io(e, (size_t)G * sizeof(double));
    if (rc) return rc;
    const int32_t* counts = e->d_counts + (int64_t)slot * e->table_elems();
    const size_t lds = (size_t)e->F * e->S * sizeof(double) + (size_t)e->F * sizeof(float);
    if (lds <= ((size_t)96 << 10)) {
        const DoneSig done = next_done(e, (unsigned)G);
        k_collapsed_groups<int32_t><<<G, 1024, lds, e->stream>>>(counts, e->d_conc, e->d_lg_conc, e->d_sum_a, e->d_lg_sum_a, d_pf,
                                                                  (double*)e->d_io, g_lo, e->F, e->S, done);
        HIPCHK(e, hipGetLastError());
        rc = wait_done(e, done);
        if (rc) return rc;
        return synced(e);
    } else {
        float* pf = d_pf;
        if (!pf) {
            rc = ensure_scratch(e, (size_t)G * e->F * sizeof(float));
            if (rc) return rc;
            pf = (float*)e->d_scratch;
        }
        k_dcl<int32_t><<<div_up((int64_t)G * e->F, 256), 256, 0, e->stream>>>(counts, e->d_conc, pf, g_lo, g_lo + G, e->F, e->S, 1);
        HIPCHK(e, hipGetLastError());
        k_group_sum_f32<<<div_up((int64_t)G * 8, 64), 64, 0, e->stream>>>(pf, (double*)e->d_io, G, e->F);
        HIPCHK(e, hipGetLastError());
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return synced(e);
}

int sbe_collapsed_loglik_all(sbe_engine* e, int slot, double* per_group_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, per_group_out);
    Slot& s = e->slots[slot];
    for (int c = 0; c < e->C; ++c) {
        if (e->G[c] == 0) continue;
        if (!s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
    }
    HIPCHK(e, hipSetDevice(e->device));
    int rc = collapsed_groups(e, slot, 0, e->Gtot, nullptr);
    if (rc) return rc;
    memcpy(per_group_out, e->h_io, (size_t)e->Gtot * sizeof(double));
    return SBE_OK;
}

int sbe_collapsed_loglik(sbe_engine* e, int slot, int component, double* per_group_out, float* per_feature_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) return SBE_OK;       // no group: Likelihood.compute_lh_clusters sums an empty cache
    CHECK_PTR(e, per_group_out);
    Slot& s = e->slots[slot];
    if (!s.counts_set[component]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, component);
    if (!e->conc_set[component]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", component);
    HIPCHK(e, hipSetDevice(e->device));
    const int G = e->G[component], g_lo = e->goff[component];
    float* d_pf = nullptr;
    int rc = SBE_OK;
    if (per_feature_out) {
        rc = ensure_scratch(e, (size_t)G * e->F * sizeof(float));
        if (rc) return rc;
        d_pf = (float*)e->d_scratch;
    }
    rc = collapsed_groups(e, slot, g_lo, G, d_pf);  // (the G doubles land in host-mapped memory: no copy-engine hop in the chain)
    if (rc) return rc;
    memcpy(per_group_out, e->h_io, (size_t)G * sizeof(double));
    if (per_feature_out) return d2h(e, per_feature_out, d_pf, (size_t)G * e->F * sizeof(float));
    return SBE_OK;
}


// =============================================================================================
// Stateless entry points: the reference's free functions, one call each, no slot involved.
// They run the same kernels on scratch buffers.
// =============================================================================================
int sbe_normalize_tables(sbe_engine* e, const float* counts, int n_groups, const double* conc, int conc_per_group,
                         double temperature, double prior_temperature, const double* unif_counts, float* out) {
    CHECK_ENGINE(e);
    if (n_groups == 0) return SBE_OK;              // empty table set (a component without groups)
    CHECK_PTR(e, counts); CHECK_PTR(e, conc); CHECK_PTR(e, out);
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    if (prior_temperature > 0.0 && !unif_counts) return fail(e, SBE_ERR_ARG, "prior_temperature given without unif_counts (conditionals.py:114)");
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S, n = (int64_t)n_groups * fs;
    const size_t cb = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    const size_t ab = ((size_t)(conc_per_group ? n : fs) * sizeof(double) + 255) / 256 * 256;
    const size_t ub = ((size_t)fs * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 2 * cb + ab + ub);
    if (rc) return rc;
    float* d_cnt = (float*)e->d_scratch;
    double* d_a = (double*)(e->d_scratch + cb);
    double* d_u = (double*)(e->d_scratch + cb + ab);
    float* d_out = (float*)(e->d_scratch + cb + ab + ub);
    { int _urc = upload(e, d_cnt, counts, (size_t)n * sizeof(float)); if (_urc) return _urc; }
    { int _urc = upload(e, d_a, conc, (size_t)(conc_per_group ? n : fs) * sizeof(double)); if (_urc) return _urc; }
    const double* d_unif = nullptr;
    if (prior_temperature > 0.0) {
        { int _urc = upload(e, d_u, unif_counts, (size_t)fs * sizeof(double)); if (_urc) return _urc; }
        d_unif = d_u;
    }
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    k_probs<float><<<div_up((int64_t)n_groups * e->F, 256), 256, 0, e->stream>>>(
        d_cnt, d_a, d_unif, d_out, 0, n_groups, e->F, e->S, temperature, prior_temperature, conc_per_group ? 1 : 0, e->d_status);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out, d_out, (size_t)n * sizeof(float));
    if (rc) return rc;
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE])
        return fail(e, SBE_ERR_DATA, "normalize: %d rows have a non-positive sum (sbayes/util.py:1006 assert)", e->h_status[ST_BAD_NORMALIZE]);
    return SBE_OK;
}

int sbe_dirichlet_logpdf(sbe_engine* e, const float* counts, int n_groups, const double* conc, int conc_per_group,
                         float* per_feature_out, double* per_group_out) {
    CHECK_ENGINE(e);
    if (n_groups == 0) return SBE_OK;
    CHECK_PTR(e, counts); CHECK_PTR(e, conc);
    if (!per_feature_out && !per_group_out) return fail(e, SBE_ERR_ARG, "no output requested");
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S, n = (int64_t)n_groups * fs;
    const size_t cb = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    const size_t ab = ((size_t)(conc_per_group ? n : fs) * sizeof(double) + 255) / 256 * 256;
    const size_t pf = ((size_t)n_groups * e->F * sizeof(float) + 255) / 256 * 256;
    int rc = ensure_scratch(e, cb + ab + pf + (size_t)n_groups * sizeof(double));
    if (rc) return rc;
    float* d_cnt = (float*)e->d_scratch;
    double* d_a = (double*)(e->d_scratch + cb);
    float* d_pf = (float*)(e->d_scratch + cb + ab);
    double* d_pg = (double*)(e->d_scratch + cb + ab + pf);
    { int _urc = upload(e, d_cnt, counts, (size_t)n * sizeof(float)); if (_urc) return _urc; }
    { int _urc = upload(e, d_a, conc, (size_t)(conc_per_group ? n : fs) * sizeof(double)); if (_urc) return _urc; }
    k_dcl<float><<<div_up((int64_t)n_groups * e->F, 256), 256, 0, e->stream>>>(d_cnt, d_a, d_pf, 0, n_groups, e->F, e->S, conc_per_group ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    if (per_group_out) {
        k_group_sum_f32<<<div_up((int64_t)n_groups * 8, 64), 64, 0, e->stream>>>(d_pf, d_pg, n_groups, e->F);
        HIPCHK(e, hipGetLastError());
        rc = d2h(e, per_group_out, d_pg, (size_t)n_groups * sizeof(double));
        if (rc) return rc;
    }
    if (per_feature_out) return d2h(e, per_feature_out, d_pf, (size_t)n_groups * e->F * sizeof(float));
    return SBE_OK;
}

int sbe_effect_counts(sbe_engine* e, const uint8_t* groups, int n_groups, const uint8_t* source_is_component,
                      const int32_t* objects, int n_subset, float* out) {
    CHECK_ENGINE(e);
    if (n_groups == 0) return SBE_OK;              // compute_effect_counts of an empty group matrix: empty counts
    CHECK_PTR(e, groups); CHECK_PTR(e, source_is_component); CHECK_PTR(e, out);
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    if (n_subset < -1 || (n_subset > 0 && !objects)) return fail(e, SBE_ERR_ARG, "bad object subset");
    for (int i = 0; i < n_subset; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, F = e->F, S = e->S;
    const int64_t n_out = (int64_t)n_groups * F * S;
    const int n_listed = n_subset < 0 ? N : n_subset;
    const size_t gb = ((size_t)n_groups * N + 255) / 256 * 256;
    const size_t mb = ((size_t)N * F + 255) / 256 * 256;
    const size_t ob = ((size_t)std::max(n_listed, 1) * sizeof(int32_t) + 255) / 256 * 256;
    const size_t cb = ((size_t)n_out * sizeof(int32_t) + 255) / 256 * 256;
    int rc = ensure_scratch(e, gb + mb + ob + 2 * cb);
    if (rc) return rc;
    uint8_t* d_groups = e->d_scratch;
    uint8_t* d_mask = e->d_scratch + gb;
    int32_t* d_obj = (int32_t*)(e->d_scratch + gb + mb);
    int32_t* d_cnt = (int32_t*)(e->d_scratch + gb + mb + ob);
    float* d_out = (float*)(e->d_scratch + gb + mb + ob + cb);
    { int _urc = upload(e, d_groups, groups, (size_t)n_groups * N); if (_urc) return _urc; }
    { int _urc = upload(e, d_mask, source_is_component, (size_t)N * F); if (_urc) return _urc; }
    if (n_subset > 0) { int _urc = upload(e, d_obj, objects, (size_t)n_subset * sizeof(int32_t)); if (_urc) return _urc; }
    HIPCHK(e, hipMemsetAsync(d_cnt, 0, (size_t)n_out * sizeof(int32_t), e->stream));
    if (n_listed > 0) {
        int ft = 32;
        auto lds_for = [&](int t) { return (size_t)n_groups * t * S * sizeof(int32_t); };
        while (ft > 1 && lds_for(ft) > 64 * 1024) ft >>= 1;
        if (lds_for(ft) > 150 * 1024) return fail(e, SBE_ERR_ARG, "effect counts: %d groups x %d states exceed the LDS histogram", n_groups, S);
        const int n_ftiles = div_up(F, ft);
        int chunks = std::max(1, std::min(div_up(n_listed, 32), div_up(2 * e->compute_units, n_ftiles)));
        const int opc = div_up(n_listed, chunks);
        chunks = div_up(n_listed, opc);
        k_effect_counts<<<dim3(n_ftiles, chunks), kBlock, lds_for(ft), e->stream>>>(
            e->d_state, d_groups, d_mask, n_subset >= 0 ? d_obj : nullptr, n_listed, opc, N, F, S, e->Fp, n_groups, ft, d_cnt);
        HIPCHK(e, hipGetLastError());
    }
    k_i32_to_f32<<<div_up(n_out, 256), 256, 0, e->stream>>>(d_cnt, d_out, n_out);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_out, (size_t)n_out * sizeof(float));
}

int sbe_normalize_weights(sbe_engine* e, const float* weights, int n_comp, const uint8_t* has_components, int n_rows,
                          float* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, weights);
    if (n_comp < 1 || n_comp > kMaxComponents) return fail(e, SBE_ERR_ARG, "n_comp=%d unsupported (1..%d)", n_comp, kMaxComponents);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, has_components); CHECK_PTR(e, out);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = n_rows, F = e->F, C = n_comp;     // the rows are whatever the caller hands over (has_components[available], operators.py:1086)
    // row form (k_normalize_weight_rows): no pattern sort on either side, one launch; a few rows read the weights out of
    // the mapped staging ring, many rows out of device memory (every block reads them once)
    const size_t wb = al256((size_t)F * C * sizeof(float)), hb = al256((size_t)N * C);
    const int64_t n_out = (int64_t)N * F * C;
    int rc = ensure_scratch(e, wb + hb + (size_t)n_out * sizeof(float));
    if (rc) return rc;
    const void *v_w = e->d_scratch, *v_hc;
    const size_t w_lds = (size_t)F * C * sizeof(float);
    const bool in_lds = w_lds <= ((size_t)64 << 10);              // (beyond that the blocks read the device copy in place)
    if (N <= 256 && in_lds) rc = stage(e, weights, w_lds, e->d_scratch, &v_w);
    else rc = upload(e, e->d_scratch, weights, w_lds);
    if (rc) return rc;
    rc = stage(e, has_components, (size_t)N * C, e->d_scratch + wb, &v_hc);
    if (rc) return rc;
    void* d_out;
    rc = out_target(e, (size_t)n_out * sizeof(float), e->d_scratch + wb + hb, &d_out);
    if (rc) return rc;
    const DoneSig done = out_done(e, d_out, (unsigned)div_up(N, kNwRows));
    k_normalize_weight_rows<<<div_up(N, kNwRows), 256, in_lds ? w_lds : 0, e->stream>>>(
        (const float*)v_w, (const uint8_t*)v_hc, (float*)d_out, N, F, C, in_lds ? 1 : 0, done);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, out, d_out, (size_t)n_out * sizeof(float), done);
}

// ---- cluster-membership marginals ---------------------------------------------
int sbe_cluster_marginals(sbe_engine* e, int slot, const float* table, const int32_t* objects, int n_objects_av,
                          double prior_temperature, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, table); CHECK_PTR(e, out);
    if (n_objects_av < 0) return fail(e, SBE_ERR_ARG, "n_objects_av=%d", n_objects_av);
    if (n_objects_av == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    if (!(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "prior_temperature must be positive");
    for (int i = 0; i < n_objects_av; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    int rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    Slot& s = e->slots[slot];
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int F = e->F, S = e->S, C = e->C;
    const size_t tb = ((size_t)F * S * sizeof(float) + 255) / 256 * 256;
    const size_t ob = ((size_t)n_objects_av * sizeof(int32_t) + 255) / 256 * 256;
    const size_t out_bytes = (size_t)2 * n_objects_av * sizeof(double);
    // table, object list and the result live in host-mapped memory (a few KB each): ONE kernel and one
    // synchronisation, no copy-engine operation in the chain
    rc = ensure_io(e, tb + ob + out_bytes);
    if (rc) return rc;
    memcpy(e->h_io, table, (size_t)F * S * sizeof(float));
    memcpy(e->h_io + tb, objects, (size_t)n_objects_av * sizeof(int32_t));
    const float* d_tab = (const float*)e->d_io;
    const int32_t* d_obj = (const int32_t*)(e->d_io + tb);
    double* d_out = (double*)(e->d_io + tb + ob);
    const double inv = 1.0 / prior_temperature;
    const DoneSig done = next_done(e, (unsigned)n_objects_av);
    k_cluster_marginals<<<n_objects_av, kBlock, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
        e->d_probs + (int64_t)slot * e->table_elems(), d_tab, e->d_weights + (int64_t)slot * F * C,
        e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0, d_obj, n_objects_av, d_out,
        reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + tb + ob, out_bytes);
    return synced(e);
}

// ---- ClusterJump.get_jump_lh / expected_confounder_features (operators.py:1679-1722, 1342-1379) -----------------
int sbe_jump_lh(sbe_engine* e, int slot, const float* pconf, const float* p_source, const float* p_target,
                const int32_t* objects, int n_members, double prior_temperature, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, p_source); CHECK_PTR(e, p_target); CHECK_PTR(e, out);
    if (n_members < 0) return fail(e, SBE_ERR_ARG, "n_members=%d", n_members);
    if (n_members == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    if (!(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "prior_temperature must be positive");
    for (int i = 0; i < n_members; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / weights not set", slot);
    const int n_conf_groups = e->Gtot - e->G[0];
    if (n_conf_groups > 0) CHECK_PTR(e, pconf);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int F = e->F, S = e->S, C = e->C;
    const size_t fs = (size_t)F * S * sizeof(float);
    const size_t cb = ((size_t)std::max(n_conf_groups, 1) * fs + 255) / 256 * 256;
    const size_t tb = (fs + 255) / 256 * 256;
    const size_t ob = ((size_t)n_members * sizeof(int32_t) + 255) / 256 * 256;
    const size_t out_bytes = (size_t)2 * n_members * sizeof(double);
    // tables, member list and result in host-mapped memory: ONE kernel, one synchronisation (as sbe_cluster_marginals)
    rc = ensure_io(e, cb + 2 * tb + ob + out_bytes);
    if (rc) return rc;
    if (n_conf_groups > 0) memcpy(e->h_io, pconf, (size_t)n_conf_groups * fs);
    memcpy(e->h_io + cb, p_source, fs);
    memcpy(e->h_io + cb + tb, p_target, fs);
    memcpy(e->h_io + cb + 2 * tb, objects, (size_t)n_members * sizeof(int32_t));
    const double inv = 1.0 / prior_temperature;
    const DoneSig done = next_done(e, (unsigned)n_members);
    k_jump_lh<<<n_members, kBlock, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
        (const float*)e->d_io, (const float*)(e->d_io + cb), (const float*)(e->d_io + cb + tb),
        e->d_weights + (int64_t)slot * F * C, e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0,
        (const int32_t*)(e->d_io + cb + 2 * tb), n_members, (double*)(e->d_io + cb + 2 * tb + ob),
        reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, e->G[0], done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + cb + 2 * tb + ob, out_bytes);
    return synced(e);
}

// ---- GibbsSampleWeights.source_lh_by_feature (operators.py:677-685) -----------------------------------------------
int sbe_source_lh_by_feature(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    rc = ensure_io(e, (size_t)e->F * sizeof(float));
    if (rc) return rc;
    const DoneSig done = next_done(e, (unsigned)div_up(e->F, kSlfFT));
    k_source_lh_by_feature<<<div_up(e->F, kSlfFT), 1024, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_pid + (int64_t)slot * e->Np,
        e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, (float*)e->d_io, e->N, e->F, e->C, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io, (size_t)e->F * sizeof(float));
    return synced(e);
}

// ---- data-parallel cores of Gibbs source resampling ---------------------------
namespace {

// Shared front end of the source-posterior family: argument checks, object upload, kernel arguments.
// Scratch layout: [objects | extra bytes requested by the caller].
int source_posterior_setup(sbe_engine* e, int slot, const int32_t* objects, int n_sub, double temperature,
                           double prior_temperature, int from_prior, size_t extra_bytes, SrcPostArgs* a,
                           uint8_t** d_extra) {
    if (n_sub < 0) return fail(e, SBE_ERR_ARG, "n_sub=%d", n_sub);
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    for (int i = 0; i < n_sub; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    int rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (e->slots[slot].patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const size_t ob = ((size_t)n_sub * sizeof(int32_t) + 255) / 256 * 256;
    rc = ensure_scratch(e, ob + extra_bytes);
    if (rc) return rc;
    *d_extra = e->d_scratch + ob;
    // the object list is read once per thread: out of the host-mapped staging ring in place (no copy operation in front of
    // the kernel); lists beyond the ring's direct size go up with a copy
    const void* v_obj;
    rc = stage(e, objects, (size_t)n_sub * sizeof(int32_t), e->d_scratch, &v_obj);
    if (rc) return rc;
    const int32_t* d_obj = (const int32_t*)v_obj;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    *a = SrcPostArgs{e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_pid + (int64_t)slot * e->Np,
                     e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C,
                     d_obj, n_sub, e->Np, e->F, e->S, e->C, e->Fp, inv_t, (float)inv_tp, inv_t != 1.0, inv_tp != 1.0,
                     from_prior != 0};
    return SBE_OK;
}

int source_posterior_status(sbe_engine* e, bool waited = false) {
    int rc = waited ? take_status(e) : read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE])
        return fail(e, SBE_ERR_DATA, "normalize: %d observations have a non-positive posterior sum (sbayes/util.py:1006 assert)", e->h_status[ST_BAD_NORMALIZE]);
    return SBE_OK;
}

// log_q = sum_i log(p_sel[i]) (fp64, fixed order) -> *out; optionally the selected probabilities themselves.
int finish_log_q(sbe_engine* e, const float* d_psel, int64_t n, double* d_partials, double* log_q_out, float* p_selected_out) {
    const int nb = (int)std::min<int64_t>(div_up(n, 4 * kBlock), 256);
    k_sum_log_f32<<<nb, kBlock, 0, e->stream>>>(d_psel, n, d_partials);
    HIPCHK(e, hipGetLastError());
    k_reduce_partials<<<1, kBlock, 0, e->stream>>>(d_partials, 0, nb, d_partials + 256, 0, 1, StepFinish{});
    HIPCHK(e, hipGetLastError());
    int rc = d2h(e, log_q_out, d_partials + 256, sizeof(double));
    if (rc) return rc;
    if (p_selected_out) { rc = d2h(e, p_selected_out, d_psel, (size_t)n * sizeof(float)); if (rc) return rc; }
    return source_posterior_status(e);
}

}  // namespace

int sbe_source_posterior(sbe_engine* e, int slot, const int32_t* objects, int n_sub, double temperature,
                         double prior_temperature, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    const int64_t n_out = (int64_t)std::max(n_sub, 0) * e->F * e->C;
    SrcPostArgs a; uint8_t* d_extra = nullptr;
    int rc = source_posterior_setup(e, slot, objects, n_sub, temperature, prior_temperature, 0, (size_t)n_out * sizeof(float), &a, &d_extra);
    if (rc || n_sub == 0) return rc;
    void* d_out;
    rc = out_target(e, (size_t)n_out * sizeof(float), d_extra, &d_out);
    if (rc) return rc;
    const unsigned nb = (unsigned)div_up((int64_t)n_sub * e->F, 256);
    const DoneSig done = out_done(e, d_out, nb);
    k_source_posterior<<<nb, 256, 0, e->stream>>>(a, (float*)d_out, e->d_status, done);
    HIPCHK(e, hipGetLastError());
    rc = out_fetch(e, out, d_out, (size_t)n_out * sizeof(float), done);
    if (rc) return rc;
    return source_posterior_status(e, /*waited=*/true);
}

int sbe_sample_source(sbe_engine* e, int slot, int dst_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z, double* log_q_out,
                      float* p_selected_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_SLOT(e, dst_slot); CHECK_PTR(e, log_q_out);
    if (!e->slots[dst_slot].source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set (rows outside the subset would be undefined)", dst_slot);
    if (n_sub == 0) { *log_q_out = 0.0; return SBE_OK; }
    const int64_t n_obs = (int64_t)std::max(n_sub, 0) * e->F;
    const size_t zb = z ? ((size_t)n_obs * sizeof(double) + 255) / 256 * 256 : 0;    // z == NULL: device Philox stream
    const size_t pb = ((size_t)n_obs * sizeof(float) + 255) / 256 * 256;
    SrcPostArgs a; uint8_t* d_extra = nullptr;
    int rc = source_posterior_setup(e, slot, objects, n_sub, temperature, prior_temperature, from_prior, zb + pb + 257 * sizeof(double), &a, &d_extra);
    if (rc) return rc;
    double* d_z = (double*)d_extra;
    float* d_psel = (float*)(d_extra + zb);
    double* d_partials = (double*)(d_extra + zb + pb);
    if (z) { int _urc = upload(e, d_z, z, (size_t)n_obs * sizeof(double)); if (_urc) return _urc; }
    k_sample_source<<<div_up(n_obs, 256), 256, 0, e->stream>>>(a, z ? d_z : nullptr, e->rng_seed, e->rng_draw,
                                                               e->d_src + (int64_t)dst_slot * e->N * e->Fp, d_psel, e->d_status, nullptr);
    bump_src(e, dst_slot);
    if (!z) ++e->rng_draw;
    HIPCHK(e, hipGetLastError());
    return finish_log_q(e, d_psel, n_obs, d_partials, log_q_out, p_selected_out);
}

int sbe_set_rng(sbe_engine* e, uint64_t seed, uint64_t draw) {
    CHECK_ENGINE(e);
    e->rng_seed = seed;
    e->rng_draw = draw;
    return SBE_OK;
}

int sbe_get_rng(sbe_engine* e, uint64_t* seed, uint64_t* draw) {
    CHECK_ENGINE(e); CHECK_PTR(e, seed); CHECK_PTR(e, draw);
    *seed = e->rng_seed;
    *draw = e->rng_draw;
    return SBE_OK;
}

int sbe_test_philox(sbe_engine* e, const uint32_t* ctr_key, int n, uint32_t* out) {
    CHECK_ENGINE(e);
    if (n <= 0) return SBE_OK;
    CHECK_PTR(e, ctr_key); CHECK_PTR(e, out);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t ib = ((size_t)n * 6 * sizeof(uint32_t) + 255) / 256 * 256;
    int rc = ensure_scratch(e, ib + (size_t)n * 4 * sizeof(uint32_t));
    if (rc) return rc;
    { int _urc = upload(e, e->d_scratch, ctr_key, (size_t)n * 6 * sizeof(uint32_t)); if (_urc) return _urc; }
    k_test_philox<<<div_up(n, 256), 256, 0, e->stream>>>((const uint32_t*)e->d_scratch, n, (uint32_t*)(e->d_scratch + ib));
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch + ib, (size_t)n * 4 * sizeof(uint32_t));
}

int sbe_source_logprob(sbe_engine* e, int slot, int src_slot, const int32_t* objects, int n_sub, double temperature,
                       double prior_temperature, int from_prior, double* log_q_out, float* p_selected_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_SLOT(e, src_slot); CHECK_PTR(e, log_q_out);
    if (!e->slots[src_slot].source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set", src_slot);
    if (n_sub == 0) { *log_q_out = 0.0; return SBE_OK; }
    const int64_t n_obs = (int64_t)std::max(n_sub, 0) * e->F;
    const size_t pb = ((size_t)n_obs * sizeof(float) + 255) / 256 * 256;
    SrcPostArgs a; uint8_t* d_extra = nullptr;
    int rc = source_posterior_setup(e, slot, objects, n_sub, temperature, prior_temperature, from_prior, pb + 257 * sizeof(double), &a, &d_extra);
    if (rc) return rc;
    float* d_psel = (float*)d_extra;
    double* d_partials = (double*)(d_extra + pb);
    k_source_logprob<<<div_up(n_obs, 256), 256, 0, e->stream>>>(a, e->d_src + (int64_t)src_slot * e->N * e->Fp, d_psel, e->d_status, nullptr);
    HIPCHK(e, hipGetLastError());
    return finish_log_q(e, d_psel, n_obs, d_partials, log_q_out, p_selected_out);
}

int sbe_subset_lh(sbe_engine* e, const int32_t* objects, int n_sub, int n_comp, const float* tables,
                  const int32_t* table_offsets, int n_tables_total, const int32_t* group_idx, double temperature,
                  float* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, out);
    if (n_sub < 0 || n_comp < 1 || n_comp > kMaxComponents || n_tables_total < 1) return fail(e, SBE_ERR_ARG, "bad sizes");
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, tables); CHECK_PTR(e, table_offsets); CHECK_PTR(e, group_idx);
    if (!(temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperature must be positive");
    for (int i = 0; i < n_sub; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    for (int c = 0; c < n_comp; ++c) {
        const int hi = (c + 1 < n_comp ? table_offsets[c + 1] : n_tables_total) - table_offsets[c];
        if (table_offsets[c] < 0 || hi < 0) return fail(e, SBE_ERR_ARG, "bad table offsets");
        for (int i = 0; i < n_sub; ++i)
            if (group_idx[(size_t)c * n_sub + i] >= hi) return fail(e, SBE_ERR_ARG, "group index out of range in component %d", c);
    }
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S;
    const size_t tb = ((size_t)n_tables_total * fs * sizeof(float) + 255) / 256 * 256;
    const size_t ob = ((size_t)n_sub * sizeof(int32_t) + 255) / 256 * 256;
    const size_t gb = ((size_t)n_comp * n_sub * sizeof(int32_t) + 255) / 256 * 256;
    const int64_t n_out = (int64_t)n_sub * e->F * n_comp;
    int rc = ensure_scratch(e, tb + ob + gb + 256 + (size_t)n_out * sizeof(float));
    if (rc) return rc;
    float* d_tab = (float*)e->d_scratch;
    int32_t* d_obj = (int32_t*)(e->d_scratch + tb);
    int32_t* d_gi = (int32_t*)(e->d_scratch + tb + ob);
    int32_t* d_off = (int32_t*)(e->d_scratch + tb + ob + gb);
    float* d_out = (float*)(e->d_scratch + tb + ob + gb + 256);
    { int _urc = upload(e, d_tab, tables, (size_t)n_tables_total * fs * sizeof(float)); if (_urc) return _urc; }
    { int _urc = upload(e, d_obj, objects, (size_t)n_sub * sizeof(int32_t)); if (_urc) return _urc; }
    { int _urc = upload(e, d_gi, group_idx, (size_t)n_comp * n_sub * sizeof(int32_t)); if (_urc) return _urc; }
    { int _urc = upload(e, d_off, table_offsets, (size_t)n_comp * sizeof(int32_t)); if (_urc) return _urc; }
    const double inv_t = 1.0 / temperature;
    k_subset_lh<<<div_up((int64_t)n_sub * e->F, 256), 256, 0, e->stream>>>(
        e->d_state, d_tab, d_off, d_gi, d_obj, n_sub, d_out, e->F, e->S, n_comp, e->Fp, (float)inv_t, inv_t != 1.0);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_out, (size_t)n_out * sizeof(float));
}

// ---- round 3: delta / resident forms for the drop-in host layer ----------------------------------------------------
// What the unchanged reference sampler asks per MCMC step goes over PCIe as object lists and a few changed rows
// (last row): no [N][F] mask, no whole [G][F][S] table.
namespace {

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


}  // namespace

int sbe_set_uniform_counts(sbe_engine* e, const double* unif_counts) {
    CHECK_ENGINE(e); CHECK_PTR(e, unif_counts);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload(e, e->d_unif_res, unif_counts, (size_t)e->F * e->S * sizeof(double));
    if (rc) return rc;
    e->unif_set = true;
    return SBE_OK;
}

// `follow_slot` >= 0 (sbe_counts_delta_apply): the slot's resident counts -- the OLD state's -- take the difference, and with
// `follow_probs` the probability rows of the touched groups are rebuilt: inside the tile kernel (the usual case), by one
// more kernel behind the general one.
static int counts_delta_impl(sbe_engine* e, int follow_slot, int follow_probs, int follow_source, const int32_t* objects, int n_subset, const int32_t* gid_old,
                             const int32_t* gid_new, const uint8_t* src_old, const uint8_t* src_new, const int32_t* touched, int n_touched,
                             float* out_diff) {
    CHECK_ENGINE(e);
    if (n_subset < 0 || n_touched < 0) return fail(e, SBE_ERR_ARG, "n_subset=%d n_touched=%d", n_subset, n_touched);
    if (n_touched == 0) return SBE_OK;
    CHECK_PTR(e, touched); CHECK_PTR(e, out_diff);
    const int F = e->F, S = e->S, C = e->C;
    const size_t out_bytes = (size_t)n_touched * F * S * sizeof(float);
    if (n_subset == 0) { memset(out_diff, 0, out_bytes); return SBE_OK; }
    CHECK_PTR(e, objects); CHECK_PTR(e, gid_old); CHECK_PTR(e, gid_new); CHECK_PTR(e, src_old); CHECK_PTR(e, src_new);
    int rc = check_objects(e, objects, n_subset);
    if (rc) return rc;
    std::vector<int32_t> comp(n_touched);
    for (int t = 0; t < n_touched; ++t) {
        if (touched[t] < 0 || touched[t] >= e->Gtot) return fail(e, SBE_ERR_ARG, "touched group %d out of range [0,%d)", touched[t], e->Gtot);
        int c = 0;
        while (c + 1 < C && touched[t] >= e->goff[c + 1]) ++c;
        comp[t] = c;
    }
    for (int64_t i = 0; i < (int64_t)C * n_subset; ++i)
        if (gid_old[i] < -1 || gid_old[i] >= e->Gtot || gid_new[i] < -1 || gid_new[i] >= e->Gtot)
            return fail(e, SBE_ERR_ARG, "group index out of range in the subset's ids");
    DeltaFollow follow{};
    if (follow_slot >= 0) {
        const Slot& sl = e->slots[follow_slot];
        for (int t = 0; t < n_touched; ++t) {
            if (!sl.counts_set[comp[t]])
                return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set (sbe_counts_delta_apply adds to resident tables)", follow_slot, comp[t]);
            if (follow_probs && (!sl.probs_set[comp[t]] || !e->conc_set[comp[t]]))
                return fail(e, SBE_ERR_STATE, "slot %d: probability tables / concentration of component %d not set (update_probs = 1 rebuilds "
                            "the rows of tables that exist: sbe_update_probs first)", follow_slot, comp[t]);
        }
        if (follow_source && !sl.source_set)
            return fail(e, SBE_ERR_STATE, "slot %d: source not set (update_source = 1 patches resident rows)", follow_slot);
        if (follow_source) follow.src = e->d_src + (int64_t)follow_slot * e->N * e->Fp;
        follow.counts = e->d_counts + (int64_t)follow_slot * e->table_elems();
        if (follow_probs) {
            follow.conc = e->d_conc; follow.probs = e->d_probs + (int64_t)follow_slot * e->table_elems();
            follow.probs_t = e->d_probs_t + (int64_t)follow_slot * e->probs_t_elems(); follow.status = e->d_status; follow.ft = e->ft;
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (follow.probs) { rc = clear_status_word(e, ST_BAD_NORMALIZE); if (rc) return rc; }
    // inputs (a few KB) in host-mapped memory, read by the kernel in place; the diff rows come back the same way when
    // they are small, through the staging copy otherwise
    const size_t ob = al256((size_t)n_subset * 4), gb = al256((size_t)C * n_subset * 4), sb = al256((size_t)n_subset * F);
    const size_t tb = al256((size_t)n_touched * 4);
    const bool mapped_out = out_bytes <= ((size_t)1 << 18);
    rc = ensure_io(e, ob + 2 * gb + 2 * sb + 2 * tb + (mapped_out ? out_bytes : 0));
    if (rc) return rc;
    uint8_t* h = e->h_io;
    size_t o = 0;
    const size_t o_obj = o;  memcpy(h + o, objects, (size_t)n_subset * 4); o += ob;
    const size_t o_go = o;   memcpy(h + o, gid_old, (size_t)C * n_subset * 4); o += gb;
    const size_t o_gn = o;   memcpy(h + o, gid_new, (size_t)C * n_subset * 4); o += gb;
    const size_t o_so = o;   memcpy(h + o, src_old, (size_t)n_subset * F); o += sb;
    const size_t o_sn = o;   memcpy(h + o, src_new, (size_t)n_subset * F); o += sb;
    const size_t o_t = o;    memcpy(h + o, touched, (size_t)n_touched * 4); o += tb;
    const size_t o_tc = o;   memcpy(h + o, comp.data(), (size_t)n_touched * 4); o += tb;
    // small subsets (the usual update_feature_counts: a few dozen objects): one launch, a block per 16-feature tile stages
    // what it needs of the mapped block into LDS in one PCIe round trip and serves every touched group (k_counts_delta_tile)
    const size_t tile_lds = ((size_t)n_touched * kDeltaFT * S + (size_t)e->Gtot + (size_t)n_touched + (size_t)n_subset * (1 + 2 * C)) * sizeof(int32_t) +
                            (size_t)2 * n_subset * kDeltaFT;
    if (mapped_out && n_subset <= kDeltaTileMaxN && tile_lds <= ((size_t)64 << 10) && e->opt_fuse_tables) {
        const uint8_t* din = e->d_io;
        float* d_out = (float*)(e->d_io + o);
        const unsigned blocks = (unsigned)div_up(F, kDeltaFT);
        const DoneSig done = next_done(e, blocks);
        k_counts_delta_tile<<<blocks, kBlock, tile_lds, e->stream>>>(
            e->d_state, (const int32_t*)(din + o_obj), n_subset, (const int32_t*)(din + o_go), (const int32_t*)(din + o_gn),
            din + o_so, din + o_sn, (const int32_t*)(din + o_t), n_touched, d_out, F, S, e->Fp, C, e->Gtot, done, follow);
        HIPCHK(e, hipGetLastError());
        rc = wait_done(e, done);                             // (the difference is complete; the slot follows behind the flag)
        if (!rc) rc = synced(e);
        if (rc) return rc;
        memcpy(out_diff, h + o, out_bytes);
        // the rebuilt rows may raise normalize's data check after the flag: reported like a setter's (deferred mode: by the
        // next call that waits for the device)
        return follow.probs ? check_after(e, ST_BAD_NORMALIZE, "sbe_counts_delta_apply") : SBE_OK;
    }
    // larger subsets: the kernel walks the listed objects one after another (ids, then the object's rows): out of
    // host-mapped memory every step of that walk would be a PCIe round trip, so the packed inputs go to device memory with
    // ONE copy from the pinned block; the diff rows come back through the mapped block (posted writes)
    const size_t in_bytes = o;
    const bool out_in_block = mapped_out && !follow.counts;     // (a following slot reads the rows after the call has returned: device memory)
    rc = ensure_scratch(e, in_bytes + (out_in_block ? 0 : out_bytes));
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(e->d_scratch, h, in_bytes, hipMemcpyHostToDevice, e->stream));
    const uint8_t* din = e->d_scratch;
    float* d_out = out_in_block ? (float*)(e->d_io + o) : (float*)(e->d_scratch + in_bytes);
    const DoneSig done = out_in_block ? next_done(e, (unsigned)(n_touched * div_up(F, kDeltaFT))) : DoneSig{};
    k_counts_delta<<<dim3(n_touched, div_up(F, kDeltaFT)), kBlock, (size_t)kDeltaFT * S * sizeof(int32_t), e->stream>>>(
        e->d_state, (const int32_t*)(din + o_obj), n_subset, (const int32_t*)(din + o_go), (const int32_t*)(din + o_gn),
        din + o_so, din + o_sn, (const int32_t*)(din + o_t), (const int32_t*)(din + o_tc), d_out, F, S, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    if (follow.counts) {
        k_add_count_rows<<<div_up((int64_t)n_touched * F, 256), 256, 0, e->stream>>>(d_out, (const int32_t*)(din + o_t), n_touched, F, S, e->Gtot, follow);
        HIPCHK(e, hipGetLastError());
    }
    if (follow.src) {
        k_set_source_ids<<<div_up((int64_t)n_subset * F, 256), 256, 0, e->stream>>>(din + o_sn, (const int32_t*)(din + o_obj), n_subset, F, e->Fp, follow.src);
        HIPCHK(e, hipGetLastError());
    }
    if (!out_in_block) {
        rc = d2h(e, out_diff, d_out, out_bytes);
        if (rc || !follow.probs) return rc;
        return sync_and_report(e);
    }
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out_diff, h + o, out_bytes);
    return synced(e);
}

int sbe_counts_delta(sbe_engine* e, const int32_t* objects, int n_subset, const int32_t* gid_old, const int32_t* gid_new,
                     const uint8_t* src_old, const uint8_t* src_new, const int32_t* touched, int n_touched, float* out_diff) {
    return counts_delta_impl(e, -1, 0, 0, objects, n_subset, gid_old, gid_new, src_old, src_new, touched, n_touched, out_diff);
}

int sbe_counts_delta_apply(sbe_engine* e, int slot, int update_probs, int update_source, const int32_t* objects, int n_subset,
                           const int32_t* gid_old, const int32_t* gid_new, const uint8_t* src_old, const uint8_t* src_new,
                           const int32_t* touched, int n_touched, float* out_diff) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    return counts_delta_impl(e, slot, update_probs, update_source, objects, n_subset, gid_old, gid_new, src_old, src_new, touched, n_touched, out_diff);
}

static int set_counts_rows_impl(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows, bool with_probs) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, group_idx); CHECK_PTR(e, rows);
    for (int i = 0; i < n_rows; ++i)
        if (group_idx[i] < 0 || group_idx[i] >= e->Gtot) return fail(e, SBE_ERR_ARG, "group index %d out of range [0,%d)", group_idx[i], e->Gtot);
    // rows PATCH a table: the component's counts must be resident already (sbe_set_counts / sbe_recount / a step),
    // else the patched rows would sit among uninitialised ones and counts_set would stay false
    for (int i = 0; i < n_rows; ++i) {
        int c = 0;
        while (c + 1 < e->C && group_idx[i] >= e->goff[c + 1]) ++c;
        if (!e->slots[slot].counts_set[c])
            return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set (sbe_set_counts_rows patches resident tables: "
                        "send the component whole with sbe_set_counts first); row for group %d refused", slot, c, group_idx[i]);
        if (with_probs && (!e->slots[slot].probs_set[c] || !e->conc_set[c]))
            return fail(e, SBE_ERR_STATE, "slot %d: probability tables / concentration of component %d not set (sbe_set_counts_rows_probs "
                        "rebuilds the rows of tables that exist: sbe_update_probs first); row for group %d refused", slot, c, group_idx[i]);
    }
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S;
    const size_t rb = al256((size_t)n_rows * fs * sizeof(float));
    int rc = ensure_scratch(e, rb + (size_t)n_rows * sizeof(int32_t));
    if (rc) return rc;
    const void *v_rows, *v_idx;
    rc = stage(e, rows, (size_t)n_rows * fs * sizeof(float), e->d_scratch, &v_rows);
    if (rc) return rc;
    rc = stage(e, group_idx, (size_t)n_rows * sizeof(int32_t), e->d_scratch + rb, &v_idx);
    if (rc) return rc;
    if (with_probs) { rc = clear_status_word(e, ST_BAD_NORMALIZE); if (rc) return rc; }
    const int kind = !with_probs ? 0 : e->S <= 8 ? 8 : e->S <= 16 ? 16 : 1;
    const int lanes_per_row = kind == 8 ? 8 : kind == 16 ? 16 : 1;
    const int64_t n_threads = kind == 0 ? (int64_t)n_rows * fs : (int64_t)n_rows * e->F * lanes_per_row;
    if (e->batch && e->batch->n_rows_blocks == 0 && v_rows != (const void*)e->d_scratch) {     // (staged in the ring: sbe_set_slot_delta launches it)
        SetterJobs& j = *e->batch;
        j.rows = CountRowsArgs{(const float*)v_rows, (const int32_t*)v_idx, e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc,
                               e->d_probs + (int64_t)slot * e->table_elems(), e->d_probs_t + (int64_t)slot * e->probs_t_elems(), n_rows,
                               e->F, e->S, e->Gtot, e->ft, e->d_status};
        j.rows_kind = kind; j.n_rows_blocks = (unsigned)div_up(n_threads, 256);
        return with_probs ? check_after(e, ST_BAD_NORMALIZE, "sbe_set_counts_rows_probs") : SBE_OK;
    }
    if (!with_probs) {
        k_set_count_rows<<<div_up((int64_t)n_rows * fs, 256), 256, 0, e->stream>>>(
            (const float*)v_rows, (const int32_t*)v_idx, e->d_counts + (int64_t)slot * e->table_elems(), n_rows, fs);
        HIPCHK(e, hipGetLastError());
        return SBE_OK;
    }
    auto launch = [&](auto kernel) {
        kernel<<<div_up(n_threads, 256), 256, 0, e->stream>>>(
            (const float*)v_rows, (const int32_t*)v_idx, e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc,
            e->d_probs + (int64_t)slot * e->table_elems(), e->d_probs_t + (int64_t)slot * e->probs_t_elems(), n_rows, e->F, e->S, e->Gtot,
            e->ft, e->d_status);
    };
    if (kind == 8) launch(k_set_count_rows_probs_x<8>);
    else if (kind == 16) launch(k_set_count_rows_probs_x<16>);
    else launch(k_set_count_rows_probs);
    HIPCHK(e, hipGetLastError());
    return check_after(e, ST_BAD_NORMALIZE, "sbe_set_counts_rows_probs");
}

int sbe_set_counts_rows(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows) {
    return set_counts_rows_impl(e, slot, group_idx, n_rows, rows, false);
}

int sbe_set_counts_rows_probs(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows) {
    return set_counts_rows_impl(e, slot, group_idx, n_rows, rows, true);
}

// Several state-setting calls of one bind as ONE launch: sbe_set_groups (groups != NULL), sbe_set_counts_rows /
// sbe_set_counts_rows_probs (n_count_rows > 0), sbe_set_source_rows (n_src_rows > 0) -- the same checks, the same host-side
// work and the same results as those calls in that order; their kernels (the three jobs touch disjoint resident arrays) are
// collected and issued together (k_apply_setters) when their inputs went through the mapped ring, one by one otherwise.
int sbe_set_slot_delta(sbe_engine* e, int slot, int groups_component, const uint8_t* groups, const int32_t* count_idx, int n_count_rows,
                       const float* count_rows, int update_probs, const int32_t* src_objects, int n_src_rows, const uint8_t* src_rows) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    SetterJobs jobs{};
    const bool batching = e->opt_deferred && e->batch == nullptr;      // (immediate data checks synchronise inside every setter)
    if (batching) e->batch = &jobs;
    int rc = SBE_OK;
    if (groups) rc = sbe_set_groups(e, slot, groups_component, groups);
    if (!rc && n_count_rows) rc = set_counts_rows_impl(e, slot, count_idx, n_count_rows, count_rows, update_probs != 0);
    if (!rc && n_src_rows) rc = sbe_set_source_rows(e, slot, src_objects, n_src_rows, src_rows);
    if (batching) {
        e->batch = nullptr;
        const unsigned n_blocks = jobs.n_group_blocks + jobs.n_rows_blocks + jobs.n_src_blocks;
        if (n_blocks) {                         // (also after a later setter's error: the earlier ones' host state counts on their launch)
            k_apply_setters<<<n_blocks, 256, 0, e->stream>>>(jobs);
            if (hipGetLastError() != hipSuccess && !rc) rc = fail(e, SBE_ERR_HIP, "k_apply_setters launch failed");
        }
    }
    return rc;
}

// k_given_unchanged_fused: LDS image of a 16-feature tile and the arguments both forms share
constexpr size_t kGuFusedLdsMax = (size_t)96 << 10;
static size_t gu_fused_lds_bytes(int R, int S, size_t in_bytes, int N) {
    return ((size_t)R * 16 * S + (size_t)(N + 31) / 32) * sizeof(int32_t) + in_bytes;
}
// `in_bytes` of the call's host-mapped block (a multiple of 256) are staged by the kernel; the arrays sit at these byte offsets
static GuFusedArgs gu_fused_args(sbe_engine* e, int slot, int i_cluster, int n_sub, int R, double temperature, double prior_temperature,
                                 const int32_t* table_offsets_host, size_t in_bytes, size_t group_idx_at, size_t hc_new_at, size_t hc_old_at) {
    GuFusedArgs fa{};
    const int C = e->C;
    fa.state = e->d_state; fa.gid = e->d_gid + (int64_t)slot * C * e->Np; fa.src = e->d_src + (int64_t)slot * e->N * e->Fp;
    fa.counts = e->d_counts + (int64_t)slot * e->table_elems();
    fa.mapped_in = reinterpret_cast<const uint32_t*>(e->d_io); fa.in_words = (int)(in_bytes / 4);
    fa.objects_word = 0; fa.group_idx_word = (int)(group_idx_at / 4); fa.hc_new_word = (int)(hc_new_at / 4); fa.hc_old_word = (int)(hc_old_at / 4);
    for (int c = 0; c < C; ++c) fa.table_offsets[c] = table_offsets_host[c];
    fa.conc = e->d_conc; fa.unif = e->d_unif_res; fa.temperature = temperature; fa.prior_temperature = prior_temperature;
    fa.status = e->d_status;
    fa.n_sub = n_sub; fa.i_cluster = i_cluster; fa.K = e->G[0]; fa.N = e->N; fa.Np = e->Np; fa.F = e->F; fa.S = e->S; fa.C = C;
    fa.Fp = e->Fp; fa.R = R;
    return fa;
}

int sbe_given_unchanged_lh(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                           double prior_temperature, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_sub < 0) return fail(e, SBE_ERR_ARG, "n_sub=%d", n_sub);
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, out);
    const int N = e->N, F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_cluster < 0 || i_cluster >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [0,%d)", i_cluster, K);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_sub);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source not set", slot);
    for (int c = 0; c < C; ++c) {
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
        if (c > 0 && !s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
    }
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    HIPCHK(e, hipSetDevice(e->device));
    const int R = 1 + e->Gtot - K;                              // table rows: the cluster + every confounder group
    const int64_t fs = (int64_t)F * S;
    const size_t out_bytes = (size_t)n_sub * F * C * sizeof(float);
    const bool mapped_out = out_bytes <= ((size_t)1 << 18);
    // host-mapped inputs: object list | table row of each (component, subset object) | table offsets
    const size_t ob = al256((size_t)n_sub * 4), mb = 0, gb = al256((size_t)C * n_sub * 4), fb = al256((size_t)C * 4);
    rc = ensure_io(e, ob + mb + gb + fb + (mapped_out ? out_bytes : 0));
    if (rc) return rc;
    uint8_t* h = e->h_io;
    memcpy(h, objects, (size_t)n_sub * 4);
    int32_t* gi = (int32_t*)(h + ob + mb);
    int32_t* off = (int32_t*)(h + ob + mb + gb);
    for (int c = 0; c < C; ++c) {
        off[c] = c == 0 ? 0 : 1 + e->goff[c] - K;
        for (int i = 0; i < n_sub; ++i) {
            if (c == 0) { gi[i] = 0; continue; }                // every subset object sees the cluster's table (operators.py:884)
            const uint16_t gg = s.h_gid[(size_t)c * N + objects[i]];
            gi[(size_t)c * n_sub + i] = gg == kNoGroup ? -1 : (int)gg - e->goff[c];
        }
    }
    const size_t cb = al256((size_t)R * fs * sizeof(float));
    rc = ensure_scratch(e, cb + (mapped_out ? 0 : out_bytes));
    if (rc) return rc;
    float* d_tab = (float*)e->d_scratch;
    float* d_out = mapped_out ? (float*)(e->d_io + ob + mb + gb + fb) : (float*)(e->d_scratch + cb);
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    // one launch (k_given_unchanged_fused: tables built per 16-feature tile in LDS and consumed there) when the tile's
    // image fits; otherwise -- or with SBE_OPT_FUSE_TABLES off -- the table kernel and the gather, two launches
    const size_t fused_lds = gu_fused_lds_bytes(R, S, ob + mb + gb, N);
    if (e->opt_fuse_tables && mapped_out && fused_lds <= kGuFusedLdsMax) {
        GuFusedArgs fa = gu_fused_args(e, slot, i_cluster, n_sub, R, temperature, prior_temperature, off, ob + mb + gb, ob + mb, 0, 0);
        const double inv_t = 1.0 / temperature;
        fa.out = d_out; fa.inv_t = (float)inv_t; fa.use_pow = inv_t != 1.0;
        const unsigned blocks = (unsigned)div_up(F, 16);
        const DoneSig done = next_done(e, blocks);
        k_given_unchanged_fused<false><<<blocks, kUnchangedBlock, fused_lds, e->stream>>>(fa, GuGibbsArgs{}, nullptr, nullptr, nullptr, done);
        HIPCHK(e, hipGetLastError());
        rc = sync_and_report(e, done);
        if (rc) return rc;
        memcpy(out, h + ob + mb + gb + fb, out_bytes);
        return SBE_OK;
    }
    // kept counts and their conditional_effect_mean (conditionals.py:105-122) in one launch: the cluster's row with the
    // cluster prior, the confounder rows with theirs
    const bool list_in_lds = (size_t)n_sub * sizeof(int32_t) <= ((size_t)32 << 10);
    if (((size_t)16 * S + (N + 31) / 32) * sizeof(int32_t) > ((size_t)96 << 10))
        return fail(e, SBE_ERR_ARG, "component_likelihood_given_unchanged: %d objects x %d states exceed the kernel's LDS image", N, S);
    k_unchanged_counts<<<dim3(R, div_up(F, 16)), kUnchangedBlock,
                         ((size_t)16 * S + (list_in_lds ? n_sub : 0) + (N + 31) / 32) * sizeof(int32_t), e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_src + (int64_t)slot * N * e->Fp,
        e->d_counts + (int64_t)slot * e->table_elems(), (const int32_t*)e->d_io, n_sub, e->d_comp_of_group,
        i_cluster, K, N, e->Np, F, S, e->Fp, e->d_conc, e->d_unif_res, temperature, prior_temperature, e->d_status, d_tab,
        list_in_lds ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    const double inv_t = 1.0 / temperature;
    const DoneSig done = mapped_out ? next_done(e, (unsigned)div_up((int64_t)n_sub * F, 256)) : DoneSig{};
    k_subset_lh<<<div_up((int64_t)n_sub * F, 256), 256, 0, e->stream>>>(
        e->d_state, d_tab, (const int32_t*)(e->d_io + ob + mb + gb), (const int32_t*)(e->d_io + ob + mb), (const int32_t*)e->d_io,
        n_sub, d_out, F, S, C, e->Fp, (float)inv_t, inv_t != 1.0, done);
    HIPCHK(e, hipGetLastError());
    if (!mapped_out) {
        rc = d2h(e, out, d_out, out_bytes);
        if (rc) return rc;
        return report_status(e);                    // (d2h synchronised)
    }
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(out, h + ob + mb + gb + fb, out_bytes);
    return SBE_OK;
}

// `gid_old` non-null: the count delta of the proposal as well (sbe_given_unchanged_gibbs_counts)
static int given_unchanged_gibbs_impl(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                                      double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                      const uint8_t* src_old, const double* z, uint8_t* src_new_out, float* sel_new_out,
                                      float* sel_back_out, const int32_t* gid_old, const int32_t* gid_new, int32_t* touched_out,
                                      int32_t* n_touched_out, float* diff_rows_out, int follow = 0, int follow_probs = 0) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    const bool with_counts = gid_old != nullptr;
    if (with_counts) { CHECK_PTR(e, gid_new); CHECK_PTR(e, touched_out); CHECK_PTR(e, n_touched_out); CHECK_PTR(e, diff_rows_out); *n_touched_out = 0; }
    if (n_sub < 0) return fail(e, SBE_ERR_ARG, "n_sub=%d", n_sub);
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, hc_new); CHECK_PTR(e, hc_old); CHECK_PTR(e, src_old); CHECK_PTR(e, z);
    CHECK_PTR(e, src_new_out); CHECK_PTR(e, sel_new_out); CHECK_PTR(e, sel_back_out);
    const int N = e->N, F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_cluster < 0 || i_cluster >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [0,%d)", i_cluster, K);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_sub);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    for (int c = 0; c < C; ++c) {
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
        if (c > 0 && !s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
    }
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    for (int64_t i = 0; i < (int64_t)n_sub * F; ++i)
        if (src_old[i] != 0xFF && src_old[i] >= C) return fail(e, SBE_ERR_ARG, "old source component %d out of range [0,%d)", src_old[i], C);
    HIPCHK(e, hipSetDevice(e->device));
    const int R = 1 + e->Gtot - K;                              // table rows: the cluster + every confounder group
    const int64_t fs = (int64_t)F * S;
    const size_t nf = (size_t)n_sub * F;
    // host-mapped block: object list | table row per (component, subset object) | table offsets | has_components rows of
    // both samples | -- outputs: drawn component ids | selected probabilities, forward and backward.  The old source ids and
    // the uniforms (n F (1 + 8) bytes, read once, element-parallel) are staged through the ring / an upload.
    const size_t ob = al256((size_t)n_sub * 4), gb = al256((size_t)C * n_sub * 4), fb = al256((size_t)C * 4);
    const size_t hb = al256((size_t)n_sub * C);
    const size_t idb = al256(nf), selb = al256(nf * sizeof(float));
    // with the count delta: + the subset's global group ids in both samples and the touched groups (in), their rows (out)
    int n_touched = 0;
    if (with_counts) {
        if (sbeh_touched_groups(gid_old, gid_new, (int64_t)C * n_sub, e->Gtot, touched_out, &n_touched) != 0)
            return fail(e, SBE_ERR_ARG, "group index out of range in the subset's ids");
        *n_touched_out = n_touched;
    }
    const size_t gidb = with_counts ? al256((size_t)C * n_sub * 4) : 0, tchb = with_counts ? al256((size_t)std::max(n_touched, 1) * 4) : 0;
    const size_t rowb = with_counts ? al256((size_t)std::max(n_touched, 1) * fs * sizeof(float)) : 0;
    const size_t in_bytes = ob + gb + fb + 2 * hb + 2 * gidb + tchb, out_bytes = idb + 2 * selb + rowb;
    if (out_bytes > ((size_t)8 << 20)) return fail(e, SBE_ERR_ARG, "sbe_given_unchanged_gibbs: %d objects x %d features exceed the mapped result block", n_sub, F);
    rc = ensure_io(e, in_bytes + out_bytes);
    if (rc) return rc;
    uint8_t* h = e->h_io;
    memcpy(h, objects, (size_t)n_sub * 4);
    int32_t* gi = (int32_t*)(h + ob);
    int32_t* off = (int32_t*)(h + ob + gb);
    for (int c = 0; c < C; ++c) {
        off[c] = c == 0 ? 0 : 1 + e->goff[c] - K;
        for (int i = 0; i < n_sub; ++i) {
            if (c == 0) { gi[i] = 0; continue; }                // every subset object sees the cluster's table (operators.py:884)
            const uint16_t gg = s.h_gid[(size_t)c * N + objects[i]];
            gi[(size_t)c * n_sub + i] = gg == kNoGroup ? -1 : (int)gg - e->goff[c];
        }
    }
    memcpy(h + ob + gb + fb, hc_new, (size_t)n_sub * C);
    memcpy(h + ob + gb + fb + hb, hc_old, (size_t)n_sub * C);
    const size_t o_gold = ob + gb + fb + 2 * hb, o_gnew = o_gold + gidb, o_tch = o_gnew + gidb;
    if (with_counts) {
        memcpy(h + o_gold, gid_old, (size_t)C * n_sub * 4);
        memcpy(h + o_gnew, gid_new, (size_t)C * n_sub * 4);
        memcpy(h + o_tch, touched_out, (size_t)n_touched * 4);
    }
    const size_t cb = al256((size_t)R * fs * sizeof(float)), zb = al256(nf * sizeof(double)), sob = al256(nf);
    rc = ensure_scratch(e, cb + zb + sob);
    if (rc) return rc;
    float* d_tab = (float*)e->d_scratch;
    const void *v_z, *v_so;
    rc = stage(e, z, nf * sizeof(double), e->d_scratch + cb, &v_z);
    if (rc) return rc;
    rc = stage(e, src_old, nf, e->d_scratch + cb + zb, &v_so);
    if (rc) return rc;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    GuGibbsArgs a{};
    a.state = e->d_state; a.tables = d_tab; a.table_offsets = (const int32_t*)(e->d_io + ob + gb);
    a.group_idx = (const int32_t*)(e->d_io + ob); a.objects = (const int32_t*)e->d_io;
    a.weights = e->d_weights + (int64_t)slot * F * C;
    a.hc_new = e->d_io + ob + gb + fb; a.hc_old = e->d_io + ob + gb + fb + hb;
    a.src_old = (const uint8_t*)v_so; a.z = (const double*)v_z;
    a.n_sub = n_sub; a.F = F; a.S = S; a.C = C; a.Fp = e->Fp;
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    a.inv_t = (float)inv_t; a.inv_tp = (float)inv_tp; a.pow_lh = inv_t != 1.0; a.pow_w = inv_tp != 1.0; a.from_prior = from_prior ? 1 : 0;
    uint8_t* d_ids = e->d_io + in_bytes;
    // the slot follows the proposal (sbe_given_unchanged_gibbs_apply): its counts take the delta, the touched groups' probability
    // rows are rebuilt, the subset's source rows become the drawn ids -- behind the completion flag of the same launch
    follow = follow && with_counts && n_touched > 0;
    DeltaFollow dfollow{};
    if (follow) {
        const Slot& sl = e->slots[slot];
        for (int t = 0; t < n_touched; ++t) {
            int c = 0;
            while (c + 1 < C && touched_out[t] >= e->goff[c + 1]) ++c;
            if (follow_probs && !sl.probs_set[c])
                return fail(e, SBE_ERR_STATE, "slot %d: probability tables of component %d not set (update_probs = 1 rebuilds the rows of "
                            "tables that exist: sbe_update_probs first)", slot, c);
        }
        dfollow.counts = e->d_counts + (int64_t)slot * e->table_elems();
        dfollow.src = e->d_src + (int64_t)slot * N * e->Fp;
        if (follow_probs) {
            dfollow.conc = e->d_conc; dfollow.probs = e->d_probs + (int64_t)slot * e->table_elems();
            dfollow.probs_t = e->d_probs_t + (int64_t)slot * e->probs_t_elems(); dfollow.status = e->d_status; dfollow.ft = e->ft;
        }
    }
    const size_t fused_lds = gu_fused_lds_bytes(R, S, in_bytes, N) +
                             (with_counts ? ((size_t)n_touched * 16 * S + (size_t)e->Gtot) * sizeof(int32_t) : 0) +
                             (follow ? (size_t)n_touched * sizeof(int32_t) + (size_t)n_sub * 16 : 0);
    if (e->opt_fuse_tables && fused_lds <= kGuFusedLdsMax) {        // one launch (see sbe_given_unchanged_lh)
        GuFusedArgs fa = gu_fused_args(e, slot, i_cluster, n_sub, R, temperature, prior_temperature, off, in_bytes, ob,
                                       ob + gb + fb, ob + gb + fb + hb);
        if (with_counts) {
            fa.gid_old_word = (int)(o_gold / 4); fa.gid_new_word = (int)(o_gnew / 4); fa.n_touched = n_touched; fa.Gtot = e->Gtot;
            fa.touched = reinterpret_cast<const int32_t*>(e->d_io + o_tch);
            fa.rows_out = reinterpret_cast<float*>(d_ids + idb + 2 * selb);
            fa.follow = dfollow;
        }
        const unsigned blocks = (unsigned)div_up(F, 16);
        const DoneSig done = next_done(e, blocks);
        k_given_unchanged_fused<true><<<blocks, kUnchangedBlock, fused_lds, e->stream>>>(
            fa, a, d_ids, (float*)(d_ids + idb), (float*)(d_ids + idb + selb), done);
        HIPCHK(e, hipGetLastError());
        rc = sync_and_report(e, done);
        if (rc) return rc;
        memcpy(src_new_out, h + in_bytes, nf);
        memcpy(sel_new_out, h + in_bytes + idb, nf * sizeof(float));
        memcpy(sel_back_out, h + in_bytes + idb + selb, nf * sizeof(float));
        if (with_counts) memcpy(diff_rows_out, h + in_bytes + idb + 2 * selb, (size_t)n_touched * fs * sizeof(float));
        // (rows rebuilt behind the flag may raise normalize's data check: reported like a setter's)
        return dfollow.probs ? check_after(e, ST_BAD_NORMALIZE, "sbe_given_unchanged_gibbs_apply") : SBE_OK;
    }
    const bool list_in_lds = (size_t)n_sub * sizeof(int32_t) <= ((size_t)32 << 10);
    if (((size_t)16 * S + (N + 31) / 32) * sizeof(int32_t) > ((size_t)96 << 10))
        return fail(e, SBE_ERR_ARG, "gibbs_sample_source: %d objects x %d states exceed the kernel's LDS image", N, S);
    k_unchanged_counts<<<dim3(R, div_up(F, 16)), kUnchangedBlock,
                         ((size_t)16 * S + (list_in_lds ? n_sub : 0) + (N + 31) / 32) * sizeof(int32_t), e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_src + (int64_t)slot * N * e->Fp,
        e->d_counts + (int64_t)slot * e->table_elems(), (const int32_t*)e->d_io, n_sub, e->d_comp_of_group,
        i_cluster, K, N, e->Np, F, S, e->Fp, e->d_conc, e->d_unif_res, temperature, prior_temperature, e->d_status, d_tab,
        list_in_lds ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    const unsigned blocks = (unsigned)div_up((int64_t)nf, 256);
    const DoneSig done = next_done(e, blocks);
    k_given_unchanged_gibbs<<<blocks, kBlock, 0, e->stream>>>(a, d_ids, (float*)(d_ids + idb), (float*)(d_ids + idb + selb), e->d_status, done);
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(src_new_out, h + in_bytes, nf);
    memcpy(sel_new_out, h + in_bytes + idb, nf * sizeof(float));
    memcpy(sel_back_out, h + in_bytes + idb + selb, nf * sizeof(float));
    if (with_counts)       // (this shape has no one-launch form: the count delta by its own call, from the ids just drawn)
        return counts_delta_impl(e, follow ? slot : -1, follow_probs, follow ? 1 : 0, objects, n_sub, gid_old, gid_new, src_old, src_new_out,
                                 touched_out, n_touched, diff_rows_out);
    return SBE_OK;
}

int sbe_given_unchanged_gibbs(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                              double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                              const uint8_t* src_old, const double* z, uint8_t* src_new_out, float* sel_new_out,
                              float* sel_back_out) {
    return given_unchanged_gibbs_impl(e, slot, i_cluster, objects, n_sub, temperature, prior_temperature, from_prior, hc_new, hc_old,
                                      src_old, z, src_new_out, sel_new_out, sel_back_out, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int sbe_given_unchanged_gibbs_counts(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                                     double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                     const uint8_t* src_old, const double* z, const int32_t* gid_old, const int32_t* gid_new,
                                     uint8_t* src_new_out, float* sel_new_out, float* sel_back_out, int32_t* touched_out,
                                     int32_t* n_touched_out, float* diff_rows_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, gid_old);
    return given_unchanged_gibbs_impl(e, slot, i_cluster, objects, n_sub, temperature, prior_temperature, from_prior, hc_new, hc_old,
                                      src_old, z, src_new_out, sel_new_out, sel_back_out, gid_old, gid_new, touched_out, n_touched_out,
                                      diff_rows_out);
}

int sbe_given_unchanged_gibbs_apply(sbe_engine* e, int slot, int update_probs, int i_cluster, const int32_t* objects, int n_sub,
                                    double temperature, double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                    const uint8_t* src_old, const double* z, const int32_t* gid_old, const int32_t* gid_new,
                                    uint8_t* src_new_out, float* sel_new_out, float* sel_back_out, int32_t* touched_out,
                                    int32_t* n_touched_out, float* diff_rows_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, gid_old);
    return given_unchanged_gibbs_impl(e, slot, i_cluster, objects, n_sub, temperature, prior_temperature, from_prior, hc_new, hc_old,
                                      src_old, z, src_new_out, sel_new_out, sel_back_out, gid_old, gid_new, touched_out, n_touched_out,
                                      diff_rows_out, 1, update_probs);
}

int sbe_cluster_posterior_marginals(sbe_engine* e, int slot, int i_cluster, double temperature, double prior_temperature,
                                    const int32_t* objects, int n_objects_av, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    if (n_objects_av < 0) return fail(e, SBE_ERR_ARG, "n_objects=%d", n_objects_av);
    if (n_objects_av == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    const int F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_cluster < 0 || i_cluster >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [0,%d)", i_cluster, K);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_objects_av);
    if (rc) return rc;
    rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.counts_set[0] || !e->conc_set[0]) return fail(e, SBE_ERR_STATE, "slot %d: cluster counts / concentration not set", slot);
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    HIPCHK(e, hipSetDevice(e->device));
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t fs = (int64_t)F * S;
    // the candidate table: conditional_effect_mean(prior, counts[[i_cluster]], unif, T_prior, T) (operators.py:1046-1052)
    // from the slot's resident counts -- nothing table-sized crosses PCIe.  Fused form (k_cluster_marginals_ws): builder
    // waves of every block put it into LDS while the block's object waves run their load chains; otherwise -- table beyond
    // 64 KB, more than kWsC components, SBE_OPT_FUSE_TABLES off -- a table kernel in front of the block-per-object kernel.
    const size_t cand_bytes = (size_t)fs * sizeof(float);
    const bool fused = e->opt_fuse_tables && C <= kWsC && cand_bytes <= ((size_t)64 << 10);
    const size_t ob = al256((size_t)n_objects_av * sizeof(int32_t));
    const size_t out_bytes = (size_t)2 * n_objects_av * sizeof(double);
    rc = ensure_io(e, ob + out_bytes);
    if (rc) return rc;
    memcpy(e->h_io, objects, (size_t)n_objects_av * sizeof(int32_t));
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const int32_t* cnt = e->d_counts + (int64_t)slot * e->table_elems();
    const double inv = 1.0 / prior_temperature;
    DoneSig done;
    if (fused) {
        InlineTables tin{};
        tin.row[0] = RowSource{cnt + (int64_t)i_cluster * fs, e->d_conc + (int64_t)i_cluster * fs};
        tin.unif = e->d_unif_res; tin.temperature = temperature; tin.prior_temperature = prior_temperature; tin.status = e->d_status;
        tin.n_rows = F;
        const unsigned blocks = (unsigned)div_up(n_objects_av, kWsObjWaves);
        done = next_done(e, blocks);
        k_cluster_marginals_ws<<<blocks, kWsBlock, cand_bytes, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
            e->d_probs + (int64_t)slot * e->table_elems(), e->d_weights + (int64_t)slot * F * C,
            e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0, (const int32_t*)e->d_io, n_objects_av,
            (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, done, tin);
    } else {
        rc = ensure_scratch(e, cand_bytes);
        if (rc) return rc;
        float* d_tab = (float*)e->d_scratch;
        k_probs<int32_t><<<div_up((int64_t)F, 256), 256, 0, e->stream>>>(
            cnt, e->d_conc, e->d_unif_res, d_tab, i_cluster, i_cluster + 1, F, S,
            temperature, prior_temperature, 1, e->d_status, -(int64_t)i_cluster * fs);
        HIPCHK(e, hipGetLastError());
        done = next_done(e, (unsigned)n_objects_av);
        k_cluster_marginals<<<n_objects_av, kBlock, 0, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
            e->d_probs + (int64_t)slot * e->table_elems(), d_tab, e->d_weights + (int64_t)slot * F * C,
            e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0, (const int32_t*)e->d_io, n_objects_av,
            (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, done);
    }
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + ob, out_bytes);
    return SBE_OK;
}

int sbe_jump_lh_resident(sbe_engine* e, int slot, int i_source, int i_target, double temperature, double prior_temperature,
                         const int32_t* objects, int n_members, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    if (n_members < 0) return fail(e, SBE_ERR_ARG, "n_members=%d", n_members);
    if (n_members == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    const int F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_source < 0 || i_source >= K || i_target < 0 || i_target >= K) return fail(e, SBE_ERR_ARG, "cluster index out of range");
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_members);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / weights not set", slot);
    for (int c = 0; c < C; ++c)
        if (!s.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", slot, c);
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    HIPCHK(e, hipSetDevice(e->device));
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t fs = (int64_t)F * S;
    const int n_conf = e->Gtot - K;
    // tempered tables of the two clusters and of every confounder group (ClusterEffectProposals.posterior_counts +
    // normalize, operators.py:1254-1259, 1364-1371) from the slot's resident counts; the reference uses the CLUSTER
    // prior's uniform concentration for every component (operators.py:1352).  Fused form (k_jump_lh_ws): builder waves
    // put them into LDS, one launch; otherwise (tables beyond 64 KB, C > kWsC, option off) three table kernels in front.
    const size_t built_bytes = (size_t)(2 + n_conf) * fs * sizeof(float);
    const bool fused = e->opt_fuse_tables && C <= kWsC && built_bytes <= ((size_t)64 << 10);
    const size_t ob = al256((size_t)n_members * sizeof(int32_t));
    const size_t out_bytes = (size_t)2 * n_members * sizeof(double);
    rc = ensure_io(e, ob + out_bytes);
    if (rc) return rc;
    memcpy(e->h_io, objects, (size_t)n_members * sizeof(int32_t));
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const int32_t* cnt = e->d_counts + (int64_t)slot * e->table_elems();
    const double inv = 1.0 / prior_temperature;
    DoneSig done;
    if (fused) {
        InlineTables tin{};
        tin.row[0] = RowSource{cnt + (int64_t)i_source * fs, e->d_conc + (int64_t)i_source * fs};
        tin.row[1] = RowSource{cnt + (int64_t)i_target * fs, e->d_conc + (int64_t)i_target * fs};
        tin.counts = cnt; tin.conc = e->d_conc; tin.first_conf_group = K;
        tin.unif = e->d_unif_res; tin.temperature = temperature; tin.prior_temperature = prior_temperature; tin.status = e->d_status;
        tin.n_rows = (2 + n_conf) * F;
        const unsigned blocks = (unsigned)div_up(n_members, kWsObjWaves);
        done = next_done(e, blocks);
        k_jump_lh_ws<<<blocks, kWsBlock, built_bytes, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
            e->d_weights + (int64_t)slot * F * C, e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0,
            (const int32_t*)e->d_io, n_members, (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C,
            e->Fp, K, done, tin);
    } else {
        rc = ensure_scratch(e, (size_t)(2 + std::max(n_conf, 1)) * fs * sizeof(float));
        if (rc) return rc;
        float* d_ps = (float*)e->d_scratch;
        float* d_pt = d_ps + fs;
        float* d_pc = d_pt + fs;
        k_probs<int32_t><<<div_up((int64_t)F, 256), 256, 0, e->stream>>>(cnt, e->d_conc, e->d_unif_res, d_ps, i_source, i_source + 1, F, S,
            temperature, prior_temperature, 1, e->d_status, -(int64_t)i_source * fs);
        HIPCHK(e, hipGetLastError());
        k_probs<int32_t><<<div_up((int64_t)F, 256), 256, 0, e->stream>>>(cnt, e->d_conc, e->d_unif_res, d_pt, i_target, i_target + 1, F, S,
            temperature, prior_temperature, 1, e->d_status, -(int64_t)i_target * fs);
        HIPCHK(e, hipGetLastError());
        if (n_conf > 0) {
            k_probs<int32_t><<<div_up((int64_t)n_conf * F, 256), 256, 0, e->stream>>>(cnt, e->d_conc, e->d_unif_res, d_pc, K, e->Gtot, F, S,
                temperature, prior_temperature, 1, e->d_status, -(int64_t)K * fs);
            HIPCHK(e, hipGetLastError());
        }
        done = next_done(e, (unsigned)n_members);
        k_jump_lh<<<n_members, kBlock, 0, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np, d_pc, d_ps, d_pt,
            e->d_weights + (int64_t)slot * F * C, e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0,
            (const int32_t*)e->d_io, n_members, (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C,
            e->Fp, K, done);
    }
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + ob, out_bytes);
    return SBE_OK;
}

// ---- source prior and the LikelihoodLogger row --------------------------------
int sbe_source_prior(sbe_engine* e, int slot, double* per_object_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, per_object_out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    rc = ensure_scratch(e, (size_t)e->N * sizeof(double));
    if (rc) return rc;
    void* d_out;
    rc = out_target(e, (size_t)e->N * sizeof(double), e->d_scratch, &d_out);
    if (rc) return rc;
    const unsigned nb = (unsigned)div_up(e->N, 1024 / kWave);
    const DoneSig done = out_done(e, d_out, nb);
    k_source_prior<<<nb, 1024, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_pid + (int64_t)slot * e->Np,
        e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, (double*)d_out, e->N, e->F, e->C, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, per_object_out, d_out, (size_t)e->N * sizeof(double), done);
}

// Model.__call__ = likelihood + prior (sbayes/model/model.py:47-51): what sbe_collapsed_loglik_all and sbe_source_prior
// return, for the same slot state, in ONE launch and one synchronisation (k_collapsed_source_prior).  Shapes whose
// group terms exceed the LDS budget take the two calls one after the other.
int sbe_collapsed_and_source_prior(sbe_engine* e, int slot, double* per_group_out, double* per_object_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, per_group_out); CHECK_PTR(e, per_object_out);
    Slot& s = e->slots[slot];
    for (int c = 0; c < e->C; ++c) {
        if (e->G[c] == 0) continue;
        if (!s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
    }
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    const size_t lds = (size_t)e->F * e->S * sizeof(double) + (size_t)e->F * sizeof(float);
    const size_t gb = al256((size_t)e->Gtot * sizeof(double)), out_bytes = gb + (size_t)e->N * sizeof(double);
    if (e->Gtot == 0 || lds > ((size_t)96 << 10) || out_bytes > kMappedOutMax || !poll_done_enabled()) {
        int rc = sbe_collapsed_loglik_all(e, slot, per_group_out);
        if (rc) return rc;
        return sbe_source_prior(e, slot, per_object_out);
    }
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    rc = ensure_io(e, out_bytes);
    if (rc) return rc;
    const unsigned nb = (unsigned)e->Gtot + (unsigned)div_up(e->N, 1024 / kWave);
    const DoneSig done = next_done(e, nb);
    const SourcePriorArgs sp{e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_pid + (int64_t)slot * e->Np,
                             e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, (double*)(e->d_io + gb), e->N, e->F, e->C, e->Fp};
    k_collapsed_source_prior<<<nb, 1024, lds, e->stream>>>(e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, e->d_lg_conc,
                                                          e->d_sum_a, e->d_lg_sum_a, (double*)e->d_io, e->Gtot, e->F, e->S, sp, done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(per_group_out, e->h_io, (size_t)e->Gtot * sizeof(double));
    memcpy(per_object_out, e->h_io + gb, (size_t)e->N * sizeof(double));
    return synced(e);
}

int sbe_observation_lh_exact(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    for (int c = 0; c < e->C; ++c)
        if (!s.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t n = (int64_t)e->N * e->F;
    rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    k_lh_exact<<<div_up(n, 256), 256, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_gid + (int64_t)slot * e->C * e->Np,
        e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C,
        e->Fp, e->d_status, e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, e->d_pid + (int64_t)slot * e->Np);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out, e->d_scratch, n * sizeof(double));
    if (rc) return rc;
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE]) return fail(e, SBE_ERR_DATA, "normalize: non-positive row sum in leave-one-out tables (sbayes/util.py:1006 assert)");
    return SBE_OK;
}

// ---- one MCMC step in one call: delta in, likelihoods out (north_star: "only the proposed cluster-
// assignment delta crosses PCIe") ---------------------------------------------------------------------
namespace {
// Inputs of k_step_core that differ between the one-call MCMC step (payload from the host) and the one-call Gibbs
// step (new source sampled on the device).
struct CoreInputs {
    const void* ids_new = nullptr;       // component-0 group ids of the candidate [Np] u16, or nullptr (unchanged)
    const void* pid = nullptr; const void* tid = nullptr; const void* toff = nullptr;      // with ids_new: the tables
    const void* tuple_g = nullptr; const void* tuple_p = nullptr; const void* patbits = nullptr;   // derived from them
    const void* weights = nullptr;       // new weights [F][C] f32, or nullptr (unchanged)
    const int16_t* row_of = nullptr;     // [Np]: >= 0 marks an object whose source changes
    const uint8_t* rows = nullptr; const int32_t* objects = nullptr; int n_changed = 0;   // payload source rows
    const uint8_t* src_new = nullptr;    // device-sampled source (the candidate's array) instead of payload rows
    const int32_t* subset = nullptr; int n_subset = 0;      // objects whose counts may change
    int P = 1;                           // has_components patterns of the candidate
    // source array of the candidate slot: the whole array is copied from the current slot (full_src_copy), or only the
    // rows in which the candidate slot is known to differ from it (sbe_engine::SrcSync)
    bool full_src_copy = true; const int32_t* stale = nullptr; int n_stale = 0;
    // delta layout (sbe_step_batch_delta): patched id arrays, per-subset-entry row / cluster id (StepCore)
    const int32_t* patch_n = nullptr; const uint16_t* patch_gid = nullptr; const uint8_t* patch_pid = nullptr;
    const uint8_t* patch_tid = nullptr; int n_patch = -1;
    const int16_t* sub_row = nullptr; const uint16_t* sub_gid0 = nullptr;
};

// the single-step calls' lane: the engine's own payload / result blocks
sbe_engine::Lane lane0(sbe_engine* e) {
    return sbe_engine::Lane{e->h_step_payload, e->d_step_payload, e->h_step, e->d_step_host, e->d_step_pf, e->d_step_stamp, e->step_id,
                            e->d_status};
}

// kernel 1 of the one-call steps: candidate slot = current slot + inputs, count delta, every table.
// build_step_core fills the kernel's argument block for one chain (lane); the caller launches it.
int build_step_core(sbe_engine* e, sbe_engine::Lane& lane, int cur_slot, int cand_slot, const CoreInputs& in,
                    StepCore& a, size_t& lds_out, int& n_blocks_out, int n_chains = 1) {
    const int N = e->N, Np = e->Np, F = e->F, C = e->C;
    const bool regroup = in.ids_new != nullptr;
    a = StepCore{};
    uint32_t run = 0;
    auto seg = [&](auto* base, int64_t elems, const void* other_src) {       // per-slot array `base`, elems per slot
        const int64_t bytes = elems * (int64_t)sizeof(*base);
        a.cs.src[a.cs.n] = other_src ? reinterpret_cast<const uint32_t*>(other_src)
                                     : reinterpret_cast<const uint32_t*>(base + (int64_t)cur_slot * elems);
        a.cs.dst[a.cs.n] = reinterpret_cast<uint32_t*>(base + (int64_t)cand_slot * elems);
        run += (uint32_t)(bytes / 4);
        a.cs.end[a.cs.n++] = run;
    };
    uint16_t* g_cur = e->d_gid + (int64_t)cur_slot * C * Np;
    const bool delta = in.n_patch >= 0;          // sbe_step_batch_delta: the per-object id arrays are patched, not copied
    if (!delta) {
        // gid: component 0 from the inputs when the clusters changed; the other components from the current slot
        uint16_t* g_cand = e->d_gid + (int64_t)cand_slot * C * Np;
        a.cs.src[a.cs.n] = reinterpret_cast<const uint32_t*>(regroup ? in.ids_new : (const void*)g_cur);
        a.cs.dst[a.cs.n] = reinterpret_cast<uint32_t*>(g_cand);
        run += (uint32_t)(Np * 2 / 4); a.cs.end[a.cs.n++] = run;
        if (C > 1) {
            a.cs.src[a.cs.n] = reinterpret_cast<const uint32_t*>(g_cur + Np);
            a.cs.dst[a.cs.n] = reinterpret_cast<uint32_t*>(g_cand + Np);
            run += (uint32_t)((int64_t)(C - 1) * Np * 2 / 4); a.cs.end[a.cs.n++] = run;
        }
        seg(e->d_pid, (int64_t)Np, regroup ? in.pid : nullptr);
        seg(e->d_tid, (int64_t)Np, regroup ? in.tid : nullptr);
        seg(e->d_toff, (int64_t)Np, regroup ? in.toff : nullptr);
    }
    a.n_patch = in.n_patch; a.patch_n = in.patch_n; a.patch_gid = in.patch_gid; a.patch_pid = in.patch_pid; a.patch_tid = in.patch_tid;
    a.gid_dst = e->d_gid + (int64_t)cand_slot * C * Np; a.pid_dst = e->d_pid + (int64_t)cand_slot * Np;
    a.tid_dst = e->d_tid + (int64_t)cand_slot * Np; a.toff_dst = e->d_toff + (int64_t)cand_slot * Np;
    a.toff_mul = (uint32_t)(e->S + 1) * 512u;
    a.sub_row = in.sub_row; a.sub_gid0 = in.sub_gid0;
    seg(e->d_tuple_g, (int64_t)kMaxTuples * kMaxComponents, in.tuple_g);
    seg(e->d_tuple_p, (int64_t)kMaxTuples, in.tuple_p);
    seg(e->d_patbits, (int64_t)e->Pmax, in.patbits);
    seg(e->d_weights, (int64_t)F * C, in.weights);
    a.src_seg = a.cs.n;
    if (in.full_src_copy) seg(e->d_src, (int64_t)N * e->Fp, nullptr);
    a.stale = in.stale; a.n_stale = in.full_src_copy ? 0 : in.n_stale;
    a.src_cur_rows = e->d_src + (int64_t)cur_slot * N * e->Fp;
    a.row_of = in.row_of;
    a.rows = in.rows;
    a.objects = in.objects;
    a.src_dst = e->d_src + (int64_t)cand_slot * N * e->Fp;
    a.n_changed = in.n_changed; a.F = F; a.C = C; a.Fp = e->Fp; a.status = lane.d_status;
    // tile blocks
    a.state = e->d_state; a.gid_cur = g_cur;
    a.ids_new = reinterpret_cast<const uint16_t*>(in.ids_new);
    a.src_cur = e->d_src + (int64_t)cur_slot * N * e->Fp;
    a.src_new = in.src_new;
    a.subset = in.subset; a.n_subset = in.n_subset;
    a.counts_cur = e->d_counts + (int64_t)cur_slot * e->table_elems();
    a.counts_new = e->d_counts + (int64_t)cand_slot * e->table_elems();
    a.conc = e->d_conc; a.lg_conc = e->d_lg_conc; a.sum_a = e->d_sum_a; a.lg_sum_a = e->d_lg_sum_a;
    a.probs = e->d_probs + (int64_t)cand_slot * e->table_elems();
    a.probs_t = e->d_probs_t + (int64_t)cand_slot * e->probs_t_elems();
    a.per_feature = lane.d_pf;
    if (++lane.step_id == 0) {                // stamp wrap-around (2^32 steps): start over from clean stamps
        HIPCHK(e, hipMemsetAsync(lane.d_stamp, 0, e->Gtot * sizeof(uint32_t), e->stream));
        lane.step_id = 1;
    }
    a.stamp = lane.d_stamp; a.step_id = lane.step_id;
    a.Np = Np; a.S = e->S; a.Gtot = e->Gtot; a.ft = e->ft;
    a.ftc = (int)std::max<int64_t>(1, std::min<int64_t>(8, 2048 / ((int64_t)e->Gtot * e->S)));
    a.n_tile_blocks = div_up(F, a.ftc);
    // weight blocks
    a.weights = in.weights ? reinterpret_cast<const float*>(in.weights) : e->d_weights + (int64_t)cur_slot * F * C;
    a.pattern_bits = in.patbits ? reinterpret_cast<const uint32_t*>(in.patbits) : e->d_patbits + (int64_t)cur_slot * e->Pmax;
    a.wpat = e->d_wpat + (int64_t)cand_slot * e->Pmax * F * C;
    a.wpat_t = e->d_wpat_t + (int64_t)cand_slot * e->wpat_t_elems();
    a.P = in.P; a.Pmax = e->Pmax; a.n_weight_blocks = div_up((int64_t)in.P * F, kBlock);
    const int64_t E = (int64_t)e->Gtot * a.ftc * e->S, R = (int64_t)e->Gtot * a.ftc;
    const size_t lds = (size_t)((E * 4 + 15) / 16 * 16) + (size_t)(2 * E + R) * sizeof(double);
    // copy blocks: enough to fill the chip for ONE chain; a batch of chains shares it (64 chains x 68 four-KB copy
    // blocks made the batched launch workgroup-dispatch bound: 6 000 blocks, 99 us)
    a.n_copy_blocks = std::max(1, (int)std::min<int64_t>(div_up(run, 1024), std::max(4, 2 * e->compute_units / std::max(1, n_chains))));
    lds_out = lds;
    n_blocks_out = a.n_tile_blocks + a.n_weight_blocks + a.n_copy_blocks;
    return SBE_OK;
}

int launch_step_core(sbe_engine* e, int cur_slot, int cand_slot, const CoreInputs& in) {
    sbe_engine::Lane lane = lane0(e);
    StepCore a; size_t lds = 0; int n_blocks = 0;
    int rc = build_step_core(e, lane, cur_slot, cand_slot, in, a, lds, n_blocks);
    e->step_id = lane.step_id;
    if (rc) return rc;
    k_step_core<<<n_blocks, kBlock, lds, e->stream>>>(a);
    HIPCHK(e, hipGetLastError());
    return SBE_OK;
}

// the step epilogue's mapped-memory block: [Gtot] f64 | [ST_WORDS] i32 | [Gtot] u8 (padded to 8) | [2] f64 (log_q, log_q_back)
inline size_t step_host_lq_offset(const sbe_engine* e) {
    return ((size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int) + (size_t)e->Gtot + 7) / 8 * 8;
}

StepFinish make_step_finish_lane(sbe_engine* e, const sbe_engine::Lane& lane) {
    StepFinish fin{};
    fin.per_feature = lane.d_pf;
    fin.group_out = reinterpret_cast<double*>(lane.d_step_host);
    fin.status = lane.d_status;
    fin.status_out = reinterpret_cast<int*>(lane.d_step_host + (size_t)e->Gtot * sizeof(double));
    fin.changed = nullptr; fin.stamp = lane.d_stamp; fin.step_id = lane.step_id;
    fin.changed_out = lane.d_step_host + (size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int);
    fin.Gtot = e->Gtot; fin.F = e->F;
    return fin;
}
StepFinish make_step_finish(sbe_engine* e) { return make_step_finish_lane(e, lane0(e)); }

// after the synchronisation that ends a one-call step: data checks, then the results out of the mapped block
int read_step_results_lane(sbe_engine* e, const uint8_t* h_step, int cand_slot, double* group_logliks_out, double* mixture_out,
                      uint8_t* changed_groups_out, const char* bad_norm_what, int* d_status = nullptr, int chain = -1) {
    const int* hst = reinterpret_cast<const int*>(h_step + (size_t)e->Gtot * sizeof(double));
    if (hst[ST_BAD_NORMALIZE] || hst[ST_MULTI_SOURCE]) {
        const int bad_norm = hst[ST_BAD_NORMALIZE], multi_src = hst[ST_MULTI_SOURCE];
        (void)hipMemsetAsync((d_status ? d_status : e->d_status) + ST_BAD_NORMALIZE, 0, 2 * sizeof(int), e->stream);
        if (!d_status || d_status == e->d_status) e->h_flag[ST_BAD_NORMALIZE] = e->h_flag[ST_MULTI_SOURCE] = 0;
        char who[32] = "";
        if (chain >= 0) snprintf(who, sizeof who, "chain %d: ", chain);
        if (bad_norm) return fail(e, SBE_ERR_DATA, "%snormalize: %d %s have a non-positive sum (sbayes/util.py:1006 assert)", who, bad_norm, bad_norm_what);
        return fail(e, SBE_ERR_DATA, "%ssource is not one-hot over components in %d observations", who, multi_src);
    }
    memcpy(group_logliks_out, h_step, (size_t)e->Gtot * sizeof(double));
    if (changed_groups_out) memcpy(changed_groups_out, h_step + (size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int), (size_t)e->Gtot);
    *mixture_out = e->h_results[cand_slot];
    return SBE_OK;
}
int read_step_results(sbe_engine* e, int cand_slot, double* group_logliks_out, double* mixture_out,
                      uint8_t* changed_groups_out, const char* bad_norm_what) {
    return read_step_results_lane(e, e->h_step, cand_slot, group_logliks_out, mixture_out, changed_groups_out, bad_norm_what);
}

}  // namespace

static int step_lean(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                     int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                     double* mixture_out, uint8_t* changed_groups_out);
static int step_general(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                        int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                        double* mixture_out, uint8_t* changed_groups_out);

int sbe_step(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
             int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
             double* mixture_out, uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    if (n_changed < 0 || (n_changed > 0 && (!changed_objects || !source_rows)))
        return fail(e, SBE_ERR_ARG, "changed_objects / source_rows missing for n_changed=%d", n_changed);
    Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    for (int c = 0; c < e->C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", cur_slot, c);
    for (int i = 0; i < n_changed; ++i)
        if (changed_objects[i] < 0 || changed_objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", changed_objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    // few-launch form (one H2D payload, four kernels, results through mapped memory) whenever the step fits its
    // payload; the call-by-call form otherwise (many changed rows, never-uploaded patterns) or on request
    static const bool force_general = getenv("SBE_STEP_GENERAL") && atoi(getenv("SBE_STEP_GENERAL")) == 1;
    if (!force_general && e->opt_step_form == 0 && n_changed <= e->step_max_rows && !cur.patterns_dirty &&
        (int64_t)e->Gtot * e->S * 28 <= 60 * 1024)        // (k_step_core's LDS image of one feature column)
        return step_lean(e, cur_slot, cand_slot, clusters, changed_objects, n_changed, source_rows, weights,
                         group_logliks_out, mixture_out, changed_groups_out);
    return step_general(e, cur_slot, cand_slot, clusters, changed_objects, n_changed, source_rows, weights,
                        group_logliks_out, mixture_out, changed_groups_out);
}

// Host half of a lean step for one chain: the candidate's host state `cd` (= current + delta) and the step's payload
// packed into the lane's host-mapped block; `in` receives the device-side views of that payload.  Touches only the lane,
// `cd` and read-only engine state, so the chains of a batch can be prepared by several host threads at once.
static int prepare_step(sbe_engine* e, const sbe_engine::Lane& lane, int cur_slot, int cand_slot, const uint8_t* clusters,
                        const int32_t* changed_objects, int n_changed, const uint8_t* source_rows, const float* weights,
                        Slot& cd, CoreInputs& in, std::string* err, std::vector<int32_t>* moved_out = nullptr) {
    const int N = e->N, Np = e->Np, F = e->F, C = e->C;
    const Slot& cur = e->slots[cur_slot];
    cd = cur;                                 // host state of the candidate (committed by the caller)
    // objects whose counts may change: listed source rows + objects whose cluster membership changed
    static thread_local std::vector<int32_t> mv;                      // (no allocation per step: the chains of a batch are
    static thread_local std::vector<uint16_t> mv_old;                  //  prepared by pool threads)
    mv.clear(); mv_old.clear();
    bool full_derive = false;
    const bool regroup = clusters != nullptr;
    if (regroup) {
        const int K = e->G[0];
        uint16_t* ids = cd.h_gid.data();      // component 0: group offset 0
        {
            char msg[320];
            if (!matrix_to_ids(clusters, K, N, 0, 0, ids, msg, sizeof msg)) { *err = msg; return SBE_ERR_DATA; }
        }
        const uint16_t* old_ids = cur.h_gid.data();
        for (int n = 0; n < N; ++n) if (ids[n] != old_ids[n]) { mv.push_back(n); mv_old.push_back(old_ids[n]); }
        // pattern ids and group tuples of the candidate: the moved objects' entries updated in place (cd holds the
        // current slot's tables), the full derivation when the set of patterns changes, the counts are not there
        // (slot never derived in full) or on request
        if (e->opt_step_derive == 1 || (int)mv.size() > N / 8 ||
            !update_patterns_and_tuples(e, cd, mv.data(), mv_old.data(), (int)mv.size())) {
            full_derive = true;                        // pattern ranks / tuple numbers of ANY object may change
            derive_patterns(e, cd);
            if ((int)cd.patterns.size() > e->Pmax) {
                char buf[160];
                snprintf(buf, sizeof buf, "%zu distinct has_components patterns exceed capacity %d", cd.patterns.size(), e->Pmax);
                *err = buf;
                return SBE_ERR_ARG;
            }
            derive_tuples(e, cd);
        }
        cd.patterns_dirty = false;
        cd.group_epoch = ++e->epoch_counter;
    }
    if (moved_out) {
        *moved_out = mv;
        if (full_derive) moved_out->assign(1, -1);     // "every entry may differ": commit_ids_sync leaves no usable record
    }
    // ---- payload: packed in host-mapped pinned memory; the kernels read it in place (a few tens of KB over
    // PCIe, no copy engine in the chain).  Every step ends with a stream synchronisation, so the lane's buffer is
    // free again when the next step starts.
    const auto& L = e->sl;
    uint8_t* st = lane.h_payload;
    int n_subset = 0;
    {   // sorted union of the objects that changed cluster and the objects with new source rows
        int32_t* sub = reinterpret_cast<int32_t*>(st + L.subset);
        if (n_changed == 0) { memcpy(sub, mv.data(), mv.size() * sizeof(int32_t)); n_subset = (int)mv.size(); }
        else {
            static thread_local std::vector<int32_t> ch;
            ch.assign(changed_objects, changed_objects + n_changed);
            std::sort(ch.begin(), ch.end());
            ch.erase(std::unique(ch.begin(), ch.end()), ch.end());
            n_subset = (int)(std::set_union(mv.begin(), mv.end(), ch.begin(), ch.end(), sub) - sub);
        }
    }
    if (regroup) {
        memcpy(st + L.ids, cd.h_gid.data(), (size_t)N * 2);
        if (Np > N) memset(st + L.ids + (size_t)N * 2, 0xFF, (size_t)(Np - N) * 2);
        memset(st + L.pid, 0, Np); memcpy(st + L.pid, cd.h_pid.data(), N);
        memcpy(st + L.tid, cd.h_tid.data(), Np);
        memcpy(st + L.toff, cd.h_toff.data(), (size_t)Np * 4);
        memcpy(st + L.tuple_g, cd.h_tuple_g.data(), (size_t)kMaxTuples * kMaxComponents * 2);
        memcpy(st + L.tuple_p, cd.h_tuple_p.data(), kMaxTuples);
        memset(st + L.patbits, 0, (size_t)e->Pmax * 4);
        memcpy(st + L.patbits, cd.patterns.data(), cd.patterns.size() * 4);
    }
    if (weights) memcpy(st + L.weights, weights, (size_t)F * C * 4);
    if (n_changed > 0) {
        int16_t* row_of = reinterpret_cast<int16_t*>(st + L.row_of);
        std::fill(row_of, row_of + Np, (int16_t)-1);
        for (int i = 0; i < n_changed; ++i) row_of[changed_objects[i]] = (int16_t)i;    // (a repeated object: last row wins)
        memcpy(st + L.objects, changed_objects, (size_t)n_changed * 4);
        memcpy(st + L.rows, source_rows, (size_t)n_changed * F * C);
    }
    const uint8_t* pl = lane.d_payload;
    in = CoreInputs{};
    if (regroup) {
        in.ids_new = pl + L.ids; in.pid = pl + L.pid; in.tid = pl + L.tid; in.toff = pl + L.toff;
        in.tuple_g = pl + L.tuple_g; in.tuple_p = pl + L.tuple_p; in.patbits = pl + L.patbits;
    }
    if (weights) in.weights = pl + L.weights;
    if (n_changed > 0) {
        in.row_of = reinterpret_cast<const int16_t*>(pl + L.row_of);
        in.rows = pl + L.rows;
        in.objects = reinterpret_cast<const int32_t*>(pl + L.objects);
        in.n_changed = n_changed;
    }
    in.subset = reinterpret_cast<const int32_t*>(pl + L.subset); in.n_subset = n_subset;
    in.P = (int)cd.patterns.size();
    {   // source array: rows to bring over from the current slot (sbe_engine::SrcSync), or the whole array
        const sbe_engine::SrcSync& rec = e->src_sync[cand_slot];
        const sbe_engine::SrcSync& cs = e->src_sync[cur_slot];
        if (rec.peer == cur_slot && rec.peer_version == cs.version && rec.own_version == rec.version &&
            (int)rec.diff.size() <= e->step_max_rows) {
            int32_t* stale = reinterpret_cast<int32_t*>(st + L.stale);
            const int16_t* row_of = n_changed > 0 ? reinterpret_cast<const int16_t*>(st + L.row_of) : nullptr;
            int ns = 0;
            for (int32_t n : rec.diff)                       // (rows this step rewrites anyway are left to it)
                if (!row_of || row_of[n] < 0) stale[ns++] = n;
            in.full_src_copy = false;
            in.stale = reinterpret_cast<const int32_t*>(pl + L.stale);
            in.n_stale = ns;
        }
    }
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    cd.weights_set = true;
    return SBE_OK;
}

// ... and its id arrays = the current slot's except the moved objects' entries
static void commit_ids_sync(sbe_engine* e, int cur_slot, int cand_slot, const std::vector<int32_t>& moved) {
    sbe_engine::SrcSync& rc = e->ids_sync[cand_slot];
    sbe_engine::SrcSync& cu = e->ids_sync[cur_slot];
    ++rc.version;
    if (moved.size() == 1 && moved[0] < 0) {           // the candidate's tables were derived afresh (prepare_step): the two
        rc.peer = cu.peer = -1;                        // slots' pattern / tuple numbering is unrelated from here on
        rc.diff.clear(); cu.diff.clear();
        return;
    }
    rc.peer = cur_slot; rc.peer_version = cu.version; rc.own_version = rc.version; rc.diff = moved;
    cu.peer = cand_slot; cu.peer_version = rc.version; cu.own_version = cu.version; cu.diff = moved;
}

// after a one-call step was enqueued: the candidate's source = the current slot's except the rows the step wrote
static void commit_src_sync(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* changed_objects, int n_changed) {
    sbe_engine::SrcSync& rc = e->src_sync[cand_slot];
    sbe_engine::SrcSync& cu = e->src_sync[cur_slot];
    ++rc.version;
    rc.peer = cur_slot; rc.peer_version = cu.version; rc.own_version = rc.version;
    rc.diff.assign(changed_objects, changed_objects + n_changed);
    cu.peer = cand_slot; cu.peer_version = rc.version; cu.own_version = cu.version;
    cu.diff = rc.diff;
}

static int step_lean(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                     int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                     double* mixture_out, uint8_t* changed_groups_out) {
    if (e->status_pending) {                  // deliver a deferred data check before this step reuses the words
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    // SBE_STEP_TIMING=1: host-side phase times (prepare / enqueue / wait), printed every 2000 steps (diagnostic)
    static const bool timing = getenv("SBE_STEP_TIMING") && atoi(getenv("SBE_STEP_TIMING")) == 1;
    static double t_acc[3] = {0, 0, 0};
    static int t_n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    Slot cd;
    CoreInputs in;
    {
        std::string err;
        int rc = prepare_step(e, lane0(e), cur_slot, cand_slot, clusters, changed_objects, n_changed, source_rows, weights, cd, in, &err,
                              &e->step_moved);
        if (rc) return fail(e, rc, "%s", err.c_str());
    }
    const auto t1 = std::chrono::steady_clock::now();
    // ---- kernel 1: candidate slot = current slot + payload, count delta, every table ---------------------------
    int rc = launch_step_core(e, cur_slot, cand_slot, in);
    if (rc) return rc;
    commit_src_sync(e, cur_slot, cand_slot, changed_objects, n_changed);
    commit_ids_sync(e, cur_slot, cand_slot, e->step_moved);
    e->slots[cand_slot] = cd;
    // ---- kernels 2 + 3: fused mixture eval, reduction + step epilogue (mapped-memory results) -------------------
    StepFinish fin = make_step_finish(e);
    DoneSig done;
    rc = launch_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, &fin, nullptr, nullptr,
                        nullptr, &done);
    if (rc) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    rc = wait_done(e, done);
    if (rc) return rc;
    if (timing) {
        const auto t3 = std::chrono::steady_clock::now();
        t_acc[0] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t_acc[1] += std::chrono::duration<double, std::micro>(t2 - t1).count();
        t_acc[2] += std::chrono::duration<double, std::micro>(t3 - t2).count();
        if (++t_n == 2000) {
            fprintf(stderr, "[sbe_step] host prepare %.1f us, enqueue %.1f us, wait %.1f us per step\n", t_acc[0] / t_n, t_acc[1] / t_n, t_acc[2] / t_n);
            t_acc[0] = t_acc[1] = t_acc[2] = 0; t_n = 0;
        }
    }
    return read_step_results(e, cand_slot, group_logliks_out, mixture_out, changed_groups_out, "rows");
}

// ---- batched multi-chain step (missing #4): B chains' deltas in ONE call ------------------------------
// The reference steps its chains one after the other in one Python loop (MCMC.generate_samples,
// sbayes/sampling/mcmc.py:237-241).  Here the chains' candidate slots are built by ONE launch of k_step_core_batch
// (chain <-> blockIdx.y), evaluated by ONE launch of the fused mixture kernel over the candidate slot list and
// finished by ONE launch of k_reduce_partials (a reduction block and an epilogue block per chain); the host halves
// (candidate host state, payload packing) run on a small pool of worker threads.  One synchronisation per batch.
namespace {
int ensure_lanes(sbe_engine* e, int n) {
    const size_t hb = step_host_lq_offset(e) + 2 * sizeof(double);
    while ((int)e->lanes.size() < n) {
        sbe_engine::Lane ln{};
        // (no payload block of its own: a batch's payloads live back to back in h_batch_payload / d_batch_payload)
        HIPCHK(e, hipHostMalloc((void**)&ln.h_step, hb, hipHostMallocMapped));
        memset(ln.h_step, 0, hb);
        HIPCHK(e, hipHostGetDevicePointer((void**)&ln.d_step_host, ln.h_step, 0));
        HIPCHK(e, hipMalloc((void**)&ln.d_pf, (size_t)e->Gtot * e->F * sizeof(float)));
        HIPCHK(e, hipMalloc((void**)&ln.d_stamp, (size_t)e->Gtot * sizeof(uint32_t)));
        HIPCHK(e, hipMemsetAsync(ln.d_stamp, 0, (size_t)e->Gtot * sizeof(uint32_t), e->stream));
        HIPCHK(e, hipMalloc((void**)&ln.d_status, ST_WORDS * sizeof(int)));
        HIPCHK(e, hipMemsetAsync(ln.d_status, 0, ST_WORDS * sizeof(int), e->stream));
        ln.step_id = 0;
        e->lanes.push_back(ln);
    }
    return SBE_OK;
}
}  // namespace

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

int sbe_step_batch(sbe_engine* e, int n_chains, const int32_t* cur_slots, const int32_t* cand_slots,
                   const uint8_t* clusters, const uint8_t* clusters_mask, const int32_t* rows_ptr,
                   const int32_t* changed_objects, const uint8_t* source_rows, const float* weights,
                   const uint8_t* weights_mask, double* group_logliks_out, double* mixture_out,
                   uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, cur_slots); CHECK_PTR(e, cand_slots); CHECK_PTR(e, rows_ptr);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (n_chains < 1 || n_chains > e->n_slots / 2) return fail(e, SBE_ERR_ARG, "n_chains=%d (1..%d: two slots per chain)", n_chains, e->n_slots / 2);
    if ((int64_t)e->Gtot * e->S * 28 > 60 * 1024) return fail(e, SBE_ERR_ARG, "sbe_step_batch: tables too large for the one-launch step (G_total=%d, S=%d)", e->Gtot, e->S);
    const int N = e->N, F = e->F, C = e->C, K = e->G[0];
    if (rows_ptr[0] != 0) return fail(e, SBE_ERR_ARG, "rows_ptr[0] must be 0");
    // SBE_STEP_TIMING=1: per-phase wall clock of this call on stderr (tools/prof_step_batch.py)
    static const bool timing = getenv("SBE_STEP_TIMING") && atoi(getenv("SBE_STEP_TIMING")) != 0;
    using clk = std::chrono::steady_clock;
    clk::time_point tp[12]; int ntp = 0;
    auto mark = [&] { if (timing && ntp < 12) tp[ntp++] = clk::now(); };
    mark();
    {   // argument checks before anything is touched
        std::vector<uint8_t> used(e->n_slots, 0);
        for (int i = 0; i < n_chains; ++i) {
            const int a = cur_slots[i], b = cand_slots[i];
            if (a < 0 || a >= e->n_slots || b < 0 || b >= e->n_slots || a == b) return fail(e, SBE_ERR_ARG, "chain %d: bad slots (%d, %d)", i, a, b);
            if (used[a] || used[b]) return fail(e, SBE_ERR_ARG, "chain %d: slot used by another chain of the batch", i);
            used[a] = used[b] = 1;
            const int nr = rows_ptr[i + 1] - rows_ptr[i];
            if (nr < 0 || nr > e->step_max_rows) return fail(e, SBE_ERR_ARG, "chain %d: %d changed source rows (0..%d per chain in a batched step)", i, nr, e->step_max_rows);
            if (nr > 0 && (!changed_objects || !source_rows)) return fail(e, SBE_ERR_ARG, "changed_objects / source_rows missing");
            for (int k = rows_ptr[i]; k < rows_ptr[i + 1]; ++k)
                if (changed_objects[k] < 0 || changed_objects[k] >= N) return fail(e, SBE_ERR_ARG, "chain %d: object index %d out of range", i, changed_objects[k]);
            const Slot& cur = e->slots[a];
            if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", a);
            for (int c = 0; c < C; ++c)
                if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", a, c);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (e->status_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    for (int i = 0; i < n_chains; ++i)       // never-uploaded patterns of a current slot (first step after set_groups)
        if (e->slots[cur_slots[i]].patterns_dirty) { int rc = upload_patterns_and_weights(e, cur_slots[i]); if (rc) return rc; }
    int rc = ensure_lanes(e, n_chains);
    if (rc) return rc;
    rc = ensure_step_pool(e);
    if (rc) return rc;
    mark();
    // ---- host halves, in parallel over the chains ---------------------------------------------------------------
    if ((int)e->batch_cands.size() < n_chains) e->batch_cands.resize(n_chains);
    if ((int)e->batch_moved.size() < n_chains) e->batch_moved.resize(n_chains);
    std::vector<Slot>& cds = e->batch_cands;
    std::vector<CoreInputs> ins(n_chains);
    // payload blocks: chain i's used prefix (fixed sections + its changed rows) at a running offset of one pinned block
    std::vector<size_t> pay_off(n_chains + 1, 0);
    for (int i = 0; i < n_chains; ++i)
        pay_off[i + 1] = pay_off[i] + (e->sl.rows + (size_t)(rows_ptr[i + 1] - rows_ptr[i]) * F * C + 255) / 256 * 256;
    if (pay_off[n_chains] > e->batch_payload_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->h_batch_payload) { HIPCHK(e, hipHostFree(e->h_batch_payload)); e->h_batch_payload = nullptr; }
        if (e->d_batch_payload) { HIPCHK(e, hipFree(e->d_batch_payload)); e->d_batch_payload = nullptr; }
        e->batch_payload_bytes = pay_off[n_chains] + pay_off[n_chains] / 2;
        HIPCHK(e, hipHostMalloc((void**)&e->h_batch_payload, e->batch_payload_bytes, hipHostMallocDefault));
        HIPCHK(e, hipMalloc((void**)&e->d_batch_payload, e->batch_payload_bytes));
    }
    // ---- device argument blocks, per part: StepCore per chain | StepFinish per chain | candidate slot list --------
    // A large batch is cut into two parts: the host halves of the second run while the device works on the first
    // (measured, headline shape: 256 chains 655 us in two parts against ~740 in one; at 64 chains the per-part
    // fixed costs of the three launches outweigh the overlap -- 245 us in one part, 271 in two).
    int n_parts = n_chains >= 128 ? 2 : 1;
    if (const char* env = getenv("SBE_STEP_PARTS")) n_parts = std::max(1, std::min(atoi(env), n_chains));
    const int per_part = div_up(n_chains, n_parts);
    const size_t part_cores = ((size_t)per_part * sizeof(StepCore) + 255) / 256 * 256;
    const size_t part_fins = ((size_t)per_part * sizeof(StepFinish) + 255) / 256 * 256;
    const size_t part_bytes = part_cores + part_fins + ((size_t)per_part * sizeof(int32_t) + 255) / 256 * 256;
    const size_t meta_bytes = part_bytes * n_parts;
    if (meta_bytes > e->batch_meta_bytes) {
        if (e->d_batch_meta) { HIPCHK(e, hipStreamSynchronize(e->stream)); HIPCHK(e, hipFree(e->d_batch_meta)); }
        e->batch_meta_bytes = meta_bytes + meta_bytes / 2;
        HIPCHK(e, hipMalloc((void**)&e->d_batch_meta, e->batch_meta_bytes));
    }
    std::vector<uint8_t> meta(meta_bytes);
    std::vector<int> rcs(n_chains, SBE_OK);
    std::vector<std::string> errs(n_chains);
    DoneSig batch_done{};
    for (int part = 0; part < n_parts; ++part) {
        const int i0 = part * per_part, i1 = std::min(n_chains, i0 + per_part), np = i1 - i0;
        if (np <= 0) break;
        // the payload goes up in chunks of kCopyChunk chains, each sent as soon as its chains are prepared: the copy
        // (1.6 MB for 64 headline chains, ~40 us) runs under the preparation of the chains behind it
        constexpr int kCopyChunk = 16;
        const int n_copy_chunks = div_up(np, kCopyChunk);
        std::vector<std::atomic<int>> chunk_done(n_copy_chunks);
        for (auto& c : chunk_done) c.store(0, std::memory_order_relaxed);
        int next_copy = 0;
        hipError_t copy_err = hipSuccess;
        auto send_ready = [&]() {
            while (next_copy < n_copy_chunks) {
                const int c0 = i0 + next_copy * kCopyChunk, c1 = std::min(i1, c0 + kCopyChunk);
                if (chunk_done[next_copy].load(std::memory_order_acquire) < c1 - c0) break;
                const hipError_t he = hipMemcpyAsync(e->d_batch_payload + pay_off[c0], e->h_batch_payload + pay_off[c0],
                                                     pay_off[c1] - pay_off[c0], hipMemcpyHostToDevice, e->stream);
                if (he != hipSuccess) copy_err = he;
                ++next_copy;
            }
        };
        const std::function<void()> poll = send_ready;
        e->pool->run(np, [&](int j) {
            const int i = i0 + j;
            const bool regroup = clusters && (!clusters_mask || clusters_mask[i]);
            const bool reweight = weights && (!weights_mask || weights_mask[i]);
            const int r0 = rows_ptr[i], nr = rows_ptr[i + 1] - r0;
            sbe_engine::Lane lane = e->lanes[i];                      // this chain's lane with its slice of the packed payload
            lane.h_payload = e->h_batch_payload + pay_off[i];
            lane.d_payload = e->d_batch_payload + pay_off[i];
            rcs[i] = prepare_step(e, lane, cur_slots[i], cand_slots[i], regroup ? clusters + (size_t)i * K * N : nullptr,
                                  nr ? changed_objects + r0 : nullptr, nr, nr ? source_rows + (size_t)r0 * F * C : nullptr,
                                  reweight ? weights + (size_t)i * F * C : nullptr, cds[i], ins[i], &errs[i], &e->batch_moved[i]);
            chunk_done[j / kCopyChunk].fetch_add(1, std::memory_order_release);
        }, &poll);
        send_ready();
        HIPCHK(e, copy_err);
        for (int i = i0; i < i1; ++i)
            if (rcs[i]) { (void)hipStreamSynchronize(e->stream); return fail(e, rcs[i], "chain %d: %s", i, errs[i].c_str()); }
        if (part == 0) mark();
        uint8_t* pm = meta.data() + (size_t)part * part_bytes;
        StepCore* cores = reinterpret_cast<StepCore*>(pm);
        StepFinish* fins = reinterpret_cast<StepFinish*>(pm + part_cores);
        int32_t* slot_list = reinterpret_cast<int32_t*>(pm + part_cores + part_fins);
        size_t lds = 0; int max_blocks = 0;
        for (int i = i0; i < i1; ++i) {
            size_t l = 0; int nb = 0;
            rc = build_step_core(e, e->lanes[i], cur_slots[i], cand_slots[i], ins[i], cores[i - i0], l, nb, n_chains);
            if (rc) return rc;
            lds = std::max(lds, l); max_blocks = std::max(max_blocks, nb);
            fins[i - i0] = make_step_finish_lane(e, e->lanes[i]);
            slot_list[i - i0] = cand_slots[i];
        }
        if (part == 0) mark();
        uint8_t* dm = e->d_batch_meta + (size_t)part * part_bytes;
        rc = upload(e, dm, pm, part_bytes);
        if (rc) return rc;
        if (part == 0) mark();
        k_step_core_batch<<<dim3(max_blocks, np), kBlock, lds, e->stream>>>(reinterpret_cast<const StepCore*>(dm));
        HIPCHK(e, hipGetLastError());
        if (part == 0) mark();
        for (int i = i0; i < i1; ++i) {
            commit_src_sync(e, cur_slots[i], cand_slots[i], changed_objects ? changed_objects + rows_ptr[i] : nullptr,
                            rows_ptr[i + 1] - rows_ptr[i]);
            commit_ids_sync(e, cur_slots[i], cand_slots[i], e->batch_moved[i]);
            std::swap(e->slots[cand_slots[i]], cds[i]);                               // (swap: both keep their storage)
        }
        if (part == 0) mark();
        rc = launch_mixture(e, 0, np, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, nullptr,
                            cand_slots + i0, reinterpret_cast<const int32_t*>(dm + part_cores + part_fins),
                            reinterpret_cast<const StepFinish*>(dm + part_cores), part == n_parts - 1 ? &batch_done : nullptr);
        if (rc) return rc;
    }
    mark();
    { int wrc = wait_done(e, batch_done); if (wrc) return wrc; }       // (the last part's reduction carries the flag)
    mark();
    // every chain's results are delivered; a chain whose proposal was malformed (its own data-check words) is reported
    // by index after that -- the other chains' outputs stay usable
    int first_bad = SBE_OK;
    std::string first_msg;
    for (int i = 0; i < n_chains; ++i) {
        rc = read_step_results_lane(e, e->lanes[i].h_step, cand_slots[i], group_logliks_out + (size_t)i * e->Gtot, mixture_out + i,
                               changed_groups_out ? changed_groups_out + (size_t)i * e->Gtot : nullptr, "rows", e->lanes[i].d_status, i);
        if (rc && !first_bad) { first_bad = rc; first_msg = e->last_error; }
    }
    if (first_bad) return fail(e, first_bad, "%s", first_msg.c_str());
    mark();
    if (timing && ntp == 10) {
        auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(tp[b] - tp[a]).count(); };
        fprintf(stderr, "[sbe_step_batch] %d chains (first part): checks %.1f | prepare (pool) %.1f | step cores %.1f | upload %.1f | launch core %.1f | "
                        "slot moves %.1f | mixture + reduce launches and the other parts %.1f | wait for the device %.1f | read results %.1f us\n", n_chains,
                us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(8, 9));
    }
    return SBE_OK;
}

// ---- batched step, DELTA form (round 3) -------------------------------------------------------
// The same step as sbe_step_batch with the proposal handed over as what it is -- a few moved objects:
//     moved_objects / moved_cluster (CSR by moved_ptr): objects that change cluster and their new cluster (-1: none)
//     changed_objects / source_rows (CSR by rows_ptr):  objects whose source rows change, each listed once
// A chain's two slots differ only in what its LAST step changed (SrcSync records for the source rows and for the id
// arrays), so the candidate is built by PATCHING: host mirror, device id arrays and source rows in O(delta); no
// [K][N] matrix is scanned, no slot state copied, no [N]-sized array packed or sent.  A chain whose records do not hold
// (first sweep, a slot touched by another call) or whose step changes the SET of has_components patterns / overflows
// the tuple table goes through sbe_step_batch itself (cluster matrix rebuilt from the ids); results are identical.
namespace {

struct DeltaPlan {           // per chain: payload section offsets (bytes from the chain's base) and capacities
    size_t subset, sub_row, sub_gid0, patch_n, patch_gid, patch_pid, patch_tid, stale, objects, tuple_g, tuple_p, patbits, weights, rows, total;
};

inline size_t al16(size_t v) { return (v + 15) / 16 * 16; }

DeltaPlan plan_delta(const sbe_engine* e, int n_mv, int n_changed, int n_last_ids, int n_last_src, bool reweight) {
    DeltaPlan p{};
    const size_t n_sub = (size_t)n_mv + n_changed, n_patch = (size_t)n_mv + n_last_ids;
    size_t o = 0;
    p.subset = o;    o = al16(o + n_sub * 4);
    p.sub_row = o;   o = al16(o + n_sub * 2);
    p.sub_gid0 = o;  o = al16(o + n_sub * 2);
    p.patch_n = o;   o = al16(o + n_patch * 4);
    p.patch_gid = o; o = al16(o + n_patch * 2);
    p.patch_pid = o; o = al16(o + n_patch);
    p.patch_tid = o; o = al16(o + n_patch);
    p.stale = o;     o = al16(o + (size_t)n_last_src * 4);
    p.objects = o;   o = al16(o + (size_t)n_changed * 4);
    p.tuple_g = o;   o = al16(o + (size_t)kMaxTuples * kMaxComponents * 2);
    p.tuple_p = o;   o = al16(o + (size_t)kMaxTuples);
    p.patbits = o;   o = al16(o + (size_t)e->Pmax * 4);
    p.weights = o;   o = al16(o + (reweight ? (size_t)e->F * e->C * 4 : 0));
    p.rows = o;      o = al16(o + (size_t)n_changed * e->F * e->C);
    p.total = (o + 255) / 256 * 256;
    return p;
}

// Host half of one chain in the delta form.  Patches e->slots[cand_slot] in place (it holds the current slot's state
// except the entries of the last step's moved objects).  Returns 1 when the chain must take the classic path instead
// (pattern set changes, tuple table full); the candidate's host state is then unspecified (the classic path rewrites it).
int prepare_step_delta(sbe_engine* e, uint8_t* h_base, const uint8_t* d_base, const DeltaPlan& L, int cur_slot, int cand_slot,
                       const int32_t* mv_objects, const int32_t* mv_cluster, int n_mv, const int32_t* changed_objects, int n_changed,
                       const uint8_t* source_rows, const float* weights, CoreInputs& in, std::vector<int32_t>& moved_out) {
    const int F = e->F, C = e->C;
    const Slot& cur = e->slots[cur_slot];
    Slot& cd = e->slots[cand_slot];
    const sbe_engine::SrcSync& irec = e->ids_sync[cand_slot];
    // 1. the candidate's host mirror back to the current slot's state: entries of the last step's moved objects
    for (int32_t n : irec.diff) {
        cd.h_gid[n] = cur.h_gid[n]; cd.h_pid[n] = cur.h_pid[n]; cd.h_tid[n] = cur.h_tid[n]; cd.h_toff[n] = cur.h_toff[n];
    }
    cd.patterns = cur.patterns; cd.n_tuples = cur.n_tuples;
    cd.h_tuple_g = cur.h_tuple_g; cd.h_tuple_p = cur.h_tuple_p; cd.pat_cnt = cur.pat_cnt; cd.tup_cnt = cur.tup_cnt;
    cd.inc_ok = cur.inc_ok; cd.patterns_dirty = false;
    cd.groups_set = cur.groups_set; cd.weights_set = true; cd.source_set = cur.source_set;
    cd.counts_set = cur.counts_set;
    // 2. this step's moves
    static thread_local std::vector<int32_t> mv;
    static thread_local std::vector<uint16_t> mv_old;
    mv.clear(); mv_old.clear();
    for (int i = 0; i < n_mv; ++i) {
        const int n = mv_objects[i];
        const uint16_t g_new = mv_cluster[i] < 0 ? kNoGroup : (uint16_t)mv_cluster[i];
        if (cd.h_gid[n] == g_new) continue;                              // (not a move)
        mv.push_back(n); mv_old.push_back(cd.h_gid[n]);
        cd.h_gid[n] = g_new;
    }
    if (!mv.empty()) {
        if (e->opt_step_derive == 1 || !update_patterns_and_tuples(e, cd, mv.data(), mv_old.data(), (int)mv.size())) return 1;
        cd.group_epoch = ++e->epoch_counter;
    } else cd.group_epoch = cur.group_epoch;
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    moved_out = mv;
    // 3. payload
    uint8_t* st = h_base;
    in = CoreInputs{};
    int32_t* sub = reinterpret_cast<int32_t*>(st + L.subset);
    int16_t* sub_row = reinterpret_cast<int16_t*>(st + L.sub_row);
    uint16_t* sub_gid0 = reinterpret_cast<uint16_t*>(st + L.sub_gid0);
    int n_subset = 0;
    {   // sorted union of moved and changed objects; per entry its row in `rows` (-1: none) and its candidate cluster id
        static thread_local std::vector<std::pair<int32_t, int32_t>> ch;     // (object, row)
        ch.clear();
        for (int i = 0; i < n_changed; ++i) ch.emplace_back(changed_objects[i], i);
        std::sort(ch.begin(), ch.end());
        std::sort(mv.begin(), mv.end());
        size_t a = 0, b = 0;
        while (a < mv.size() || b < ch.size()) {
            int32_t n; int r = -1;
            if (b == ch.size() || (a < mv.size() && mv[a] < ch[b].first)) n = mv[a++];
            else { n = ch[b].first; r = ch[b].second; if (a < mv.size() && mv[a] == n) ++a; ++b; }
            sub[n_subset] = n; sub_row[n_subset] = (int16_t)r; sub_gid0[n_subset] = cd.h_gid[n];
            ++n_subset;
        }
    }
    int n_patch = 0;
    {   // id entries to (re)write in the candidate's device arrays: last step's leftovers and this step's moves
        int32_t* pn = reinterpret_cast<int32_t*>(st + L.patch_n);
        uint16_t* pg = reinterpret_cast<uint16_t*>(st + L.patch_gid);
        uint8_t* pp = st + L.patch_pid; uint8_t* pt = st + L.patch_tid;
        auto put = [&](int32_t n) { pn[n_patch] = n; pg[n_patch] = cd.h_gid[n]; pp[n_patch] = cd.h_pid[n]; pt[n_patch] = cd.h_tid[n]; ++n_patch; };
        for (int32_t n : irec.diff) put(n);
        for (int32_t n : mv) put(n);                                      // (an object in both lists: same value twice)
    }
    const bool tables_changed = cd.patterns != cur.patterns || cd.h_tuple_p != cur.h_tuple_p || cd.h_tuple_g != cur.h_tuple_g;
    if (tables_changed) {
        memcpy(st + L.tuple_g, cd.h_tuple_g.data(), (size_t)kMaxTuples * kMaxComponents * 2);
        memcpy(st + L.tuple_p, cd.h_tuple_p.data(), kMaxTuples);
        memset(st + L.patbits, 0, (size_t)e->Pmax * 4);
        memcpy(st + L.patbits, cd.patterns.data(), cd.patterns.size() * 4);
        in.tuple_g = d_base + L.tuple_g; in.tuple_p = d_base + L.tuple_p; in.patbits = d_base + L.patbits;
    }
    if (weights) { memcpy(st + L.weights, weights, (size_t)F * C * 4); in.weights = d_base + L.weights; }
    if (n_changed > 0) {
        memcpy(st + L.objects, changed_objects, (size_t)n_changed * 4);
        memcpy(st + L.rows, source_rows, (size_t)n_changed * F * C);
        in.rows = d_base + L.rows;
        in.objects = reinterpret_cast<const int32_t*>(d_base + L.objects);
        in.n_changed = n_changed;
    }
    in.subset = reinterpret_cast<const int32_t*>(d_base + L.subset); in.n_subset = n_subset;
    in.sub_row = reinterpret_cast<const int16_t*>(d_base + L.sub_row);
    in.sub_gid0 = reinterpret_cast<const uint16_t*>(d_base + L.sub_gid0);
    in.patch_n = reinterpret_cast<const int32_t*>(d_base + L.patch_n);
    in.patch_gid = reinterpret_cast<const uint16_t*>(d_base + L.patch_gid);
    in.patch_pid = d_base + L.patch_pid; in.patch_tid = d_base + L.patch_tid; in.n_patch = n_patch;
    in.P = (int)cd.patterns.size();
    {   // source rows to bring over from the current slot (the last step's rows that this step does not rewrite)
        const sbe_engine::SrcSync& rec = e->src_sync[cand_slot];
        int32_t* stale = reinterpret_cast<int32_t*>(st + L.stale);
        int ns = 0;
        for (int32_t n : rec.diff) {
            bool rewritten = false;
            for (int i = 0; i < n_changed && !rewritten; ++i) rewritten = changed_objects[i] == n;
            if (!rewritten) stale[ns++] = n;
        }
        in.full_src_copy = false;
        in.stale = reinterpret_cast<const int32_t*>(d_base + L.stale);
        in.n_stale = ns;
    }
    return 0;
}

bool sync_valid(const sbe_engine::SrcSync& rec, const sbe_engine::SrcSync& peer, int peer_slot, int cap) {
    return rec.peer == peer_slot && rec.peer_version == peer.version && rec.own_version == rec.version && (int)rec.diff.size() <= cap;
}

}  // namespace

int sbe_step_batch_delta(sbe_engine* e, int n_chains, const int32_t* cur_slots, const int32_t* cand_slots,
                         const int32_t* moved_ptr, const int32_t* moved_objects, const int32_t* moved_cluster,
                         const int32_t* rows_ptr, const int32_t* changed_objects, const uint8_t* source_rows,
                         const float* weights, const uint8_t* weights_mask, double* group_logliks_out, double* mixture_out,
                         uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, cur_slots); CHECK_PTR(e, cand_slots); CHECK_PTR(e, moved_ptr); CHECK_PTR(e, rows_ptr);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    const auto t_start = std::chrono::steady_clock::now();
    std::chrono::steady_clock::time_point tq[8]; int nq = 0;
    auto markd = [&] { if (nq < 8) tq[nq++] = std::chrono::steady_clock::now(); };
    if (n_chains < 1 || n_chains > e->n_slots / 2) return fail(e, SBE_ERR_ARG, "n_chains=%d (1..%d: two slots per chain)", n_chains, e->n_slots / 2);
    if ((int64_t)e->Gtot * e->S * 28 > 60 * 1024) return fail(e, SBE_ERR_ARG, "sbe_step_batch_delta: tables too large for the one-launch step (G_total=%d, S=%d)", e->Gtot, e->S);
    const int N = e->N, F = e->F, C = e->C, K = e->G[0];
    if (rows_ptr[0] != 0 || moved_ptr[0] != 0) return fail(e, SBE_ERR_ARG, "rows_ptr[0] / moved_ptr[0] must be 0");
    {
        std::vector<uint8_t> used(e->n_slots, 0);
        std::vector<uint32_t> seen(N, 0), seen_mv(N, 0);
        for (int i = 0; i < n_chains; ++i) {
            const int a = cur_slots[i], b = cand_slots[i];
            if (a < 0 || a >= e->n_slots || b < 0 || b >= e->n_slots || a == b) return fail(e, SBE_ERR_ARG, "chain %d: bad slots (%d, %d)", i, a, b);
            if (used[a] || used[b]) return fail(e, SBE_ERR_ARG, "chain %d: slot used by another chain of the batch", i);
            used[a] = used[b] = 1;
            const int nr = rows_ptr[i + 1] - rows_ptr[i], nm = moved_ptr[i + 1] - moved_ptr[i];
            if (nr < 0 || nr > e->step_max_rows) return fail(e, SBE_ERR_ARG, "chain %d: %d changed source rows (0..%d per chain in a batched step)", i, nr, e->step_max_rows);
            if (nm < 0 || nm > N) return fail(e, SBE_ERR_ARG, "chain %d: %d moved objects", i, nm);
            if ((nr > 0 && (!changed_objects || !source_rows)) || (nm > 0 && (!moved_objects || !moved_cluster))) return fail(e, SBE_ERR_ARG, "chain %d: delta arrays missing", i);
            for (int k = rows_ptr[i]; k < rows_ptr[i + 1]; ++k) {
                const int n = changed_objects[k];
                if (n < 0 || n >= N) return fail(e, SBE_ERR_ARG, "chain %d: object index %d out of range", i, n);
                if (seen[n] == (uint32_t)(2 * i + 1)) return fail(e, SBE_ERR_ARG, "chain %d: object %d listed twice in changed_objects", i, n);
                seen[n] = (uint32_t)(2 * i + 1);
            }
            for (int k = moved_ptr[i]; k < moved_ptr[i + 1]; ++k) {
                const int n = moved_objects[k];
                if (n < 0 || n >= N) return fail(e, SBE_ERR_ARG, "chain %d: moved object index %d out of range", i, n);
                if (moved_cluster[k] < -1 || moved_cluster[k] >= K) return fail(e, SBE_ERR_ARG, "chain %d: cluster %d out of range [-1,%d)", i, moved_cluster[k], K);
                // a repeated moved object would be patched twice (pattern counts decremented for a pattern the object
                // was never in, its count delta added twice) while the matrix form resolves it last-wins: rejected
                if (seen_mv[n] == (uint32_t)(i + 1)) return fail(e, SBE_ERR_ARG, "chain %d: object %d listed twice in moved_objects", i, n);
                seen_mv[n] = (uint32_t)(i + 1);
            }
            const Slot& cur = e->slots[a];
            if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", a);
            for (int c = 0; c < C; ++c)
                if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", a, c);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (e->status_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    for (int i = 0; i < n_chains; ++i)
        if (e->slots[cur_slots[i]].patterns_dirty) { int rc = upload_patterns_and_weights(e, cur_slots[i]); if (rc) return rc; }
    int rc = ensure_lanes(e, n_chains);
    if (rc) return rc;
    rc = ensure_step_pool(e);
    if (rc) return rc;
    if ((int)e->batch_moved.size() < n_chains) e->batch_moved.resize(n_chains);
    markd();                                     // 0: checks done
    // ---- which chains can be patched; their payload plans -------------------------------------------------------------
    std::vector<int> fast, slow;
    std::vector<DeltaPlan> plans(n_chains);
    std::vector<size_t> pay_off(n_chains + 1, 0);
    for (int i = 0; i < n_chains; ++i) {
        const int a = cur_slots[i], b = cand_slots[i];
        const bool ok = sync_valid(e->ids_sync[b], e->ids_sync[a], a, e->step_max_rows) &&
                        sync_valid(e->src_sync[b], e->src_sync[a], a, e->step_max_rows) &&
                        e->slots[a].inc_ok && e->slots[a].n_tuples > 0 && e->slots[b].h_gid.size() == e->slots[a].h_gid.size();
        if (ok) {
            const bool reweight = weights && (!weights_mask || weights_mask[i]);
            plans[i] = plan_delta(e, moved_ptr[i + 1] - moved_ptr[i], rows_ptr[i + 1] - rows_ptr[i], (int)e->ids_sync[b].diff.size(),
                                  (int)e->src_sync[b].diff.size(), reweight);
            pay_off[i + 1] = pay_off[i] + plans[i].total;
            fast.push_back(i);
        } else { pay_off[i + 1] = pay_off[i]; slow.push_back(i); }
    }
    // Chains that cannot be patched run through the classic entry point (cluster matrices rebuilt from the ids).  That call
    // uses the lanes, the payload block and the per-chain scratch of ITS chain numbering and synchronises: it runs either
    // before the patched chains are prepared or after their results have been read, never in between.
    auto run_classic = [&](const std::vector<int>& which, int& rc_out, std::string& msg_out) {
        rc_out = SBE_OK;
        if (which.empty()) return;
        const int ns = (int)which.size();
        std::vector<int32_t> s_cur(ns), s_cand(ns), s_ptr(ns + 1, 0), s_objs;
        std::vector<uint8_t> s_cl((size_t)ns * K * N, 0), s_rows, s_wm(ns, 0);
        std::vector<float> s_w(weights ? (size_t)ns * F * C : 0);
        for (int j = 0; j < ns; ++j) {
            const int i = which[j];
            s_cur[j] = cur_slots[i]; s_cand[j] = cand_slots[i];
            const Slot& cur = e->slots[cur_slots[i]];
            std::vector<uint16_t> ids(cur.h_gid.begin(), cur.h_gid.begin() + N);
            for (int k = moved_ptr[i]; k < moved_ptr[i + 1]; ++k) ids[moved_objects[k]] = moved_cluster[k] < 0 ? kNoGroup : (uint16_t)moved_cluster[k];
            uint8_t* cl = s_cl.data() + (size_t)j * K * N;
            for (int n = 0; n < N; ++n) if (ids[n] != kNoGroup) cl[(size_t)ids[n] * N + n] = 1;
            const int r0 = rows_ptr[i], nr = rows_ptr[i + 1] - r0;
            s_ptr[j + 1] = s_ptr[j] + nr;
            if (nr) {
                s_objs.insert(s_objs.end(), changed_objects + r0, changed_objects + r0 + nr);
                s_rows.insert(s_rows.end(), source_rows + (size_t)r0 * F * C, source_rows + (size_t)(r0 + nr) * F * C);
            }
            if (weights && (!weights_mask || weights_mask[i])) { s_wm[j] = 1; memcpy(&s_w[(size_t)j * F * C], weights + (size_t)i * F * C, (size_t)F * C * 4); }
        }
        std::vector<double> s_glh((size_t)ns * e->Gtot), s_mix(ns);
        std::vector<uint8_t> s_chg((size_t)ns * e->Gtot);
        rc_out = sbe_step_batch(e, ns, s_cur.data(), s_cand.data(), s_cl.data(), nullptr, s_ptr.data(), s_objs.empty() ? nullptr : s_objs.data(),
                                s_rows.empty() ? nullptr : s_rows.data(), weights ? s_w.data() : nullptr, weights ? s_wm.data() : nullptr,
                                s_glh.data(), s_mix.data(), s_chg.data());
        if (rc_out) { msg_out = e->last_error; return; }
        for (int j = 0; j < ns; ++j) {
            const int i = which[j];
            memcpy(group_logliks_out + (size_t)i * e->Gtot, &s_glh[(size_t)j * e->Gtot], (size_t)e->Gtot * sizeof(double));
            mixture_out[i] = s_mix[j];
            if (changed_groups_out) memcpy(changed_groups_out + (size_t)i * e->Gtot, &s_chg[(size_t)j * e->Gtot], (size_t)e->Gtot);
        }
    };
    int slow_rc = SBE_OK; std::string slow_msg;
    run_classic(slow, slow_rc, slow_msg);        // (chains without usable records: before the patched chains touch anything)
    markd();                                     // 1: plans (+ the unpatched chains)
    // ---- host halves of the patched chains (pool) ----------------------------------------------------------------------
    // The delta payload is small (~10 KB per chain): it goes up in ONE copy together with the launch arguments (every
    // copy-engine operation costs ~10 us of latency in the stream; the chunked upload of the matrix form paid five).
    std::vector<CoreInputs> ins(n_chains);
    std::vector<int> fb(n_chains, 0);
    const int nf = (int)fast.size();
    const size_t part_cores = ((size_t)std::max(nf, 1) * sizeof(StepCore) + 255) / 256 * 256;
    const size_t part_fins = ((size_t)std::max(nf, 1) * sizeof(StepFinish) + 255) / 256 * 256;
    const size_t meta_bytes = part_cores + part_fins + ((size_t)std::max(nf, 1) * sizeof(int32_t) + 255) / 256 * 256;
    const size_t meta_off = pay_off[n_chains];
    if (meta_off + meta_bytes > e->batch_payload_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->h_batch_payload) { HIPCHK(e, hipHostFree(e->h_batch_payload)); e->h_batch_payload = nullptr; }
        if (e->d_batch_payload) { HIPCHK(e, hipFree(e->d_batch_payload)); e->d_batch_payload = nullptr; }
        e->batch_payload_bytes = (meta_off + meta_bytes) * 3 / 2;
        HIPCHK(e, hipHostMalloc((void**)&e->h_batch_payload, e->batch_payload_bytes, hipHostMallocDefault));
        HIPCHK(e, hipMalloc((void**)&e->d_batch_payload, e->batch_payload_bytes));
    }
    if (nf > 0) {
        e->pool->run(nf, [&](int j) {
            const int i = fast[j];
            const bool reweight = weights && (!weights_mask || weights_mask[i]);
            const int r0 = rows_ptr[i], nr = rows_ptr[i + 1] - r0, m0 = moved_ptr[i], nm = moved_ptr[i + 1] - m0;
            fb[i] = prepare_step_delta(e, e->h_batch_payload + pay_off[i], e->d_batch_payload + pay_off[i], plans[i], cur_slots[i], cand_slots[i],
                                       nm ? moved_objects + m0 : nullptr, nm ? moved_cluster + m0 : nullptr, nm,
                                       nr ? changed_objects + r0 : nullptr, nr, nr ? source_rows + (size_t)r0 * F * C : nullptr,
                                       reweight ? weights + (size_t)i * F * C : nullptr, ins[i], e->batch_moved[i]);
        });
    }
    markd();                                     // 2: host halves
    std::vector<int> go, late;                             // patched chains that stay on the fast path / that turned out not to
    for (int i : fast) { if (fb[i]) { late.push_back(i); bump_ids(e, cand_slots[i]); } else go.push_back(i); }
    const int ng = (int)go.size();
    DoneSig fast_done{};
    if (ng > 0) {
        uint8_t* pm = e->h_batch_payload + meta_off;
        uint8_t* dm = e->d_batch_payload + meta_off;
        StepCore* cores = reinterpret_cast<StepCore*>(pm);
        StepFinish* fins = reinterpret_cast<StepFinish*>(pm + part_cores);
        int32_t* slot_list = reinterpret_cast<int32_t*>(pm + part_cores + part_fins);
        std::vector<int32_t> cand_go(ng);
        size_t lds = 0; int max_blocks = 0;
        for (int j = 0; j < ng; ++j) {
            const int i = go[j];
            size_t l = 0; int nb = 0;
            rc = build_step_core(e, e->lanes[i], cur_slots[i], cand_slots[i], ins[i], cores[j], l, nb, ng);
            if (rc) return rc;
            lds = std::max(lds, l); max_blocks = std::max(max_blocks, nb);
            fins[j] = make_step_finish_lane(e, e->lanes[i]);
            slot_list[j] = cand_go[j] = cand_slots[i];
        }
        HIPCHK(e, hipMemcpyAsync(e->d_batch_payload, e->h_batch_payload, meta_off + meta_bytes, hipMemcpyHostToDevice, e->stream));
        k_step_core_batch<<<dim3(max_blocks, ng), kBlock, lds, e->stream>>>(reinterpret_cast<const StepCore*>(dm));
        HIPCHK(e, hipGetLastError());
        markd();                                 // 3: step cores built, uploaded, launched
        rc = launch_mixture(e, 0, ng, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, nullptr,
                            cand_go.data(), reinterpret_cast<const int32_t*>(dm + part_cores + part_fins),
                            reinterpret_cast<const StepFinish*>(dm + part_cores), &fast_done);
        for (int j = 0; j < ng; ++j) {           // (bookkeeping under the device work: everything is enqueued)
            const int i = go[j];
            commit_src_sync(e, cur_slots[i], cand_slots[i], changed_objects ? changed_objects + rows_ptr[i] : nullptr, rows_ptr[i + 1] - rows_ptr[i]);
            commit_ids_sync(e, cur_slots[i], cand_slots[i], e->batch_moved[i]);
        }
        if (rc) return rc;
    }
    static const bool timing_d = getenv("SBE_STEP_TIMING") && atoi(getenv("SBE_STEP_TIMING")) != 0;
    const auto t_enq = std::chrono::steady_clock::now();
    { int wrc = wait_done(e, fast_done); if (wrc) return wrc; }        // (no patched chain: nothing was launched, plain wait)
    if (timing_d) {
        const auto t_done = std::chrono::steady_clock::now();
        auto us = [&](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[sbe_step_batch_delta] %d chains: %d patched, %d + %d through sbe_step_batch | checks %.1f | plans %.1f | host halves (pool) %.1f | cores + launch %.1f | "
                        "mixture launches + records %.1f | wait %.1f us\n", n_chains, ng, (int)slow.size(), (int)late.size(), nq > 0 ? us(t_start, tq[0]) : 0.0,
                nq > 1 ? us(tq[0], tq[1]) : 0.0, nq > 2 ? us(tq[1], tq[2]) : 0.0, nq > 3 ? us(tq[2], tq[3]) : 0.0, nq > 3 ? us(tq[3], t_enq) : 0.0, us(t_enq, t_done));
    }
    int first_bad = slow_rc; std::string first_msg = slow_msg;
    for (int i : go) {
        rc = read_step_results_lane(e, e->lanes[i].h_step, cand_slots[i], group_logliks_out + (size_t)i * e->Gtot, mixture_out + i,
                                    changed_groups_out ? changed_groups_out + (size_t)i * e->Gtot : nullptr, "rows", e->lanes[i].d_status, i);
        if (rc && !first_bad) { first_bad = rc; first_msg = e->last_error; }
    }
    // ---- chains whose step turned out to need the full derivation: now that the patched chains' results are out ---------
    if (!late.empty()) {
        int late_rc = SBE_OK; std::string late_msg;
        run_classic(late, late_rc, late_msg);
        if (late_rc && !first_bad) { first_bad = late_rc; first_msg = late_msg; }
    }
    if (first_bad) return fail(e, first_bad, "%s", first_msg.c_str());
    return SBE_OK;
}

// The single-chain step in delta form: sbe_step with the proposal as moved objects + changed rows (see
// sbe_step_batch_delta); the payload sits in the lane's host-mapped block, so no copy-engine operation is in the chain.
int sbe_step_delta(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* moved_objects, const int32_t* moved_cluster, int n_moved,
                   const int32_t* changed_objects, int n_changed, const uint8_t* source_rows, const float* weights,
                   double* group_logliks_out, double* mixture_out, uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    const int N = e->N, C = e->C, K = e->G[0];
    if (n_moved < 0 || n_moved > N || (n_moved > 0 && (!moved_objects || !moved_cluster))) return fail(e, SBE_ERR_ARG, "moved_objects / moved_cluster missing for n_moved=%d", n_moved);
    if (n_changed < 0 || (n_changed > 0 && (!changed_objects || !source_rows))) return fail(e, SBE_ERR_ARG, "changed_objects / source_rows missing for n_changed=%d", n_changed);
    bool dup = false;                  // a repeated object in either list: the matrix form resolves it (last entry wins)
    {
        static thread_local std::vector<uint32_t> stamp;
        static thread_local uint32_t epoch = 0;
        if ((int)stamp.size() < N || ++epoch == 0) { stamp.assign(N, 0); epoch = 1; }
        for (int i = 0; i < n_moved; ++i) {
            if (moved_objects[i] < 0 || moved_objects[i] >= N) return fail(e, SBE_ERR_ARG, "moved object index %d out of range", moved_objects[i]);
            if (moved_cluster[i] < -1 || moved_cluster[i] >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [-1,%d)", moved_cluster[i], K);
            dup = dup || stamp[moved_objects[i]] == epoch;
            stamp[moved_objects[i]] = epoch;
        }
    }
    for (int i = 0; i < n_changed; ++i) {
        if (changed_objects[i] < 0 || changed_objects[i] >= N) return fail(e, SBE_ERR_ARG, "object index %d out of range", changed_objects[i]);
        for (int j = 0; j < i && !dup && n_changed <= 64; ++j) dup = changed_objects[j] == changed_objects[i];
    }
    const Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    for (int c = 0; c < C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", cur_slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    // the matrix form serves whatever the patching form cannot: no usable records, a large or repeated row list, a
    // step that needs the full derivation, tables too large for the one-launch step
    auto classic = [&]() {
        std::vector<uint16_t> ids(cur.h_gid.begin(), cur.h_gid.begin() + N);
        for (int k = 0; k < n_moved; ++k) ids[moved_objects[k]] = moved_cluster[k] < 0 ? kNoGroup : (uint16_t)moved_cluster[k];
        std::vector<uint8_t> cl((size_t)std::max(K, 1) * N, 0);
        for (int n = 0; n < N; ++n) if (ids[n] != kNoGroup) cl[(size_t)ids[n] * N + n] = 1;
        return sbe_step(e, cur_slot, cand_slot, n_moved > 0 ? cl.data() : nullptr, changed_objects, n_changed, source_rows, weights,
                        group_logliks_out, mixture_out, changed_groups_out);
    };
    const bool ok = !dup && n_changed <= std::min(e->step_max_rows, 64) && e->opt_step_form == 0 && !cur.patterns_dirty &&
                    (int64_t)e->Gtot * e->S * 28 <= 60 * 1024 &&
                    sync_valid(e->ids_sync[cand_slot], e->ids_sync[cur_slot], cur_slot, e->step_max_rows) &&
                    sync_valid(e->src_sync[cand_slot], e->src_sync[cur_slot], cur_slot, e->step_max_rows) &&
                    cur.inc_ok && cur.n_tuples > 0 && e->slots[cand_slot].h_gid.size() == cur.h_gid.size();
    if (!ok) return classic();
    const DeltaPlan plan = plan_delta(e, n_moved, n_changed, (int)e->ids_sync[cand_slot].diff.size(), (int)e->src_sync[cand_slot].diff.size(), weights != nullptr);
    if (plan.total > e->sl.total) return classic();
    if (e->status_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    CoreInputs in;
    if (prepare_step_delta(e, e->h_step_payload, e->d_step_payload, plan, cur_slot, cand_slot, moved_objects, moved_cluster, n_moved,
                           changed_objects, n_changed, source_rows, weights, in, e->step_moved)) {
        bump_ids(e, cand_slot);
        return classic();
    }
    int rc = launch_step_core(e, cur_slot, cand_slot, in);
    if (rc) return rc;
    commit_src_sync(e, cur_slot, cand_slot, changed_objects, n_changed);
    commit_ids_sync(e, cur_slot, cand_slot, e->step_moved);
    StepFinish fin = make_step_finish(e);
    DoneSig done;
    rc = launch_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, &fin, nullptr, nullptr,
                        nullptr, &done);
    if (rc) return rc;
    rc = wait_done(e, done);
    if (rc) return rc;
    return read_step_results(e, cand_slot, group_logliks_out, mixture_out, changed_groups_out, "rows");
}

// ---- one-call Gibbs source step (GibbsSampleSource._propose, operators.py:495-552, on the resident state) ------
// candidate = current with the source of the listed objects redrawn from its posterior ON THE DEVICE; count delta,
// every table, both transition log-probabilities, collapsed per-group and mixture log-likelihood of the candidate:
// five launches, one synchronisation; objects (and the caller's uniforms) are read from host-mapped memory, all
// results arrive through host-mapped memory.
int sbe_gibbs_step(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                   double prior_temperature, int from_prior, const double* z, double* log_q_out, double* log_q_back_out,
                   double* group_logliks_out, double* mixture_out, uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, log_q_out); CHECK_PTR(e, log_q_back_out); CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    if (n_sub < 1) return fail(e, SBE_ERR_ARG, "n_sub=%d (nothing to resample)", n_sub);
    CHECK_PTR(e, objects);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    for (int c = 0; c < e->C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c] || !cur.probs_set[c])
            return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration / probability tables of component %d not set", cur_slot, c);
    for (int i = 0; i < n_sub; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    if ((int64_t)e->Gtot * e->S * 28 > 60 * 1024)
        return fail(e, SBE_ERR_ARG, "one-call Gibbs step: tables too large for the fused table kernel (G_total=%d, S=%d)", e->Gtot, e->S);
    HIPCHK(e, hipSetDevice(e->device));
    if (cur.patterns_dirty) { int rc = upload_patterns_and_weights(e, cur_slot); if (rc) return rc; }
    if (e->status_pending) {                  // deliver a deferred data check before this step reuses the words
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    const int N = e->N, Np = e->Np, F = e->F, C = e->C;
    const int64_t n_obs = (int64_t)n_sub * F;
    // host-mapped inputs: objects | row_of marks | uniforms (when they are few; a large block is copied instead)
    const size_t ob = ((size_t)n_sub * sizeof(int32_t) + 255) / 256 * 256;
    const size_t rb = ((size_t)Np * sizeof(int16_t) + 255) / 256 * 256;
    const size_t zbytes = z ? (size_t)n_obs * sizeof(double) : 0;
    const bool z_mapped = zbytes <= ((size_t)1 << 19);
    int rc = ensure_io(e, ob + rb + (z_mapped ? zbytes : 0));
    if (rc) return rc;
    memcpy(e->h_io, objects, (size_t)n_sub * sizeof(int32_t));
    int16_t* row_of = reinterpret_cast<int16_t*>(e->h_io + ob);
    std::fill(row_of, row_of + Np, (int16_t)-1);
    for (int i = 0; i < n_sub; ++i) row_of[objects[i]] = 0;
    const int32_t* d_obj = reinterpret_cast<const int32_t*>(e->d_io);
    // device scratch: selected probabilities (forward / back), their partial log sums, uniforms if copied
    const size_t pb = ((size_t)n_obs * sizeof(float) + 255) / 256 * 256;
    const size_t zb = (z && !z_mapped) ? (zbytes + 255) / 256 * 256 : 0;
    const int nblk = div_up(n_obs, kBlock);                   // one log-sum partial per block of the two posterior kernels
    const size_t qb_bytes = ((size_t)nblk * sizeof(double) + 255) / 256 * 256;
    rc = ensure_scratch(e, 2 * pb + 2 * qb_bytes + zb);
    if (rc) return rc;
    float* d_psel_f = (float*)e->d_scratch;
    float* d_psel_b = (float*)(e->d_scratch + pb);
    double* d_part_f = (double*)(e->d_scratch + 2 * pb);
    double* d_part_b = (double*)(e->d_scratch + 2 * pb + qb_bytes);
    const double* d_z = nullptr;
    if (z && z_mapped) { memcpy(e->h_io + ob + rb, z, zbytes); d_z = reinterpret_cast<const double*>(e->d_io + ob + rb); }
    else if (z) {
        double* dz = (double*)(e->d_scratch + 2 * pb + 2 * qb_bytes);
        int _urc = upload(e, dz, z, zbytes); if (_urc) return _urc;
        d_z = dz;
    }
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    auto post_args = [&](int slot) {
        return SrcPostArgs{e->d_state, e->d_gid + (int64_t)slot * C * Np, e->d_pid + (int64_t)slot * Np,
                           e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * F * C,
                           d_obj, n_sub, Np, F, e->S, C, e->Fp, inv_t, (float)inv_tp, inv_t != 1.0, inv_tp != 1.0,
                           from_prior != 0};
    };
    uint8_t* src_cand = e->d_src + (int64_t)cand_slot * N * e->Fp;
    bump_src(e, cand_slot);                  // (the Gibbs step draws into the candidate's array and copies the rest in full)
    bump_ids(e, cand_slot);
    // 1: the draw (posterior from the current tables) -> the candidate's source rows of the listed objects, and the
    //    per-block partial sums of log_q
    k_sample_source<<<nblk, kBlock, 0, e->stream>>>(post_args(cur_slot), d_z, e->rng_seed, e->rng_draw, src_cand, d_psel_f, e->d_status, d_part_f);
    if (!z) ++e->rng_draw;
    HIPCHK(e, hipGetLastError());
    // 2: the rest of the candidate slot, its count delta and every one of its tables
    Slot cd = cur;
    {
        CoreInputs in;
        in.row_of = reinterpret_cast<const int16_t*>(e->d_io + ob);
        in.src_new = src_cand;
        in.subset = d_obj; in.n_subset = n_sub;
        in.P = (int)cd.patterns.size();
        rc = launch_step_core(e, cur_slot, cand_slot, in);
        if (rc) return rc;
    }
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    e->slots[cand_slot] = cd;
    // 3: log_q_back -- the candidate's posterior evaluated at the CURRENT source assignment (+ its partial sums)
    k_source_logprob<<<nblk, kBlock, 0, e->stream>>>(post_args(cand_slot), e->d_src + (int64_t)cur_slot * N * e->Fp, d_psel_b, e->d_status, d_part_b);
    HIPCHK(e, hipGetLastError());
    // 4 + 5: fused mixture eval, reduction + epilogue (per-group collapsed values, flags, checks, log_q / log_q_back)
    StepFinish fin = make_step_finish(e);
    fin.lq_partials[0] = d_part_f; fin.lq_partials[1] = d_part_b; fin.lq_n[0] = fin.lq_n[1] = nblk;
    fin.lq_out = reinterpret_cast<double*>(e->d_step_host + step_host_lq_offset(e));
    DoneSig done;
    rc = launch_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, &fin, nullptr, nullptr,
                        nullptr, &done);
    if (rc) return rc;
    rc = wait_done(e, done);
    if (rc) return rc;
    rc = read_step_results(e, cand_slot, group_logliks_out, mixture_out, changed_groups_out, "posterior rows / table rows");
    if (rc) return rc;
    const double* lq = reinterpret_cast<const double*>(e->h_step + step_host_lq_offset(e));
    *log_q_out = lq[0];
    *log_q_back_out = lq[1];
    return SBE_OK;
}

static int step_general(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                        int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                        double* mixture_out, uint8_t* changed_groups_out) {
    Slot& cur = e->slots[cur_slot];
    const int saved_deferred = e->opt_deferred;
    e->opt_deferred = 1;                       // no intermediate synchronisation inside the step
    auto done = [&](int rc) { e->opt_deferred = saved_deferred; return rc; };
    int rc = sbe_copy_slot(e, cand_slot, cur_slot);
    if (rc) return done(rc);
    const int N = e->N;
    // objects whose counts may change: listed source rows + objects whose cluster membership changed
    std::vector<uint8_t> moved(N, 0);
    for (int i = 0; i < n_changed; ++i) moved[changed_objects[i]] = 1;
    if (clusters) {
        const int K = e->G[0];
        std::vector<uint16_t> ids(N, kNoGroup);
        char msg[320];
        if (!matrix_to_ids(clusters, K, N, 0, 0, ids.data(), msg, sizeof msg)) return done(fail(e, SBE_ERR_DATA, "%s", msg));   // component 0: offset 0
        for (int n = 0; n < N; ++n) if (ids[n] != cur.h_gid[n]) moved[n] = 1;
        rc = set_gid_common(e, cand_slot, 0, ids);
        if (rc) return done(rc);
    }
    if (n_changed > 0) {
        rc = sbe_set_source_rows(e, cand_slot, changed_objects, n_changed, source_rows);
        if (rc) return done(rc);
    }
    if (weights) rc = sbe_set_weights(e, cand_slot, weights);
    else if (e->slots[cand_slot].patterns_dirty) rc = upload_patterns_and_weights(e, cand_slot);
    if (rc) return done(rc);
    std::vector<int32_t> subset;
    for (int n = 0; n < N; ++n) if (moved[n]) subset.push_back(n);
    rc = sbe_update_counts(e, cand_slot, cur_slot, subset.data(), (int)subset.size(), nullptr);
    if (rc) return done(rc);
    // probability tables of every component in one launch (+ their tile-transposed copy)
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return done(rc);
    k_probs<int32_t><<<div_up((int64_t)e->Gtot * e->F, 256), 256, 0, e->stream>>>(
        e->d_counts + (int64_t)cand_slot * e->table_elems(), e->d_conc, nullptr,
        e->d_probs + (int64_t)cand_slot * e->table_elems(), 0, e->Gtot, e->F, e->S, 0.0, 0.0, 1, e->d_status);
    k_tile_probs<<<div_up((int64_t)e->Gtot * e->S * e->ft * e->n_ftiles, 256), 256, 0, e->stream>>>(
        e->d_probs + (int64_t)cand_slot * e->table_elems(), e->d_probs_t + (int64_t)cand_slot * e->probs_t_elems(),
        0, e->Gtot, e->Gtot, e->F, e->S, e->ft, e->n_ftiles);
    HIPCHK(e, hipGetLastError());
    std::fill(e->slots[cand_slot].probs_set.begin(), e->slots[cand_slot].probs_set.end(), 1);
    rc = check_after(e, ST_BAD_NORMALIZE);
    if (rc) return done(rc);
    // collapsed likelihood of every group (a7/a8) into a device buffer
    k_dcl<int32_t><<<div_up((int64_t)e->Gtot * e->F, 256), 256, 0, e->stream>>>(
        e->d_counts + (int64_t)cand_slot * e->table_elems(), e->d_conc, e->d_step_pf, 0, e->Gtot, e->F, e->S, 1);
    k_group_sum_f32<<<div_up((int64_t)e->Gtot * 8, 64), 64, 0, e->stream>>>(e->d_step_pf, e->d_step_pg, e->Gtot, e->F);
    HIPCHK(e, hipGetLastError());
    rc = enqueue_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS);
    if (rc) return done(rc);
    // one read-back, one synchronisation
    const size_t pg_bytes = (size_t)e->Gtot * sizeof(double);
    rc = ensure_pinned(e, pg_bytes + (size_t)e->Gtot);
    if (rc) return done(rc);
    HIPCHK(e, hipMemcpyAsync(e->h_pinned, e->d_step_pg, pg_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->h_pinned + pg_bytes, e->d_changed, (size_t)e->Gtot, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    memcpy(group_logliks_out, e->h_pinned, pg_bytes);
    if (changed_groups_out) memcpy(changed_groups_out, e->h_pinned + pg_bytes, (size_t)e->Gtot);
    *mixture_out = e->h_results[cand_slot];
    return done(synced(e));
}

// ---- self-test hook: the table-build log against the device library's log -------------------------------
// GibbsSampleSource._propose (operators.py:495-552) for the drop-in layer in ONE call: sbe_gibbs_step's device chain -- the
// draw into the candidate slot, the rest of the slot with its count delta and tables (k_step_core), the backward
// probabilities -- and then, instead of the likelihoods a resident chain wants, what the reference's sample bookkeeping
// wants: the drawn ids, both selected-probability arrays and the count rows that changed, all in the host-mapped block,
// one completion flag.  (As eight engine calls -- copy_slot, sample_source, update_counts, update_probs, source_logprob,
// get_source_rows, counts_delta -- the same work cost 190 us per proposal in the sampler replay, seven stream
// synchronisations among them.)
int sbe_gibbs_propose_supported(sbe_engine* e) {                       // 1: the CHAIN form fits (the tile form is tried first, per call)
    CHECK_ENGINE(e);
    return ((int64_t)e->Gtot * e->S * 28 <= 60 * 1024) ? 1 : 0;          // (the fused table kernel of the step core)
}

// `follow` (sbe_gibbs_propose_apply): when the proposal touches any group, the CURRENT slot takes it -- counts, the touched
// groups' tables, the drawn source rows -- inside the tile kernel (tables and ids behind its completion flag), or as a copy of
// the candidate slot behind the chain form.
static int gibbs_propose_impl(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                              double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                              float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out, int follow) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, src_new_out); CHECK_PTR(e, sel_out); CHECK_PTR(e, sel_back_out); CHECK_PTR(e, touched_out); CHECK_PTR(e, n_touched_out);
    CHECK_PTR(e, diff_rows_out); CHECK_PTR(e, z);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    if (n_sub < 1) return fail(e, SBE_ERR_ARG, "n_sub=%d (nothing to resample)", n_sub);
    CHECK_PTR(e, objects);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    for (int c = 0; c < e->C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c] || !cur.probs_set[c])
            return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration / probability tables of component %d not set", cur_slot, c);
    int rc = check_objects(e, objects, n_sub);
    if (rc) return rc;
    const int N = e->N, Np = e->Np, F = e->F, C = e->C, S = e->S;
    const int64_t n_obs = (int64_t)n_sub * F, fs = (int64_t)F * S;
    // the groups the subset's objects are in (their count rows are the only ones the redraw can change), ascending
    std::vector<uint8_t> seen((size_t)e->Gtot, 0);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < n_sub; ++i) {
            const uint16_t gg = cur.h_gid[(size_t)c * N + objects[i]];
            if (gg != kNoGroup) seen[gg] = 1;
        }
    int n_touched = 0;
    for (int g = 0; g < e->Gtot; ++g) if (seen[g]) touched_out[n_touched++] = g;
    *n_touched_out = n_touched;
    HIPCHK(e, hipSetDevice(e->device));
    if (cur.patterns_dirty) { rc = upload_patterns_and_weights(e, cur_slot); if (rc) return rc; }
    if (e->status_pending) {                  // deliver a deferred data check before this call reuses the words
        HIPCHK(e, hipStreamSynchronize(e->stream));
        rc = synced(e);
        if (rc) return rc;
    }
    // ---- tile form: the whole proposal in ONE kernel (k_gibbs_propose_tile), no candidate slot built ----
    {
        const size_t t_ob = al256((size_t)n_sub * sizeof(int32_t)), t_gb = al256((size_t)C * n_sub * sizeof(int32_t));
        const size_t t_tb = al256((size_t)std::max(n_touched, 1) * sizeof(int32_t));
        const size_t t_in = t_ob + t_gb + t_tb;
        const size_t t_lds = t_in + ((size_t)e->Gtot + (size_t)n_touched * 16 * S) * sizeof(int32_t) + (size_t)n_sub * 16;
        const size_t zbytes_t = (size_t)n_obs * sizeof(double);
        const bool z_map = zbytes_t <= ((size_t)1 << 19);
        const size_t t_zb = z_map ? al256(zbytes_t) : 0;
        const size_t t_idb = al256((size_t)n_obs), t_selb = al256((size_t)n_obs * sizeof(float));
        const size_t t_rowb = al256((size_t)std::max(n_touched, 1) * fs * sizeof(float));
        const size_t t_out = t_idb + 2 * t_selb + t_rowb;
        // (a block serves ALL listed objects for its 16 features: beyond ~128 objects the chain form's grid over every
        //  observation is the faster one -- 154 us against 64 us at 1000 objects, 13 us against 64 us at 30)
        const bool chain_possible = sbe_gibbs_propose_supported(e) == 1;
        if (e->opt_fuse_tables && t_lds <= kGuFusedLdsMax && t_out <= ((size_t)8 << 20) && (n_sub <= 128 || !chain_possible)) {
            rc = ensure_io(e, t_in + t_zb + t_out);
            if (rc) return rc;
            uint8_t* h = e->h_io;
            memcpy(h, objects, (size_t)n_sub * sizeof(int32_t));
            int32_t* gl = reinterpret_cast<int32_t*>(h + t_ob);
            for (int c = 0; c < C; ++c)
                for (int i = 0; i < n_sub; ++i) {
                    const uint16_t gg = cur.h_gid[(size_t)c * N + objects[i]];
                    gl[(size_t)c * n_sub + i] = gg == kNoGroup ? -1 : (int32_t)gg;
                }
            memcpy(h + t_ob + t_gb, touched_out, (size_t)n_touched * sizeof(int32_t));
            const double* d_zt;
            if (z_map) { memcpy(h + t_in, z, zbytes_t); d_zt = reinterpret_cast<const double*>(e->d_io + t_in); }
            else {
                rc = ensure_scratch(e, al256(zbytes_t));
                if (rc) return rc;
                int urc = upload(e, e->d_scratch, z, zbytes_t); if (urc) return urc;
                d_zt = reinterpret_cast<const double*>(e->d_scratch);
            }
            rc = clear_status_word(e, ST_BAD_NORMALIZE);
            if (rc) return rc;
            uint8_t* d_o = e->d_io + t_in + t_zb;
            GibbsTileArgs ta{};
            ta.state = e->d_state; ta.gid = e->d_gid + (int64_t)cur_slot * C * Np; ta.pid = e->d_pid + (int64_t)cur_slot * Np;
            ta.src = e->d_src + (int64_t)cur_slot * N * e->Fp; ta.probs = e->d_probs + (int64_t)cur_slot * e->table_elems();
            ta.wpat = e->d_wpat + (int64_t)cur_slot * e->Pmax * F * C; ta.counts = e->d_counts + (int64_t)cur_slot * e->table_elems();
            ta.conc = e->d_conc;
            ta.mapped_in = reinterpret_cast<const uint32_t*>(e->d_io); ta.in_words = (int)(t_in / 4);
            ta.objects_word = 0; ta.gid_word = (int)(t_ob / 4); ta.touched_word = (int)((t_ob + t_gb) / 4);
            ta.z = d_zt;
            ta.ids_out = d_o; ta.sel_out = (float*)(d_o + t_idb); ta.back_out = (float*)(d_o + t_idb + t_selb);
            ta.rows_out = (float*)(d_o + t_idb + 2 * t_selb);
            ta.n_sub = n_sub; ta.n_touched = n_touched; ta.Gtot = e->Gtot; ta.Np = Np; ta.F = F; ta.S = S; ta.C = C; ta.Fp = e->Fp;
            const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
            ta.inv_t = inv_t; ta.inv_tp = (float)inv_tp; ta.pow_lh = inv_t != 1.0; ta.pow_w = inv_tp != 1.0; ta.from_prior = from_prior != 0;
            ta.status = e->d_status;
            if (follow && n_touched > 0) {
                ta.follow.counts = e->d_counts + (int64_t)cur_slot * e->table_elems();
                ta.follow.probs = e->d_probs + (int64_t)cur_slot * e->table_elems();
                ta.follow.probs_t = e->d_probs_t + (int64_t)cur_slot * e->probs_t_elems();
                ta.follow.ft = e->ft;
                ta.follow.src = e->d_src + (int64_t)cur_slot * N * e->Fp;
            }
            const unsigned blocks = (unsigned)div_up(F, 16);
            const DoneSig done = next_done(e, blocks);
            k_gibbs_propose_tile<<<blocks, kTileBlock, t_lds, e->stream>>>(ta, done);
            HIPCHK(e, hipGetLastError());
            rc = sync_and_report(e, done);
            if (rc) return rc;
            const uint8_t* ho = h + t_in + t_zb;
            memcpy(src_new_out, ho, (size_t)n_obs);
            memcpy(sel_out, ho + t_idb, (size_t)n_obs * sizeof(float));
            memcpy(sel_back_out, ho + t_idb + t_selb, (size_t)n_obs * sizeof(float));
            memcpy(diff_rows_out, ho + t_idb + 2 * t_selb, (size_t)n_touched * fs * sizeof(float));
            return SBE_OK;
        }
    }
    // ---- chain form (tables beyond the tile kernel's LDS image, or SBE_OPT_FUSE_TABLES off): the candidate slot is built ----
    if (!sbe_gibbs_propose_supported(e))
        return fail(e, SBE_ERR_ARG, "sbe_gibbs_propose: tables too large for the fused table kernel (G_total=%d, S=%d)", e->Gtot, e->S);
    // host-mapped block: objects | row_of marks | touched | uniforms (when few) || ids | sel | sel_back | count rows
    const size_t ob = al256((size_t)n_sub * sizeof(int32_t)), rb = al256((size_t)Np * sizeof(int16_t));
    const size_t tb = al256((size_t)std::max(n_touched, 1) * sizeof(int32_t));
    const size_t zbytes = (size_t)n_obs * sizeof(double);
    const bool z_mapped = zbytes <= ((size_t)1 << 19);
    const size_t zb = z_mapped ? al256(zbytes) : 0;
    const size_t idb = al256((size_t)n_obs), selb = al256((size_t)n_obs * sizeof(float));
    const size_t rowb = al256((size_t)std::max(n_touched, 1) * fs * sizeof(float));
    const size_t in_bytes = ob + rb + tb + zb, out_bytes = idb + 2 * selb + rowb;
    if (out_bytes > ((size_t)8 << 20)) return fail(e, SBE_ERR_ARG, "sbe_gibbs_propose: %d objects x %d features exceed the mapped result block", n_sub, F);
    rc = ensure_io(e, in_bytes + out_bytes);
    if (rc) return rc;
    uint8_t* h = e->h_io;
    memcpy(h, objects, (size_t)n_sub * sizeof(int32_t));
    int16_t* row_of = reinterpret_cast<int16_t*>(h + ob);
    std::fill(row_of, row_of + Np, (int16_t)-1);
    for (int i = 0; i < n_sub; ++i) row_of[objects[i]] = 0;
    memcpy(h + ob + rb, touched_out, (size_t)n_touched * sizeof(int32_t));
    const int32_t* d_obj = reinterpret_cast<const int32_t*>(e->d_io);
    const int nblk = div_up(n_obs, kBlock);
    const size_t qb_bytes = al256((size_t)nblk * sizeof(double));
    rc = ensure_scratch(e, 2 * qb_bytes + (z_mapped ? 0 : al256(zbytes)));
    if (rc) return rc;
    double* d_part_f = (double*)e->d_scratch;
    double* d_part_b = (double*)(e->d_scratch + qb_bytes);
    const double* d_z;
    if (z_mapped) { memcpy(h + ob + rb + tb, z, zbytes); d_z = reinterpret_cast<const double*>(e->d_io + ob + rb + tb); }
    else {
        double* dz = (double*)(e->d_scratch + 2 * qb_bytes);
        int urc = upload(e, dz, z, zbytes); if (urc) return urc;
        d_z = dz;
    }
    uint8_t* d_ids = e->d_io + in_bytes;
    float* d_psel_f = (float*)(d_ids + idb);
    float* d_psel_b = (float*)(d_ids + idb + selb);
    float* d_rows = (float*)(d_ids + idb + 2 * selb);
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    auto post_args = [&](int slot) {
        return SrcPostArgs{e->d_state, e->d_gid + (int64_t)slot * C * Np, e->d_pid + (int64_t)slot * Np,
                           e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * F * C,
                           d_obj, n_sub, Np, F, S, C, e->Fp, inv_t, (float)inv_tp, inv_t != 1.0, inv_tp != 1.0,
                           from_prior != 0};
    };
    uint8_t* src_cand = e->d_src + (int64_t)cand_slot * N * e->Fp;
    bump_src(e, cand_slot);
    bump_ids(e, cand_slot);
    // 1: the draw (posterior from the current tables) -> the candidate's source rows of the listed objects; p[drawn] out
    k_sample_source<<<nblk, kBlock, 0, e->stream>>>(post_args(cur_slot), d_z, e->rng_seed, e->rng_draw, src_cand, d_psel_f, e->d_status, d_part_f);
    HIPCHK(e, hipGetLastError());
    // 2: the rest of the candidate slot, its count delta and every one of its tables
    Slot cd = cur;
    {
        CoreInputs in;
        in.row_of = reinterpret_cast<const int16_t*>(e->d_io + ob);
        in.src_new = src_cand;
        in.subset = d_obj; in.n_subset = n_sub;
        in.P = (int)cd.patterns.size();
        rc = launch_step_core(e, cur_slot, cand_slot, in);
        if (rc) return rc;
    }
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    e->slots[cand_slot] = cd;
    // 3: the candidate's posterior evaluated at the CURRENT source assignment; p_back[old source] out
    k_source_logprob<<<nblk, kBlock, 0, e->stream>>>(post_args(cand_slot), e->d_src + (int64_t)cur_slot * N * e->Fp, d_psel_b, e->d_status, d_part_b);
    HIPCHK(e, hipGetLastError());
    // 4: drawn ids and changed count rows, completion
    const int64_t n_el = n_obs + (int64_t)n_touched * fs;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n_el, 256), 256);
    const DoneSig done = next_done(e, blocks);
    k_gibbs_fetch<<<blocks, 256, 0, e->stream>>>(src_cand, d_obj, n_sub, d_ids, e->d_counts + (int64_t)cur_slot * e->table_elems(),
                                                e->d_counts + (int64_t)cand_slot * e->table_elems(),
                                                reinterpret_cast<const int32_t*>(e->d_io + ob + rb), n_touched, d_rows, F, S, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(src_new_out, h + in_bytes, (size_t)n_obs);
    memcpy(sel_out, h + in_bytes + idb, (size_t)n_obs * sizeof(float));
    memcpy(sel_back_out, h + in_bytes + idb + selb, (size_t)n_obs * sizeof(float));
    memcpy(diff_rows_out, h + in_bytes + idb + 2 * selb, (size_t)n_touched * fs * sizeof(float));
    if (follow && n_touched > 0) return sbe_copy_slot(e, cur_slot, cand_slot);       // (the candidate IS the proposal: one copy launch)
    return SBE_OK;
}

int sbe_gibbs_propose(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                      float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out) {
    return gibbs_propose_impl(e, cur_slot, cand_slot, objects, n_sub, temperature, prior_temperature, from_prior, z, src_new_out, sel_out,
                              sel_back_out, touched_out, n_touched_out, diff_rows_out, 0);
}

int sbe_gibbs_propose_apply(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                            double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                            float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out) {
    return gibbs_propose_impl(e, cur_slot, cand_slot, objects, n_sub, temperature, prior_temperature, from_prior, z, src_new_out, sel_out,
                              sel_back_out, touched_out, n_touched_out, diff_rows_out, 1);
}

int sbe_test_fast_log(sbe_engine* e, const double* in, int n, double* out_fast, double* out_lib) {
    CHECK_ENGINE(e); CHECK_PTR(e, in); CHECK_PTR(e, out_fast); CHECK_PTR(e, out_lib);
    if (n < 1) return fail(e, SBE_ERR_ARG, "n=%d", n);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t b = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 3 * b);
    if (rc) return rc;
    double* d_in = (double*)e->d_scratch;
    double* d_f = (double*)(e->d_scratch + b);
    double* d_l = (double*)(e->d_scratch + 2 * b);
    { int _urc = upload(e, d_in, in, (size_t)n * sizeof(double)); if (_urc) return _urc; }
    k_test_fast_log<<<div_up(n, 256), 256, 0, e->stream>>>(d_in, d_f, d_l, n);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out_fast, d_f, (size_t)n * sizeof(double));
    if (rc) return rc;
    return d2h(e, out_lib, d_l, (size_t)n * sizeof(double));
}

int sbe_test_lgamma(sbe_engine* e, const double* in, int n, double* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, in); CHECK_PTR(e, out);
    if (n < 1) return fail(e, SBE_ERR_ARG, "n=%d", n);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t b = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 2 * b);
    if (rc) return rc;
    double* d_in = (double*)e->d_scratch;
    double* d_o = (double*)(e->d_scratch + b);
    { int _urc = upload(e, d_in, in, (size_t)n * sizeof(double)); if (_urc) return _urc; }
    k_test_lgamma<<<div_up(n, 256), 256, 0, e->stream>>>(d_in, d_o, n);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_o, (size_t)n * sizeof(double));
}

#ifdef SBE_WS_CLOCK
// debug build only: the wall-clock stamps block 0 of k_given_unchanged_fused left (100 MHz ticks)
int sbe_debug_gu_clk(sbe_engine* e, unsigned long long* out16) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gu_clk), 16 * sizeof(unsigned long long)));
    return SBE_OK;
}
#endif

int sbe_test_roundtrip(sbe_engine* e, int n_blocks, int mode) {
    CHECK_ENGINE(e);
    if (n_blocks < 1 || n_blocks > 65535) return fail(e, SBE_ERR_ARG, "n_blocks=%d", n_blocks);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = ensure_io(e, 256 + (size_t)n_blocks * sizeof(double));
    if (rc) return rc;
    const DoneSig done = next_done(e, (unsigned)n_blocks);
    k_test_roundtrip<<<n_blocks, 64, 0, e->stream>>>((const int32_t*)e->d_io, (double*)(e->d_io + 256), mode, done);
    HIPCHK(e, hipGetLastError());
    return sync_and_report(e, done);
}

int sbe_test_tab_log(sbe_engine* e, const double* in, int n, double* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, in); CHECK_PTR(e, out);
    if (n < 1) return fail(e, SBE_ERR_ARG, "n=%d", n);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t b = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 2 * b);
    if (rc) return rc;
    double* d_in = (double*)e->d_scratch;
    double* d_out = (double*)(e->d_scratch + b);
    { int _urc = upload(e, d_in, in, (size_t)n * sizeof(double)); if (_urc) return _urc; }
    k_test_tab_log<<<div_up(n, 256), 256, 0, e->stream>>>(d_in, e->d_logtab, d_out, n);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_out, (size_t)n * sizeof(double));
}

// ---- slots ------------------------------------------------------------------------------------------
int sbe_copy_slot(sbe_engine* e, int dst, int src) {
    CHECK_ENGINE(e); CHECK_SLOT(e, dst); CHECK_SLOT(e, src);
    if (dst == src) return SBE_OK;
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t N = e->N, F = e->F, C = e->C, T = e->table_elems();
    // every per-slot array in ONE launch (twelve hipMemcpyAsync calls cost ~40 us of host time per step)
    CopySegs cs{};
    uint32_t run = 0;
    auto seg = [&](auto* ptr, int64_t elems) {
        const int64_t bytes = elems * (int64_t)sizeof(*ptr);
        cs.src[cs.n] = reinterpret_cast<const uint32_t*>(ptr + (int64_t)src * elems);
        cs.dst[cs.n] = reinterpret_cast<uint32_t*>(ptr + (int64_t)dst * elems);
        run += (uint32_t)(bytes / 4);
        cs.end[cs.n++] = run;
    };
    seg(e->d_gid, C * e->Np); seg(e->d_pid, (int64_t)e->Np); seg(e->d_src, N * e->Fp); seg(e->d_counts, T); seg(e->d_probs, T);
    seg(e->d_probs_t, e->probs_t_elems()); seg(e->d_wpat_t, e->wpat_t_elems());
    seg(e->d_tid, (int64_t)e->Np); seg(e->d_toff, (int64_t)e->Np); seg(e->d_tuple_g, (int64_t)kMaxTuples * kMaxComponents); seg(e->d_tuple_p, (int64_t)kMaxTuples);
    seg(e->d_weights, F * C); seg(e->d_wpat, (int64_t)e->Pmax * F * C); seg(e->d_patbits, (int64_t)e->Pmax);
    k_multi_copy<<<std::min<int64_t>(div_up(run, 256), 4 * e->compute_units), 256, 0, e->stream>>>(cs);
    HIPCHK(e, hipGetLastError());
    bump_src(e, dst);
    bump_ids(e, dst);
    e->slots[dst] = e->slots[src];
    return SBE_OK;
}

// ---- measurement -------------------------------------------------------------------------------------
int sbe_timer_start(sbe_engine* e) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    return SBE_OK;
}

int sbe_timer_stop(sbe_engine* e, float* elapsed_ms) {
    CHECK_ENGINE(e); CHECK_PTR(e, elapsed_ms);
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return SBE_OK;
}

int sbe_kernel_timing(sbe_engine* e, int enable, int* n_launches, float* main_kernel_avg_ms) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipSetDevice(e->device));
    if (enable == 1) { e->ev_timing = true; e->ev_used = 0; return SBE_OK; }     // start: forget earlier pairs
    if (enable == 2) { e->ev_timing = false; return SBE_OK; }                    // pause: keep the recorded pairs
    if (enable == 3) { e->ev_timing = true; return SBE_OK; }                     // resume
    if (enable != 0) return fail(e, SBE_ERR_ARG, "sbe_kernel_timing: enable=%d", enable);
    CHECK_PTR(e, n_launches); CHECK_PTR(e, main_kernel_avg_ms);
    e->ev_timing = false;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    double acc = 0.0;
    for (int it = 0; it < e->ev_used; ++it) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[2 * it], e->ev_pool[2 * it + 1]));
        acc += ms;
    }
    *n_launches = e->ev_used;
    *main_kernel_avg_ms = e->ev_used ? (float)(acc / e->ev_used) : 0.f;
    e->ev_used = 0;
    return synced(e);
}

const char* sbe_last_mixture_kernel(const sbe_engine* e) { return e ? e->last_kernel : "none"; }

int sbe_profile_mixture(sbe_engine* e, int first_slot, int n, int iters, float* total_ms, float* main_kernel_avg_ms) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot); CHECK_PTR(e, total_ms); CHECK_PTR(e, main_kernel_avg_ms);
    if (iters < 1 || iters > 100000) return fail(e, SBE_ERR_ARG, "iters=%d", iters);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range out of range");
    HIPCHK(e, hipSetDevice(e->device));
    // one event pair per launch around the dominant kernel only (the fused gather/log/reduce
    // kernel); the small fixed-order partial reduction that follows is outside the pair.
    while ((int)e->ev_pool.size() < 2 * iters) {
        hipEvent_t ev;
        HIPCHK(e, hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    const int mode = e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS;
    int rc = enqueue_mixture(e, first_slot, n, mode);     // resolves lazily-built state
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    for (int it = 0; it < iters; ++it) {
        rc = launch_mixture(e, first_slot, n, mode, e->ev_pool[2 * it], e->ev_pool[2 * it + 1]);
        if (rc) return rc;
    }
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(total_ms, e->ev0, e->ev1));
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[2 * it], e->ev_pool[2 * it + 1]));
        acc += ms;
    }
    *main_kernel_avg_ms = (float)(acc / iters);
    return SBE_OK;
}

}  // extern "C"
