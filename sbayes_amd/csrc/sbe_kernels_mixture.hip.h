// sbe_kernels_mixture.hip.h -- the fused mixture log-likelihood kernels (SURVEY.md 8(d) "one eval").
// Included by sbe_mixture.hip only.
#pragma once
#include "sbe_mixture.hip.h"

namespace sbe {

// ==========================================================================================
// Fused mixture log-likelihood, v2: the GENERAL packed form (single evals, many group tuples, large tables).
//
// Mapping (chosen from rocprof/HIP-event measurements of v1):
//   lane   <-> feature inside a tile of FT features (FT = 64: one wave spans the tile)
//   dword  <-> 4 consecutive objects of that feature: the state block is kept in an
//              object-quad-interleaved layout state_q[N/4][Fq] (uint32), so one coalesced
//              256-byte wave load brings 4 objects x 64 features
//   LDS    <-> tab[(g*S + x)*FT + fl]: state-major tile image => the gather's bank is fl mod 32
//              for every x and g (conflict-free); image is a straight float4 copy of the
//              tile-transposed global table probs_t[tile][Gtot+1][S][FT] (row Gtot = zeros:
//              "object in no group" needs no branch).  Weights sit in LDS as float64
//              wl[(p*C + c)*FT + fl] (conflict-free ds_read_b64, no conversion per use).
//   ids    <-> FT = 64: the object quad is wave-uniform, group / pattern ids are 8-byte /
//              4-byte scalar loads; FT < 64: 64/FT quads per wave, per-lane broadcast loads.
// ==========================================================================================

// End of a block of the Mix2Params kernels, called by the 64 lanes of the block's FIRST wave with the block's partial sum in
// lane 0.  Without p.results the sum goes to partials[slot][work] and k_reduce_partials adds them up (the step forms: their
// epilogue block rides on that launch).  With p.results the kernel finishes by itself -- one launch per eval instead of two,
// 3 us of a 14 us host-synchronous eval: the blocks of a slot take tickets (arrive[slot], left at 0), the LAST one adds the
// slot's n_work partial sums in a fixed order (k_reduce_partials' own: the same bits whichever block is last) and
// writes the result; p.done (optional) is signalled by those blocks, one per slot.  The partial sums travel as agent-scope
// atomic stores / loads ordered by s_waitcnt around the ticket: coherent across the XCDs' L2s without a release fence (which
// writes the whole L2 back -- measured on the matrix-pipe form, profiles/r5/mfma_kernel_experiments_session2.log).
// MEMORY-MODEL NOTE: every access below is "relaxed"; there is no release / acquire pair of the HIP / LLVM memory
// model here.  The order store -> s_waitcnt vmcnt(0) -> ticket RMW -> (last block) loads holds in hardware on gfx942 / gfx950:
// agent-scope atomic stores are sc1 write-through and acknowledged at the coherence point the XCDs share, agent-scope atomic
// loads bypass the reader's own L2.  Other targets are refused at compile time; SBE_REDUCE_IN_KERNEL=0 selects the two-launch
// form with no such assumption (kept exercised by tests/test_gpu_shapes.py).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "finish_partial orders relaxed agent-scope atomics by s_waitcnt (gfx942 / gfx950 cache behaviour); not valid for this target"
#endif
__device__ __forceinline__ void finish_partial(const Mix2Params& p, int slot, int work, double total) {
    const int lane = threadIdx.x & (kWave - 1);
    double* const my = p.partials + (int64_t)slot * p.partials_stride;
    if (!p.results) {
        if (lane == 0) my[work] = total;
        return;
    }
    if (p.n_work > 1) {
        unsigned t = 0;
        if (lane == 0) {
            __hip_atomic_store(my + work, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t = __hip_atomic_fetch_add(p.arrive + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned)p.n_work - 1u) __hip_atomic_store(p.arrive + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t != (unsigned)p.n_work - 1u) return;                         // (wave-uniform)
        asm volatile("" ::: "memory");
        // the additions of k_reduce_partials, in its order (256 threads stride the partial sums, wave trees, then
        // (w0 + w1) + (w2 + w3)): lane l stands for its threads l, l + 64, l + 128, l + 192 -- the step forms, which keep
        // that kernel for their epilogue, and this path return the same bits
        double ws[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = 0.0;
            for (int i = lane + k * kWave; i < p.n_work; i += 4 * kWave) v += __hip_atomic_load(my + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ws[k] = wave_sum(v);
        }
        total = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    }
    if (lane == 0) p.results[slot] = total;
    signal_done(p.done);
}

// LOG_PRODUCT, branch-free form used by the v2 kernel: the four observation likelihoods of a
// step are multiplied into the running mantissa and the binary exponent is stripped once per
// step.  Anything that is not a positive normal double (an observed state with probability 0,
// NaN / inf tables, or an underflowing product of pathologically small factors) leaves exponent
// bits 0 or 0x7FF or a set sign bit in the product; that raises `bad`, and the thread then
// recomputes its whole contribution with the per-observation log (exactly NumPy's path), so the
// fast path never has to be right for such inputs.
struct ProdAcc {
    double mant;     // in [1, 2) while good
    int expo;        // sum of stripped biased exponents
    int steps;       // number of strips (bias correction 1023 * steps)
    uint32_t bad;    // sticky
};

__device__ __forceinline__ void prod_add4(ProdAcc& a, double v0, double v1, double v2, double v3) {
    const double m = a.mant * ((v0 * v1) * (v2 * v3));
    const uint32_t hi = (uint32_t)(__double_as_longlong(m) >> 32);
    const uint32_t ex = (hi >> 20) & 0xFFFu;                 // sign + exponent field
    // good: positive, finite, and >= 2^-1019 so that (factors <= 2) neither the 4-product nor
    // its two pair products were subnormal (no silent precision loss)
    a.bad |= (ex - 4u) >= 0x7FBu;
    a.expo += (int)ex;
    const uint32_t hi2 = (hi & 0x000FFFFFu) | 0x3FF00000u;
    a.mant = __hiloint2double((int)hi2, __double2loint(m));
}

// Shared prologue of the v2 kernels: LDS image of the tile (tables, weights) and of the chunk's ids.
struct V2Lds {
    float* tab;          // [(Gtot+1)][S][FT]
    double* wl;          // [P][C][FT]
    uint16_t* ids_g;     // [C][4*quads_per_chunk]   (one u16 per object)
    uint8_t* ids_p;      // [4*quads_per_chunk]
};

// DIRECT = the tables of a tile do not fit LDS (very many groups x states): tab / wl then point at the
// tile-transposed GLOBAL tables (served by L2) and only the ids are staged.
template <int FT, bool DIRECT>
__device__ __forceinline__ V2Lds v2_stage(const Mix2Params& p, unsigned char* lds_raw, int slot, int tile, int C,
                                          int q0, int nq) {
    V2Lds L;
    const int tab_elems = (p.Gtot + 1) * p.S * FT;            // multiple of 16
    const float* g_tab = p.probs_t + (int64_t)slot * p.probs_t_stride + (int64_t)tile * tab_elems;
    const double* g_wl = p.wpat_t + (int64_t)slot * p.wpat_t_stride + (int64_t)tile * p.wpat_tile_stride;
    uint64_t* ids_g64;
    if (DIRECT) {
        L.tab = const_cast<float*>(g_tab);
        L.wl = const_cast<double*>(g_wl);
        ids_g64 = reinterpret_cast<uint64_t*>(lds_raw);
    } else {
        L.tab = reinterpret_cast<float*>(lds_raw);
        L.wl = reinterpret_cast<double*>(lds_raw + (size_t)tab_elems * sizeof(float));
        ids_g64 = reinterpret_cast<uint64_t*>(L.wl + (size_t)p.P * C * FT);            // [C][quads_per_chunk]
    }
    uint32_t* ids_p32 = reinterpret_cast<uint32_t*>(ids_g64 + (size_t)C * p.quads_per_chunk);
    L.ids_g = reinterpret_cast<uint16_t*>(ids_g64);
    L.ids_p = reinterpret_cast<uint8_t*>(ids_p32);
    if (!DIRECT) {
        // tile image: contiguous float4 copy, 4 loads in flight per thread
        const float4* src = reinterpret_cast<const float4*>(g_tab);
        float4* dst = reinterpret_cast<float4*>(L.tab);
        const int n4 = tab_elems >> 2;
        int i = threadIdx.x;
        for (; i + 3 * kBlock < n4; i += 4 * kBlock) {
            const float4 a = src[i], b = src[i + kBlock], c = src[i + 2 * kBlock], d = src[i + 3 * kBlock];
            dst[i] = a; dst[i + kBlock] = b; dst[i + 2 * kBlock] = c; dst[i + 3 * kBlock] = d;
        }
        for (; i < n4; i += kBlock) dst[i] = src[i];
        const double2* wsrc = reinterpret_cast<const double2*>(g_wl);
        double2* wdst = reinterpret_cast<double2*>(L.wl);
        const int w2 = (p.P * C * FT) >> 1;
        for (int k = threadIdx.x; k < w2; k += kBlock) wdst[k] = wsrc[k];
    }
    // the chunk's ids: group ids (4 x u16 per quad and component) and pattern ids (LDS reads are
    // in-order on lgkmcnt and ~64 cycles; scalar loads would serialise behind every LDS wait)
    const uint16_t* gid = p.gid + (int64_t)slot * p.gid_stride;
    const uint8_t* pid = p.pid + (int64_t)slot * p.pid_stride;
    for (int k = threadIdx.x; k < nq * C; k += kBlock) {
        const int c = k / nq, qi = k - c * nq;
        ids_g64[c * p.quads_per_chunk + qi] = *reinterpret_cast<const uint64_t*>(gid + (int64_t)c * p.Np + 4 * (q0 + qi));
    }
    for (int k = threadIdx.x; k < nq; k += kBlock)
        ids_p32[k] = *reinterpret_cast<const uint32_t*>(pid + 4 * (q0 + k));
    __syncthreads();
    return L;
}

template <int MODE, int FT, int CT, bool DIRECT>   // CT: compile-time component count (1..4), 0 = runtime (<= 8)
__global__ __launch_bounds__(kBlock) void k_mixture_v2(Mix2Params p) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red4[4];
    // XCD-aware 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch; b % 8 labels the
    // group, speed only).  The slot-blocks of one (tile, chunk) work item get adjacent positions on
    // one XCD: they run at about the same time and share the streamed feature bytes in its L2.
    // Units = (work item, slot group); unit u lives on XCD u % 8 (consecutive work items are
    // interleaved over the XCDs -- neighbouring tiles of the same rows stream together, which the
    // measured stress-shape runs prefer -- and small grids are balanced by splitting the slots of a
    // work item into `slot_groups` groups); the slots of a unit are adjacent on that XCD.
    const int unit = ((int)(blockIdx.x >> 3) / p.slots_per_group) * 8 + (int)(blockIdx.x & 7);
    const int slot_i = (unit % p.slot_groups) * p.slots_per_group + (int)(blockIdx.x >> 3) % p.slots_per_group;
    const int work = unit / p.slot_groups;                 // (tile, chunk) index
    if (work >= p.n_work || slot_i >= p.n_batch) return;   // padding blocks (before any barrier)
    const int slot = p.slot_list ? p.slot_list[slot_i] : p.first_slot + slot_i;
    const int tile = work % p.n_ftiles, chunk = work / p.n_ftiles;
    const int S = p.S;
    const int C = CT ? CT : p.C;
    constexpr int CU = CT ? CT : kMaxComponents;
    const int q0 = chunk * p.quads_per_chunk;
    const int q1 = min(p.NQ, q0 + p.quads_per_chunk);
    const int nq = q1 - q0;                                 // >= 1
    const V2Lds L = v2_stage<FT, DIRECT>(p, lds_raw, slot, tile, C, q0, nq);
    const uint64_t* ids_g = reinterpret_cast<const uint64_t*>(L.ids_g);
    const uint32_t* ids_p = reinterpret_cast<const uint32_t*>(L.ids_p);

    constexpr int ROWS = kWave / FT;                        // object quads per wave step
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    const int fl = lane % FT, sub = lane / FT;
    const int f = tile * FT + fl;
    const float* tab_l = L.tab + fl;
    const double* wl_l = L.wl + fl;
    const uint32_t gmax = (uint32_t)p.Gtot;
    const uint32_t row = (uint32_t)S * FT;                  // floats per group row

    const int n_steps = (nq + 4 * ROWS - 1) / (4 * ROWS);   // same for every wave: uniform loop
    const uint32_t* sq = p.state_q + (int64_t)q0 * p.Fq + f;

    auto local_quad = [&](int k) { return (k * 4 + wid) * ROWS + sub; };
    auto load_state = [&](int k) -> uint32_t {
        const int i = local_quad(k);
        const uint32_t xs = sq[(int64_t)min(i, nq - 1) * p.Fq];        // always in bounds
        return i < nq ? xs : 0xFFFFFFFFu;                             // past the chunk: 4 x NA
    };
    // the four observation likelihoods of step k (NA -> exactly 1.0)
    auto step_values = [&](int k, uint32_t xs, double (&v)[4]) {
        const int qi = min(local_quad(k), nq - 1);
        uint64_t gq[CU];
        uint32_t pq = ids_p[qi];
#pragma unroll
        for (int c = 0; c < CU; ++c) gq[c] = (CT || c < C) ? ids_g[c * p.quads_per_chunk + qi] : ~0ull;
        if (FT == kWave) {                                   // wave-uniform quad: ids to SGPRs
            pq = __builtin_amdgcn_readfirstlane(pq);
#pragma unroll
            for (int c = 0; c < CU; ++c) {
                if (CT || c < C) {
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)gq[c]);
                    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(gq[c] >> 32));
                    gq[c] = ((uint64_t)hi << 32) | lo;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = (xs >> (8 * j)) & 0xFFu;
            const bool valid = x < (uint32_t)S;                     // NA byte: S (data) or 0xFF (padding)
            const uint32_t xc = valid ? x : 0u;
            const uint32_t pj = (pq >> (8 * j)) & 0xFFu;
            const float* tj = tab_l + xc * FT;
            const double* wj = wl_l + __umul24(pj, (uint32_t)(C * FT));
            double vj = 0.0;
#pragma unroll
            for (int c = 0; c < CU; ++c) {
                if (CT || c < C) {
                    uint32_t g = (uint32_t)(gq[c] >> (16 * j)) & 0xFFFFu;
                    g = g < gmax ? g : gmax;                        // no group -> zero row
                    const double t = wj[c * FT] * (double)tj[__umul24(g, row)];   // (24-bit multiply: full rate; g < 2^16, row < 2^24)
                    vj = c == 0 ? t : vj + t;                       // NumPy order, no FMA
                }
            }
            v[j] = valid ? vj : 1.0;                                // NA: log 1 = 0
        }
    };

    double thread_ll;
    if (MODE == LOG_PRODUCT) {
        ProdAcc pa{1.0, 0, 0, 0u};
        uint32_t xs_next = load_state(0);
        for (int k = 0; k < n_steps; ++k) {
            const uint32_t xs = xs_next;
            xs_next = load_state(k + 1);                     // in flight while this step computes
            double v[4];
            step_values(k, xs, v);
            prod_add4(pa, v[0], v[1], v[2], v[3]);
        }
        pa.steps = n_steps;
        thread_ll = log(pa.mant) + (double)(pa.expo - 1023 * pa.steps) * 0.693147180559945309417232;
        if (__builtin_expect(pa.bad != 0u, 0)) {             // rare: redo this thread per observation
            double sum = 0.0;
            for (int k = 0; k < n_steps; ++k) {
                double v[4];
                step_values(k, load_state(k), v);
                sum += log(v[0]); sum += log(v[1]); sum += log(v[2]); sum += log(v[3]);
            }
            thread_ll = sum;
        }
    } else {
        double sum = 0.0;
        uint32_t xs_next = load_state(0);
        for (int k = 0; k < n_steps; ++k) {
            const uint32_t xs = xs_next;
            xs_next = load_state(k + 1);
            double v[4];
            step_values(k, xs, v);
            sum += log(v[0]); sum += log(v[1]); sum += log(v[2]); sum += log(v[3]);
        }
        thread_ll = sum;
    }
    const double total = block_sum(thread_ll, red4);
    if (threadIdx.x < kWave) finish_partial(p, slot, work, total);
}

// ------------------------------------------------------------------------------------------
// One-hot variant of the v2 kernel: streams the one-hot block exactly as the reference hands it
// over ([N][F][S] bool, N*F*S bytes per eval -- the contract figure of SURVEY.md 8(d)) with
// coalesced 16-byte lane loads, and shares the LDS image, the id staging and the log
// accumulation with the packed kernel.
//   item  <-> (object, 16-byte chunk of the object's tile row segment of FT*S bytes)
//   flags <-> bytes are 0/1, so m = d.x | d.y<<1 | d.z<<2 | d.w<<3 holds the 16 flags at distinct
//             bit positions (bit 8k+i = byte k of dword i); set bytes are walked with ffbl / m&(m-1)
//   (fl,x) <-> byte offset j in the row segment: fl = floor((j + 0.5)/S) in f32 (exact for
//             j < 2^16, S <= 254), x = j - fl*S
// ------------------------------------------------------------------------------------------
template <int MODE, int FT, int CT, bool DIRECT>
__global__ __launch_bounds__(kBlock) void k_mixture_onehot_v2(Mix2Params p) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red4[4];
    // XCD-aware 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch; b % 8 labels the
    // group, speed only).  The slot-blocks of one (tile, chunk) work item get adjacent positions on
    // one XCD: they run at about the same time and share the streamed feature bytes in its L2.
    // Units = (work item, slot group); unit u lives on XCD u % 8 (consecutive work items are
    // interleaved over the XCDs -- neighbouring tiles of the same rows stream together, which the
    // measured stress-shape runs prefer -- and small grids are balanced by splitting the slots of a
    // work item into `slot_groups` groups); the slots of a unit are adjacent on that XCD.
    const int unit = ((int)(blockIdx.x >> 3) / p.slots_per_group) * 8 + (int)(blockIdx.x & 7);
    const int slot_i = (unit % p.slot_groups) * p.slots_per_group + (int)(blockIdx.x >> 3) % p.slots_per_group;
    const int work = unit / p.slot_groups;                 // (tile, chunk) index
    if (work >= p.n_work || slot_i >= p.n_batch) return;   // padding blocks (before any barrier)
    const int slot = p.slot_list ? p.slot_list[slot_i] : p.first_slot + slot_i;
    const int tile = work % p.n_ftiles, chunk = work / p.n_ftiles;
    const int S = p.S;
    const int C = CT ? CT : p.C;
    constexpr int CU = CT ? CT : kMaxComponents;
    const int q0 = chunk * p.quads_per_chunk;
    const int q1 = min(p.NQ, q0 + p.quads_per_chunk);
    const int nq = q1 - q0;
    const V2Lds L = v2_stage<FT, DIRECT>(p, lds_raw, slot, tile, C, q0, nq);

    const int n0 = 4 * q0;
    const int n_obj = min(4 * nq, p.N - n0);
    const int seg_off = tile * FT * S;                             // byte offset of the tile in a row (mult. of 16)
    const int seg16 = min(FT * S, p.rs_pitch - seg_off) >> 4;      // 16-byte chunks of the segment inside the row
    const int n_items = n_obj * seg16;
    const int n_steps = (n_items + kBlock - 1) / kBlock;
    const uint32_t gmax = (uint32_t)p.Gtot;
    const uint32_t row = (uint32_t)S * FT;
    const float inv_s = 1.0f / (float)S, half_inv_s = 0.5f / (float)S;
    const float inv_seg = 1.0f / (float)seg16, half_inv_seg = 0.5f / (float)seg16;
    const int ids_pitch = 4 * p.quads_per_chunk;
    const uint8_t* oh = p.onehot + (int64_t)n0 * p.rs_pitch + seg_off;

    struct Item { uint4 d; int nl, ch; };
    auto fetch = [&](int k) -> Item {
        Item it;
        const int i = k * kBlock + (int)threadIdx.x;
        const int ic = min(i, n_items - 1);
        it.nl = (int)(((float)ic + 0.5f) * inv_seg);               // ic / seg16, exact (see header)
        it.ch = ic - it.nl * seg16;
        it.d = *reinterpret_cast<const uint4*>(oh + (int64_t)it.nl * p.rs_pitch + it.ch * 16);
        if (i >= n_items) it.d = make_uint4(0u, 0u, 0u, 0u);
        return it;
    };
    (void)half_inv_seg;

    // likelihood of the observation whose set byte is the lowest flag of m (1.0 if m == 0)
    auto observe = [&](uint32_t m, const Item& it, const uint32_t (&g)[CU], uint32_t pidn) -> double {
        const bool has = m != 0u;
        const int b = has ? __builtin_ctz(m) : 0;
        const int j = it.ch * 16 + ((b & 7) << 2) + (b >> 3);       // byte offset in the row segment
        const int fl = (int)((float)j * inv_s + half_inv_s);        // j / S
        const int x = j - fl * S;
        const float* tj = L.tab + x * FT + fl;
        const double* wj = L.wl + __umul24(pidn, (uint32_t)(C * FT)) + fl;
        double vj = 0.0;
#pragma unroll
        for (int c = 0; c < CU; ++c) {
            if (CT || c < C) {
                const double t = wj[c * FT] * (double)tj[__umul24(g[c], row)];
                vj = c == 0 ? t : vj + t;
            }
        }
        return has ? vj : 1.0;
    };
    auto item_ids = [&](const Item& it, uint32_t (&g)[CU], uint32_t& pidn) {
        pidn = L.ids_p[it.nl];
#pragma unroll
        for (int c = 0; c < CU; ++c) {
            if (CT || c < C) {
                const uint32_t gg = L.ids_g[c * ids_pitch + it.nl];
                g[c] = gg < gmax ? gg : gmax;
            } else g[c] = gmax;
        }
    };

    double thread_ll;
    if (MODE == LOG_PRODUCT) {
        ProdAcc pa{1.0, 0, 0, 0u};
        int n_strips = 0;
        Item nxt = fetch(0);
        for (int k = 0; k < n_steps; ++k) {
            const Item it = nxt;
            nxt = fetch(k + 1);
            uint32_t m = it.d.x | (it.d.y << 1) | (it.d.z << 2) | (it.d.w << 3);
            uint32_t g[CU], pidn;
            item_ids(it, g, pidn);
            // walk the set bytes; the trip count is wave-uniform (max over lanes), idle lanes multiply by 1
            while (__builtin_amdgcn_ballot_w64(m != 0u)) {
                const double a = observe(m, it, g, pidn);
                m &= m - 1u;
                const double b = observe(m, it, g, pidn);
                m &= m - 1u;
                prod_add4(pa, a, b, 1.0, 1.0);
                ++n_strips;
            }
        }
        thread_ll = log(pa.mant) + (double)(pa.expo - 1023 * n_strips) * 0.693147180559945309417232;
        if (__builtin_expect(pa.bad != 0u, 0)) {
            double sum = 0.0;
            for (int k = 0; k < n_steps; ++k) {
                const Item it = fetch(k);
                uint32_t m = it.d.x | (it.d.y << 1) | (it.d.z << 2) | (it.d.w << 3);
                uint32_t g[CU], pidn;
                item_ids(it, g, pidn);
                while (m) { sum += log(observe(m, it, g, pidn)); m &= m - 1u; }
            }
            thread_ll = sum;
        }
    } else {
        double sum = 0.0;
        Item nxt = fetch(0);
        for (int k = 0; k < n_steps; ++k) {
            const Item it = nxt;
            nxt = fetch(k + 1);
            uint32_t m = it.d.x | (it.d.y << 1) | (it.d.z << 2) | (it.d.w << 3);
            uint32_t g[CU], pidn;
            item_ids(it, g, pidn);
            while (m) { sum += log(observe(m, it, g, pidn)); m &= m - 1u; }
        }
        thread_ll = sum;
    }
    const double total = block_sum(thread_ll, red4);
    if (threadIdx.x < kWave) finish_partial(p, slot, work, total);
}

// ==========================================================================================
// Fused mixture log-likelihood, "rows" form: the GENERAL packed kernel for samples with many distinct group tuples
// (stress shape: 10 clusters x 2 confounders of 20 groups => thousands of tuples, no group-tuple table) and for
// every launch the tuple kernels do not take.  Same value per observation as the reference, same order:
//        v(n, f) = ((w0*p0 + w1*p1) + w2*p2) + ...   (fp64; w*p of two float32 is exact in fp64, so the fma chain below
//                                                      rounds exactly where NumPy's mul-then-add does)
// What differs from k_mixture_v2 is the machine mapping:
//   block   1024 threads = 16 waves sharing ONE LDS image of a 32-feature tile (16 features when 32 do not fit):
//           at the stress shape the image (54 group rows x 21 state rows x 32 features x 4 B = 145 KB + weights)
//           fills the CU's LDS once instead of twice per 16 features, and 16 waves hide the gather latency
//   lane    <-> feature of the tile; a half-wave (FT = 32) is one object quad: its 32 lanes read 32 consecutive
//           banks of one table row -- conflict-free for any row (k_mixture_v2 at FT = 16 put four objects in a
//           wave: two rows per half-wave, a 2-way bank conflict whenever their parities agree)
//   tables  f32 [(Gtot+1)][S+1][FT]: row S of every group is the NA row (zeros), row block Gtot is "no group" (zeros)
//   weights f64 planes [P][ceil(C/2)][FT][2]: one ds_read_b128 brings two components' weights, conflict-free
//   offsets the LDS byte offset of every object's group row per component and of its weight pattern come precomputed
//           (k_rowoff, [NQ][C+1][4] u32 per slot): one add per (observation, component), no id decode, no multiply.
//           They are per OBJECT, i.e. the same for the 32 lanes of a half-wave: loading them as five 16-byte lane
//           loads per step made the kernel texture-addresser bound (a wave-wide dwordx4 load occupies the TA for 64
//           lanes however few distinct addresses it carries: measured, removing every LDS gather changed nothing).
//           So a wave fetches the 2 x (C+1) x 4 dwords of its step with ONE coalesced dword load (40 lanes), parks
//           them in a private LDS slot and every lane reads its quad's five 16-byte rows back (LDS broadcast reads)
//   NA      the accumulator starts at 1.0 for an NA observation and its table row is zero: v = 1 exactly, log 1 = 0,
//           no select at the end
// ==========================================================================================


// SORTED (round 5, stress shape): the slot's objects come in an order in which every wave step (SUBS quads = 8 objects at
// FT = 32) holds ONE has_components pattern (k_rowsort: counting sort by pattern id, runs padded to whole steps with null
// objects), so the pattern's weights live in registers -- no per-observation weight read, a third of the loop's LDS traffic --
// and are re-read only where a run ends.  The price is the state stream: it can no longer be the shared quad-interleaved
// dword (four CONSECUTIVE objects), each observation's state byte is gathered from the object-major state block through the
// permutation: the offsets stream carries, instead of the pattern's weight offset, (pattern << 24 | byte offset of the
// object's state row), parked two steps ahead so that the four byte loads of a step are in flight for two steps.
template <int MODE, int FT, int CT, bool SORTED = false>
__global__ __launch_bounds__(kRowsBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mixture_rows(Mix2Params p) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red[kRowsWaves];
    // XCD-aware 1-D grid, same scheme as k_mixture_v2 (b % 8 labels the XCD group; speed only)
    const int unit = ((int)(blockIdx.x >> 3) / p.slots_per_group) * 8 + (int)(blockIdx.x & 7);
    const int slot_i = (unit % p.slot_groups) * p.slots_per_group + (int)(blockIdx.x >> 3) % p.slots_per_group;
    const int work = unit / p.slot_groups;                 // (tile, chunk) index
    if (work >= p.n_work || slot_i >= p.n_batch) return;   // padding blocks (before any barrier)
    const int slot = p.slot_list ? p.slot_list[slot_i] : p.first_slot + slot_i;
    const int tile = work % p.n_ftiles, chunk = work / p.n_ftiles;       // (n_ftiles: tiles of FT features here)
    constexpr int C = CT, CP = (CT + 1) / 2;
    const int S = p.S, S1 = p.S + 1;
    const uint32_t state_bytes = FT * 4;                                  // one state row of a group block
    const uint32_t row_bytes = (uint32_t)S1 * state_bytes;
    const uint32_t tab_bytes = (uint32_t)(p.Gtot + 1) * row_bytes;
    const int NQ_slot = SORTED ? p.rs_nq[slot] : p.NQ;                    // (sorted: the padded length of THIS slot's order)
    const int q0 = chunk * p.quads_per_chunk;
    const int q1 = min(NQ_slot, q0 + p.quads_per_chunk);
    const int nq = max(0, q1 - q0);                                       // (sorted: a chunk beyond the slot's length is empty)
    if (SORTED && nq == 0) {                                              // (block-uniform, before any barrier: nothing to stage for)
        if (threadIdx.x < kWave) finish_partial(p, slot, work, 0.0);
        return;
    }

    // ---- LDS image of the tile ---------------------------------------------------------------------------------
    // Tables: 16-byte pieces straight from the engine's tile-transposed copy probs_t[tile_e][g][s][eft] (a 32-feature
    // tile is half a 64-wide engine tile, one 32-wide one or two 16-wide ones).  A wave owns the groups w, w+16, ..;
    // four groups' loads are in flight together per lane (no division: a group's [S][FT] block is walked by piece).
    {
        const int lane_s = threadIdx.x & (kWave - 1), wave_s = threadIdx.x >> 6;
        constexpr int PPR = FT / 4;                                       // 16-byte pieces per state row
        const float* probs_t = p.probs_t + (int64_t)slot * p.probs_t_stride;
        const int n_tiles_e = (p.F + p.eft - 1) / p.eft;
        const int per_g = S * PPR;
        constexpr int GU = 4, JU = 3;                                     // 12 sixteen-byte loads in flight per lane
        for (int gb = wave_s; gb <= p.Gtot; gb += GU * kRowsWaves) {
            for (int jb = lane_s; jb < per_g; jb += JU * kWave) {
                uint4 v[GU][JU];
#pragma unroll
                for (int ju = 0; ju < JU; ++ju) {
                    const int j = jb + ju * kWave;
                    const int srow = j / PPR, part = j % PPR;             // (compile-time divisor)
                    const int f0 = tile * FT + part * 4;
                    const int te = f0 / p.eft, fle = f0 % p.eft;          // eft is 16 / 32 / 64: shifts
#pragma unroll
                    for (int u = 0; u < GU; ++u) {
                        const int g = gb + u * kRowsWaves;
                        v[u][ju] = (j < per_g && g <= p.Gtot && te < n_tiles_e)
                            ? *reinterpret_cast<const uint4*>(probs_t + (((int64_t)te * (p.Gtot + 1) + g) * S + srow) * p.eft + fle)
                            : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int ju = 0; ju < JU; ++ju) {
                    const int j = jb + ju * kWave;
                    const int srow = j / PPR, part = j % PPR;
#pragma unroll
                    for (int u = 0; u < GU; ++u) {
                        const int g = gb + u * kRowsWaves;
                        if (j < per_g && g <= p.Gtot)
                            *reinterpret_cast<uint4*>(lds_raw + (uint32_t)(g * S1 + srow) * state_bytes + (uint32_t)part * 16u) = v[u][ju];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {                                // NA row of each of the four groups: zeros
                const int g = gb + u * kRowsWaves;
                if (g <= p.Gtot && lane_s < PPR)
                    *reinterpret_cast<uint4*>(lds_raw + (uint32_t)(g * S1 + S) * state_bytes + (uint32_t)lane_s * 16u) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        double* wl = reinterpret_cast<double*>(lds_raw + tab_bytes);      // [P][CP][FT][2]
        const float* wpat = p.wpat + (int64_t)slot * p.wpat_stride;
        for (int k = threadIdx.x; k < p.P * CP * FT * 2; k += kRowsBlock) {
            const int e = k & 1, fl2 = (k >> 1) % FT, hp = (k >> 1) / FT, h = hp % CP, pp = hp / CP;
            const int c = 2 * h + e, f2 = tile * FT + fl2;
            wl[k] = (c < C && f2 < p.F) ? (double)wpat[((int64_t)pp * p.F + f2) * C + c] : 0.0;
        }
    }
    __syncthreads();

    constexpr int SUBS = kWave / FT;                                      // object quads per wave step
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: the step range and loop bounds derived from it stay on the scalar unit)
    const int fl = lane % FT, sub = lane / FT;
    const int f = tile * FT + fl;                                         // < Fq (Fq = F rounded up to 64)
    const uint32_t lane_tab = (uint32_t)fl * 4u;
    const uint32_t lane_w = tab_bytes + (uint32_t)fl * 16u;
    const uint32_t na4 = (uint32_t)S * 0x01010101u;                        // four NA state bytes
    // Steps of the chunk (one step = SUBS object quads x FT features) are NOT dealt evenly to the 16 waves.  The SIMD
    // issues oldest-first: with equal shares waves 0-3 (the oldest on their SIMDs) left the loop after 17.8 us, waves
    // 4-7 after 23.9, 8-11 after 30.4 and 12-15 after 36.0 us (in-kernel stamps), so the last third of every block ran
    // with one or two waves per SIMD while the block held the CU.  Age class a = wave / 4 gets the share rows_cum[a+1] -
    // rows_cum[a] (per mille; host: 45 / 27 / 17 / 11 %, measured best) of the steps, as one contiguous range dealt round-robin to its
    // four waves.  Static, so results stay run-to-run deterministic.
    const int total_steps = (nq + SUBS - 1) / SUBS;
    const int cls = wave >> 2, wic = wave & 3;
    const int sb0 = (int)(((int64_t)total_steps * p.rows_cum[cls]) / 1000), sb1 = (int)(((int64_t)total_steps * p.rows_cum[cls + 1]) / 1000);
    const int n_steps = max(0, (sb1 - sb0 - wic + 3) / 4);                 // this wave's steps: sb0 + wic + 4k < sb1
    auto step_of = [&](int k) { return sb0 + wic + 4 * k; };
    // streamed operands through buffer descriptors (plain buffer loads: no flat path, no 64-bit address arithmetic):
    // the quad-interleaved state block (shared by every slot) and the slot's per-object row offsets
    double thread_ll;
    if constexpr (!SORTED) {
    const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(p.state_q), 0, (int)((uint32_t)p.NQ * (uint32_t)p.Fq * 4u), 0x00020000);
    constexpr int QD = (CT + 1) * 4;                                      // offset dwords per quad
    constexpr int WD = SUBS * QD;                                         // ... per wave step
    constexpr int VPL = (WD + kWave - 1) / kWave;                         // dwords a lane fetches per step (1 or 2)
    const __amdgpu_buffer_rsrc_t ro_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(p.rowoff + (int64_t)slot * p.rowoff_stride), 0, (int)((uint32_t)p.NQ * QD * 4u), 0x00020000);
    const uint32_t st_row = (uint32_t)p.Fq * 4u, st_col = (uint32_t)f * 4u;
    // the wave's private offset slot: WD dwords behind the tables and the weights
    const uint32_t slot_lds = tab_bytes + (uint32_t)p.P * CP * FT * 16u + (uint32_t)wave * (WD * 4u);
    const uint32_t my_rows = slot_lds + (uint32_t)sub * (QD * 4u);        // this lane's quad inside the slot

    auto local_quad = [&](int k) { return step_of(k) * SUBS + sub; };
    struct Raw { uint32_t xs; uint32_t ro[VPL]; };
    auto load_raw = [&](int k) -> Raw {                                   // global loads of step k (state dword, offsets)
        Raw r;
        const int i = local_quad(k);
        const uint32_t q = (uint32_t)(q0 + min(i, nq - 1));               // always in bounds
        const uint32_t xs = __builtin_amdgcn_raw_buffer_load_b32(st_rsrc, (int)(q * st_row + st_col), 0, 0);
        r.xs = i < nq ? xs : na4;                                         // past the chunk: four NA observations
        // the wave's first quad of step k is step_of(k)*SUBS; lane l fetches dword(s) l, l+64 of the run
        const uint32_t run0 = (uint32_t)(q0 + step_of(k) * SUBS) * (QD * 4u);
#pragma unroll
        for (int u = 0; u < VPL; ++u)                                     // (past the array: the descriptor returns 0)
            r.ro[u] = __builtin_amdgcn_raw_buffer_load_b32(ro_rsrc, (int)(run0 + (uint32_t)(lane + u * kWave) * 4u), 0, 0);
        return r;
    };
    auto park = [&](const Raw& r) {                                       // offsets of a step -> the wave's LDS slot
#pragma unroll
        for (int u = 0; u < VPL; ++u)
            if (lane + u * kWave < WD)
                *reinterpret_cast<uint32_t*>(lds_raw + slot_lds + (uint32_t)(lane + u * kWave) * 4u) = r.ro[u];
    };
    struct Offs { u32x4_t ro[CT]; u32x4_t po; };
    auto fetch_offsets = [&]() -> Offs {                                  // this lane's quad: (C+1) broadcast reads
        Offs o;
#pragma unroll
        for (int c = 0; c < CT; ++c) o.ro[c] = *reinterpret_cast<const u32x4_t*>(lds_raw + my_rows + (uint32_t)c * 16u);
        o.po = *reinterpret_cast<const u32x4_t*>(lds_raw + my_rows + (uint32_t)CT * 16u);
        return o;
    };
    // the four observation likelihoods of a step (NA -> exactly 1.0)
    auto step_values = [&](uint32_t xs, const Offs& o, double (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = (xs >> (8 * j)) & 0xFFu;                    // <= S (NA / padding byte is S)
            const uint32_t xo = x * state_bytes + lane_tab;
            const uint32_t pj = o.po[j];
            double w[2 * CP];
#pragma unroll
            for (int h = 0; h < CP; ++h) {
                const f64x2_t ww = *reinterpret_cast<const f64x2_t*>(lds_raw + lane_w + pj + (uint32_t)h * (FT * 16u));
                w[2 * h] = ww.x; w[2 * h + 1] = ww.y;
            }
            double acc = x >= (uint32_t)S ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float t = *reinterpret_cast<const float*>(lds_raw + o.ro[c][j] + xo);
                acc = fma(w[c], (double)t, acc);                          // exact product, one rounding: NumPy's order
            }
            v[j] = acc;
        }
    };

    // Software pipeline: global loads two steps ahead (one dword of state, VPL dwords of offsets per lane), the
    // offsets of step k+1 parked in LDS and read back while step k's arithmetic runs.  LDS operations of a wave
    // complete in order, so the slot needs no barrier: the write of step k+1's offsets is issued after the reads
    // of step k's (already in registers), and its read-back after that write.
    {
        ProdAcc pa{1.0, 0, 0, 0u};
        double sum = 0.0;
        constexpr int D = 4;                                             // global loads D steps ahead (2-3 VGPRs per step)
        Raw g0 = load_raw(0), g1 = load_raw(1), g2 = load_raw(2), g3 = load_raw(3);      // (named: no runtime indexing)
        park(g0);
        uint32_t xs_cur = g0.xs;
        Offs o_cur = fetch_offsets();
        auto one_step = [&](int k, Raw& slot_next, Raw& slot_refill) __attribute__((always_inline)) {
            // slot_next holds step k+1, slot_refill (the one step k used) is refilled with step k+D
            slot_refill = load_raw(k + D);
            __builtin_amdgcn_sched_barrier(0);
            double v[4];
            step_values(xs_cur, o_cur, v);
            park(slot_next);                                             // offsets of step k+1 (o_cur is in registers)
            const Offs o_next = fetch_offsets();
            if (MODE == LOG_PRODUCT) prod_add4(pa, v[0], v[1], v[2], v[3]);
            else { sum += log(v[0]); sum += log(v[1]); sum += log(v[2]); sum += log(v[3]); }
            __builtin_amdgcn_sched_barrier(0);
            xs_cur = slot_next.xs; o_cur = o_next;
        };
        for (int k = 0; k < n_steps; k += D) {                           // (n_steps is wave-uniform: uniform branches)
            one_step(k + 0, g1, g0);
            if (k + 1 < n_steps) one_step(k + 1, g2, g1);
            if (k + 2 < n_steps) one_step(k + 2, g3, g2);
            if (k + 3 < n_steps) one_step(k + 3, g0, g3);
        }
        if (MODE == LOG_PRODUCT) {
            thread_ll = log(pa.mant) + (double)(pa.expo - 1023 * n_steps) * 0.693147180559945309417232;
            // rare: a thread whose product left the positive normal range redoes its sum per observation.  Every lane
            // of the wave takes part in the slot traffic, so the whole wave walks the steps again (wave-uniform branch)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pa.bad != 0u) != 0ull, 0)) {
                double s2 = 0.0;
                for (int kk = 0; kk < n_steps; ++kk) {
                    const Raw r = load_raw(kk);
                    park(r);
                    const Offs o = fetch_offsets();
                    double v[4];
                    step_values(r.xs, o, v);
                    s2 += log(v[0]); s2 += log(v[1]); s2 += log(v[2]); s2 += log(v[3]);
                }
                if (pa.bad != 0u) thread_ll = s2;
            }
        } else thread_ll = sum;
    }
    } else {
    // ================= SORTED: pattern-uniform steps, weights in registers, gathered state bytes =================
    constexpr int QD = (CT + 1) * 4;                                      // offset dwords per quad: C table rows + the state row
    constexpr int WD = SUBS * QD;                                         // ... per wave step (<= 64: one dword per lane)
    static_assert(WD <= kWave, "one offsets dword per lane and step");
    const __amdgpu_buffer_rsrc_t ro_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(p.rowoff_s + (int64_t)slot * p.rowoff_s_stride), 0, (int)((uint32_t)NQ_slot * QD * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t st8_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(p.state_s), 0, (int)((uint32_t)(p.N + 1) * (uint32_t)p.state_s_pitch), 0x00020000);
    // two offset slots per wave (steps of even / odd parity): the state rows of step k+2 are read back while the table
    // rows of step k+1 still wait in the other slot
    const uint32_t slots_lds = tab_bytes + (uint32_t)p.P * CP * FT * 16u + (uint32_t)wave * (2u * WD * 4u);
    auto slot_at = [&](int k) -> uint32_t { return slots_lds + (uint32_t)(k & 1) * (WD * 4u); };
    const uint32_t my_quad = (uint32_t)sub * (QD * 4u);
    auto load_ro = [&](int k) -> uint32_t {                               // offsets dword of step k for this lane
        const uint32_t run0 = (uint32_t)(q0 + step_of(k) * SUBS) * (QD * 4u);
        return __builtin_amdgcn_raw_buffer_load_b32(ro_rsrc, (int)(run0 + (uint32_t)lane * 4u), 0, 0);      // (past the slot's order: 0)
    };
    auto park = [&](int k, uint32_t ro) {
        if (lane < WD) *reinterpret_cast<uint32_t*>(lds_raw + slot_at(k) + (uint32_t)lane * 4u) = ro;
    };
    struct StateRows { u32x4_t po; };
    auto read_state_rows = [&](int k) -> u32x4_t {                       // (pattern << 24 | state row offset) of this lane's quad
        return *reinterpret_cast<const u32x4_t*>(lds_raw + slot_at(k) + my_quad + (uint32_t)CT * 16u);
    };
    struct Xs { uint32_t x[4]; uint32_t pat; };
    auto load_states = [&](int k, const u32x4_t po) -> Xs {              // the four state bytes of step k
        // (a step past the wave's range is only ever PREFETCHED, never evaluated, and inside the range every quad of a step
        //  exists -- runs and chunks are whole steps -- so whatever row the offsets name is read: no bounds selects.  Only
        //  object 0 of the quad carries the pattern id above its 24-bit row offset.)
        Xs r;
        r.x[0] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(st8_rsrc, (int)((po[0] & 0x00FFFFFFu) + (uint32_t)f), 0, 0);
#pragma unroll
        for (int j = 1; j < 4; ++j)
            r.x[j] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(st8_rsrc, (int)(po[j] + (uint32_t)f), 0, 0);
        r.pat = po[0] >> 24;
        return r;
    };
    struct Offs { u32x4_t ro[CT]; };
    auto read_table_rows = [&](int k) -> Offs {
        Offs o;
#pragma unroll
        for (int c = 0; c < CT; ++c) o.ro[c] = *reinterpret_cast<const u32x4_t*>(lds_raw + slot_at(k) + my_quad + (uint32_t)c * 16u);
        return o;
    };
    double w[2 * CP];                                                     // the current pattern's weights of this lane's feature
    uint32_t w_pat = 0xFFFFFFFFu;
    auto load_weights = [&](uint32_t pat) {
#pragma unroll
        for (int h = 0; h < CP; ++h) {
            const f64x2_t ww = *reinterpret_cast<const f64x2_t*>(lds_raw + lane_w + pat * (uint32_t)(CP * FT * 16) + (uint32_t)h * (FT * 16u));
            w[2 * h] = ww.x; w[2 * h + 1] = ww.y;
        }
    };
#pragma unroll
    for (int i = 0; i < 2 * CP; ++i) w[i] = 0.0;
    auto step_values = [&](const Xs& xs, const Offs& o, double (&v)[4]) {
        const uint32_t pat = (uint32_t)__builtin_amdgcn_readfirstlane((int)xs.pat);           // (a step holds ONE pattern: wave-uniform)
        if (pat != w_pat) { w_pat = pat; load_weights(pat); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = xs.x[j];                                    // <= S (NA byte is S in the sorted form's state block)
            const uint32_t xo = x * state_bytes + lane_tab;
            double acc = x >= (uint32_t)S ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float t = *reinterpret_cast<const float*>(lds_raw + o.ro[c][j] + xo);
                acc = fma(w[c], (double)t, acc);
            }
            v[j] = acc;
        }
    };
    {
        ProdAcc pa{1.0, 0, 0, 0u};
        double sum = 0.0;
        constexpr int D = 4;
        // prologue: offsets of steps 0 and 1 parked, their state bytes asked for, table rows of step 0 read
        park(0, load_ro(0));
        park(1, load_ro(1));
        uint32_t g0 = load_ro(2), g1 = load_ro(3), g2 = load_ro(4), g3 = load_ro(5);       // (named: no runtime indexing)
        Xs xs_cur = load_states(0, read_state_rows(0));
        Xs xs_nx1 = load_states(1, read_state_rows(1));
        Offs o_cur = read_table_rows(0);
        auto one_step = [&](int k, uint32_t& g_k2) __attribute__((always_inline)) {
            // g_k2 holds the offsets of step k+2; it is parked now (its slot's previous tenant, step k, is in registers) and
            // refilled with step k+2+D
            double v[4];
            step_values(xs_cur, o_cur, v);
            park(k + 2, g_k2);
            g_k2 = load_ro(k + 2 + D);
            const u32x4_t po2 = read_state_rows(k + 2);
            const Xs xs_nx2 = load_states(k + 2, po2);
            const Offs o_next = read_table_rows(k + 1);
            if (MODE == LOG_PRODUCT) prod_add4(pa, v[0], v[1], v[2], v[3]);
            else { sum += log(v[0]); sum += log(v[1]); sum += log(v[2]); sum += log(v[3]); }
            __builtin_amdgcn_sched_barrier(0);
            xs_cur = xs_nx1; xs_nx1 = xs_nx2; o_cur = o_next;
        };
        for (int k = 0; k < n_steps; k += D) {
            one_step(k + 0, g0);
            if (k + 1 < n_steps) one_step(k + 1, g1);
            if (k + 2 < n_steps) one_step(k + 2, g2);
            if (k + 3 < n_steps) one_step(k + 3, g3);
        }
        if (MODE == LOG_PRODUCT) {
            thread_ll = log(pa.mant) + (double)(pa.expo - 1023 * n_steps) * 0.693147180559945309417232;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pa.bad != 0u) != 0ull, 0)) {      // rare: the sum per observation
                double s2 = 0.0;
                for (int kk = 0; kk < n_steps; ++kk) {
                    park(kk, load_ro(kk));
                    const Xs xs = load_states(kk, read_state_rows(kk));
                    const Offs o = read_table_rows(kk);
                    double v[4];
                    step_values(xs, o, v);
                    s2 += log(v[0]); s2 += log(v[1]); s2 += log(v[2]); s2 += log(v[3]);
                }
                if (pa.bad != 0u) thread_ll = s2;
            }
        } else thread_ll = sum;
    }
    }
    const double wsum = wave_sum(thread_ll);
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    if (threadIdx.x < kWave) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kRowsWaves; ++w) total += red[w];            // fixed order: run-to-run deterministic
        finish_partial(p, slot, work, total);
    }
}

// ==========================================================================================
// Fused mixture log-likelihood, group-tuple form with the tuple metadata in LDS (tile widths 32 / 16, S > 127, the
// one-hot stream; at tile width 64 the packed stream runs k_mixture_tuple64 below).
//
// Observation (n, f) contributes log v with v = sum_c w[pat(n)][f][c] * p_c[g_c(n)][f][x(n,f)]:
// v depends on n only through the tuple of group indices t(n) = (g_0(n), .., g_{C-1}(n)), and the
// objects of a sample realise few distinct tuples (headline: 6 = 5 clusters + "no cluster";
// south_america: <= 28).  The block therefore evaluates
//        T[t][x][f] = log( sum_c w[pat(t)][f][c] * p_c[g_c(t)][f][x] )
// once per (tuple, state, feature of its tile) -- same products, same NumPy order, same fp64 log as
// the general kernel -- into LDS, and the per-observation work collapses to ONE conflict-free 8-byte
// LDS gather and ONE fp64 add.  Common-subexpression elimination, not an approximation: every
// eval still recomputes T from the slot's tables and weights (nothing is cached across evals).
// Row x = S of every tuple holds 0.0: NA observations (state byte 0xFF, clamped to S) add log 1.
// Eligibility (host): KT <= kMaxTuples, the image fits LDS and a block has enough observations to
// amortise the KT*S*FT logs; otherwise the general k_mixture_v2 runs.
// ==========================================================================================
template <int FT, int CT, bool ONEHOT>
__global__ __launch_bounds__(kBlock) void k_mixture_combo(Mix2Params p) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red4[4];
    const int unit = ((int)(blockIdx.x >> 3) / p.slots_per_group) * 8 + (int)(blockIdx.x & 7);
    const int slot_i = (unit % p.slot_groups) * p.slots_per_group + (int)(blockIdx.x >> 3) % p.slots_per_group;
    const int work = unit / p.slot_groups;
    if (work >= p.n_work || slot_i >= p.n_batch) return;
    const int slot = p.slot_list ? p.slot_list[slot_i] : p.first_slot + slot_i;
    const int tile = work % p.n_ftiles, chunk = work / p.n_ftiles;
    const int S = p.S, S1 = p.S + 1;
    const int C = CT ? CT : p.C;
    constexpr int CU = CT ? CT : kMaxComponents;
    const int q0 = chunk * p.quads_per_chunk;
    const int q1 = min(p.NQ, q0 + p.quads_per_chunk);
    const int nq = q1 - q0;
    const int KT = p.KT;

    double* T = reinterpret_cast<double*>(lds_raw);                       // [KT][S+1][FT]
    uint32_t* tq = reinterpret_cast<uint32_t*>(T + (size_t)KT * S1 * FT);  // [quads_per_chunk] 4 tuple ids per quad

    // ---- build the tile's log table: one entry per (tuple, state, feature) --------------------------
    // small per-slot metadata first (tuple -> group rows / pattern, the tile's weights, tuple ids of the
    // chunk), then the table itself with the probability loads of 4 entries in flight per thread
    uint16_t* tgl = reinterpret_cast<uint16_t*>(tq + p.quads_per_chunk);   // [KT][CU] group row of each tuple
    uint32_t* tpl = reinterpret_cast<uint32_t*>(tgl + (size_t)KT * CU + ((KT * CU) & 1));   // [KT] pattern id
    double* wls = reinterpret_cast<double*>(lds_raw + p.combo_w_off);      // [P][C][FT] weights of the tile
    {
        const uint16_t* tg = p.tuple_g + (int64_t)slot * p.tuple_g_stride;
        const uint8_t* tp = p.tuple_p + (int64_t)slot * p.tuple_p_stride;
        for (int k = threadIdx.x; k < KT * CU; k += kBlock) tgl[k] = tg[(k / CU) * kMaxComponents + (k % CU)];
        for (int k = threadIdx.x; k < KT; k += kBlock) tpl[k] = tp[k];
        const double* wpat_t = p.wpat_t + (int64_t)slot * p.wpat_t_stride + (int64_t)tile * p.wpat_tile_stride;
        for (int k = threadIdx.x; k < p.P * C * FT; k += kBlock) wls[k] = wpat_t[k];
        const uint8_t* tid = p.tid + (int64_t)slot * p.tid_stride;
        for (int k = threadIdx.x; k < nq; k += kBlock)
            tq[k] = *reinterpret_cast<const uint32_t*>(tid + 4 * (q0 + k));
        for (int e = threadIdx.x; e < KT * FT; e += kBlock)                 // NA row
            T[((e / FT) * S1 + S) * FT + (e % FT)] = 0.0;
    }
    __syncthreads();
    {
        const float* probs_t = p.probs_t + (int64_t)slot * p.probs_t_stride + (int64_t)tile * ((int64_t)(p.Gtot + 1) * S * FT);
        const int n_ent = KT * S * FT;
        constexpr int U = 8;
        for (int e0 = threadIdx.x; e0 < n_ent; e0 += U * kBlock) {
            float pr[U][CU];
            int dst[U], tt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                                   // issue every load of the group
                const int e = e0 + u * kBlock;
                const int ec = e < n_ent ? e : threadIdx.x;                 // in-range stand-in, result unused
                const int fl = ec % FT, r = ec / FT;
                const int s = r % S, t = r / S;
                tt[u] = t;
                // features past F in the last tile are never gathered (their state bytes are NA): no log for them
                // ... nor for tuples this slot does not have (pattern 0xFF: the launch's KT is the batch maximum)
                dst[u] = (e < n_ent && tile * FT + fl < p.F && tpl[t] != 0xFFu) ? (t * S1 + s) * FT + fl : -1;
#pragma unroll
                for (int c = 0; c < CU; ++c)
                    pr[u][c] = (CT || c < C) ? probs_t[((int64_t)tgl[t * CU + c] * S + s) * FT + fl] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] < 0) continue;
                const double* w = wls + tpl[tt[u]] * (uint32_t)(C * FT) + (dst[u] % FT);
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < CU; ++c) {
                    if (CT || c < C) {
                        const double term = w[c * FT] * (double)pr[u][c];
                        v = c == 0 ? term : v + term;                       // NumPy order, no FMA
                    }
                }
                T[dst[u]] = fast_log_pos(v);
            }
        }
    }
    __syncthreads();

    double sum0 = 0.0, sum1 = 0.0;
    if constexpr (!ONEHOT) {
    constexpr int ROWS = kWave / FT;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    const int fl = lane % FT, sub = lane / FT;
    const int f = tile * FT + fl;
    const int n_steps = (nq + 4 * ROWS - 1) / (4 * ROWS);
    const uint32_t* sq = p.state_q + (int64_t)q0 * p.Fq + f;
    const double* T_l = T + fl;
    auto local_quad = [&](int k) { return (k * 4 + wid) * ROWS + sub; };
    // running 32-bit offsets of the state stream (no 64-bit multiply per step): quad i of the chunk sits at
    // dword i * Fq; steps advance by 4*ROWS quads; reads past the chunk are clamped to its last quad and masked
    const uint32_t off_step = (uint32_t)(4 * ROWS) * (uint32_t)p.Fq, off_last = (uint32_t)(nq - 1) * (uint32_t)p.Fq;
    uint32_t pre_off = (uint32_t)local_quad(0) * (uint32_t)p.Fq;
    int pre_i = local_quad(0);
    auto next_state = [&]() -> uint32_t {
        const uint32_t xs = sq[min(pre_off, off_last)];
        const uint32_t r = pre_i < nq ? xs : 0xFFFFFFFFu;
        pre_off += off_step;
        pre_i += 4 * ROWS;
        return r;
    };
    // two steps (8 objects per lane) per trip, four independent accumulators: the LDS gathers of a trip
    // are all in flight together and the add chains are short; state dwords are fetched one trip ahead
    auto gather4 = [&](uint32_t xs, int k, double& a, double& b) {
        uint32_t t4 = tq[min(local_quad(k), nq - 1)];
        if (FT == kWave) t4 = __builtin_amdgcn_readfirstlane(t4);
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = min((xs >> (8 * j)) & 0xFFu, (uint32_t)S);   // NA (0xFF) -> zero row
            const uint32_t t = (t4 >> (8 * j)) & 0xFFu;
            v[j] = T_l[(t * S1 + x) * FT];
        }
        a += v[0] + v[2];
        b += v[1] + v[3];
    };
    double sum2 = 0.0, sum3 = 0.0;
    uint32_t xa_next = next_state(), xb_next = next_state();
    int k = 0;
    for (; k + 1 < n_steps; k += 2) {
        const uint32_t xa = xa_next, xb = xb_next;
        xa_next = next_state();
        xb_next = next_state();
        gather4(xa, k, sum0, sum1);
        gather4(xb, k + 1, sum2, sum3);
    }
    if (k < n_steps) gather4(xa_next, k, sum0, sum1);
    sum0 += sum2;
    sum1 += sum3;
    } else {
        // one-hot stream (the block as the reference hands it over).  A lane owns one 16-byte chunk
        // position `ch` of the tile row segment and walks down the objects, R = 256 / seg16 objects per
        // step, so everything that depends on the byte position is lane-constant: the 16 byte flags
        // are OR-shifted into one word (bit 8k+i = byte k of dword i) and, for each set bit b found with
        // ffbl, the LDS offset (state, feature) -> x*FT + f of that byte comes from a per-block lookup
        // table offtab[ch][b] -- no division, no decode arithmetic: one u16 LDS read, one 8-byte LDS
        // gather and one fp64 add per observation.
        const int n0 = 4 * q0;
        const int n_obj = min(4 * nq, p.N - n0);
        const int seg_off = tile * FT * S;
        const int seg16 = min(FT * S, p.rs_pitch - seg_off) >> 4;       // <= kBlock (host-checked)
        const int R = kBlock / seg16;
        uint16_t* offtab = reinterpret_cast<uint16_t*>(lds_raw + p.combo_tab_off);    // [seg16][32]
        for (int e = threadIdx.x; e < seg16 * 32; e += kBlock) {
            const int b = e & 31, chh = e >> 5;
            const int j = chh * 16 + ((b & 7) << 2) + (b >> 3);
            const int fj = j / S;
            offtab[e] = (uint16_t)((j - fj * S) * FT + min(fj, FT - 1));
        }
        __syncthreads();
        const bool active = (int)threadIdx.x < R * seg16;
        const int ch = active ? (int)threadIdx.x % seg16 : 0, row_off = active ? (int)threadIdx.x / seg16 : 0;
        const int n_steps = (n_obj + R - 1) / R;
        const uint8_t* oh = p.onehot + (int64_t)n0 * p.rs_pitch + seg_off + ch * 16;
        const uint8_t* tidl = reinterpret_cast<const uint8_t*>(tq);       // tuple id per object of the chunk
        const uint16_t* off_l = offtab + ch * 32;
        auto fetch = [&](int k) -> uint4 {
            const int nl = k * R + row_off;
            const uint4 d = *reinterpret_cast<const uint4*>(oh + (int64_t)min(nl, n_obj - 1) * p.rs_pitch);
            return (active && nl < n_obj) ? d : make_uint4(0u, 0u, 0u, 0u);
        };
        uint4 nxt = fetch(0);
        for (int k = 0; k < n_steps; ++k) {
            const uint4 d = nxt;
            nxt = fetch(k + 1);
            uint32_t m = d.x | (d.y << 1) | (d.z << 2) | (d.w << 3);
            const double* Tt = T + (uint32_t)tidl[min(k * R + row_off, n_obj - 1)] * (uint32_t)(S1 * FT);
            while (__builtin_amdgcn_ballot_w64(m != 0u)) {                // wave-uniform trip count
                const bool has = m != 0u;
                const int b = has ? __builtin_ctz(m) : 0;
                const double v = Tt[off_l[b]];
                sum0 += has ? v : 0.0;
                m &= m - 1u;
            }
        }
    }
    const double total = block_sum(sum0 + sum1, red4);
    if (threadIdx.x < kWave) finish_partial(p, slot, work, total);
}

// ==========================================================================================
// Group-tuple form, 64-feature tiles, packed stream: the scalar-unit version of k_mixture_combo
// (default PACKED path at tile width 64 and S <= 127).
//
// With FT = 64 a wave step is one object quad, so everything that depends on the object -- its tuple, the
// tuple's table rows -- is wave-uniform and belongs on the scalar side; the vector side is left with, per
// observation, ONE address add, ONE conflict-free 8-byte LDS gather and ONE fp64 add:
//   * state stream `state_h` [NQ][Fq] of 4 x u16 (built once by k_ingest_onehot): the entry of (object n,
//     feature f) is the observation's byte offset inside a tuple's table block, x*512 + (f%64)*8
//     (x = S for NA / padding: the block's zero row) -- the lane-constant part of the LDS address is folded
//     into the data, so no shift, no lane term at run time;
//   * `toff` [Np] u32 per slot (built by the host with the tuple ids): byte offset of the object's tuple
//     block, tid*(S+1)*512.  The 16 offsets of a batch of 4 quads arrive with one scalar load
//     (s_load_dwordx16) straight into SGPRs;
//   * address = v_add_u32_sdwa(SGPR offset, u16 half of the state dword): one VALU op;
//   * the state stream is read through a buffer descriptor (constant per-lane offset + scalar row offset: no
//     vector address arithmetic), one batch ahead;
//   * each wave owns a contiguous range of the chunk's quads.
// Table build: a wave owns rows r = w, w+4, .. of the KT*S (tuple, state) rows; per-row offsets (probability
// rows of the C components, weight pattern, destination) are computed 64 rows at a time on the vector side
// (lane l <-> row l of the wave) and handed to the scalar side with v_readlane; the probability loads of a
// batch of rows are in flight before the first log.
// LDS is addressed absolutely (the kernel has no static LDS, so the dynamic block starts at 0; checked).
// Same products, same NumPy order, same fp64 log per table entry as the other forms.
// ==========================================================================================
typedef __attribute__((address_space(3))) const double lds_cdouble_t;
typedef __attribute__((address_space(3))) double lds_double_t;
typedef __attribute__((address_space(3))) unsigned char lds_uchar_t;

#define SBE_SDWA_ADD(dst, soff, vx, sel0, sel1)                                                             \
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel0 " src1_sel:" sel1   \
        : "=v"(dst) : "s"(soff), "v"(vx))

template <int CT, bool OFF16, int NW>           // NW: waves per block (4; 8-wave blocks were tried twice and lose: 80 or 64 VGPRs spill)
__global__ __launch_bounds__(NW * kWave, 4) void k_mixture_tuple64(Mix2Params p) {
    constexpr int kThreads = NW * kWave;
    constexpr int FT = 64;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // Block order.  Work items = (tile, chunk); the last tile is LIGHT when it runs in sub-row mode (ragged_w).
    // Slots are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8).  Inside an XCD the blocks come in
    // generations of `gen_slots` slots (one generation = as many blocks as the XCD's CUs hold at once), and
    // inside a generation all heavy work items come before all light ones, so that whatever order the
    // dispatcher fills the CUs in, every CU ends up with the same mix.  Fewer than 8 slots: slot-major order.
    int slot_i, work;
    {
        const int n_chunks = p.n_work / p.n_ftiles;
        const int n_light = p.ragged_w ? n_chunks : 0, n_heavy = p.n_work - n_light;
        if (p.n_batch >= 8) {
            const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
            const int slots_here = (p.n_batch - xcd + 7) >> 3;               // slots xcd, xcd + 8, ...
            const int gen = j / (p.gen_slots * p.n_work), jj = j - gen * (p.gen_slots * p.n_work);
            const int s0 = gen * p.gen_slots, s_gen = min(p.gen_slots, slots_here - s0);   // this generation's slots
            if (s_gen <= 0) return;                                          // padding blocks (before any barrier)
            const int heavy_gen = s_gen * n_heavy, light_gen = s_gen * n_light;
            int sl, wk;
            const int jh = jj, jl = jj - heavy_gen;
            if (jh < heavy_gen) {
                const int ht = p.n_ftiles - (n_light ? 1 : 0);               // heavy tiles
                sl = jh / n_heavy; wk = jh - sl * n_heavy; wk = (wk / ht) * p.n_ftiles + wk % ht;
            } else {
                if (jl >= light_gen) return;                                 // padding
                sl = jl / n_light; wk = (jl - sl * n_light) * p.n_ftiles + (p.n_ftiles - 1);
            }
            slot_i = (s0 + sl) * 8 + xcd; work = wk;
        } else {
            slot_i = (int)blockIdx.x / p.n_work; work = (int)blockIdx.x - slot_i * p.n_work;
            if (slot_i >= p.n_batch) return;
        }
    }
    const int slot = p.slot_list ? p.slot_list[slot_i] : p.first_slot + slot_i;
    const int tile = work % p.n_ftiles, chunk = work / p.n_ftiles;
    const int S = p.S, S1 = p.S + 1;
    const int C = CT ? CT : p.C;
    constexpr int CU = CT ? CT : kMaxComponents;
    const int q0 = chunk * p.quads_per_chunk;
    const int nq = min(p.NQ, q0 + p.quads_per_chunk) - q0;
    const int KT = p.KT;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane8 = (uint32_t)lane * 8u;
    if ((uint32_t)(uintptr_t)(lds_uchar_t*)lds_raw != 0u) {       // absolute LDS addressing needs base 0
        if (threadIdx.x < kWave) finish_partial(p, slot, work, __longlong_as_double(0x7FF8000000000000ll));
        return;
    }
    // LDS map (bytes): T [KT][S+1][64] f64 at 0 | weights [P][C][64] f64 at combo_w_off | 4 doubles (reduction)
    //                  | 128 x {1/c, log c} (tab_log_pos)
    const uint32_t w_off = (uint32_t)p.combo_w_off;
    const uint32_t red_off = w_off + (uint32_t)(p.P * C * FT) * 8u;
    const uint32_t tab_off = red_off + (uint32_t)NW * 8u;
    double* red4 = reinterpret_cast<double*>(lds_raw + red_off);
    const bool ragged = p.ragged_w != 0 && tile == p.n_ftiles - 1;           // block-uniform

    // this wave's quads [qa, qb) of the chunk
    const int per = (nq + NW - 1) / NW;
    const int qa = min(nq, w * per), qb = min(nq, qa + per);
    const int n_my = qb - qa;
    const uint4* toff4 = reinterpret_cast<const uint4*>(p.toff + (int64_t)slot * p.toff_stride) + (q0 + qa);
    const float* probs_tile = p.probs_t + (int64_t)slot * p.probs_t_stride + (int64_t)tile * ((int64_t)(p.Gtot + 1) * S * FT);
    const uint16_t* tuple_g = p.tuple_g + (int64_t)slot * p.tuple_g_stride;
    const uint8_t* tuple_p = p.tuple_p + (int64_t)slot * p.tuple_p_stride;
    const int n_rows = KT * S;

    // ---- gather operands that can be in flight during the table build (full-tile mode) -----------------
    // tuple-block offsets of this wave's objects: lane l <-> quad l of a group of 64 quads (one coalesced
    // 16-byte load per lane and group, handed to the scalar side quad by quad with v_readlane)
    auto load_offsets = [&](int g0) -> uint4 {                            // g0: first quad of the group (wave-relative)
        return (g0 + lane < n_my) ? toff4[g0 + lane] : make_uint4(0u, 0u, 0u, 0u);
    };
    const __amdgpu_buffer_rsrc_t sh_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(p.state_h)), 0, (int)((uint32_t)p.NQ * (uint32_t)p.Fq * 8u), 0x00020000);
    const int fcol8 = (tile * FT + lane) * 8;
    const uint32_t qrow_bytes = (uint32_t)p.Fq * 8u;
    auto load_quad = [&](int q_wave) -> u32x2_t {                         // 4 x u16 of one quad (wave-relative index,
        const int q = min(q0 + qa + q_wave, q0 + nq - 1);                 //  clamped to the chunk: always in bounds)
        return __builtin_amdgcn_raw_buffer_load_b64(sh_rsrc, fcol8, (int)((uint32_t)q * qrow_bytes), 0);
    };
    // two batches of QB quads of the state stream are kept in flight (measured: a deeper rolling window does
    // not pay -- the gather runs at ~80 % of its VALU issue bound and wants its 32 LDS reads per trip batched)
    constexpr int QB = 4;
    uint4 offv = make_uint4(0u, 0u, 0u, 0u);
    u32x2_t xa[QB], xb[QB];

    // ---- table-build operands of the wave's first rows, also in flight before the first barrier --------
    // A wave owns rows r = w, w+4, .. of the KT*S (tuple, state) rows; lane l <-> its l-th row (64 rows per
    // pass): the offsets of everything a row needs are computed on the vector side and handed to the scalar
    // side row by row with v_readlane; the probability loads of a batch of U rows are issued together.
    const __amdgpu_buffer_rsrc_t pr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(probs_tile), 0, (int)((uint32_t)(p.Gtot + 1) * (uint32_t)S * FT * 4u), 0x00020000);
    const int lane4 = lane * 4;
    const int my_rows = (!ragged && n_rows > w) ? (n_rows - w + NW - 1) / NW : 0;
    constexpr int U = CU <= 2 ? 16 : (CU <= 4 ? 8 : 4);
    uint32_t v_goff[CU], v_woff = 0u, v_doff = 0u;
    auto row_offsets = [&](int base) {
        const uint32_t r = (uint32_t)(w + NW * min(base + lane, my_rows - 1));
        const uint32_t t = r / (uint32_t)S, st = r - t * (uint32_t)S;
#pragma unroll
        for (int c = 0; c < CU; ++c)
            v_goff[c] = (CT || c < C) ? ((uint32_t)tuple_g[t * kMaxComponents + c] * (uint32_t)S + st) * (FT * 4u) : 0u;
        const uint32_t pat = tuple_p[t];
        v_woff = w_off + pat * (uint32_t)(C * FT * 8);
        // a tuple this slot does not have (pattern 0xFF: the launch's KT is the batch maximum): row skipped
        v_doff = pat != 0xFFu ? (t * (uint32_t)S1 + st) * (FT * 8u) : 0xFFFFFFFFu;
    };
    float pr[U][CU];
    auto row_loads = [&](int i0, int n_here) {                             // rows i0 .. i0+U-1 of the current pass
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < n_here) {
#pragma unroll
                for (int c = 0; c < CU; ++c)
                    if (CT || c < C)
                        pr[u][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            pr_rsrc, lane4, __builtin_amdgcn_readlane((int)v_goff[c], i0 + u), 0));
            }
        }
    };
    // Start-up loads in TWO dependent levels (a block's start is pure memory latency, ~1 us per level under
    // load): level 1 = everything that needs no other load -- the tuple rows / patterns of the wave's table rows
    // (issued first: level 2 waits for them only), the tile's weights, the log table, the gather operands --
    // level 2 = the probability rows; the LDS fills follow.
    if (my_rows > 0) row_offsets(0);
    const double* wpat_t = p.wpat_t + (int64_t)slot * p.wpat_t_stride + (int64_t)tile * p.wpat_tile_stride;
    const int n_wl = p.P * C * FT;
    constexpr int WPRE = 2;                                                  // weight loads per thread held in registers
    double w_pre[WPRE];
#pragma unroll
    for (int j = 0; j < WPRE; ++j) {
        const int k = (int)threadIdx.x + j * kThreads;
        w_pre[j] = k < n_wl ? wpat_t[k] : 0.0;
    }
    const int lt_i = (int)threadIdx.x & (2 * kLogTabEntries - 1);            // (8-wave blocks: the upper half repeats)
    const double lt_pre = reinterpret_cast<const double*>(p.logtab)[lt_i];   // 256 doubles
    if (!ragged) {
        offv = load_offsets(0);
#pragma unroll
        for (int i = 0; i < QB; ++i) xa[i] = load_quad(i);
#pragma unroll
        for (int i = 0; i < QB; ++i) xb[i] = load_quad(QB + i);
    }
    if (my_rows > 0) row_loads(0, min(kWave, my_rows));

    {   // weights of the tile, the NA rows, the log table
        double* wls = reinterpret_cast<double*>(lds_raw + w_off);
#pragma unroll
        for (int j = 0; j < WPRE; ++j) {
            const int k = (int)threadIdx.x + j * kThreads;
            if (k < n_wl) wls[k] = w_pre[j];
        }
        for (int k = (int)threadIdx.x + WPRE * kThreads; k < n_wl; k += kThreads) wls[k] = wpat_t[k];
        double* T = reinterpret_cast<double*>(lds_raw);
        for (int e = threadIdx.x; e < KT * FT; e += kThreads) T[((e >> 6) * S1 + S) * FT + (e & 63)] = 0.0;
        if (threadIdx.x < 2 * kLogTabEntries) reinterpret_cast<double*>(lds_raw + tab_off)[threadIdx.x] = lt_pre;
    }
    __syncthreads();

    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (!ragged) {
        // ---- log table ------------------------------------------------------------------------------------
        {
            const bool live = tile * FT + lane < p.F;
            for (int base = 0; base < my_rows; base += kWave) {
                const int n_here = min(kWave, my_rows - base);
                if (base) row_offsets(base);
                for (int i0 = 0; i0 < n_here; i0 += U) {
                    if (base || i0) row_loads(i0, n_here);                      // (the first batch is already in flight)
                    // G rows at a time: the G log chains (each ~20 dependent FMAs) are straight-line code the
                    // scheduler interleaves -- a wave that walks its rows one by one issues one vector instruction per
                    // dependent-issue latency and four such waves do not fill a SIMD (measured: a block alone on
                    // its CU is only 20 % faster than one of four)
                    constexpr int G = U < 4 ? U : 4;
#pragma unroll
                    for (int u0 = 0; u0 < U; u0 += G) {
                        double vv[G];
                        uint32_t dd[G];
                        bool ok[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const int u = u0 + g;                                // (i0 + u <= 63: U divides 64)
                            const uint32_t doff_s = (uint32_t)__builtin_amdgcn_readlane((int)v_doff, i0 + u);
                            ok[g] = i0 + u < n_here && doff_s != 0xFFFFFFFFu;     // wave-uniform
                            const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)v_woff, i0 + u) + lane8;
                            dd[g] = doff_s + lane8;
                            double v = 0.0;
#pragma unroll
                            for (int c = 0; c < CU; ++c) {
                                if (CT || c < C) {
                                    const double wc = *(lds_cdouble_t*)(uintptr_t)(woff + (uint32_t)c * (FT * 8u));
                                    const double term = wc * (double)pr[u][c];
                                    v = c == 0 ? term : v + term;               // NumPy order, no FMA
                                }
                            }
                            vv[g] = (ok[g] && live) ? v : 1.0;                   // dead lanes of the last tile, rows not there: log 1
                        }
                        double lg[G];
                        bool special = false;
#pragma unroll
                        for (int g = 0; g < G; ++g) special |= tab_log_special(vv[g]);
                        tab_log_core_n<G>(vv, lg, tab_off);
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {     // rare: library log
#pragma unroll
                            for (int g = 0; g < G; ++g) if (tab_log_special(vv[g])) lg[g] = lib_log(vv[g]);
                        }
#pragma unroll
                        for (int g = 0; g < G; ++g)
                            if (ok[g]) *(lds_double_t*)(uintptr_t)dd[g] = lg[g];
                    }
                }
            }
        }
        __syncthreads();

        // ---- gather: one address add + one LDS read + one fp64 add per observation ----------------------
        // OFF16: every tuple block starts below 64 KiB, so two offsets share an SGPR (2 v_readlane per quad
        // instead of 4) and the address add selects the halves of BOTH operands
        uint32_t olo = OFF16 ? (offv.x | (offv.y << 16)) : 0u, ohi = OFF16 ? (offv.z | (offv.w << 16)) : 0u;
        auto quad = [&](const u32x2_t x, int ql) {                          // ql: quad's lane in the offset group
            uint32_t ad0, ad1, ad2, ad3;
            if (OFF16) {
                const uint32_t s01 = (uint32_t)__builtin_amdgcn_readlane((int)olo, ql);
                const uint32_t s23 = (uint32_t)__builtin_amdgcn_readlane((int)ohi, ql);
                SBE_SDWA_ADD(ad0, s01, x.x, "WORD_0", "WORD_0"); SBE_SDWA_ADD(ad1, s01, x.x, "WORD_1", "WORD_1");
                SBE_SDWA_ADD(ad2, s23, x.y, "WORD_0", "WORD_0"); SBE_SDWA_ADD(ad3, s23, x.y, "WORD_1", "WORD_1");
            } else {
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)offv.x, ql);
                const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)offv.y, ql);
                const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)offv.z, ql);
                const uint32_t s3 = (uint32_t)__builtin_amdgcn_readlane((int)offv.w, ql);
                SBE_SDWA_ADD(ad0, s0, x.x, "DWORD", "WORD_0"); SBE_SDWA_ADD(ad1, s1, x.x, "DWORD", "WORD_1");
                SBE_SDWA_ADD(ad2, s2, x.y, "DWORD", "WORD_0"); SBE_SDWA_ADD(ad3, s3, x.y, "DWORD", "WORD_1");
            }
            a0 += *(lds_cdouble_t*)(uintptr_t)ad0; a1 += *(lds_cdouble_t*)(uintptr_t)ad1;
            a2 += *(lds_cdouble_t*)(uintptr_t)ad2; a3 += *(lds_cdouble_t*)(uintptr_t)ad3;
        };
        for (int g0 = 0; g0 < n_my; g0 += kWave) {                          // groups of 64 quads (one offset load)
            const int ng = min(kWave, n_my - g0);
            if (g0) {
                offv = load_offsets(g0);
                if (OFF16) { olo = offv.x | (offv.y << 16); ohi = offv.z | (offv.w << 16); }
#pragma unroll
                for (int i = 0; i < QB; ++i) xa[i] = load_quad(g0 + i);
#pragma unroll
                for (int i = 0; i < QB; ++i) xb[i] = load_quad(g0 + QB + i);
            }
            // batches alternate between the two register sets; a set is refilled (two batches ahead) as soon
            // as it has been consumed
            int i = 0;
            for (; i + 2 * QB <= ng; i += 2 * QB) {
#pragma unroll
                for (int j = 0; j < QB; ++j) quad(xa[j], i + j);
#pragma unroll
                for (int j = 0; j < QB; ++j) xa[j] = load_quad(g0 + i + 2 * QB + j);
#pragma unroll
                for (int j = 0; j < QB; ++j) quad(xb[j], i + QB + j);
#pragma unroll
                for (int j = 0; j < QB; ++j) xb[j] = load_quad(g0 + i + 3 * QB + j);
            }
            // tail (< 2*QB quads): already loaded, in xa then xb
            for (int j = 0; i < ng; ++i, ++j) {
                u32x2_t x = xa[0];
#pragma unroll
                for (int k = 1; k < QB; ++k) x = j == k ? xa[k] : x;
#pragma unroll
                for (int k = 0; k < QB; ++k) x = j == QB + k ? xb[k] : x;
                quad(x, i);
            }
        }
    } else {
        // ---- sub-row mode for a narrow last tile (ragged_w <= 32 valid features): RW lanes per row, 64/RW
        // rows per wave step, everything per lane (vector side).  ~64/RW times less work than a full tile, so the
        // block is all memory latency: every load that does not depend on another is issued up front -- the first
        // batch of gather operands before the table build, the build's operands for RI rows per lane in two
        // levels (tuple rows / patterns, then probabilities) instead of three levels per row. --
        int sh = 0;
        while ((1 << sh) < p.ragged_w) ++sh;                                 // RW = 1 << sh lanes per row
        const int RW = 1 << sh, SUB = kWave >> sh;
        const int fl = lane & (RW - 1), sub = lane >> sh;
        const bool live = fl < p.ragged_w;
        const uint2* sh2 = p.state_h + (int64_t)(q0 + qa) * p.Fq + (tile * FT + fl);
        uint2 x[8]; uint4 o[8];
        auto load_batch = [&](int ql0) {                                    // 8 steps' gather operands
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ql = max(0, min(ql0 + u * SUB + sub, n_my - 1));
                x[u] = sh2[(int64_t)ql * p.Fq];
                o[u] = toff4[ql];
            }
        };
        if (n_my > 0) load_batch(0);
        constexpr int RI = CU <= 2 ? 4 : (CU <= 4 ? 2 : 1);                 // rows per lane and pass
        for (int r0 = w * SUB; r0 < n_rows; r0 += NW * SUB * RI) {
            uint32_t t[RI], st[RI], pat[RI], gr[RI][CU];
            float pc[RI][CU];
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                const uint32_t r = (uint32_t)min(r0 + i * NW * SUB + sub, n_rows - 1);
                t[i] = r / (uint32_t)S; st[i] = r - t[i] * (uint32_t)S;
                pat[i] = tuple_p[t[i]];
#pragma unroll
                for (int c = 0; c < CU; ++c)
                    gr[i][c] = (CT || c < C) ? (uint32_t)tuple_g[t[i] * kMaxComponents + c] : 0u;
            }
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int c = 0; c < CU; ++c)
                    pc[i][c] = (CT || c < C) ? probs_tile[(gr[i][c] * (uint32_t)S + st[i]) * FT + fl] : 0.0f;
            double vv[RI], lg[RI];
            bool ok[RI], special = false;
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                // (a tuple this slot does not have -- pattern 0xFF -- has no row: skipped)
                ok[i] = r0 + i * NW * SUB + sub < n_rows && pat[i] != 0xFFu;
                const double* wr = reinterpret_cast<const double*>(lds_raw + w_off) + (ok[i] ? pat[i] : 0u) * (uint32_t)(C * FT) + fl;
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < CU; ++c) {
                    if (CT || c < C) {
                        const double term = wr[c * FT] * (double)pc[i][c];
                        v = c == 0 ? term : v + term;                           // NumPy order, no FMA
                    }
                }
                vv[i] = (ok[i] && live) ? v : 1.0;
            }
#pragma unroll
            for (int i = 0; i < RI; ++i) special |= tab_log_special(vv[i]);
            tab_log_core_n<RI>(vv, lg, tab_off);                                                       // interleaved chains
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {                 // rare: library log
#pragma unroll
                for (int i = 0; i < RI; ++i) if (tab_log_special(vv[i])) lg[i] = lib_log(vv[i]);
            }
#pragma unroll
            for (int i = 0; i < RI; ++i)
                if (ok[i]) *(lds_double_t*)(uintptr_t)((t[i] * (uint32_t)S1 + st[i]) * (FT * 8u) + (uint32_t)fl * 8u) = lg[i];
        }
        __syncthreads();
        for (int ql0 = 0; ql0 < n_my; ql0 += 8 * SUB) {
            if (ql0) load_batch(ql0);                                        // (the first batch is already in flight)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ql0 + u * SUB + sub < n_my) {
                    a0 += *(lds_cdouble_t*)(uintptr_t)(o[u].x + (x[u].x & 0xFFFFu));
                    a1 += *(lds_cdouble_t*)(uintptr_t)(o[u].y + (x[u].x >> 16));
                    a2 += *(lds_cdouble_t*)(uintptr_t)(o[u].z + (x[u].y & 0xFFFFu));
                    a3 += *(lds_cdouble_t*)(uintptr_t)(o[u].w + (x[u].y >> 16));
                }
            }
        }
    }
    {   // fixed-order block reduction over NW waves
        const double wsum = wave_sum((a0 + a2) + (a1 + a3));
        if (lane == 0) red4[w] = wsum;
        __syncthreads();
        if (threadIdx.x < kWave) {
            double total = 0.0;
            if (NW == 4) total = (red4[0] + red4[1]) + (red4[2] + red4[3]);
            else {
#pragma unroll
                for (int i = 0; i < NW; ++i) total += red4[i];
            }
            finish_partial(p, slot, work, total);
        }
    }
}
#undef SBE_SDWA_ADD

}  // namespace sbe
