// sbe_mixture_mfma_ws.hip -- the batched group-tuple form on the matrix pipe, WAVE-SPECIALISED (round 6).
//
// Same mathematics as k_mixture_tuple_mfma (sbe_mixture_mfma.hip: LL[b] = sum_{t,f,s} cnt_b[t][(f,s)] * log v_b(t,f,s), counts by an
// exact FP4 0/1 contraction; reference expression sbayes/sampling/loggers.py:355-357 over sbayes/model/likelihood.py:104-133,
// 171-190), another division of labour.  There every wave holds 2 x MT accumulator tiles (96 registers at MT = 3), which keeps
// the kernel at two waves per SIMD, and a wave's timeline is serial: counts, then its epilogue at the issue rate of a lone wave.
// Here a block is PW PRODUCER waves and PW x MT CONSUMER waves (4 + 12 at MT = 3: four waves per SIMD at <= 128 registers):
//   * a producer owns one 32-column tile per generation: MT accumulator tiles (A fragments from LDS, X fragments from L2), then it
//     hands the finished counts to LDS as u16 (a 32 x 32 tile = 2 KB; counts <= N < 65 536), in accumulator order;
//   * a consumer owns one (column tile, M tile) pair per generation: it reads the 16 counts of its lanes back and runs the
//     epilogue of k_mixture_tuple_mfma on them (table operands one step ahead, the table log of the mantissa, exponents summed as
//     integers) -- three consumers and one producer share a SIMD, so the matrix pipe works under the vector instructions of
//     the previous generation's epilogue and a consumer's load / LDS latencies are covered by its neighbours;
//   * generations are separated by ONE block barrier: in generation g the producers fill half g & 1 of the ring while the
//     consumers drain half (g - 1) & 1.
// Final reduction: the exponent sums leave the lanes as integers (64-bit across lanes and waves), the bias of the biased
// exponents is taken off ONCE per (slot, column split) -- 1023 x the number of observations in the split's columns, a property of
// the data (tile_prefix) -- and ln 2 is applied once.  Partial sums / tickets / results as in k_mixture_tuple_mfma.
// Limits: FP4 operands only; LDS = log table 16 KB + A image MT x ceil(N / 64) KB + ring 2 x PW x MT x 2 KB (<= 160 KB: N <= ~1 900
// at MT = 3); otherwise the unspecialised kernel runs.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "sbe_mixture_mfma.hip.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "sbe_mixture_mfma_ws.hip: the in-kernel final reduction orders relaxed agent-scope atomics by s_waitcnt (gfx942 / gfx950 cache behaviour); not valid for this target"
#endif

namespace sbe {

template <int MT, int CT, int PW>
__global__ __launch_bounds__((PW + PW * MT) * kWave, 1) void k_mixture_tuple_mfma_ws(MfmaMixParams p) {
    constexpr int NCONS = PW * MT, NW = PW + NCONS, NTHR = NW * kWave;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = (int)blockIdx.x % p.n_split, sg = (int)blockIdx.x / p.n_split;
    const int KBp = p.KBp;
    // LDS map: log table at ABSOLUTE address 0 (its index is the whole address) | A fragments [MT][KBp][64] x 16 B | meta [16][2 MT] |
    // ring [2][PW][MT][4 quads][64 lanes] x 8 B | reduction [NCONS][16] f64 + [NCONS][16] i64
    constexpr uint32_t tab_off = 0u;
    constexpr uint32_t a_off = kFineLogEntries * 16u;
    const uint32_t a_bytes = (uint32_t)MT * (uint32_t)KBp * 1024u;
    const uint32_t meta_off = a_off + a_bytes;
    const uint32_t ring_off = meta_off + (uint32_t)(kMfmaSlots * 2 * MT * sizeof(TupleMeta<CT>));
    constexpr uint32_t kTileBytes = 4u * 64u * 8u;
    const uint32_t red_off = ring_off + 2u * PW * MT * kTileBytes;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw;
    typedef TupleMeta<CT> Meta;
    Meta* meta = reinterpret_cast<Meta*>(lds_raw + meta_off);
    double* redl = reinterpret_cast<double*>(lds_raw + red_off);
    long long* redk = reinterpret_cast<long long*>(lds_raw + red_off + NCONS * kMfmaSlots * sizeof(double));

    auto slot_of = [&](int sl) -> int {          // absolute slot of the block's sl-th slot, or -1
        const int i = sg * kMfmaSlots + sl;
        if (i >= p.n_batch) return -1;
        return p.slot_list ? p.slot_list[i] : p.first_slot + i;
    };
    const int nt_lo = split * p.nt_per_split, nt_hi = min(p.NT, nt_lo + p.nt_per_split);
    const int n_gen = (nt_hi - nt_lo + PW - 1) / PW;             // generations: PW column tiles each

    if (lds_base != 0u) {                                    // (block-uniform; before any barrier; the host checks this too)
        if ((int)threadIdx.x < kMfmaSlots) {
            const int slot = slot_of((int)threadIdx.x);
            if (slot >= 0) {
                p.partials[(int64_t)slot * p.partials_stride + split] = __longlong_as_double(0x7FF8000000000000ll);
                if (p.results) p.results[slot] = __longlong_as_double(0x7FF8000000000000ll);
            }
        }
        return;
    }
    mfma_phase0<MT, CT, true, NTHR>(lds_raw, p, slot_of, tab_off, a_off, meta, KBp);
    __syncthreads();

    const int h = lane >> 5, cl = lane & 31;
    if (w < PW) {
        // ================================ producer: counts of one column tile per generation ==================================
        // k-blocks in flight: a producer has no partner wave of its own kind on its SIMD, so its X fragments must be asked for a
        // whole L2 round trip ahead (PF x MT MFMAs of 32 cycles); KBp is a multiple of 4, so PF = 8 needs the tail handled below
        constexpr int PF = 8;
        const __amdgpu_buffer_rsrc_t xt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.xt), 0, (int)p.xt_bytes, 0x00020000);
        const int lane16 = lane * 16;
        auto tile_off = [&](int nt) -> int { return (nt < nt_hi ? nt : p.NT) * KBp * 1024; };       // scalar; the zero tile behind the array
        auto load_b = [&](int toff, int kb) -> v4i_t {
            const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(xt_rsrc, lane16, toff + kb * 1024, 0);
            v4i_t r; r.x = (int)d.x; r.y = (int)d.y; r.z = (int)d.z; r.w = (int)d.w;
            return r;
        };
        const uint32_t a_lane = a_off + (uint32_t)lane * 16u;
        v4i_t bq[PF];
        {
            const int toff0 = tile_off(nt_lo + w);
#pragma unroll
            for (int i = 0; i < PF; ++i) bq[i] = load_b(toff0, i);
        }
        for (int g = 0; g <= n_gen; ++g) {
            if (g < n_gen) {
                const int toff = tile_off(nt_lo + g * PW + w);
                v16f_t acc[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
                v4i_t a_cur[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a_cur[m] = *(lds_cv4i_t*)(uintptr_t)(a_lane + ((uint32_t)m * (uint32_t)KBp) * 1024u);
                for (int kb0 = 0; kb0 < KBp; kb0 += PF) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const int kb = kb0 + i;
                        if (i >= 4 && kb >= KBp) break;     // (KBp = 4 (mod 8): the second half of the last round does not exist; uniform)
                        v4i_t a_nxt[MT];
#pragma unroll
                        for (int m = 0; m < MT; ++m)        // (the read behind the last k-block lands in the next M tile / the metadata: valid LDS, unused)
                            a_nxt[m] = *(lds_cv4i_t*)(uintptr_t)(a_lane + ((uint32_t)m * (uint32_t)KBp + (uint32_t)(kb + 1)) * 1024u);
                        const v4i_t b = bq[i];
                        bq[i] = load_b(toff, kb + PF);      // (behind the tile's last k-block: the next tile's first fragments, discarded)
                        const v8i_t b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const v8i_t a8 = {a_cur[m].x, a_cur[m].y, a_cur[m].z, a_cur[m].w, 0, 0, 0, 0};
                            acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[m], 4, 4, 0, 0, 0, 0);
                        }
#pragma unroll
                        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
                        __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);                 // the next k-block's A fragments
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // the X fragment PF k-blocks ahead
                        __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);                 // this k-block's MFMAs
                    }
                }
                // the next generation's first X fragments are on their way while this one's counts leave for LDS
                {
                    const int toff_n = tile_off(nt_lo + (g + 1) * PW + w);
#pragma unroll
                    for (int i = 0; i < PF; ++i) bq[i] = load_b(toff_n, i);
                }
                // counts -> ring half g & 1 as u16, accumulator order: quad j of lane l = registers 4 j .. 4 j + 3
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const uint32_t tile = ring_off + (uint32_t)((((g & 1) * PW + w) * MT + m)) * kTileBytes + (uint32_t)lane * 8u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t c0 = (uint32_t)acc[m][4 * j + 0], c1 = (uint32_t)acc[m][4 * j + 1];
                        const uint32_t c2 = (uint32_t)acc[m][4 * j + 2], c3 = (uint32_t)acc[m][4 * j + 3];
                        uint2 v;
                        v.x = c0 | (c1 << 16); v.y = c2 | (c3 << 16);
                        *reinterpret_cast<uint2*>(lds_raw + tile + (uint32_t)j * 512u) = v;
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ================================ consumer: epilogue of one (column tile, M tile) per generation =======================
        const int c = w - PW, pw = c / MT, m = c % MT;
        double lsum[8];                                               // per slot of this lane: sum of cnt * log(mantissa part)
        int ksum[8];                                                  // ... and of cnt * biased binary exponent (exact)
#pragma unroll
        for (int i = 0; i < 8; ++i) { lsum[i] = 0.0; ksum[i] = 0; }
        uint32_t one_hi = 0x3FF00000u;
        asm volatile("" : "+v"(one_hi));                              // (a VGPR operand of tab_log4_n's v_bfi_b32)
        const __amdgpu_buffer_rsrc_t pr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.probs), 0, (int)p.probs_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpat), 0, (int)p.wpat_bytes, 0x00020000);
        constexpr int G = 4;                                          // entries per step: one register quad (tuple t, slots sl0 .. sl0 + 3)
        uint32_t col4 = 0, fw4 = 0;
        Meta mdn[2][G];                                               // metadata of the next two steps (LDS reads a step ahead of their use)
        float prq[2][G][CT], wrq[2][G][CT];
        int f_next = 0;                                               // feature of this lane's column in the NEXT tile (asked for a generation ahead)
        auto colc_of = [&](int nt) -> uint32_t { return (uint32_t)min(min(nt, p.NT - 1) * 32 + cl, p.FS - 1); };   // (beyond F * S: no counts, any valid address)
        auto st_cols = [&](int nt, int f) { col4 = colc_of(nt) * 4u; fw4 = (uint32_t)f * (uint32_t)(CT * 4); };
        auto st_meta = [&](int j, int buf) {                          // step j of a tile: tuple 2 m + (j >> 1), slots 8 (j & 1) + 4 h + 0..3
            const int t = 2 * m + (j >> 1), sl0 = 8 * (j & 1) + 4 * h;
#pragma unroll
            for (int i = 0; i < G; ++i) mdn[buf][i] = meta[(sl0 + i) * 2 * MT + t];
        };
        auto st_load = [&](int q) {                                   // step q's table operands; its metadata sits in mdn[q & 1]
            const Meta (&mdq)[G] = mdn[q & 1];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t wo = mdq[i].woff + fw4;
                if constexpr (CT == 2) {
                    const u32x2_t v2 = __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, (int)wo, 0, 0);
                    const uint32_t e0 = v2.x, e1 = v2.y;             // (copies: see k_mixture_tuple_mfma)
                    wrq[q & 1][i][0] = __uint_as_float(e0); wrq[q & 1][i][1] = __uint_as_float(e1);
                } else if constexpr (CT == 4) {
                    const u32x4_t v4 = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)wo, 0, 0);
                    const uint32_t e0 = v4.x, e1 = v4.y, e2 = v4.z, e3 = v4.w;
                    wrq[q & 1][i][0] = __uint_as_float(e0); wrq[q & 1][i][1] = __uint_as_float(e1);
                    wrq[q & 1][i][2] = __uint_as_float(e2); wrq[q & 1][i][3] = __uint_as_float(e3);
                } else {
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc)
                        wrq[q & 1][i][cc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w_rsrc, (int)(wo + 4u * cc), 0, 0));
                }
#pragma unroll
                for (int cc = 0; cc < CT; ++cc)
                    prq[q & 1][i][cc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr_rsrc, (int)(mdq[i].goff[cc] + col4), 0, 0));
            }
        };
        auto st_comp = [&](int j, const uint2 (&tile)[4]) __attribute__((always_inline)) {
            double vv[G], lg[G], cntd[G];
            int cnt[G];
            bool special = false;
            const uint32_t lo = tile[j].x, hi2 = tile[j].y;
            cnt[0] = (int)(lo & 0xFFFFu); cnt[1] = (int)(lo >> 16); cnt[2] = (int)(hi2 & 0xFFFFu); cnt[3] = (int)(hi2 >> 16);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                cntd[i] = (double)cnt[i];
                double v = 0.0;                                       // sum_c w_c * p_c in NumPy's order (products of float32 are exact in fp64)
#pragma unroll
                for (int cc = 0; cc < CT; ++cc) {
                    const double wc = (double)wrq[j & 1][i][cc], pc = (double)prq[j & 1][i][cc];
                    v = cc == 0 ? wc * pc : fma(wc, pc, v);
                }
                vv[i] = v;
                special |= __builtin_amdgcn_class(v, 0x2FF);            // anything but a positive normal double
            }
            int kx[G];
            tab_log4_n<G>(vv, lg, kx, tab_off, one_hi);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {      // rare: see k_mixture_tuple_mfma
#pragma unroll
                for (int i = 0; i < G; ++i)
                    if (__builtin_amdgcn_class(vv[i], 0x2FF)) { lg[i] = cnt[i] != 0 ? lib_log(vv[i]) : 0.0; kx[i] = kFineLogBias; }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int k = (j & 1) * 4 + i;
                lsum[k] = fma(cntd[i], lg[i], lsum[k]);
                ksum[k] = __mul24(cnt[i], kx[i]) + ksum[k];            // (biased exponents; host: generations * N * 2100 < 2^31)
                asm volatile("" : "+v"(ksum[k]));
                asm volatile("" : "+v"(lsum[k]));
            }
        };
        // the first tile's operands are asked for before the first barrier
        st_cols(nt_lo + pw, p.colfeat[colc_of(nt_lo + pw)]);
        f_next = p.colfeat[colc_of(nt_lo + pw + PW)];
        st_meta(0, 0);
        st_meta(1, 1);
        st_load(0);
        for (int g = 0; g <= n_gen; ++g) {
            if (g > 0) {
                const int nt = nt_lo + (g - 1) * PW + pw;             // this generation's tile of the consumer
                const uint32_t tbase = ring_off + (uint32_t)(((((g - 1) & 1) * PW + pw) * MT + m)) * kTileBytes + (uint32_t)lane * 8u;
                uint2 tile[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[j] = *reinterpret_cast<const uint2*>(lds_raw + tbase + (uint32_t)j * 512u);
                const bool valid = nt < nt_hi;                         // (wave-uniform: the last generation may be short)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // step j + 1's operands (its metadata was read a step ago), then the metadata of step j + 2 into the buffer
                    // step j's loads have just left; behind a tile's last step come the next generation's first two
                    if (j == 3) { st_cols(nt + PW, f_next); f_next = p.colfeat[colc_of(nt + 2 * PW)]; }
                    st_load(j + 1);
                    st_meta((j + 2) & 3, j & 1);
                    if (valid) st_comp(j, tile);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
        // ---- 32 columns of a lane half; the consumer's sums per slot go to LDS ------------------------------------------------
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double lv = lsum[i];
            long long kv = (long long)ksum[i];
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) { lv += __shfl_xor(lv, off, 64); kv += __shfl_xor(kv, off, 64); }
            if (cl == 0) {
                const int sl = 8 * (i >> 2) + 4 * h + (i & 3);
                redl[c * kMfmaSlots + sl] = lv; redk[c * kMfmaSlots + sl] = kv;
            }
        }
    }
    __syncthreads();
    if (w != 0) return;                                                    // the rest is wave 0's (no block barrier below)
    const int my_slot = lane < kMfmaSlots ? slot_of(lane) : -1;
    double total = 0.0;
    if (my_slot >= 0) {
        double lt = 0.0;
        long long kt = 0;
#pragma unroll
        for (int cc = 0; cc < NCONS; ++cc) { lt += redl[cc * kMfmaSlots + lane]; kt += redk[cc * kMfmaSlots + lane]; }
        // every object is in exactly one tuple of a slot, so the split's columns counted tile_prefix[hi] - tile_prefix[lo] observations
        kt -= (long long)kFineLogBias * (long long)(p.tile_prefix[nt_hi] - p.tile_prefix[nt_lo]);
        total = fma((double)kt, 6.93147180559945286227e-01, lt);
    }
    double* const my_partials = p.partials + (int64_t)max(my_slot, 0) * p.partials_stride;
    if (!p.results) {
        if (my_slot >= 0) my_partials[split] = total;
        return;
    }
    // cross-split reduction by the group's last block: see the MEMORY-MODEL NOTE in sbe_mixture_mfma.hip (relaxed agent-scope atomics
    // ordered by s_waitcnt on gfx942 / gfx950; SBE_REDUCE_IN_KERNEL=0 takes the two-launch form)
    if (p.n_split > 1) {
        if (my_slot >= 0) __hip_atomic_store(my_partials + split, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = 0;
        if (lane == 0) {
            t = __hip_atomic_fetch_add(p.arrive + sg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned)p.n_split - 1u) __hip_atomic_store(p.arrive + sg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t != (unsigned)p.n_split - 1u) return;                         // (wave-uniform)
        asm volatile("" ::: "memory");
        if (my_slot >= 0) {
            total = 0.0;
            for (int k = 0; k < p.n_split; ++k) total += __hip_atomic_load(my_partials + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (my_slot >= 0) p.results[my_slot] = total;
    signal_done(p.done);
}

// producers per block for MT M tiles: PW + PW * MT waves <= 16
constexpr int ws_producers(int MT) { return MT == 1 ? 8 : MT == 2 ? 5 : MT == 3 ? 4 : 3; }

size_t tuple_mfma_ws_lds_bytes(int MT, int C, int KBp) {   // log table | A fragments | meta | ring | reduction
    const int PW = ws_producers(MT);
    const size_t meta = (size_t)kMfmaSlots * 2 * MT * (C <= 1 ? 8 : (C <= 3 ? 16 : 32));
    return (size_t)MT * KBp * 1024 + kFineLogEntries * 16 + meta + (size_t)2 * PW * MT * 2048 + (size_t)PW * MT * kMfmaSlots * 16;
}

template <int MT, int CT>
static bool ws_prepare() {
    const void* fn = reinterpret_cast<const void*>(&k_mixture_tuple_mfma_ws<MT, CT, ws_producers(MT)>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncAttributes attr{};
    if (hipFuncGetAttributes(&attr, fn) != hipSuccess) { (void)hipGetLastError(); return true; }
    return attr.sharedSizeBytes == 0;
}

template <int MT>
static void launch_ws_mt(int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st) {
    constexpr int PW = ws_producers(MT);
    constexpr int threads = (PW + PW * MT) * kWave;
    switch (C) {
        case 1: k_mixture_tuple_mfma_ws<MT, 1, PW><<<grid, threads, lds, st>>>(p); break;
        case 2: k_mixture_tuple_mfma_ws<MT, 2, PW><<<grid, threads, lds, st>>>(p); break;
        case 3: k_mixture_tuple_mfma_ws<MT, 3, PW><<<grid, threads, lds, st>>>(p); break;
        default: k_mixture_tuple_mfma_ws<MT, 4, PW><<<grid, threads, lds, st>>>(p); break;
    }
}

// false (nothing launched): an instance carries static LDS (its log table must sit at LDS address 0)
bool launch_tuple_mfma_ws(int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st) {
    static const bool no_static_lds = [] {
        bool ok = true;
        ok &= ws_prepare<1, 1>(); ok &= ws_prepare<1, 2>(); ok &= ws_prepare<1, 3>(); ok &= ws_prepare<1, 4>();
        ok &= ws_prepare<2, 1>(); ok &= ws_prepare<2, 2>(); ok &= ws_prepare<2, 3>(); ok &= ws_prepare<2, 4>();
        ok &= ws_prepare<3, 1>(); ok &= ws_prepare<3, 2>(); ok &= ws_prepare<3, 3>(); ok &= ws_prepare<3, 4>();
        ok &= ws_prepare<4, 1>(); ok &= ws_prepare<4, 2>(); ok &= ws_prepare<4, 3>(); ok &= ws_prepare<4, 4>();
        return ok;
    }();
    if (!no_static_lds) return false;
    const int MT = (p.KT + 1) / 2;
    switch (MT) {
        case 1: launch_ws_mt<1>(C, p, grid, lds, st); break;
        case 2: launch_ws_mt<2>(C, p, grid, lds, st); break;
        case 3: launch_ws_mt<3>(C, p, grid, lds, st); break;
        default: launch_ws_mt<4>(C, p, grid, lds, st); break;
    }
    return true;
}

}  // namespace sbe
