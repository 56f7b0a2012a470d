// sbe_mixture_mfma.hip -- the group-tuple form of the fused mixture log-likelihood for LARGE BATCHES of resident states,
// with the per-observation gather replaced by an exact integer contraction on the matrix pipe (round 5).
//
// What it computes (reference expression: sbayes/sampling/loggers.py:355-357 over sbayes/model/likelihood.py:104-133,
// 171-190): for every slot b of the launch
//     LL[b] = sum_{n, f: not NA} log sum_c w[pat(n), f, c] * p_c[g_c(n), f, x(n, f)]
// In the group-tuple form an object's (g_0 .. g_{C-1}) is one of KT tuples, so the log argument depends on
// (tuple t, feature f, state s) only -- the table T_b[t][f][s] that k_mixture_tuple64 builds in LDS and then gathers
// from, once per observation.  All B slots of a launch share ONE feature block; what differs is which tuple an object
// belongs to.  Hence
//     LL[b] = sum_{t, f, s} cnt_b[t][f][s] * T_b[t][f][s],      cnt_b[t][(f, s)] = sum_n [tid_b(n) = t] * X[n][(f, s)]
// and cnt is a product of two 0/1 byte matrices: A[(b, t)][n] = [tid_b(n) = t] (built in LDS from the slots' tuple ids)
// and X (the one-hot block, transposed once into MFMA fragment order: k_xt_frags).  v_mfma_i32_32x32x32_i8 accumulates it
// exactly (counts <= N fit i32), the table entry is computed ONCE per (slot, tuple, feature, state) in the lane that holds
// its count -- same products, same NumPy order of the component sum, same table-driven fp64 log as the other forms -- and
// never goes through LDS.  NA observations have an all-zero one-hot row and are counted nowhere.
//
// Block = 16 slots x a range of 32-column tiles, 8 waves.  M tile m (32 rows) = tuples 2m, 2m+1 x the 16 slots
// (row = (t & 1) * 16 + slot): the 32x32 accumulator layout (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
// then gives every lane ONE column and, per register quad, ONE tuple of four consecutive slots.  A wave owns pairs of
// column tiles: 2 x MT accumulator tiles, A fragments from LDS (shared by all waves), X fragments straight from L2 in
// 1 KB fully coalesced pieces (fragment order in memory), four k-blocks ahead.  Both operands take their k index from the
// same (lane half, byte) position, so the product does not depend on the instruction's internal k order.
//
// Operand format (round 6): FP4.  0 and 1.0 are exact in e2m1 (0x0, 0x2) and a count <= N < 2^24 is exact in an f32 accumulator, so
// v_mfma_f32_32x32x64_f8f6f4 (cbsz = blgp = 4, scale operands 0: the unscaled form) computes the SAME integers at twice the i8
// instruction's depth for the same issue time (tools/probe/mfma_fp4_probe.hip: 35.1 against 35.1 counter ticks per instruction,
// all 1024 outputs of an asymmetric 0/1 contraction exact): half the MFMAs, half the A image in LDS (48 KB instead of 96 KB at the
// headline shape: N up to ~2 900 fits), half the X-fragment bytes from L2.  A k-block is then 64 objects; a lane's 16 bytes hold
// 32 nibbles, byte b of dword w: low nibble = object 8 w + b, high nibble = object 8 w + 4 + b of the lane half's 32 objects --
// the order in which the A build gets them out of the tuple-id bytes with two shifts; the X fragments (k_xt_frags) use the same
// one.  The i8 form stays selectable (SBE_MFMA_FP4=0) for same-box comparisons; both are oracle-checked.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "sbe_mixture_mfma.hip.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "sbe_mixture_mfma.hip: the in-kernel final reduction orders relaxed agent-scope atomics by s_waitcnt (gfx942 / gfx950 cache behaviour); not valid for this target"
#endif

namespace sbe {

constexpr int kMfmaWaves = 8;
constexpr int kMfmaThreads = kMfmaWaves * kWave;
constexpr int kMfmaRN = kTupleMfmaColsPerPass;   // column tiles per wave pass (2; the host's overflow bound uses the same constant)
// (Other block shapes were measured in round 5 and lost: 16 waves x 1 column tile at 128 registers, +9 %; 8 waves x 1 tile with
//  the next tile's MFMAs issued inside the epilogue, +7 %: profiles/r5/mfma_kernel_experiments_session2.log.)

// One-hot block -> fragment order.  i8 operands: fragment (nt, kb): lane l holds bytes j = 0..15 = X[n = 32 kb + 16 (l >> 5) + j]
// [col = 32 nt + (l & 31)] with col = f * S + s.  FP4 operands: a k-block is 64 objects; lane l holds 32 nibbles (1.0 = 0x2) of
// the objects n0 = 64 kb + 32 (l >> 5) + ..: byte b of dword w, low nibble = object n0 + 8 w + b, high nibble = n0 + 8 w + 4 + b.
// Zero for n >= N, col >= F * S and NA observations.
__global__ void k_xt_frags(const uint8_t* __restrict__ state /* [N][Fp], 0xFF = NA */, uint8_t* __restrict__ xt,
                           int N, int F, int S, int Fp, int NT, int KBp, int fp4) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (int64_t)NT * KBp * 64) return;
    const int l = (int)(u & 63), kb = (int)((u >> 6) % KBp), nt = (int)((u >> 6) / KBp);
    const int col = nt * 32 + (l & 31), f = col / S, s = col - f * S;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (col < F * S) {
        if (fp4) {
            const int n0 = kb * 64 + 32 * (l >> 5);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int n = n0 + 8 * (j >> 3) + (j & 7);                     // object j of the lane half, in order
                const int b = j & 3, hi = (j >> 2) & 1;                       // -> byte b of dword j >> 3, low / high nibble
                if (n < N && state[(int64_t)n * Fp + f] == (uint8_t)s) w[j >> 3] |= 0x2u << (8 * b + 4 * hi);
            }
        } else {
            const int n0 = kb * 32 + 16 * (l >> 5);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = n0 + j;
                if (n < N && state[(int64_t)n * Fp + f] == (uint8_t)s) w[j >> 2] |= 1u << (8 * (j & 3));
            }
        }
    }
    reinterpret_cast<uint4*>(xt)[u] = make_uint4(w[0], w[1], w[2], w[3]);
}

// colcount[col] = objects whose feature f is in state s (col = f * S + s), zero for the padding columns and the whole tile NT;
// colfeat[col] = f (the last feature for padding columns); tile_prefix[t] = observations counted in the column tiles before t
// (t = 0 .. NT + 1).  Data only; built once with the fragment image.
__global__ void k_colcount(const uint8_t* __restrict__ state, int32_t* __restrict__ colcount, int32_t* __restrict__ colfeat,
                           int N, int F, int S, int Fp, int NT) {
    const int col = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (col >= (NT + 1) * 32) return;
    int c = 0;
    const int f = min(col / S, F - 1);
    if (col < F * S) {
        const int s = col - f * S;
        for (int n = 0; n < N; ++n) c += state[(int64_t)n * Fp + f] == (uint8_t)s;
    }
    colcount[col] = c;
    colfeat[col] = f;
}

__global__ void k_tile_prefix(const int32_t* __restrict__ colcount, int32_t* __restrict__ tile_prefix, int NT) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int t = 0; t <= NT + 1; ++t) {
        tile_prefix[t] = acc;
        if (t <= NT) for (int i = 0; i < 32; ++i) acc += colcount[t * 32 + i];
    }
}

// `out`: colcount [(NT + 1) * 32] | colfeat [(NT + 1) * 32] | tile_prefix [NT + 2]
size_t column_tables_bytes(int NT) { return ((size_t)(NT + 1) * 64 + (size_t)NT + 2) * sizeof(int32_t); }
void launch_column_tables(const uint8_t* state, int32_t* out, int N, int F, int S, int Fp, int NT, hipStream_t st) {
    int32_t* colcount = out;
    int32_t* colfeat = out + (size_t)(NT + 1) * 32;
    int32_t* tile_prefix = out + (size_t)(NT + 1) * 64;
    k_colcount<<<(unsigned)(((NT + 1) * 32 + 255) / 256), 256, 0, st>>>(state, colcount, colfeat, N, F, S, Fp, NT);
    k_tile_prefix<<<1, 64, 0, st>>>(colcount, tile_prefix, NT);
}

void launch_xt_frags(const uint8_t* state, uint8_t* xt, int N, int F, int S, int Fp, int NT, int KBp, bool fp4, hipStream_t st) {
    const int64_t units = (int64_t)NT * KBp * 64;
    k_xt_frags<<<(unsigned)((units + 255) / 256), 256, 0, st>>>(state, xt, N, F, S, Fp, NT, KBp, fp4 ? 1 : 0);
}

// the table of tab_log4_n (host side: ensure_xt uploads it)
void fine_log_table(double* tab /* [2 * kFineLogEntries] */) {
    for (int i = 0; i < kFineLogEntries; ++i) {
        const long double c = i == 0 ? 1.0L : i == kFineLogEntries - 1 ? 2.0L : 1.0L + (i + 0.5L) / kFineLogEntries;
        const double inv_c = (double)(1.0L / c);
        // log of the centre the kernel actually divides by (1 / inv_c), halved from the split on
        const long double lc = -logl((long double)inv_c) - (i >= kFineLogSplit ? logl(2.0L) : 0.0L);
        tab[2 * i] = 0.5 * inv_c;                              // (exact: the kernel's first FMA yields s = r / 2)
        tab[2 * i + 1] = (i == 0 || i == kFineLogEntries - 1) ? 0.0 : (double)lc;
    }
}

// SLB = log2(slots per block): 4 (16 slots x <= 8 tuples: the form of round 5), 2 (4 slots x <= 32 tuples), 1 (2 slots x <= 64 tuples:
// the whole tuple table) -- an M tile's 32 rows are 32 / SL tuples x SL slots (row = (tuple in tile) * SL + slot), so a block of
// MT <= 4 tiles holds MT * 32 / SL tuples.  The wide forms (round 6) put datasets with several confounders on the matrix pipe.
template <int MT, int CT, int GT = 4, bool FP4 = true, int SLB = 4>
__global__ __launch_bounds__(kMfmaThreads, 1) void k_mixture_tuple_mfma(MfmaMixParams p) {
    typedef typename std::conditional<FP4, v16f_t, v16i_t>::type acc_t;      // counts: exact integers either way
    constexpr int SL = 1 << SLB, TPT = 32 / SL, TPB = MT * TPT;              // slots per block; tuples per M tile / per block
    constexpr int NK = SL == 16 ? 8 : SL;                                    // slots a lane sums for
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = (int)blockIdx.x % p.n_split, sg = (int)blockIdx.x / p.n_split;
    const int KBp = p.KBp;
    // LDS map: log table 128 x 16 B at ABSOLUTE address 0 (its index is the whole address: the kernel has no static LDS, so
    // the dynamic block starts at 0 -- checked) | A fragments [MT][KBp][64] x 16 B | meta [16][2 MT] | reduction [8 waves][16] f64
    constexpr uint32_t tab_off = 0u;
    constexpr uint32_t a_off = kFineLogEntries * 16u;
    const uint32_t a_bytes = (uint32_t)MT * (uint32_t)KBp * 1024u;
    const uint32_t meta_off = a_off + a_bytes;
    const uint32_t red_off = meta_off + (uint32_t)(SL * TPB * sizeof(TupleMeta<CT>));
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw;
    typedef TupleMeta<CT> Meta;
    Meta* meta = reinterpret_cast<Meta*>(lds_raw + meta_off);
    double* red = reinterpret_cast<double*>(lds_raw + red_off);

    auto slot_of = [&](int sl) -> int {          // absolute slot of the block's sl-th slot, or -1
        const int i = sg * SL + sl;
        if (i >= p.n_batch) return -1;
        return p.slot_list ? p.slot_list[i] : p.first_slot + i;
    };

    // ---- column tiles of this wave; the first X fragments are asked for before anything else --------------------------
    // X fragments come through a buffer descriptor: lane-constant vector offset, scalar fragment offset, no address
    // arithmetic and no bounds branches -- a tile beyond the split's range reads the zero tile behind the array
    // (index NT), and the PF fragments read ahead past a tile's last k-block are the next tile's first (discarded).
    const int nt_lo = split * p.nt_per_split, nt_hi = min(p.NT, nt_lo + p.nt_per_split);
    constexpr int PF = 4;                         // k-blocks in flight (KBp is a multiple of PF)
    const __amdgpu_buffer_rsrc_t xt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.xt), 0, (int)p.xt_bytes, 0x00020000);
    const int lane16 = lane * 16;
    auto tile_off = [&](int nt) -> int { return (nt < nt_hi ? nt : p.NT) * KBp * 1024; };       // scalar
    auto load_b = [&](int toff, int kb) -> v4i_t {
        const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(xt_rsrc, lane16, toff + kb * 1024, 0);
        v4i_t r; r.x = (int)d.x; r.y = (int)d.y; r.z = (int)d.z; r.w = (int)d.w;
        return r;
    };
    v4i_t bq[PF][kMfmaRN];
    {
        const int nt0 = nt_lo + w * kMfmaRN;
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int r = 0; r < kMfmaRN; ++r) bq[i][r] = load_b(tile_off(nt0 + r), i);
    }

    if (lds_base != 0u) {                                    // (block-uniform; before any barrier)
        if ((int)threadIdx.x < SL) {
            const int slot = slot_of((int)threadIdx.x);
            if (slot >= 0) {
                p.partials[(int64_t)slot * p.partials_stride + split] = __longlong_as_double(0x7FF8000000000000ll);
                if (p.results) p.results[slot] = __longlong_as_double(0x7FF8000000000000ll);
            }
        }
        return;
    }
    // ---- phase 0: tuple metadata, log table, A fragments (sbe_mixture_mfma.hip.h) -----------------------------------------
    mfma_phase0<MT, CT, FP4, kMfmaThreads, decltype(slot_of), SL>(lds_raw, p, slot_of, tab_off, a_off, meta, KBp);
    __syncthreads();

    // ---- phase 1: counts on the matrix pipe, table entries + log + dot product on the vector pipe ----------------------
    const int h = lane >> 5, cl = lane & 31;
    double lsum[NK];                                              // per slot of this lane: sum of cnt * log(mantissa part)
    int ksum[NK];                                                 // ... and of cnt * binary exponent (exact)
#pragma unroll
    for (int i = 0; i < NK; ++i) { lsum[i] = 0.0; ksum[i] = 0; }
    // entry = register e (0..15) of a count tile in lane half h: row = (e & 3) + 8 (e >> 2) + 4 h of the tile's 32 rows,
    // slot = row % SL, tuple = row / SL (+ the tile's first tuple).  A register quad (4 consecutive rows) is one tuple's four
    // consecutive slots (SL = 16, 4) or two tuples x two slots (SL = 2).
    auto ent_slot = [&](int e) -> int { return ((e & 3) + 8 * (e >> 2) + 4 * h) & (SL - 1); };
    auto ent_tuple = [&](int m, int e) -> int { return m * TPT + (((e & 3) + 8 * (e >> 2) + 4 * h) >> SLB); };
    auto ent_k = [&](int e) -> int { return SL == 16 ? (e & 3) + 4 * ((e >> 2) & 1) : (e & (SL - 1)); };      // index into lsum / ksum (no h: static)
    int csum = 0;                                                 // objects counted in this lane's columns so far (the same for every slot)
    uint32_t one_hi = 0x3FF00000u;
    asm volatile("" : "+v"(one_hi));                              // (a VGPR operand of tab_log4_n's v_bfi_b32)
    const __amdgpu_buffer_rsrc_t pr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.probs), 0, (int)p.probs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpat), 0, (int)p.wpat_bytes, 0x00020000);
    const uint32_t a_lane = a_off + (uint32_t)lane * 16u;

    // counts of one pass (RN column tiles from the scalar fragment offsets toff[]) into acc; the first PF X fragments of the
    // pass are in bq already
    auto counts_pass = [&](acc_t (&acc)[MT][kMfmaRN], const int (&toff)[kMfmaRN]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < kMfmaRN; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][r][i] = 0;
        // A fragments one k-block ahead in their own registers (the read after the last k-block lands in the next M tile
        // or in the log table: valid LDS, unused), X fragments PF k-blocks ahead; per k-block 3 LDS reads, 2 loads, 2 MT MFMAs
        v4i_t a_cur[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = *(lds_cv4i_t*)(uintptr_t)(a_lane + ((uint32_t)m * (uint32_t)KBp) * 1024u);
        for (int kb0 = 0; kb0 < KBp; kb0 += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int kb = kb0 + i;
                v4i_t a_nxt[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a_nxt[m] = *(lds_cv4i_t*)(uintptr_t)(a_lane + ((uint32_t)m * (uint32_t)KBp + (uint32_t)(kb + 1)) * 1024u);
                v4i_t b[kMfmaRN];
#pragma unroll
                for (int r = 0; r < kMfmaRN; ++r) b[r] = bq[i][r];
#pragma unroll
                for (int r = 0; r < kMfmaRN; ++r) bq[i][r] = load_b(toff[r], kb + PF);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < kMfmaRN; ++r)
                        if constexpr (FP4) {
                            // (an FP4 operand is the first four registers of the eight-register operand)
                            const v8i_t a8 = {a_cur[m].x, a_cur[m].y, a_cur[m].z, a_cur[m].w, 0, 0, 0, 0};
                            const v8i_t b8 = {b[r].x, b[r].y, b[r].z, b[r].w, 0, 0, 0, 0};
                            acc[m][r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[m][r], 4, 4, 0, 0, 0, 0);
                        } else {
                            acc[m][r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[m], b[r], acc[m][r], 0, 0, 0);
                        }
#pragma unroll
                for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
                __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);                 // the next k-block's A fragments
                __builtin_amdgcn_sched_group_barrier(0x020, kMfmaRN, 0);            // X fragments PF k-blocks ahead
                __builtin_amdgcn_sched_group_barrier(0x008, MT * kMfmaRN, 0);       // this k-block's MFMAs
            }
        }
    };

    // epilogue: LL += cnt * log(sum_c w * p) over the wave's RN * MT count tiles.  A "quad" = the four entries
    // (tuple t, slots sl0 .. sl0+3) of one register quad of one tile; software pipeline over the quads: the tuple
    // metadata (LDS) two quads ahead, the table operands (L2 / HBM) one quad ahead.  Columns beyond F*S and tiles
    // beyond the split have no counts (their X fragments are zero), so nothing needs a bounds condition here.
    // A step = G entries of one register quad (G = 4: the whole quad; G = 2: half of it); steps run r-minor.
    constexpr int G = GT;
    constexpr int HQ = 4 / G;                                    // steps per quad
    constexpr int NST = MT * 4 * HQ * kMfmaRN;                   // steps: (m, j, half) major, r minor
    uint32_t col4[kMfmaRN], fw4[kMfmaRN];
    Meta mdq[G];
    float prq[2][G][CT], wrq[2][G][CT];
    auto st_cols = [&](int nt0) {
#pragma unroll
        for (int r = 0; r < kMfmaRN; ++r) {
            const uint32_t colc = (uint32_t)min((nt0 + r) * 32 + cl, p.FS - 1);
            const uint32_t f = colc / (uint32_t)p.S;
            col4[r] = colc * 4u; fw4[r] = f * (uint32_t)(CT * 4);
        }
    };
    auto st_meta = [&](int u) {                                 // u = (m * 4 + j) * HQ + half
        const int mj = u / HQ, half = u % HQ, m = mj >> 2, j = mj & 3;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = 4 * j + half * G + i;
            mdq[i] = meta[ent_slot(e) * TPB + ent_tuple(m, e)];
        }
    };
    auto st_load = [&](int q) {
        const int r = q % kMfmaRN;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const uint32_t wo = mdq[i].woff + fw4[r];
            if constexpr (CT == 2) {
                const u32x2_t v2 = __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, (int)wo, 0, 0);
                // (__uint_as_float of a copy: __builtin_bit_cast on a vector ELEMENT lvalue reads element 0 whatever the
                //  element -- clang 22 / ROCm 7.2; found with the debug dump of this kernel)
                const uint32_t e0 = v2.x, e1 = v2.y;
                wrq[q & 1][i][0] = __uint_as_float(e0); wrq[q & 1][i][1] = __uint_as_float(e1);
            } else if constexpr (CT == 4) {
                const u32x4_t v4 = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)wo, 0, 0);
                const uint32_t e0 = v4.x, e1 = v4.y, e2 = v4.z, e3 = v4.w;
                wrq[q & 1][i][0] = __uint_as_float(e0); wrq[q & 1][i][1] = __uint_as_float(e1);
                wrq[q & 1][i][2] = __uint_as_float(e2); wrq[q & 1][i][3] = __uint_as_float(e3);
            } else {
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    wrq[q & 1][i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w_rsrc, (int)(wo + 4u * c), 0, 0));
            }
#pragma unroll
            for (int c = 0; c < CT; ++c)
                prq[q & 1][i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr_rsrc, (int)(mdq[i].goff[c] + col4[r]), 0, 0));
        }
    };
    auto st_comp = [&](int q, const acc_t (&acc)[MT][kMfmaRN]) __attribute__((always_inline)) {
        const int r = q % kMfmaRN, u = q / kMfmaRN, mj = u / HQ, half = u % HQ, m = mj >> 2, j = mj & 3;
        double vv[G], lg[G], cntd[G];
        int cnt[G];
        bool special = false;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if constexpr (FP4) {
                const float c = acc[m][r][4 * j + half * G + i];          // an exact integer <= N
                cnt[i] = (int)c; cntd[i] = (double)c;
            } else {
                cnt[i] = acc[m][r][4 * j + half * G + i]; cntd[i] = (double)cnt[i];
            }
            // sum_c w_c * p_c in NumPy's order.  The product of two float32 values is exact in fp64, so fma(w, p, v)
            // rounds exactly like the reference's multiply-then-add: the same bits as the other kernel forms
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const double wc = (double)wrq[q & 1][i][c], pc = (double)prq[q & 1][i][c];
                v = c == 0 ? wc * pc : fma(wc, pc, v);
            }
            vv[i] = v;
            special |= __builtin_amdgcn_class(v, 0x2FF);            // anything but a positive normal double
        }
        int kx[G];
        tab_log4_n<G>(vv, lg, kx, tab_off, one_hi);
        // Rare: a table entry that is not a positive normal number -- the zero probability of an inapplicable state,
        // which no observation falls on (contributes nothing, whatever it is), or of an observed one (log 0 = -inf,
        // like the reference), or corrupt input (library log).
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (__builtin_amdgcn_class(vv[i], 0x2FF)) { lg[i] = cnt[i] != 0 ? lib_log(vv[i]) : 0.0; kx[i] = kFineLogBias; }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = ent_k(4 * j + half * G + i);
            lsum[k] = fma(cntd[i], lg[i], lsum[k]);
            ksum[k] = __mul24(cnt[i], kx[i]) + ksum[k];            // (biased exponents; host: passes * columns per lane * N * 2100 < 2^31)
            asm volatile("" : "+v"(ksum[k]));
            // (pins the sum in this step's block: the rare-path branch above splits the epilogue into basic blocks and
            //  the compiler otherwise sinks the whole chain of sums to the last one, keeping every log alive: 150 spills)
            asm volatile("" : "+v"(lsum[k]));
        }
    };

    for (int nt0 = nt_lo + w * kMfmaRN; nt0 < nt_hi; nt0 += kMfmaWaves * kMfmaRN) {
        int toff[kMfmaRN];
#pragma unroll
        for (int r = 0; r < kMfmaRN; ++r) toff[r] = tile_off(nt0 + r);
        const bool first_pass = nt0 == nt_lo + w * kMfmaRN;
        if (!first_pass) {
#pragma unroll
            for (int i = 0; i < PF; ++i)
#pragma unroll
                for (int r = 0; r < kMfmaRN; ++r) bq[i][r] = load_b(toff[r], i);
        }
        // the column's object count over all tuples (data only: every object is in exactly one tuple of every slot), for the
        // exponent bias; tiles beyond the split read the zero row behind the array
        int ccnt[kMfmaRN];
#pragma unroll
        for (int r = 0; r < kMfmaRN; ++r) ccnt[r] = p.colcount[(nt0 + r < nt_hi ? nt0 + r : p.NT) * 32 + cl];
        acc_t acc[MT][kMfmaRN];
        counts_pass(acc, toff);
#pragma unroll
        for (int r = 0; r < kMfmaRN; ++r) csum += ccnt[r];
        st_cols(nt0);
        // the single metadata buffer is refilled as soon as the loads of its last step (r = RN - 1) are out
        st_meta(0);
        st_load(0);
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            if (q + 1 < NST) {
                st_load(q + 1);
                if ((q + 2) % kMfmaRN == 0 && q + 2 < NST) st_meta((q + 2) / kMfmaRN);
            }
            st_comp(q, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- phase 2: fixed-order reduction: 32 columns of a lane half, then the lane halves and the 8 waves --------------------
    // red[wave][lane half][16 slots].  SL = 16: a slot's tuples all sit in ONE lane half (bit 2 of the slot = h), whose lanes saw
    // every object of their columns: the exponent bias leaves per lane, and the other half's entry of the slot is a zero.  Wide
    // forms: a lane half sees the tuples of half the rows, so the two halves' exponent sums are added as integers first and the
    // bias leaves once, in half 0.
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        double v;
        if constexpr (SL == 16) {
            // (slots beyond the batch counted nothing: their sums are discarded below, whatever the bias makes of them)
            v = fma((double)(ksum[i] - kFineLogBias * csum), 6.93147180559945286227e-01, lsum[i]);
        } else {
            const int ks = ksum[i] + __shfl_xor(ksum[i], 32, 64);
            v = h == 0 ? fma((double)(ks - kFineLogBias * csum), 6.93147180559945286227e-01, lsum[i]) : lsum[i];
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        lsum[i] = v;
    }
    if (cl == 0) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            if constexpr (SL == 16) {
                const int sl = 8 * (i >> 2) + 4 * h + (i & 3);
                red[(w * 2 + h) * 16 + sl] = lsum[i];
                red[(w * 2 + h) * 16 + (sl ^ 4)] = 0.0;             // the slots of the other lane half
            } else {
                red[(w * 2 + h) * 16 + i] = lsum[i];
            }
        }
    }
    __syncthreads();
    if (w != 0) return;                                                    // the rest is wave 0's (no block barrier below)
    const int my_slot = lane < SL ? slot_of(lane) : -1;
    double total = 0.0;
    if (my_slot >= 0) {
#pragma unroll
        for (int ww = 0; ww < kMfmaWaves; ++ww) { total += red[(ww * 2 + 0) * 16 + lane]; total += red[(ww * 2 + 1) * 16 + lane]; }
    }
    double* const my_partials = p.partials + (int64_t)max(my_slot, 0) * p.partials_stride;
    if (!p.results) {
        if (my_slot >= 0) my_partials[split] = total;
        return;
    }
    // MEMORY-MODEL NOTE: what follows is NOT a release / acquire pair of the HIP / LLVM memory model.  It rests on
    // gfx942 / gfx950 behaviour: an agent-scope atomic store is issued sc1 write-through and is acknowledged (vmcnt) only at
    // the coherence point shared by the XCDs; an agent-scope atomic load bypasses the reader XCD's L2.  So store -> s_waitcnt
    // vmcnt(0) -> ticket RMW -> (last block) loads is ordered in hardware although every access is "relaxed".  Another
    // target must not compile this (the #error at the top of the unit); SBE_REDUCE_IN_KERNEL=0 takes the fence-free
    // two-launch form, which tests/test_gpu_shapes.py::test_mfma_kernel_final_reduction_in_kernel keeps exercised.
    // The group's last block adds the partial sums (fixed order: run-to-run deterministic whichever block that is).  The
    // partial sums travel as agent-scope atomic stores / loads (write-through, coherent across the XCDs' L2s) ordered by
    // s_waitcnt around the ticket: a release FENCE at agent scope writes the whole L2 back -- every wave doing that cost 26 us
    // per launch, one wave per block still as much as the reduction kernel it replaces (measured)
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

size_t tuple_mfma_lds_bytes(int MT, int C, int KBp) {     // log table | A fragments | meta (32 MT entries whatever the form) | reduction
    const size_t meta = (size_t)32 * MT * (C <= 1 ? 8 : (C <= 3 ? 16 : 32));
    return (size_t)MT * KBp * 1024 + kFineLogEntries * 16 + meta + (size_t)kMfmaWaves * 2 * 16 * sizeof(double);
}

// entries per epilogue step: a whole register quad where the registers allow it, half a quad for the widest instances
template <int MT, int CT> constexpr int mfma_gt() { return (MT >= 4 || (MT == 3 && CT >= 3)) ? 2 : 4; }

template <int MT, bool FP4, int SLB>
static void launch_mfma_mt(int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_tuple_mfma<MT, 1, mfma_gt<MT, 1>(), FP4, SLB><<<grid, kMfmaThreads, lds, st>>>(p); break;
        case 2: k_mixture_tuple_mfma<MT, 2, mfma_gt<MT, 2>(), FP4, SLB><<<grid, kMfmaThreads, lds, st>>>(p); break;
        case 3: k_mixture_tuple_mfma<MT, 3, mfma_gt<MT, 3>(), FP4, SLB><<<grid, kMfmaThreads, lds, st>>>(p); break;
        default: k_mixture_tuple_mfma<MT, 4, mfma_gt<MT, 4>(), FP4, SLB><<<grid, kMfmaThreads, lds, st>>>(p); break;
    }
}

// one-time: the kernels ask for up to the whole 160 KB of a CU's LDS.  The log table's index is its whole LDS address
// (tab_off = 0), which holds only while the kernel has NO static LDS: checked here, once, on the host -- the guard inside the
// kernel would store NaNs but not take its completion tickets, and a host-synchronous caller would wait for them.
template <int MT, int CT, bool FP4, int SLB>
static bool allow_lds() {
    const void* fn = reinterpret_cast<const void*>(&k_mixture_tuple_mfma<MT, CT, mfma_gt<MT, CT>(), FP4, SLB>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncAttributes attr{};
    if (hipFuncGetAttributes(&attr, fn) != hipSuccess) { (void)hipGetLastError(); return true; }   // (not answerable: the kernel's own guard stays)
    return attr.sharedSizeBytes == 0;
}

template <bool FP4, int SLB>
static bool allow_lds_all() {
    bool ok = true;
    ok &= allow_lds<1, 1, FP4, SLB>(); ok &= allow_lds<1, 2, FP4, SLB>(); ok &= allow_lds<1, 3, FP4, SLB>(); ok &= allow_lds<1, 4, FP4, SLB>();
    ok &= allow_lds<2, 1, FP4, SLB>(); ok &= allow_lds<2, 2, FP4, SLB>(); ok &= allow_lds<2, 3, FP4, SLB>(); ok &= allow_lds<2, 4, FP4, SLB>();
    ok &= allow_lds<3, 1, FP4, SLB>(); ok &= allow_lds<3, 2, FP4, SLB>(); ok &= allow_lds<3, 3, FP4, SLB>(); ok &= allow_lds<3, 4, FP4, SLB>();
    ok &= allow_lds<4, 1, FP4, SLB>(); ok &= allow_lds<4, 2, FP4, SLB>(); ok &= allow_lds<4, 3, FP4, SLB>(); ok &= allow_lds<4, 4, FP4, SLB>();
    return ok;
}

// operand format of the count contraction: FP4 (default) or i8 (SBE_MFMA_FP4=0: same-box comparisons); fixed for the process
bool tuple_mfma_fp4() {
    static const bool on = [] { const char* v = getenv("SBE_MFMA_FP4"); return !(v && atoi(v) == 0); }();
    return on;
}

// slots per block for KT tuples (0: the form does not apply): 16 up to 8 tuples, 4 up to 32, 2 up to 64 -- the wide forms with FP4
// operands only
int tuple_mfma_slots_per_block(int KT) {
    if (KT < 1 || KT > 64) return 0;
    if (KT <= 8) return 16;
    if (!tuple_mfma_fp4()) return 0;
    return KT <= 32 ? 4 : 2;
}

template <bool FP4, int SLB>
static void launch_mfma_slb(int MT, int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (MT) {
        case 1: launch_mfma_mt<1, FP4, SLB>(C, p, grid, lds, st); break;
        case 2: launch_mfma_mt<2, FP4, SLB>(C, p, grid, lds, st); break;
        case 3: launch_mfma_mt<3, FP4, SLB>(C, p, grid, lds, st); break;
        default: launch_mfma_mt<4, FP4, SLB>(C, p, grid, lds, st); break;
    }
}

bool launch_tuple_mfma(int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st) {
    const bool fp4 = tuple_mfma_fp4();
    static const bool no_static_lds = tuple_mfma_fp4() ? (allow_lds_all<true, 4>() & allow_lds_all<true, 2>() & allow_lds_all<true, 1>())
                                                       : allow_lds_all<false, 4>();
    if (!no_static_lds) return false;
    const int SL = p.SL;                                 // (chosen by the host: mfma_geometry)
    const int MT = (p.KT + 32 / SL - 1) / (32 / SL);
    if (!fp4) launch_mfma_slb<false, 4>(MT, C, p, grid, lds, st);
    else if (SL == 16) launch_mfma_slb<true, 4>(MT, C, p, grid, lds, st);
    else if (SL == 4) launch_mfma_slb<true, 2>(MT, C, p, grid, lds, st);
    else launch_mfma_slb<true, 1>(MT, C, p, grid, lds, st);
    return true;
}

}  // namespace sbe
