// sbe_kernels_operators.hip.h -- device code of the operator-side evaluations (SURVEY.md 8(f) rank 1 and friends): cluster-membership
// marginals (first form and the wave-specialised one-launch form), ClusterJump's scores, GibbsSampleWeights' per-feature source
// likelihood.  Included through sbe_kernels.hip.h (which holds the shared device routines these kernels call).
#pragma once
#include "sbe_kernels.hip.h"

namespace sbe {

// ==========================================================================================
// cluster-membership marginals
//   (operators.py:1035-1095 AlterCluster.compute_cluster_posterior / compute_feature_weights_
//    with_and_without, operators.py:1420-1472 AlterClusterWide.compute_raw_cluster_probs)
// For every available object n and z in {0 = outside, 1 = inside the cluster}:
//     log m_z(n) = sum_f log sum_c lh_c(n,f) * w_z(n)[f][c]
// with lh_0 taken from the candidate cluster table, lh_{c>=1} from the slot's tables, NA -> 1.
// The reference multiplies in linear space (np.prod) and underflows beyond F ~ 300 (SURVEY.md
// H5); the sums of logs here do not, and exp() of them reproduces the reference's products to
// ~1e-14 relative where those exist.
// ==========================================================================================
// per-pattern weight tables of compute_feature_weights_with_and_without (operators.py:1075-1095):
//   wcur  = normalize( normalize_weights(weights, pattern) ** (1/T_prior) )
//   wflip = normalize_weights( weights ** (1/T_prior), pattern with the cluster bit flipped )
// float32 throughout, NumPy reduction order; x ** 1.0 is exact, other exponents go through powf.
// One (pattern, feature) row of those tables: wc[c] = wcur, wf[c] = wflip.
__device__ __forceinline__ void weight_tables_z_row(const float* __restrict__ w, uint32_t bits, int C, float inv_tp,
                                                    int use_pow, float* wc, float* wf) {
    const uint32_t fbits = bits ^ 1u;
    auto masked = [&](int c) -> float { return ((bits >> c) & 1u) ? w[c] : 0.0f * w[c]; };
    const float tot = np_pairwise_sum<float>(masked, C);
    auto powd = [&](int c) -> float { const float a = masked(c) / tot; return use_pow ? powf(a, inv_tp) : a; };
    const float tot2 = np_pairwise_sum<float>(powd, C);
    auto fl = [&](int c) -> float {
        const float pw = use_pow ? powf(w[c], inv_tp) : w[c];
        return ((fbits >> c) & 1u) ? pw : 0.0f * pw;
    };
    const float tot3 = np_pairwise_sum<float>(fl, C);
    for (int c = 0; c < C; ++c) { wc[c] = powd(c) / tot2; wf[c] = fl(c) / tot3; }
}

// Tables a resident operator call builds for itself (sbe_cluster_posterior_marginals, sbe_jump_lh_resident; one launch per
// call): k_cluster_marginals_ws / k_jump_lh_ws below.  A block is 16 waves with two jobs.  Waves
// 0..3 each take one object and run its dependent load chain -- object id over PCIe, group ids and pattern, state bytes,
// the other components' table entries, the weight rows -- keeping what they loaded in registers.  Waves 4..15 meanwhile
// build the call's tables into LDS, thread <-> table row (probs_row: k_probs' arithmetic, the same bits, the same data
// checks; block 0 reports them).  One barrier joins the two, then the object waves look their entries up in LDS and
// finish.  The table build runs UNDER the load chain instead of in front of it, and nothing crosses a block.
// Measured forms that lost (profiles/r4/fused_operator_forms.log): every thread building the entry it reads (587
// redundant builds: 23 us); one table per block with a wave per object, build first (a serial chain: 25 us); builder
// BLOCKS with a device-wide flag the other blocks wait on (agent-scope acquire / release across the eight XCD L2s:
// 21 us at 8 objects, 45 us at 587); the table kernel in front costs 6 us + a launch gap on top of the 9.5 us consumer.
struct InlineTables {
    RowSource row[2];                // candidate cluster (marginals) / source and target cluster (jump)
    const int32_t* counts;           // the slot's whole [Gtot][F][S] count table (jump: confounder rows)
    const double* conc;              // [Gtot][F][S]
    const double* unif;              // [F][S] the cluster prior's uniform concentration
    double temperature, prior_temperature;
    int* status;
    int n_rows;                      // table rows = (1 or 2 + confounder groups) * F
    int first_conf_group;            // jump: rows 2 F.. are the groups first_conf_group.. in order
};
constexpr int kWsObjWaves = 4, kWsWaves = 16, kWsBlock = kWsWaves * kWave, kWsC = 4;   // (object waves; components held in registers)
static_assert(kLogTabEntries == 2 * kWave, "the first builder wave copies the log table with two loads per lane");

// Builder waves of a wave-specialised operator kernel: rows (wave - kWsObjWaves) * 64 + lane, + 768, ... into `built` (LDS).
__device__ __forceinline__ void ws_build_rows(const InlineTables& tin, float* __restrict__ built, int F, int S) {
    for (int t = (int)threadIdx.x - kWsObjWaves * kWave; t < tin.n_rows; t += kWsBlock - kWsObjWaves * kWave) {
        const int r = t / F, f = t % F;
        const int32_t* cnt;
        const double* conc;
        if (r < 2 && tin.row[r].counts) { cnt = tin.row[r].counts + (int64_t)f * S; conc = tin.row[r].conc + (int64_t)f * S; }
        else {
            const int64_t ro = ((int64_t)(tin.first_conf_group + r - 2) * F + f) * S;
            cnt = tin.counts + ro; conc = tin.conc + ro;
        }
        float* row = built + (int64_t)t * S;
        probs_row([&](int s) { return (float)cnt[s]; }, conc, tin.unif + (int64_t)f * S, S, tin.temperature, tin.prior_temperature,
                  blockIdx.x == 0 ? tin.status : nullptr, [&](int s, float v) { row[s] = v; });
    }
}

// The same row computed in REGISTERS (C <= CM < 8: NumPy's sum of fewer than eight terms is the plain left-to-right chain):
// the C weights are loaded once, together, and every intermediate is a statically indexed register.  The general form
// above reads w[c] from memory inside loops with a run-time trip count -- a dozen loads, each waited for before the
// next -- which was 11 of the 19 us of the fused marginals kernel (profiles/r4/ws_kernel_clock.log).  Same arithmetic.
template <int CM>
__device__ __forceinline__ void weight_tables_z_row_reg(const float* __restrict__ w, uint32_t bits, int C, float inv_tp, int use_pow,
                                                        float (&wc)[CM], float (&wf)[CM]) {
    static_assert(CM <= 8, "one NumPy leaf of at most eight terms (np_sum_regs: the plain chain below eight, the tree at eight)");
    const uint32_t fbits = bits ^ 1u;
    float wr[CM], m[8], pd[8], fl[8];
#pragma unroll
    for (int c = 0; c < CM; ++c) wr[c] = c < C ? w[c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float v = wr[c < CM ? c : 0];                          // (c >= CM: padding of the eight-wide sum, never counted)
        m[c] = c < CM ? (((bits >> c) & 1u) ? v : 0.0f * v) : 0.0f;
    }
    const float tot = np_sum_regs<float, 8>(m, C);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float a = m[c] / tot;
        pd[c] = c < CM ? (use_pow ? lib_powf(a, inv_tp) : a) : 0.0f;
    }
    const float tot2 = np_sum_regs<float, 8>(pd, C);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float base = c < CM ? wr[c < CM ? c : 0] : 0.0f;
        const float pw = c < CM ? (use_pow ? lib_powf(base, inv_tp) : base) : 0.0f;
        fl[c] = ((fbits >> c) & 1u) ? pw : 0.0f * pw;
    }
    const float tot3 = np_sum_regs<float, 8>(fl, C);
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        wc[c] = c < C ? pd[c] / tot2 : 0.0f;
        wf[c] = c < C ? fl[c] / tot3 : 0.0f;
    }
}

// One BLOCK per available object, thread <-> feature: the with/without weight row of the object's pattern is
// computed in place (no table kernel in front), a lane's loads (state byte, weights, table entries of every
// component) are all in flight together and there is one dependent-load chain per object instead of one per
// 64 features; the two fp64 logs per observation are table-driven (tab_log_pos; the sums of logs carry far more
// accuracy than the reference's linear-space products).  Fixed-order block reduction: deterministic.
static __global__ __launch_bounds__(kBlock) void k_cluster_marginals(
    const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid,
    const float* __restrict__ probs, const float* __restrict__ table0, const float* __restrict__ weights,
    const uint32_t* __restrict__ pattern_bits, float inv_tp, int use_pow, const int32_t* __restrict__ objects,
    int n_av, double* __restrict__ out, const f64x2_t* __restrict__ logtab, int Np, int F, int S, int C, int Fp,
    DoneSig done = DoneSig{}) {
    __shared__ f64x2_t tab[kLogTabEntries];
    __shared__ double red[8];
    if (threadIdx.x < kLogTabEntries) tab[threadIdx.x] = logtab[threadIdx.x];
    __syncthreads();
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f64x2_t*)tab;
    const int i = blockIdx.x;
    const int n = objects[i];
    const bool inside = gid[n] != kNoGroup;                                  // component 0 = clusters
    const uint32_t bits = pattern_bits[pid[n]];
    double acc0 = 0.0, acc1 = 0.0;
    for (int f = threadIdx.x; f < F; f += kBlock) {
        const uint8_t x = state[(int64_t)n * Fp + f];
        float wc[kMaxComponents], wf[kMaxComponents], lhc[kMaxComponents];
        weight_tables_z_row_reg<kMaxComponents>(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, wc, wf);   // this object's pattern
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) {                           // (every entry asked for before any is used)
            lhc[c] = 1.0f;
            if (c < C && x != kNA) {
                if (c == 0) lhc[c] = table0[(int64_t)f * S + x];
                else {
                    const uint16_t gg = gid[(int64_t)c * Np + n];
                    lhc[c] = gg == kNoGroup ? 0.0f : probs[((int64_t)gg * F + f) * S + x];
                }
            }
        }
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) {
            if (c < C) {
                const double lh = (double)lhc[c], a = (double)wc[c], b = (double)wf[c];
                v1 = v1 + lh * (inside ? a : b);     // z = 1: the object is (or becomes) a cluster member
                v0 = v0 + lh * (inside ? b : a);
            }
        }
        acc0 += tab_log_pos(v0, tab_addr);
        acc1 += tab_log_pos(v1, tab_addr);
    }
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid] = acc0; red[4 + wid] = acc1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[i] = (red[0] + red[1]) + (red[2] + red[3]);
        out[(int64_t)n_av + i] = (red[4] + red[5]) + (red[6] + red[7]);
    }
    signal_done(done);
}

// The same marginals with the candidate table built by the block's builder waves (InlineTables above;
// sbe_cluster_posterior_marginals).  An object wave holds, per lane, the four features the four waves of the block form
// above would have given that lane (w * 64 + lane, + 256 k beyond), keeps the block form's four accumulators and reduces
// them by the same wave tree in the same order: the result is that form's bit for bit.  C <= kWsC.
static __global__ __launch_bounds__(kWsBlock) void k_cluster_marginals_ws(
    const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid,
    const float* __restrict__ probs, const float* __restrict__ weights, const uint32_t* __restrict__ pattern_bits, float inv_tp,
    int use_pow, const int32_t* __restrict__ objects, int n_av, double* __restrict__ out, const f64x2_t* __restrict__ logtab,
    int Np, int F, int S, int C, int Fp, DoneSig done, InlineTables tin) {
    __shared__ f64x2_t tab[kLogTabEntries];
    extern __shared__ float cand[];                                          // the candidate table [F][S]
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    const int i = blockIdx.x * kWsObjWaves + wid;
    const bool object_wave = wid < kWsObjWaves, active = object_wave && i < n_av;
    uint8_t x[4];
    float lhc[4][kWsC], wc[4][kWsC], wf[4][kWsC];
    int n = 0;
    bool inside = false;
    uint32_t bits = 0;
    uint16_t gg[kWsC];
    // one (object, feature): the other components' entries and the weight rows, into registers
    auto fetch = [&](int f, uint8_t xx, float (&l)[kWsC], float (&a)[kWsC], float (&b)[kWsC]) {
        weight_tables_z_row_reg<kWsC>(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, a, b);   // this object's pattern
#pragma unroll
        for (int c = 0; c < kWsC; ++c) {
            l[c] = 1.0f;
            if (c >= 1 && c < C && xx != kNA) l[c] = gg[c] == kNoGroup ? 0.0f : probs[((int64_t)gg[c] * F + f) * S + xx];
        }
    };
    if (!object_wave) {
        if (wid == kWsObjWaves) { tab[lane] = logtab[lane]; tab[lane + kWave] = logtab[lane + kWave]; }     // (kLogTabEntries = 128)
        ws_build_rows(tin, cand, F, S);
    } else if (active) {
        n = objects[i];
        inside = gid[n] != kNoGroup;                                         // component 0 = clusters
        bits = pattern_bits[pid[n]];
#pragma unroll
        for (int c = 0; c < kWsC; ++c) gg[c] = (c >= 1 && c < C) ? gid[(int64_t)c * Np + n] : kNoGroup;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int f = w * kWave + lane;
            x[w] = f < F ? state[(int64_t)n * Fp + f] : kNA;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int f = w * kWave + lane;
            if (f < F) fetch(f, x[w], lhc[w], wc[w], wf[w]);
        }
    }
    __syncthreads();
    if (active) {
        const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f64x2_t*)tab;
        auto term = [&](const float (&l)[kWsC], const float (&a)[kWsC], const float (&b)[kWsC], double& acc0, double& acc1) {
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int c = 0; c < kWsC; ++c) {
                if (c < C) {
                    const double lh = (double)l[c], wa = (double)a[c], wb = (double)b[c];
                    v1 = v1 + lh * (inside ? wa : wb);   // z = 1: the object is (or becomes) a cluster member
                    v0 = v0 + lh * (inside ? wb : wa);
                }
            }
            acc0 += tab_log_pos(v0, tab_addr);
            acc1 += tab_log_pos(v1, tab_addr);
        };
        double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int f = w * kWave + lane;
            if (f < F) {
                lhc[w][0] = x[w] != kNA ? cand[(int64_t)f * S + x[w]] : 1.0f;
                term(lhc[w], wc[w], wf[w], a0[w], a1[w]);
            }
        }
        for (int fb = kBlock; fb < F; fb += kBlock) {                        // (F > 256: the block form's later passes, in order)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int f = fb + w * kWave + lane;
                if (f < F) {
                    const uint8_t xx = state[(int64_t)n * Fp + f];
                    float l[kWsC], a[kWsC], b[kWsC];
                    fetch(f, xx, l, a, b);
                    l[0] = xx != kNA ? cand[(int64_t)f * S + xx] : 1.0f;
                    term(l, a, b, a0[w], a1[w]);
                }
            }
        }
        double r0[4], r1[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { r0[w] = wave_sum(a0[w]); r1[w] = wave_sum(a1[w]); }
        if (lane == 0) {
            out[i] = (r0[0] + r0[1]) + (r0[2] + r0[3]);
            out[(int64_t)n_av + i] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        }
    }
    signal_done(done);
}

// ------------------------------------------------------------------------------------------
// ClusterJump.get_jump_lh (sbayes/sampling/operators.py:1679-1722) with
// ClusterEffectProposals.expected_confounder_features (:1342-1379): for every member n of the source cluster
//     p_conf(n,f)  = sum_{c>=1, n in group g of c} wh(n)[f][c] * pconf[g][f][x]     float32, components in order
//     stay(n,f)    = p_conf + wh(n)[f][0] * p_source[f][x]                          float32 (mul, then add)
//     jump(n,f)    = p_conf + wh(n)[f][0] * p_target[f][x]
//     out[0][i]    = sum_{f not NA} log stay,   out[1][i] = sum_{f not NA} log jump  (fp64 logs and sums)
// wh = normalize(update_weights(sample) ** (1/T_prior)) (the `wcur` of weight_tables_z_row), x = the observed state.
// The float32 arithmetic is the reference's, operation for operation; only the product over features -- np.prod in
// float32 there, which underflows to 0 beyond F ~ 75 -- is replaced by a sum of logs.  `pconf` are the
// caller's tempered tables of every confounder group (global group index minus the cluster count), `p_source` /
// `p_target` the two clusters' tempered tables (conditional_effect_mean, conditionals.py:105-122).
// One block per member, thread <-> feature, fixed-order reduction (as k_cluster_marginals).
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kBlock) void k_jump_lh(
    const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid,
    const float* __restrict__ pconf /* [Gtot - G0][F][S] */, const float* __restrict__ p_source,
    const float* __restrict__ p_target, const float* __restrict__ weights, const uint32_t* __restrict__ pattern_bits,
    float inv_tp, int use_pow, const int32_t* __restrict__ objects, int n_members, double* __restrict__ out,
    const f64x2_t* __restrict__ logtab, int Np, int F, int S, int C, int Fp, int G0,
    DoneSig done = DoneSig{}) {
    __shared__ f64x2_t tab[kLogTabEntries];
    __shared__ double red[8];
    if (threadIdx.x < kLogTabEntries) tab[threadIdx.x] = logtab[threadIdx.x];
    __syncthreads();
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f64x2_t*)tab;
    const int i = blockIdx.x;
    const int n = objects[i];
    const uint32_t bits = pattern_bits[pid[n]];
    double acc0 = 0.0, acc1 = 0.0;
    for (int f = threadIdx.x; f < F; f += kBlock) {
        const uint8_t x = state[(int64_t)n * Fp + f];
        if (x == kNA) continue;                                       // np.prod(..., where=~NAs): factor 1
        float wc[kMaxComponents], wf[kMaxComponents], pcf[kMaxComponents];
        uint16_t gg[kMaxComponents];
        weight_tables_z_row_reg<kMaxComponents>(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, wc, wf);   // wc = weights_heated row
#pragma unroll
        for (int c = 1; c < kMaxComponents; ++c) gg[c] = c < C ? gid[(int64_t)c * Np + n] : kNoGroup;
#pragma unroll
        for (int c = 1; c < kMaxComponents; ++c)                              // (every entry asked for before any is used)
            pcf[c] = gg[c] != kNoGroup ? pconf[((int64_t)(gg[c] - G0) * F + f) * S + x] : 0.0f;
        const float e_src = p_source[(int64_t)f * S + x], e_tgt = p_target[(int64_t)f * S + x];
        float pc = 0.0f;
#pragma unroll
        for (int c = 1; c < kMaxComponents; ++c)
            if (gg[c] != kNoGroup) pc = pc + wc[c] * pcf[c];
        const float ps = pc + wc[0] * e_src;
        const float pt = pc + wc[0] * e_tgt;
        acc0 += tab_log_pos((double)ps, tab_addr);
        acc1 += tab_log_pos((double)pt, tab_addr);
    }
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid] = acc0; red[4 + wid] = acc1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[i] = (red[0] + red[1]) + (red[2] + red[3]);
        out[(int64_t)n_members + i] = (red[4] + red[5]) + (red[6] + red[7]);
    }
    signal_done(done);
}

// The same scores with the tempered tables of the two clusters and of every confounder group built by the block's builder
// waves from the slot's resident counts (InlineTables: rows 0..F-1 source, F..2F-1 target, then the confounder groups in
// order; sbe_jump_lh_resident).  Object waves as in k_cluster_marginals_ws: the block form's bits.  C <= kWsC.
static __global__ __launch_bounds__(kWsBlock) void k_jump_lh_ws(
    const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid,
    const float* __restrict__ weights, const uint32_t* __restrict__ pattern_bits, float inv_tp, int use_pow,
    const int32_t* __restrict__ objects, int n_members, double* __restrict__ out, const f64x2_t* __restrict__ logtab, int Np, int F,
    int S, int C, int Fp, int G0, DoneSig done, InlineTables tin) {
    __shared__ f64x2_t tab[kLogTabEntries];
    extern __shared__ float built[];                                         // source | target | confounder tables
    const int64_t fs = (int64_t)F * S;
    const float* ps = built;
    const float* pt = built + fs;
    const float* pc = built + 2 * fs;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    const int i = blockIdx.x * kWsObjWaves + wid;
    const bool object_wave = wid < kWsObjWaves, active = object_wave && i < n_members;
    uint8_t x[4];
    float wc[4][kWsC];
    uint16_t gg[kWsC];
    int n = 0;
    uint32_t bits = 0;
    if (!object_wave) {
        if (wid == kWsObjWaves) { tab[lane] = logtab[lane]; tab[lane + kWave] = logtab[lane + kWave]; }     // (kLogTabEntries = 128)
        ws_build_rows(tin, built, F, S);
    } else if (active) {
        n = objects[i];
        bits = pattern_bits[pid[n]];
#pragma unroll
        for (int c = 0; c < kWsC; ++c) gg[c] = (c >= 1 && c < C) ? gid[(int64_t)c * Np + n] : kNoGroup;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int f = w * kWave + lane;
            x[w] = f < F ? state[(int64_t)n * Fp + f] : kNA;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int f = w * kWave + lane;
            float unused[kWsC];
            if (f < F && x[w] != kNA) weight_tables_z_row_reg<kWsC>(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, wc[w], unused);
        }
    }
    __syncthreads();
    if (active) {
        const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f64x2_t*)tab;
        auto term = [&](uint8_t xx, int f, const float (&a)[kWsC], double& acc0, double& acc1) {
            float p = 0.0f;
#pragma unroll
            for (int c = 1; c < kWsC; ++c)
                if (c < C && gg[c] != kNoGroup) p = p + a[c] * pc[((int64_t)(gg[c] - G0) * F + f) * S + xx];
            const float s_ = p + a[0] * ps[(int64_t)f * S + xx];
            const float t_ = p + a[0] * pt[(int64_t)f * S + xx];
            acc0 += tab_log_pos((double)s_, tab_addr);
            acc1 += tab_log_pos((double)t_, tab_addr);
        };
        double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int f = w * kWave + lane;
            if (f < F && x[w] != kNA) term(x[w], f, wc[w], a0[w], a1[w]);    // np.prod(..., where=~NAs): factor 1
        }
        for (int fb = kBlock; fb < F; fb += kBlock) {                        // (F > 256: the block form's later passes, in order)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int f = fb + w * kWave + lane;
                if (f < F) {
                    const uint8_t xx = state[(int64_t)n * Fp + f];
                    if (xx != kNA) {
                        float a[kWsC], unused[kWsC];
                        weight_tables_z_row_reg<kWsC>(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, a, unused);
                        term(xx, f, a, a0[w], a1[w]);
                    }
                }
            }
        }
        double r0[4], r1[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { r0[w] = wave_sum(a0[w]); r1[w] = wave_sum(a1[w]); }
        if (lane == 0) {
            out[i] = (r0[0] + r0[1]) + (r0[2] + r0[3]);
            out[(int64_t)n_members + i] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        }
    }
    signal_done(done);
}

// ------------------------------------------------------------------------------------------
// GibbsSampleWeights.source_lh_by_feature (sbayes/sampling/operators.py:677-685): per feature
//     out[f] = float32( sum_n log p(n,f) ),   p = sum_c source[n,f,c] * w[n,f,c]  (one-hot source: the weight of the
//     observation's source component; 0 if none is set), p = 1 for NA observations
// from the slot's resident source, patterns and normalised weights: the [N, F, C] weight array the reference
// materialises (twice per call of the operator) never exists.  float32 logs like the reference's.  NumPy's own float32
// log is not bit-reproducible here, so the result is compared at float32 accuracy (tests/_call_log.py:COMPARE), and the
// sum over the objects is taken where it is cheap AND no less accurate than the reference's: the reference adds the N
// float32 logs in float32, object after object (np.sum(axis=0) of a C-ordered float32 [N, F] array: a serial chain
// with up to N/2 ulp of accumulated rounding); here every lane adds its objects' logs in float64, the 64 lanes of a
// feature are combined by a fixed tree, and the total is rounded to float32 once.  (Round 3, first form: the serial
// float32 chain itself, 18 us per call at N = 1000 whatever the tiling -- a thousand dependent adds per feature.)
// Block = 16 features x 64 object lanes; loads unconditional and grouped by level so that they are in flight together.
// ------------------------------------------------------------------------------------------
constexpr int kSlfFT = 16;
static __global__ __launch_bounds__(1024) void k_source_lh_by_feature(
    const uint8_t* __restrict__ state, const uint8_t* __restrict__ src, const uint8_t* __restrict__ pid,
    const float* __restrict__ wpat, float* __restrict__ out, int N, int F, int C, int Fp, DoneSig done = DoneSig{}) {
    constexpr int OL = 1024 / kSlfFT, PER = 8;
    __shared__ double part[OL][kSlfFT];                    // 8 KB
    const int fl = threadIdx.x & (kSlfFT - 1), ol = threadIdx.x / kSlfFT;
    const int f = blockIdx.x * kSlfFT + fl;
    const int fc = min(f, F - 1);                          // (clamped: every load below is unconditional)
    double acc = 0.0;
    for (int n0 = 0; n0 < N; n0 += OL * PER) {
        float w[PER];
        uint8_t x[PER], sc[PER], pp[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int n = min(n0 + ol + OL * j, N - 1);
            x[j] = state[(int64_t)n * Fp + fc];
            sc[j] = src[(int64_t)n * Fp + fc];
            pp[j] = pid[n];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) w[j] = wpat[((int64_t)pp[j] * F + fc) * C + min((int)sc[j], C - 1)];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int n = n0 + ol + OL * j;
            if (!(n < N && f < F) || x[j] == kNA) w[j] = 1.0f;          // NA (and padding): p = 1, log p = 0
            else if (sc[j] >= C) w[j] = 0.0f;                           // no source component set: log 0 = -inf
            acc += (double)logf(w[j]);
        }
    }
    part[ol][fl] = acc;
    __syncthreads();
    for (int half = OL / 2; half > 0; half >>= 1) {        // fixed tree over the object lanes
        if (ol < half) part[ol][fl] += part[ol + half][fl];
        __syncthreads();
    }
    if (ol == 0 && f < F) out[f] = (float)part[0][fl];
    signal_done(done);
}

}  // namespace sbe
