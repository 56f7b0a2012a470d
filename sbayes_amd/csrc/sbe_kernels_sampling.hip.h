// sbe_kernels_sampling.hip.h -- device code of the Gibbs source resampling family and of the one-call MCMC steps (SURVEY.md 8(f)
// ranks 3, 2): Philox, source posterior / draw / transition probability, the step core (counts delta, tables, collapsed
// likelihood), count deltas and row setters of the drop-in host layer, the fused kept-observations kernels.
// Included through sbe_kernels.hip.h.
#pragma once
#include "sbe_kernels.hip.h"

namespace sbe {

// ==========================================================================================
// data-parallel cores of Gibbs source resampling
// ==========================================================================================
// GibbsSampleSource.calculate_source_posterior (operators.py:554-574): for the listed objects
//   p[i][f][:] = normalize( lh[n_i][f][:] ** (1/T) * w[n_i][f][:] ** (1/T_prior) )  -> float32
// lh as in likelihood_per_component (NA -> 1, no group -> 0), w = normalised weights of the slot.
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based,
// so uniform i of draw d under seed k is a pure function philox((i, d), k) -- no RNG state in HBM, any
// grid shape gives the same numbers.  oracle/sbayes_oracle.py restates it (and its known-answer vectors).
__device__ __host__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53-bit uniform in [0, 1) from two words, like MT19937's genrand_res53 that np.random.random uses.
__device__ inline double philox_uniform(uint64_t seed, uint64_t draw, uint64_t i) {
    uint32_t r[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    return ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
}

static __global__ void k_test_philox(const uint32_t* __restrict__ ctr_key, int n, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* p = ctr_key + 6 * (int64_t)i;
    philox4x32_10(p[0], p[1], p[2], p[3], p[4], p[5], out + 4 * (int64_t)i);
}

struct SrcPostArgs {
    const uint8_t* state; const uint16_t* gid; const uint8_t* pid; const float* probs; const float* wpat;
    const int32_t* objects; int n_sub, Np, F, S, C, Fp;
    double inv_t; float inv_tp; int pow_lh, pow_w, from_prior;
};

// One observation's posterior row p[0..C) (float32).  from_prior (operators.py:520-522): p =
// normalize(w ** (1/T_prior)) entirely in float32, the likelihood plays no part.
// (Register form: the C weights, group ids and table entries are loaded up front, together, and every term is computed
// once -- the first form evaluated term(c) twice from memory inside run-time loops, a chain of waited-for loads.)
// Core: `group_of(c)` = the object's group of component c as the caller indexes its tables (kNoGroup: none),
// `table_at(c, g)` = that group's table entry for this feature and the observed state x.
template <class GroupOf, class TableAt>
__device__ __forceinline__ bool posterior_row_core(uint8_t x, const float* __restrict__ w, int C, bool from_prior, int pow_lh, double inv_t,
                                                   int pow_w, float inv_tp, GroupOf group_of, TableAt table_at, float* p) {
    constexpr int CM = kMaxComponents;
    float wr[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) wr[c] = c < C ? w[c] : 0.0f;
    if (from_prior) {
        float t[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) t[c] = pow_w ? lib_powf(wr[c], inv_tp) : wr[c];
        const float total = np_sum_regs<float, CM>(t, C);
#pragma unroll
        for (int c = 0; c < CM; ++c) if (c < C) p[c] = t[c] / total;
        return total > 0.0f;
    }
    uint32_t gg[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) gg[c] = (c < C && x != kNA) ? (uint32_t)group_of(c) : (uint32_t)kNoGroup;
    double t[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        double lh = 1.0;
        if (x != kNA) lh = gg[c] == (uint32_t)kNoGroup ? 0.0 : (double)table_at(c, gg[c]);
        if (pow_lh) lh = lib_pow(lh, inv_t);
        const float wc = pow_w ? lib_powf(wr[c], inv_tp) : wr[c];
        t[c] = lh * (double)wc;
    }
    const double total = np_sum_regs<double, CM>(t, C);
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) p[c] = (float)(t[c] / total);
    return total > 0.0;
}

__device__ inline bool source_posterior_row(const SrcPostArgs& a, int n, int f, float* p) {
    const uint8_t x = a.state[(int64_t)n * a.Fp + f];
    const float* w = a.wpat + ((int64_t)a.pid[n] * a.F + f) * a.C;
    return posterior_row_core(x, w, a.C, a.from_prior, a.pow_lh, a.inv_t, a.pow_w, a.inv_tp,
                              [&](int c) { return a.gid[(int64_t)c * a.Np + n]; },
                              [&](int c, uint32_t g) { return a.probs[((int64_t)g * a.F + f) * a.S + x]; }, p);
}

static __global__ void k_source_posterior(SrcPostArgs a, float* __restrict__ out, int* __restrict__ status, DoneSig done = DoneSig{}) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)a.n_sub * a.F) {
        const int r = (int)(i / a.F), f = (int)(i % a.F);
        float p[kMaxComponents];
        if (!source_posterior_row(a, a.objects[r], f, p)) raise_status(status, ST_BAD_NORMALIZE, 1);
        float* o = out + i * a.C;
        for (int c = 0; c < a.C; ++c) o[c] = p[c];
    }
    signal_done(done);
}

// GibbsSampleSource._propose (operators.py:495-552), the draw: sample_categorical
// (preprocessing.py:224-256) on the posterior row with the caller's uniform z[r][f] --
//   cdf = cumsum(p) (float32, sequential), cdf /= cdf[-1], k = first c with z < cdf[c] (0 if none)
// (z == nullptr: uniform i of the engine's Philox stream, SURVEY.md 8(f) rank 3 "device RNG")
// -- writes component k (0xFF for NA observations, operators.py:527) into the destination slot's
// source and keeps p[k] (1 for NA) for the transition log-probability log_q = sum log p[k].
// log_partials != nullptr (one-call Gibbs step): the block's sum of log(selected probability) goes to
// log_partials[blockIdx.x] (fixed order inside the block; the step epilogue adds the blocks in order) -- no separate
// k_sum_log_f32 launch.  Blocks are 256 threads.
__device__ __forceinline__ void block_log_partial(float sel, bool active, double* __restrict__ log_partials) {
    if (!log_partials) return;                                   // kernel-uniform
    __shared__ double red4[4];
    const double total = block_sum(active ? log((double)sel) : 0.0, red4);
    if (threadIdx.x == 0) log_partials[blockIdx.x] = total;
}

static __global__ __launch_bounds__(kBlock) void k_sample_source(SrcPostArgs a, const double* __restrict__ z, uint64_t seed,
                                                         uint64_t draw, uint8_t* __restrict__ src_dst,
                                                         float* __restrict__ p_sel, int* __restrict__ status,
                                                         double* __restrict__ log_partials) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < (int64_t)a.n_sub * a.F;
    float sel = 1.0f;
    if (active) {
        const int r = (int)(i / a.F), f = (int)(i % a.F);
        const int n = a.objects[r];
        float p[kMaxComponents];
        if (!source_posterior_row(a, n, f, p)) raise_status(status, ST_BAD_NORMALIZE, 1);
        float cdf[kMaxComponents];
        float run = p[0];
        cdf[0] = run;
        for (int c = 1; c < a.C; ++c) { run = run + p[c]; cdf[c] = run; }
        const float last = cdf[a.C - 1];
        const double zz = z ? z[i] : philox_uniform(seed, draw, (uint64_t)i);     // z == nullptr: the engine's own stream
        int k = 0;
        for (int c = a.C - 1; c >= 0; --c)
            if (zz < (double)(cdf[c] / last)) k = c;
        const bool na = a.state[(int64_t)n * a.Fp + f] == kNA;
        src_dst[(int64_t)n * a.Fp + f] = na ? (uint8_t)kNA : (uint8_t)k;
        for (int c = 0; c < a.C; ++c) sel = (!na && c == k) ? p[c] : sel;
        p_sel[i] = sel;
    }
    block_log_partial(sel, active, log_partials);        // (one convergent call: it contains a barrier)
}

// log_q_back (operators.py:544-550): p of `a`'s state evaluated at ANOTHER slot's source assignment.
static __global__ __launch_bounds__(kBlock) void k_source_logprob(SrcPostArgs a, const uint8_t* __restrict__ src,
                                                          float* __restrict__ p_sel, int* __restrict__ status,
                                                          double* __restrict__ log_partials) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < (int64_t)a.n_sub * a.F;
    float sel = 1.0f;
    if (active) {
        const int r = (int)(i / a.F), f = (int)(i % a.F);
        const int n = a.objects[r];
        float p[kMaxComponents];
        if (!source_posterior_row(a, n, f, p)) raise_status(status, ST_BAD_NORMALIZE, 1);
        const int id = src[(int64_t)n * a.Fp + f];
        for (int c = 0; c < a.C; ++c) sel = (c == id) ? p[c] : sel;
        p_sel[i] = sel;
    }
    block_log_partial(sel, active, log_partials);        // (one convergent call: it contains a barrier)
}

// partials[b] = sum of log(v[i]) over block b's grid-stride share (fp64 logs, fixed order);
// k_reduce_partials finishes.
static __global__ __launch_bounds__(kBlock) void k_sum_log_f32(const float* __restrict__ v, int64_t n,
                                                       double* __restrict__ partials) {
    __shared__ double red4[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        acc += log((double)v[i]);
    const double total = block_sum(acc, red4);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// component_likelihood_given_unchanged (operators.py:863-928), gather part: float32 likelihood of
// the listed objects' observations under caller-supplied tables (built from the observations that
// are not being resampled); group_idx[c][i] = table row of object i in component c (-1: none -> 0);
// NA -> 1; finally ** (1/T) in float32.
static __global__ void k_subset_lh(const uint8_t* __restrict__ state, const float* __restrict__ tables,
                            const int32_t* __restrict__ table_offsets, const int32_t* __restrict__ group_idx,
                            const int32_t* __restrict__ objects, int n_sub, float* __restrict__ out, int F, int S,
                            int C, int Fp, float inv_t, int use_pow, DoneSig done = DoneSig{}) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)n_sub * F) {
        const int r = (int)(i / F), f = (int)(i % F);
        const uint8_t x = state[(int64_t)objects[r] * Fp + f];
        float* o = out + i * C;
        for (int c = 0; c < C; ++c) {
            float v = 1.0f;
            if (x != kNA) {
                const int g = group_idx[(int64_t)c * n_sub + r];
                v = g < 0 ? 0.0f : tables[((int64_t)(table_offsets[c] + g) * F + f) * S + x];
            }
            o[c] = use_pow ? powf(v, inv_t) : v;
        }
    }
    signal_done(done);
}

// ClusterOperator.gibbs_sample_source (sbayes/sampling/operators.py:796-851), everything between the likelihood under the
// kept observations (k_unchanged_counts' tempered tables, gathered here like k_subset_lh does) and the sample bookkeeping:
//   lh[c]      = table value of the observation under component c, ** (1/T)                          float32
//   w_new[c]   = normalize_weights(weights, has_components_NEW[n])[f][c] ** (1/T_prior)              float32 (k_normalize_weight_rows' arithmetic)
//   p          = normalize(w_new * lh)          (sample_from_prior: p = w_new, NOT renormalised: operators.py:815-816)
//   k          = sample_categorical(p) with the caller's uniform z[r][f]   (k_sample_source's draw)
//   p_back     = normalize(w_old * lh) with has_components_OLD[n]          (operators.py:838-844)
// out: the drawn component (0xFF for an NA observation), p[k] (1 for NA) and p_back[old source component] (1 when the
// old source has none) -- the host sums their float32 logs the way the reference does.
struct GuGibbsArgs {
    const uint8_t* state; const float* tables; const int32_t* table_offsets; const int32_t* group_idx; const int32_t* objects;
    const float* weights;            // the slot's [F][C] float32 mixture weights
    const uint8_t* hc_new; const uint8_t* hc_old;   // [n_sub][C] has_components rows of the two samples
    const uint8_t* src_old;          // [n_sub][F] old source component per observation (0xFF: none)
    const double* z;                 // [n_sub][F] uniforms
    int n_sub, F, S, C, Fp;
    float inv_t, inv_tp; int pow_lh, pow_w, from_prior;
};

// One observation (subset row r, feature f; i = r * F + f) of the above.  `table_at(c, g)` = the kept-observations table
// entry of component c, group g (>= 0) for this feature and the observed state x; `group_of(c)` = the object's group in
// component c (-1: none).
// Register form: the has_components bytes (host-mapped: a PCIe read each), the weights and the table entries are loaded
// once, up front and together; every array is indexed statically (unrolled, guarded by c < C).
template <class GroupOf, class TableAt>
__device__ __forceinline__ int gu_gibbs_obs(const GuGibbsArgs& a, int64_t i, int r, int f, uint8_t x, double zz, int id_old,
                                             GroupOf group_of, TableAt table_at,
                                             uint8_t* __restrict__ src_new, float* __restrict__ sel_new, float* __restrict__ sel_back,
                                             int* __restrict__ status) {
    constexpr int CM = kMaxComponents;
    const bool na = x == kNA;
    const int C = a.C;
    const float* w = a.weights + (int64_t)f * C;
    const uint8_t* hcn = a.hc_new + (int64_t)r * C;
    const uint8_t* hco = a.hc_old + (int64_t)r * C;
    float wr[CM], lh[CM];
    uint8_t hn[CM], ho[CM];
    int g[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const bool on = c < C;
        wr[c] = on ? w[c] : 0.0f;
        hn[c] = on ? hcn[c] : (uint8_t)0;
        ho[c] = on ? hco[c] : (uint8_t)0;
        g[c] = (on && !na) ? group_of(c) : -1;
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        float v = 1.0f;
        if (!na) v = g[c] < 0 ? 0.0f : table_at(c, g[c]);
        lh[c] = a.pow_lh ? lib_powf(v, a.inv_t) : v;
    }
    float p[2][CM];
    bool ok = true;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        float m[CM], t[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) m[c] = (side == 0 ? hn[c] : ho[c]) ? wr[c] : 0.0f * wr[c];
        const float wtot = np_sum_regs<float, CM>(m, C);
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            float wc = m[c] / wtot;                                 // normalize_weights (likelihood.py:171-190)
            if (a.pow_w) wc = lib_powf(wc, a.inv_tp);
            t[c] = a.from_prior ? wc : wc * lh[c];
        }
        if (a.from_prior) {
#pragma unroll
            for (int c = 0; c < CM; ++c) p[side][c] = t[c];
            continue;
        }
        const float tot = np_sum_regs<float, CM>(t, C);
        ok = ok && tot > 0.0f;                                      // normalize's assert (util.py:1006)
#pragma unroll
        for (int c = 0; c < CM; ++c) p[side][c] = t[c] / tot;
    }
    if (!ok) raise_status(status, ST_BAD_NORMALIZE, 1);
    // sample_categorical (preprocessing.py:224-256): float32 cumulative sums, divided by the last, first c with z < cdf[c]
    float cdf[CM];
    float run = p[0][0];
    cdf[0] = run;
#pragma unroll
    for (int c = 1; c < CM; ++c) { if (c < C) run = run + p[0][c]; cdf[c] = run; }
    const float last = run;                                         // (= cdf[C - 1])
    int k = 0;
#pragma unroll
    for (int c = CM - 1; c >= 0; --c)
        if (c < C && zz < (double)(cdf[c] / last)) k = c;
    src_new[i] = na ? (uint8_t)kNA : (uint8_t)k;
    float sn = 1.0f, sb = 1.0f;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (c < C) {
            sn = (!na && c == k) ? p[0][c] : sn;
            sb = (c == id_old) ? p[1][c] : sb;
        }
    }
    sel_new[i] = sn;
    sel_back[i] = sb;
    return na ? -1 : k;
}

static __global__ __launch_bounds__(kBlock) void k_given_unchanged_gibbs(GuGibbsArgs a, uint8_t* __restrict__ src_new,
                                                                 float* __restrict__ sel_new, float* __restrict__ sel_back,
                                                                 int* __restrict__ status, DoneSig done = DoneSig{}) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)a.n_sub * a.F) {
        const int r = (int)(i / a.F), f = (int)(i % a.F);
        const uint8_t x = a.state[(int64_t)a.objects[r] * a.Fp + f];
        (void)gu_gibbs_obs(a, i, r, f, x, a.z[i], (int)a.src_old[i], [&](int c) { return a.group_idx[(int64_t)c * a.n_sub + r]; },
                     [&](int c, int g) { return a.tables[((int64_t)(a.table_offsets[c] + g) * a.F + f) * a.S + x]; },
                     src_new, sel_new, sel_back, status);
    }
    signal_done(done);
}

// Last kernel of sbe_gibbs_propose (GibbsSampleSource._propose in one call): what the host bookkeeping needs of the candidate
// slot, written straight into the host-mapped result block -- the drawn source component of every observation of the subset
// (0xFF: none, i.e. an NA observation) and, for the groups the subset's objects are in, the rows of candidate counts minus
// current counts (what update_feature_counts would have computed, counts.py:55-95).  Element-parallel; carries the call's
// completion flag.
static __global__ void k_gibbs_fetch(const uint8_t* __restrict__ src_cand /* [N][Fp] */, const int32_t* __restrict__ objects, int n_sub,
                              uint8_t* __restrict__ ids_out /* [n_sub][F] */, const int32_t* __restrict__ counts_cur,
                              const int32_t* __restrict__ counts_cand, const int32_t* __restrict__ touched, int n_touched,
                              float* __restrict__ rows_out /* [n_touched][F][S] */, int F, int S, int Fp, DoneSig done) {
    const int64_t n_ids = (int64_t)n_sub * F, fs = (int64_t)F * S, n_rows = (int64_t)n_touched * fs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ids + n_rows; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_ids) {
            const int r = (int)(i / F), f = (int)(i % F);
            ids_out[i] = src_cand[(int64_t)objects[r] * Fp + f];
        } else {
            const int64_t j = i - n_ids;
            const int64_t at = (int64_t)touched[j / fs] * fs + j % fs;
            rows_out[j] = (float)(counts_cand[at] - counts_cur[at]);
        }
    }
    signal_done(done);
}

// GibbsSampleSource._propose (operators.py:495-552) in ONE kernel (sbe_gibbs_propose's tile form): everything the proposal
// needs is independent from feature to feature, so a block owns a 16-feature tile and does, for the listed objects,
//   1. the draw: posterior of every observation under the slot's tables (posterior_row_core: k_sample_source's arithmetic),
//      the caller's uniform picks the component; drawn id and p[drawn] go to the mapped result block;
//   2. the count delta: one more in the object's group of the drawn component, one less in its group of the old one, per
//      touched group in LDS; the rows go out as float32 (update_feature_counts, counts.py:55-95);
//   3. the tables of the touched groups from counts + delta (probs_row_x16: update_probs' untempered arithmetic), in LDS --
//      every group an object of the subset is in IS touched, so the backward pass reads nothing else;
//   4. the backward probabilities: posterior under those tables, p_back[old source] (k_source_logprob's arithmetic).
// No candidate slot is built (the chain form: k_sample_source, k_step_core -- a whole slot copied and every table rebuilt,
// 33 us -- k_source_logprob, k_gibbs_fetch).  LDS: staged input block | pos [Gtot] | delta / tables [T][16][S] | drawn ids [n][16].
constexpr int kTileBlock = 1024;                 // (= kUnchangedBlock below: the 16-feature-tile operator kernels)
// A slot that FOLLOWS a call on the device (sbe_counts_delta_apply, sbe_given_unchanged_gibbs_apply, sbe_gibbs_propose_apply):
// its resident counts take the call's count delta, the touched groups' probability rows are rebuilt (probs != nullptr), its
// source rows of the subset become the new ids (src != nullptr).
struct DeltaFollow { int32_t* counts; const double* conc; float* probs; float* probs_t; int* status; int ft;
                     uint8_t* src; /* or nullptr: the slot's [N][Fp] source ids take the subset's new rows */ };

struct GibbsTileArgs {
    const uint8_t* state; const uint16_t* gid; const uint8_t* pid; const uint8_t* src; const float* probs; const float* wpat;
    const int32_t* counts; const double* conc;
    const uint32_t* mapped_in; int in_words, objects_word, gid_word /* [C][n] GLOBAL group ids, -1 none */, touched_word;
    const double* z;                 // [n][F] uniforms
    uint8_t* ids_out; float* sel_out; float* back_out; float* rows_out;
    int n_sub, n_touched, Gtot, Np, F, S, C, Fp;
    double inv_t; float inv_tp; int pow_lh, pow_w, from_prior;
    int* status;
    DeltaFollow follow;              // sbe_gibbs_propose_apply: the CURRENT slot takes the proposal (counts in step 3, tables and ids behind the flag)
};

static __global__ __launch_bounds__(kTileBlock) void k_gibbs_propose_tile(GibbsTileArgs a, DoneSig done) {
    constexpr int FTU = 16;
    extern __shared__ int32_t gl[];
    const int S = a.S, C = a.C, n_sub = a.n_sub, T = a.n_touched;
    uint32_t* stage = reinterpret_cast<uint32_t*>(gl);
    int32_t* pos = reinterpret_cast<int32_t*>(stage + a.in_words);       // [Gtot]
    int32_t* dhist = pos + a.Gtot;                                       // [T][FTU][S]: delta, then the new tables (float)
    uint8_t* knew = reinterpret_cast<uint8_t*>(dhist + T * FTU * S);     // [n_sub][FTU] drawn component (0xFF: NA)
    const int32_t* obj = reinterpret_cast<const int32_t*>(stage + a.objects_word);
    const int32_t* gidl = reinterpret_cast<const int32_t*>(stage + a.gid_word);
    const int32_t* touched = reinterpret_cast<const int32_t*>(stage + a.touched_word);
    const int f0 = blockIdx.x * FTU;
    for (int i0 = threadIdx.x; i0 < a.in_words; i0 += 4 * kTileBlock) {       // one PCIe round trip (k_given_unchanged_fused)
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = i0 + j * kTileBlock; v[j] = i < a.in_words ? a.mapped_in[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = i0 + j * kTileBlock; if (i < a.in_words) stage[i] = v[j]; }
    }
    // the uniform of the observation this thread draws first: asked for now (host-mapped or device memory)
    double z_first = 0.0;
    {
        const int r = threadIdx.x / FTU, ff = f0 + (threadIdx.x & (FTU - 1));
        if (r < n_sub && ff < a.F) z_first = a.z[(int64_t)r * a.F + ff];
    }
    for (int i = threadIdx.x; i < T * FTU * S; i += kTileBlock) dhist[i] = 0;
    for (int i = threadIdx.x; i < a.Gtot; i += kTileBlock) pos[i] = -1;
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += kTileBlock) pos[touched[t]] = t;
    __syncthreads();
    // 1 + 2: the draw and the count delta
    for (int t = threadIdx.x; t < n_sub * FTU; t += kTileBlock) {
        const int r = t / FTU, tf = t % FTU, ff = f0 + tf;
        if (ff >= a.F) continue;
        const int n = obj[r];
        const uint8_t x = a.state[(int64_t)n * a.Fp + ff];
        const int64_t i = (int64_t)r * a.F + ff;
        const float* w = a.wpat + ((int64_t)a.pid[n] * a.F + ff) * C;
        float p[kMaxComponents];
        const bool ok = posterior_row_core(x, w, C, a.from_prior, a.pow_lh, a.inv_t, a.pow_w, a.inv_tp,
                                           [&](int c) { const int g = gidl[c * n_sub + r]; return g < 0 ? (uint32_t)kNoGroup : (uint32_t)g; },
                                           [&](int, uint32_t g) { return a.probs[((int64_t)g * a.F + ff) * S + x]; }, p);
        if (!ok) raise_status(a.status, ST_BAD_NORMALIZE, 1);
        // sample_categorical (preprocessing.py:224-256), as k_sample_source draws: float32 cumulative sums / the last, first c with z < cdf[c]
        float cdf[kMaxComponents];
        float run = p[0];
        cdf[0] = run;
#pragma unroll
        for (int c = 1; c < kMaxComponents; ++c) { if (c < C) run = run + p[c]; cdf[c] = run; }
        const float last = run;
        const double zz = t == (int)threadIdx.x ? z_first : a.z[i];
        int k = 0;
#pragma unroll
        for (int c = kMaxComponents - 1; c >= 0; --c)
            if (c < C && zz < (double)(cdf[c] / last)) k = c;
        const bool na = x == kNA;
        float sel = 1.0f;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) if (c < C) sel = (!na && c == k) ? p[c] : sel;
        a.ids_out[i] = na ? (uint8_t)kNA : (uint8_t)k;
        a.sel_out[i] = sel;
        knew[t] = na ? (uint8_t)kNA : (uint8_t)k;
        if (!na) {
            const int g_new = gidl[k * n_sub + r];
            if (g_new >= 0 && pos[g_new] >= 0) atomicAdd(&dhist[(pos[g_new] * FTU + tf) * S + x], 1);
            const int so = a.src[(int64_t)n * a.Fp + ff];
            if (so < C) {
                const int g_old = gidl[so * n_sub + r];
                if (g_old >= 0 && pos[g_old] >= 0) atomicAdd(&dhist[(pos[g_old] * FTU + tf) * S + x], -1);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * FTU * S; e += kTileBlock) {
        const int tt = e / (FTU * S), q = e % (FTU * S), ff = f0 + q / S;
        if (ff < a.F) a.rows_out[((int64_t)tt * a.F + ff) * S + q % S] = (float)dhist[e];
    }
    __syncthreads();
    // 3: the touched groups' tables from counts + delta, in place (update_probs' arithmetic: untempered)
    auto rows_by_lane_groups = [&](auto width) {
        constexpr int W = decltype(width)::value;
        for (int row0 = 0; row0 < T * FTU; row0 += kTileBlock / W) {
            const int row = row0 + (int)threadIdx.x / W, j = threadIdx.x & (W - 1);
            const int tt = row / FTU, tf = row % FTU, ff = f0 + tf;
            const bool row_on = row < T * FTU && ff < a.F;
            const int64_t at = row_on ? ((int64_t)touched[tt] * a.F + ff) * S : 0;
            int32_t* h = dhist + (row_on ? (tt * FTU + tf) * S : 0);
            float cj = 0.0f;
            if (row_on && j < S) {
                const int32_t v = a.counts[at + j] + h[j];
                if (a.follow.counts) a.follow.counts[at + j] = v;         // (the following slot: this block owns the row)
                cj = (float)v;
            }
            probs_row_x16<W>(j, row_on, [&](int) { return cj; }, a.conc + at, nullptr, S, 0.0, 0.0, a.status,
                             [&](int s, float v) { h[s] = __float_as_int(v); });
        }
    };
    if (S <= 8) rows_by_lane_groups(std::integral_constant<int, 8>{});
    else if (S <= 16) rows_by_lane_groups(std::integral_constant<int, 16>{});
    else {
        for (int t = threadIdx.x; t < T * FTU; t += kTileBlock) {
            const int tt = t / FTU, tf = t % FTU, ff = f0 + tf;
            if (ff >= a.F) continue;
            const int64_t at = ((int64_t)touched[tt] * a.F + ff) * S;
            int32_t* h = dhist + (tt * FTU + tf) * S;
            if (a.follow.counts) {
                for (int s = 0; s < S; ++s) a.follow.counts[at + s] = a.counts[at + s] + h[s];
                probs_row([&](int s) { return (float)a.follow.counts[at + s]; }, a.conc + at, nullptr, S, 0.0, 0.0, a.status,
                          [&](int s, float v) { h[s] = __float_as_int(v); });
                continue;
            }
            probs_row([&](int s) { return (float)(a.counts[at + s] + h[s]); }, a.conc + at, nullptr, S, 0.0, 0.0, a.status,
                      [&](int s, float v) { h[s] = __float_as_int(v); });
        }
    }
    __syncthreads();
    // 4: the backward probabilities under the new tables
    const float* tab = reinterpret_cast<const float*>(dhist);
    for (int t = threadIdx.x; t < n_sub * FTU; t += kTileBlock) {
        const int r = t / FTU, tf = t % FTU, ff = f0 + tf;
        if (ff >= a.F) continue;
        const int n = obj[r];
        const uint8_t x = a.state[(int64_t)n * a.Fp + ff];
        const float* w = a.wpat + ((int64_t)a.pid[n] * a.F + ff) * C;
        float p[kMaxComponents];
        const bool ok = posterior_row_core(x, w, C, a.from_prior, a.pow_lh, a.inv_t, a.pow_w, a.inv_tp,
                                           [&](int c) { const int g = gidl[c * n_sub + r]; return g < 0 ? (uint32_t)kNoGroup : (uint32_t)g; },
                                           [&](int, uint32_t g) { return tab[(pos[g] * FTU + tf) * S + x]; }, p);
        if (!ok) raise_status(a.status, ST_BAD_NORMALIZE, 1);
        const int id = a.src[(int64_t)n * a.Fp + ff];
        float sel = 1.0f;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) if (c < C) sel = (c == id) ? p[c] : sel;
        a.back_out[(int64_t)r * a.F + ff] = sel;
    }
    signal_done(done);
    if (a.follow.counts) {                               // behind the flag: the touched groups' new tables (LDS) and the drawn ids
        if (!done.flag) __syncthreads();
        if (a.follow.probs) {
            for (int e = threadIdx.x; e < T * FTU * S; e += kTileBlock) {
                const int tt = e / (FTU * S), q = e % (FTU * S), tf = q / S, s = q % S, ff = f0 + tf;
                if (ff >= a.F) continue;
                const int g = touched[tt];
                a.follow.probs[((int64_t)g * a.F + ff) * S + s] = tab[e];
                a.follow.probs_t[((((int64_t)(ff / a.follow.ft) * (a.Gtot + 1) + g) * S + s)) * a.follow.ft + ff % a.follow.ft] = tab[e];
            }
        }
        if (a.follow.src) {
            for (int t = threadIdx.x; t < n_sub * FTU; t += kTileBlock) {
                const int r = t / FTU, tf = t % FTU;
                if (f0 + tf < a.F) a.follow.src[(int64_t)obj[r] * a.Fp + f0 + tf] = knew[t];
            }
        }
    }
}

// SourcePrior.__call__ (prior.py:573-611), per-object values:
//   sp[n] = float32( sum_{f valid} log( w[pat(n)][f][source(n,f)] ) )     (float32 logs)
// One wave per object, lanes over features, wave64 shuffle reduce.  An observation whose source has no
// component set contributes log(0) = -inf like the reference's sum(w * s) = 0.
__device__ __forceinline__ void source_prior_block(const uint8_t* __restrict__ state, const uint8_t* __restrict__ src,
                                                   const uint8_t* __restrict__ pid, const float* __restrict__ wpat,
                                                   double* __restrict__ out, int N, int F, int C, int Fp, int block) {
    const int lane = threadIdx.x & (kWave - 1);
    const int n = block * (blockDim.x / kWave) + (threadIdx.x >> 6);           // (16 objects per 1024-thread block: a block's
    if (n < N) {                                           // (wave-uniform)           completion fence is per block)
        const float* w = wpat + (int64_t)pid[n] * F * C;
        double acc = 0.0;
        for (int f = lane; f < F; f += kWave) {
            if (state[(int64_t)n * Fp + f] == kNA) continue;
            const uint8_t c = src[(int64_t)n * Fp + f];
            const float ow = c < C ? w[(int64_t)f * C + c] : 0.0f;
            acc += (double)logf(ow);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[n] = (double)(float)acc;
    }
}
static __global__ __launch_bounds__(1024) void k_source_prior(const uint8_t* __restrict__ state,
                                                        const uint8_t* __restrict__ src,
                                                        const uint8_t* __restrict__ pid,
                                                        const float* __restrict__ wpat, double* __restrict__ out,
                                                        int N, int F, int C, int Fp, DoneSig done = DoneSig{}) {
    source_prior_block(state, src, pid, wpat, out, N, F, C, Fp, (int)blockIdx.x);
    signal_done(done);
}

// Model.__call__ = likelihood + prior (sbayes/model/model.py:47-51): the collapsed log-likelihood of every group
// (k_collapsed_groups' blocks) and the per-object source prior (k_source_prior's blocks) of the same slot state in ONE
// launch -- the two halves share nothing but the completion flag.  Blocks [0, G): groups; the rest: 16 objects each.
struct SourcePriorArgs { const uint8_t* state; const uint8_t* src; const uint8_t* pid; const float* wpat; double* out; int N, F, C, Fp; };
static __global__ __launch_bounds__(1024) void k_collapsed_source_prior(
    const int32_t* __restrict__ counts, const double* __restrict__ conc, const double* __restrict__ lg_conc,
    const double* __restrict__ sum_a, const double* __restrict__ lg_sum_a, double* __restrict__ per_group, int G, int F, int S,
    SourcePriorArgs sp, DoneSig done) {
    extern __shared__ __align__(16) unsigned char cg_lds[];
    if ((int)blockIdx.x < G)                                   // (block-uniform: the barriers inside are reached by the whole block)
        collapsed_group_block<int32_t>(counts, conc, lg_conc, sum_a, lg_sum_a, nullptr, per_group, 0, F, S, (int)blockIdx.x, cg_lds);
    else
        source_prior_block(sp.state, sp.src, sp.pid, sp.wpat, sp.out, sp.N, sp.F, sp.C, sp.Fp, (int)blockIdx.x - G);
    signal_done(done);
}

// One-launch copy of all per-slot arrays (sbe_copy_slot): up to 16 dword-granular segments.
struct CopySegs {
    const uint32_t* src[16];
    uint32_t* dst[16];
    uint32_t end[16];        // exclusive prefix end of each segment, in dwords
    int n;
};

static __global__ void k_multi_copy(CopySegs cs) {
    const uint32_t total = cs.end[cs.n - 1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = 0;
        while (i >= cs.end[k]) ++k;
        const uint32_t base = k ? cs.end[k - 1] : 0u;
        cs.dst[k][i - base] = cs.src[k][i - base];
    }
}

// ------------------------------------------------------------------------------------------
// One-call MCMC step (sbe_step), kernel 1 of 3: the candidate slot = current slot + the step's payload (new
// cluster / pattern / tuple ids, changed source rows, weights; host-mapped pinned memory read in place), its count
// delta and every one of its tables in ONE launch -- no kernel boundary is needed between "apply", "count" and
// "tables" once every block works from the CURRENT slot and the payload only:
//   tile blocks   [0, n_tile_blocks): a block owns `ftc` features of every group: count delta of the moved
//                 objects (LDS histogram; the new group / source of an object come from the payload where it
//                 carries them, else from the current slot), candidate counts = current + delta, then the
//                 block's rows of every table (k_step_tables' arithmetic).  Changed groups are stamped with
//                 the step number (no flag array to clear).
//   weight blocks: normalised weights per pattern (from the payload's / the current slot's weights and patterns)
//   copy blocks  : the other per-slot arrays, current slot or payload -> candidate (k_step_apply without counts)
// ------------------------------------------------------------------------------------------
// lgamma terms of the concentration tables that do not depend on the counts (one-call steps, round 3): per element
// lgamma(conc) (0 where conc <= 0: that state is not applicable), per (group, feature) row sum_a = the NumPy-order sum
// of the row and lgamma(sum_a).  Built once per sbe_set_concentration; k_step_core then evaluates ONE lgamma per
// element and per row instead of two (same function, same arguments: the values are the ones it computed itself).
static __global__ void k_conc_lgamma(const double* __restrict__ conc, double* __restrict__ lg_conc, double* __restrict__ sum_a,
                              double* __restrict__ lg_sum_a, int g_lo, int g_hi, int F, int S) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (int64_t)(g_hi - g_lo) * F) return;
    const int64_t r = (int64_t)g_lo * F + row, base = r * S;
    for (int s = 0; s < S; ++s) {
        const double a = conc[base + s];
        lg_conc[base + s] = a > 0.0 ? sbe_lgamma_pos(a) : 0.0;
    }
    auto conc_at = [&](int k) -> double { return conc[base + k]; };
    const double sa = np_pairwise_sum<double>(conc_at, S);
    sum_a[r] = sa;
    lg_sum_a[r] = sbe_lgamma_pos(sa);
}

struct StepCore {
    // copy blocks
    CopySegs cs; int src_seg;
    const int16_t* row_of; const uint8_t* rows; const int32_t* objects; uint8_t* src_dst;
    int n_changed, F, C, Fp;
    int* status;
    // tile blocks
    const uint8_t* state; const uint16_t* gid_cur; const uint16_t* ids_new; const uint8_t* src_cur;
    const uint8_t* src_new;        // one-call Gibbs step: the new source of the marked objects was SAMPLED into this
                                   // [N][Fp] array (the candidate's) by k_sample_source; nullptr: payload rows
    const int32_t* subset; int n_subset;
    const int32_t* counts_cur; int32_t* counts_new;
    const double* conc; float* probs; float* probs_t; float* per_feature;
    const double* lg_conc; const double* sum_a; const double* lg_sum_a;   // k_conc_lgamma's tables of `conc`
    uint32_t* stamp; uint32_t step_id;
    int Np, S, Gtot, ft, ftc, n_tile_blocks;
    // weight blocks
    const float* weights; const uint32_t* pattern_bits; float* wpat; double* wpat_t;
    int P, Pmax, n_weight_blocks;
    int n_copy_blocks;             // (batched launch: the grid is sized for the largest chain; surplus blocks exit)
    // round 3: when the candidate slot's source array is known to differ from the current slot's only in `stale` rows
    // (the previous step's changed rows), those rows are copied instead of the whole array (no src segment in `cs`)
    const int32_t* stale; int n_stale; const uint8_t* src_cur_rows;
    // round 3, delta layout (sbe_step_batch_delta): the candidate slot's per-object id arrays are PATCHED, not copied --
    // entry i of the patch lists holds the candidate's (gid of component 0, pattern id, tuple id) of object patch_n[i]
    // (the objects this step moves and the ones the previous step left different in this slot); `sub_row` / `sub_gid0`
    // give, per entry of `subset`, the object's row in `rows` (-1: its source does not change) and its candidate
    // component-0 group id, in place of the [Np] arrays row_of / ids_new.  n_patch < 0: classic layout.
    const int32_t* patch_n; const uint16_t* patch_gid; const uint8_t* patch_pid; const uint8_t* patch_tid; int n_patch;
    uint16_t* gid_dst; uint8_t* pid_dst; uint8_t* tid_dst; uint32_t* toff_dst; uint32_t toff_mul;
    const int16_t* sub_row; const uint16_t* sub_gid0;
};

__device__ __forceinline__ void step_core_body(const StepCore& a, unsigned char* core_lds, const int bx) {
    const int S = a.S, C = a.C, F = a.F;
    if (bx >= a.n_tile_blocks + a.n_weight_blocks + a.n_copy_blocks) return;      // surplus block of a batched launch
    if (bx < a.n_tile_blocks) {
        const int ftc = a.ftc, f0 = bx * ftc;
        const int E = a.Gtot * ftc * S, R = a.Gtot * ftc;
        int32_t* hist = reinterpret_cast<int32_t*>(core_lds);                       // [Gtot][ftc][S] delta, then counts
        double* sh_post = reinterpret_cast<double*>(core_lds + ((size_t)E * 4 + 15) / 16 * 16);
        double* sh_ser = sh_post + E;
        double* sh_total = sh_ser + E;                                              // [Gtot][ftc]
        for (int e = threadIdx.x; e < E; e += kBlock) hist[e] = 0;
        __syncthreads();
        for (int k = threadIdx.x; k < a.n_subset * ftc; k += kBlock) {
            const int i = k / ftc, fl = k - i * ftc, f = f0 + fl;
            if (f >= F) continue;
            const int n = a.subset[i];
            const uint8_t x = a.state[(int64_t)n * a.Fp + f];
            if (x == kNA) continue;
            const int c_old = a.src_cur[(int64_t)n * a.Fp + f];
            if (c_old < C) {
                const uint16_t g = a.gid_cur[(int64_t)c_old * a.Np + n];
                if (g != kNoGroup) atomicAdd(&hist[((int)g * ftc + fl) * S + x], -1);
            }
            int c_new = c_old;
            const int r = a.sub_row ? a.sub_row[i] : (a.row_of ? a.row_of[n] : -1);
            if (r >= 0 && a.src_new) c_new = a.src_new[(int64_t)n * a.Fp + f];
            else if (r >= 0) {
                const uint8_t* pr = a.rows + ((int64_t)r * F + f) * C;
                c_new = kNA;
                for (int c = 0; c < C; ++c) if (pr[c]) c_new = c;
            }
            if (c_new < C) {
                const uint16_t g = c_new == 0 ? (a.sub_gid0 ? a.sub_gid0[i] : (a.ids_new ? a.ids_new[n] : a.gid_cur[n]))
                                              : a.gid_cur[(int64_t)c_new * a.Np + n];
                if (g != kNoGroup) atomicAdd(&hist[((int)g * ftc + fl) * S + x], 1);
            }
        }
        __syncthreads();
        // candidate counts; posterior counts and lgamma terms of every element
        for (int e = threadIdx.x; e < E; e += kBlock) {
            const int g = e / (ftc * S), rem = e - g * (ftc * S), fl = rem / S, s = rem - fl * S;
            const int f = f0 + fl;
            if (f >= F) continue;
            const int64_t gi = ((int64_t)g * F + f) * S + s;
            const int d = hist[e];
            const int cn = a.counts_cur[gi] + d;
            a.counts_new[gi] = cn;
            if (d != 0) a.stamp[g] = a.step_id;
            hist[e] = cn;
            const float cf = (float)cn;
            const double conc = a.conc[gi];
            sh_post[e] = (double)cf + conc;
            sh_ser[e] = conc > 0.0 ? sbe_lgamma_pos((double)cf + conc) - a.lg_conc[gi] : 0.0;
        }
        __syncthreads();
        for (int r = threadIdx.x; r < R; r += kBlock) {                              // ordered sums of a row
            const int g = r / ftc, fl = r - g * ftc, f = f0 + fl;
            if (f >= F) continue;
            const int e0 = r * S;
            auto post_at = [&](int k) -> double { return sh_post[e0 + k]; };
            auto ser_at = [&](int k) -> double { return sh_ser[e0 + k]; };
            auto cnt_at = [&](int k) -> float { return (float)hist[e0 + k]; };
            const double total = np_pairwise_sum<double>(post_at, S);
            if (!(total > 0.0)) raise_status(a.status, ST_BAD_NORMALIZE, 1);
            sh_total[r] = total;
            const float n = np_pairwise_sum<float>(cnt_at, S);
            const double sum_a = a.sum_a[(int64_t)g * F + f];
            const double cst = a.lg_sum_a[(int64_t)g * F + f] - sbe_lgamma_pos((double)n + sum_a);
            a.per_feature[(int64_t)g * F + f] = (float)(cst + np_pairwise_sum<double>(ser_at, S));
        }
        __syncthreads();
        for (int e = threadIdx.x; e < E; e += kBlock) {
            const int g = e / (ftc * S), rem = e - g * (ftc * S), fl = rem / S, s = rem - fl * S;
            const int f = f0 + fl;
            if (f >= F) continue;
            const float pr = (float)(sh_post[e] / sh_total[g * ftc + fl]);
            a.probs[((int64_t)g * F + f) * S + s] = pr;
            const int tile = f / a.ft, tl = f % a.ft;
            a.probs_t[((((int64_t)tile * (a.Gtot + 1) + g) * S) + s) * a.ft + tl] = pr;
        }
        return;
    }
    const int wb = bx - a.n_tile_blocks;
    if (wb < a.n_weight_blocks) {
        const int64_t j = (int64_t)wb * kBlock + threadIdx.x;
        if (j >= (int64_t)a.P * F) return;
        const int p = (int)(j / F), f = (int)(j % F);
        const uint32_t bits = a.pattern_bits[p];
        const float* w = a.weights + (int64_t)f * C;
        auto masked = [&](int c) -> float { return ((bits >> c) & 1u) ? w[c] : 0.0f * w[c]; };
        const float total = np_pairwise_sum<float>(masked, C);
        float* out = a.wpat + ((int64_t)p * F + f) * C;
        const int tile = f / a.ft, tl = f % a.ft;
        double* ot = a.wpat_t + (((int64_t)tile * a.Pmax + p) * C) * a.ft + tl;
        for (int c = 0; c < C; ++c) {
            const float v = masked(c) / total;
            out[c] = v;
            ot[(int64_t)c * a.ft] = (double)v;
        }
        return;
    }
    // copy blocks
    const uint32_t n_copy = (uint32_t)a.n_copy_blocks;
    const uint32_t tid = (uint32_t)(wb - a.n_weight_blocks) * kBlock + threadIdx.x, nthreads = n_copy * kBlock;
    // segment by segment, 16 bytes per lane where both ends allow it (the source-assignment array, 95 % of the bytes,
    // always does: its rows are Fp = 64k bytes); rows of the source array that the payload replaces are skipped
    for (int k = 0; k < a.cs.n; ++k) {
        const uint32_t len = a.cs.end[k] - (k ? a.cs.end[k - 1] : 0u);              // dwords
        const uint32_t* src = a.cs.src[k];
        uint32_t* dst = a.cs.dst[k];
        const bool filtered = k == a.src_seg && a.row_of != nullptr;
        if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0 && (len & 3u) == 0 && (!filtered || (a.Fp & 15) == 0)) {
            const uint4* s4 = reinterpret_cast<const uint4*>(src);
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            for (uint32_t i = tid; i < len / 4u; i += nthreads) {
                if (filtered && a.row_of[(i * 16u) / (uint32_t)a.Fp] >= 0) continue;   // converted below
                d4[i] = s4[i];
            }
        } else {
            for (uint32_t j = tid; j < len; j += nthreads) {
                if (filtered && a.row_of[(j * 4u) / (uint32_t)a.Fp] >= 0) continue;
                dst[j] = src[j];
            }
        }
    }
    for (uint32_t i = tid; i < (uint32_t)max(a.n_patch, 0); i += nthreads) {      // delta layout: patched id entries
        const int n = a.patch_n[i];
        a.gid_dst[n] = a.patch_gid[i];
        a.pid_dst[n] = a.patch_pid[i];
        a.tid_dst[n] = a.patch_tid[i];
        a.toff_dst[n] = (uint32_t)a.patch_tid[i] * a.toff_mul;
    }
    if (a.n_stale > 0) {                                     // stale rows of the candidate's source <- the current slot's
        const uint32_t per_row = (uint32_t)a.Fp / 16u;       // (Fp is a multiple of 64)
        for (uint32_t i = tid; i < (uint32_t)a.n_stale * per_row; i += nthreads) {
            const uint32_t r = i / per_row, k = i - r * per_row;
            const int64_t off = (int64_t)a.stale[r] * a.Fp + (int64_t)k * 16;
            *reinterpret_cast<uint4*>(a.src_dst + off) = *reinterpret_cast<const uint4*>(a.src_cur_rows + off);
        }
    }
    int multi = 0;
    for (uint32_t i = tid; i < (uint32_t)a.n_changed * (uint32_t)F; i += nthreads) {
        const int r = (int)(i / (uint32_t)F), f = (int)(i % (uint32_t)F);
        const int n = a.objects[r];
        if (a.row_of && a.row_of[n] != r) continue;          // an object listed twice: the row the tile blocks use
        const uint8_t* p = a.rows + (int64_t)i * C;
        int id = kNA, cnt = 0;
        for (int c = 0; c < C; ++c)
            if (p[c]) { id = c; ++cnt; }
        a.src_dst[(int64_t)n * a.Fp + f] = (uint8_t)id;
        multi += cnt > 1;
    }
    if (multi) raise_status(a.status, ST_MULTI_SOURCE, multi);
}

static __global__ __launch_bounds__(kBlock) void k_step_core(StepCore a) {
    extern __shared__ __align__(16) unsigned char core_lds[];
    step_core_body(a, core_lds, (int)blockIdx.x);
}

// sbe_step_batch: one chain per blockIdx.y, each with its own StepCore (current / candidate slot, payload, counts ...)
// (the single-chain kernel takes what registers it wants -- 255, its lgamma expansions are wide -- at two blocks per CU;
//  a batch has thousands of blocks and is better off at 128 VGPRs / four blocks per CU despite more spills: 225 against
//  231 us per 64-chain sweep; 64 VGPRs: 268 us)
static __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_step_core_batch(const StepCore* __restrict__ cores) {
    extern __shared__ __align__(16) unsigned char core_lds[];
    // the chain's descriptor into LDS with one coalesced pass instead of scalar loads through the pointer, field by field
    // (27.5 -> 26.2 us per 64-chain launch, profiles/r3/ab_step_core_lds_args.log)
    __shared__ __align__(16) StepCore sc;
    static_assert(sizeof(StepCore) % 4 == 0, "StepCore is copied as dwords");
    const uint32_t* src = reinterpret_cast<const uint32_t*>(cores + blockIdx.y);
    for (int i = threadIdx.x; i < (int)(sizeof(StepCore) / 4); i += kBlock) reinterpret_cast<uint32_t*>(&sc)[i] = src[i];
    __syncthreads();
    step_core_body(sc, core_lds, (int)blockIdx.x);
}

// canonical probs [Gtot][F][S] -> tile-transposed probs_t [n_ftiles][Gtot+1][S][FT] for the
// groups [g_lo, g_hi).  Row Gtot and features >= F stay zero (set once at creation).
static __global__ void k_tile_probs(const float* __restrict__ probs, float* __restrict__ probs_t, int g_lo,
                             int g_hi, int Gtot, int F, int S, int ft, int n_ftiles) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_tile = (int64_t)(g_hi - g_lo) * S * ft;
    if (i >= per_tile * n_ftiles) return;
    const int tile = (int)(i / per_tile);
    int64_t r = i % per_tile;
    const int g = g_lo + (int)(r / ((int64_t)S * ft));
    r %= (int64_t)S * ft;
    const int s = (int)(r / ft), fl = (int)(r % ft);
    const int f = tile * ft + fl;
    const float v = f < F ? probs[((int64_t)g * F + f) * S + s] : 0.0f;
    probs_t[(((int64_t)tile * (Gtot + 1) + g) * S + s) * ft + fl] = v;
}

// ------------------------------------------------------------------------------------------
// Round 3: delta forms for the drop-in host layer -- what the UNCHANGED reference sampler asks per MCMC step crosses
// PCIe as object lists and a few changed rows, never as [N][F] masks or whole [G][F][S] tables (SURVEY.md 8(b),
// last row).
//
// k_counts_delta: update_feature_counts(sample_old, sample_new, features, object_subset) (counts.py:55-95), stateless.
// For the listed objects the caller hands over both states -- global group index per component (-1: none) and source
// component id per observation (0xFF: none) -- and the list of groups any of them is in ("touched"); the kernel
// writes, for every touched group,
//     diff[t][f][s] = #{i: new state counts (object i, f) at state s in group t} - #{i: old state ...}
// = the rows of the reference's `new_counts - old_counts` that can be non-zero (as float32: FLOAT_TYPE, counts.py:20).
// Block = (touched group, 16-feature tile); 16 features x 16 object lanes (a subset can be a whole cluster: the object
// axis gets the lanes, and every step of a lane's walk is a chain of dependent loads); LDS histogram [16][S].
// ------------------------------------------------------------------------------------------
constexpr int kDeltaFT = 16;
static __global__ __launch_bounds__(kBlock) void k_counts_delta(
    const uint8_t* __restrict__ state, const int32_t* __restrict__ objects, int n,
    const int32_t* __restrict__ gid_old /* [C][n] */, const int32_t* __restrict__ gid_new,
    const uint8_t* __restrict__ src_old /* [n][F] */, const uint8_t* __restrict__ src_new,
    const int32_t* __restrict__ touched /* [T] global group index */, const int32_t* __restrict__ touched_comp /* [T] */,
    float* __restrict__ out /* [T][F][S] */, int F, int S, int Fp, DoneSig done = DoneSig{}) {
    constexpr int FTU = kDeltaFT, OL = kBlock / FTU;
    extern __shared__ int32_t hist[];
    const int t = blockIdx.x, f0 = blockIdx.y * FTU;
    for (int i = threadIdx.x; i < FTU * S; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int fl = threadIdx.x & (FTU - 1), ol = threadIdx.x / FTU;
    const int f = f0 + fl;
    const int gg = touched[t], c = touched_comp[t];
    if (f < F) {
        for (int i0 = ol; i0 < n; i0 += 4 * OL) {                   // four objects per lane and step: their loads overlap
            bool in_new[4], in_old[4];
            int obj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(i0 + j * OL, n - 1);
                const bool live = i0 + j * OL < n;
                in_new[j] = live && gid_new[(int64_t)c * n + i] == gg;
                in_old[j] = live && gid_old[(int64_t)c * n + i] == gg;
                obj[j] = objects[i];
            }
            uint8_t x[4], sn[4], so[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(i0 + j * OL, n - 1);
                x[j] = state[(int64_t)obj[j] * Fp + f];
                sn[j] = src_new[(int64_t)i * F + f];
                so[j] = src_old[(int64_t)i * F + f];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (x[j] == kNA) continue;
                const int d = (int)(in_new[j] && sn[j] == c) - (int)(in_old[j] && so[j] == c);
                if (d) atomicAdd(&hist[fl * S + x[j]], d);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FTU * S; i += kBlock) {
        const int ff = f0 + i / S;
        if (ff < F) out[((int64_t)t * F + ff) * S + i % S] = (float)hist[i];
    }
    signal_done(done);
}

// The same difference for SMALL subsets in one launch, no copy in front (sbe_counts_delta, n <= kDeltaTileMaxN): a block owns
// a 16-feature tile and ALL touched groups.  It stages what it needs of the call's host-mapped input block into LDS in one
// PCIe round trip -- object list, both id arrays, its 16 columns of both source-id arrays (13 blocks read 20 KB in all at
// the headline shape; the grid over (touched group, tile) would have read every array 78 times) -- and then walks the
// staged subset: an observation whose new source is component c adds one to its new group of c, one whose old source is c
// takes one off its old group of c (counts.py:55-95: new counts - old counts over the subset, group by group).
// LDS: hist [T][16][S] | pos [Gtot] (touched index of a group, -1) | objects [n] | gid_old, gid_new [C][n] | src_old, src_new [n][16].
constexpr int kDeltaTileMaxN = 256;
// `follow` (sbe_counts_delta_apply): a slot whose resident counts are the OLD state's takes the difference in the same launch
// -- counts[touched rows] += delta, and (probs != nullptr) the probability rows of those groups rebuilt (update_probs'
// arithmetic, k_set_count_rows_probs_x's row form) -- so that the host does not send back the rows it has just received.
// The following slot's rows of one 16-feature tile (the calling block owns these features of every touched group): counts +=
// the LDS histograms `hist` [T][16][S], and -- follow.probs -- the probability rows of those groups rebuilt (update_probs'
// arithmetic in k_set_count_rows_probs_x's row form).  `tgl` [T]: the touched groups (LDS).  Every thread of the block calls.
__device__ __forceinline__ void follow_tile_rows(const DeltaFollow& follow, const int32_t* hist, const int32_t* tgl, int n_touched, int f0,
                                                 int F, int S, int Gtot) {
    constexpr int FTU = 16;
    if (!follow.counts) return;
    const int nthr = blockDim.x;
    auto rows_by_lane_groups = [&](auto width) {
        constexpr int W = decltype(width)::value;
        for (int row0 = 0; row0 < n_touched * FTU; row0 += nthr / W) {
            const int row = row0 + (int)threadIdx.x / W, j = threadIdx.x & (W - 1);
            const int tt = row / FTU, tf = row % FTU, ff = f0 + tf;
            const bool row_on = row < n_touched * FTU && ff < F;
            const int g = row_on ? tgl[tt] : 0;
            const int64_t at = row_on ? ((int64_t)g * F + ff) * S : 0;
            float cj = 0.0f;
            if (row_on && j < S) {
                const int32_t v = follow.counts[at + j] + hist[(tt * FTU + tf) * S + j];
                follow.counts[at + j] = v;
                cj = (float)v;
            }
            if (follow.probs) {
                float* out_row = follow.probs + at;
                float* out_t = follow.probs_t + (((int64_t)(ff / follow.ft) * (Gtot + 1) + g) * S) * follow.ft + ff % follow.ft;   // (k_probs' tile layout)
                probs_row_x16<W>(j, row_on, [&](int) { return cj; }, follow.conc + at, nullptr, S, 0.0, 0.0, follow.status,
                                 [&](int s, float v) { out_row[s] = v; out_t[(int64_t)s * follow.ft] = v; });
            }
        }
    };
    if (S <= 8) rows_by_lane_groups(std::integral_constant<int, 8>{});
    else if (S <= 16) rows_by_lane_groups(std::integral_constant<int, 16>{});
    else {
        for (int t = threadIdx.x; t < n_touched * FTU; t += nthr) {
            const int tt = t / FTU, tf = t % FTU, ff = f0 + tf;
            if (ff >= F) continue;
            const int g = tgl[tt];
            const int64_t at = ((int64_t)g * F + ff) * S;
            const int32_t* h = hist + (tt * FTU + tf) * S;
            for (int k = 0; k < S; ++k) follow.counts[at + k] += h[k];
            if (follow.probs) {
                float* out_row = follow.probs + at;
                float* out_t = follow.probs_t + (((int64_t)(ff / follow.ft) * (Gtot + 1) + g) * S) * follow.ft + ff % follow.ft;
                probs_row([&](int k) { return (float)follow.counts[at + k]; }, follow.conc + at, nullptr, S, 0.0, 0.0, follow.status,
                          [&](int k, float v) { out_row[k] = v; out_t[(int64_t)k * follow.ft] = v; });
            }
        }
    }
}

static __global__ __launch_bounds__(kBlock) void k_counts_delta_tile(
    const uint8_t* __restrict__ state, const int32_t* __restrict__ objects, int n, const int32_t* __restrict__ gid_old,
    const int32_t* __restrict__ gid_new, const uint8_t* __restrict__ src_old, const uint8_t* __restrict__ src_new,
    const int32_t* __restrict__ touched, int n_touched, float* __restrict__ out /* [T][F][S] */, int F, int S, int Fp, int C,
    int Gtot, DoneSig done, DeltaFollow follow = DeltaFollow{}) {
    constexpr int FTU = kDeltaFT, OL = kBlock / FTU;
    extern __shared__ int32_t dl[];
    int32_t* hist = dl;                                  // [T][FTU][S]
    int32_t* pos = hist + n_touched * FTU * S;           // [Gtot]
    int32_t* tgl = pos + Gtot;                           // [T] the touched groups
    int32_t* obj = tgl + n_touched;                      // [n]
    int32_t* go = obj + n;                               // [C][n]
    int32_t* gn = go + C * n;                            // [C][n]
    uint8_t* so = reinterpret_cast<uint8_t*>(gn + C * n);            // [n][FTU]
    uint8_t* sn = so + n * FTU;                                      // [n][FTU]
    const int f0 = blockIdx.x * FTU;
    // one round trip: every mapped load of this thread is issued before the first LDS store waits
    int32_t v_touched = -1;
    {
        int32_t v_obj = 0, v_go[kMaxComponents], v_gn[kMaxComponents];
        const int i = threadIdx.x;                       // n <= kDeltaTileMaxN = blockDim: one object per thread
        const bool on = i < n;
        if (on) v_obj = objects[i];
        v_touched = (int)threadIdx.x < n_touched ? touched[threadIdx.x] : -1;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) {
            v_go[c] = (on && c < C) ? gid_old[c * n + i] : -1;
            v_gn[c] = (on && c < C) ? gid_new[c * n + i] : -1;
        }
        uint32_t w_so[4], w_sn[4];                       // this thread's share of the source columns: n * 16 bytes = n * 4 words per array
        int widx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = threadIdx.x + k * kBlock;      // word w <-> object w / 4, bytes (w % 4) * 4 .. + 3 of the tile
            widx[k] = w;
            w_so[k] = w_sn[k] = 0xFFFFFFFFu;
            if (w < n * 4) {
                const int oi = w >> 2, b0 = (w & 3) * 4;
                if ((F & 3) == 0) {                      // rows of a multiple of four features: the four ids are one aligned word
                    if (f0 + b0 < F) {
                        w_so[k] = *reinterpret_cast<const uint32_t*>(src_old + (int64_t)oi * F + f0 + b0);
                        w_sn[k] = *reinterpret_cast<const uint32_t*>(src_new + (int64_t)oi * F + f0 + b0);
                    }
                } else {
                    uint32_t a = 0, b = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = f0 + b0 + q;
                        const uint32_t x_o = f < F ? src_old[(int64_t)oi * F + f] : 0xFFu;
                        const uint32_t x_n = f < F ? src_new[(int64_t)oi * F + f] : 0xFFu;
                        a |= x_o << (8 * q);
                        b |= x_n << (8 * q);
                    }
                    w_so[k] = a; w_sn[k] = b;
                }
            }
        }
        for (int t = threadIdx.x; t < n_touched * FTU * S; t += kBlock) hist[t] = 0;
        for (int g = threadIdx.x; g < Gtot; g += kBlock) pos[g] = -1;
        if (on) obj[i] = v_obj;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) if (on && c < C) { go[c * n + i] = v_go[c]; gn[c * n + i] = v_gn[c]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (widx[k] < n * 4) { reinterpret_cast<uint32_t*>(so)[widx[k]] = w_so[k]; reinterpret_cast<uint32_t*>(sn)[widx[k]] = w_sn[k]; }
    }
    __syncthreads();
    if (v_touched >= 0) { pos[v_touched] = threadIdx.x; tgl[threadIdx.x] = v_touched; }
    for (int t = threadIdx.x + kBlock; t < n_touched; t += kBlock) { const int g = touched[t]; pos[g] = t; tgl[t] = g; }   // (more than 256 touched groups)
    __syncthreads();
    const int fl = threadIdx.x & (FTU - 1), ol = threadIdx.x / FTU;
    const int f = f0 + fl;
    if (f < F) {
        for (int i0 = ol; i0 < n; i0 += 4 * OL) {                   // four objects per lane and pass: their state loads overlap
            uint8_t x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int i = i0 + j * OL; x[j] = i < n ? state[(int64_t)obj[i] * Fp + f] : kNA; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * OL;
                if (i >= n || x[j] == kNA) continue;
                const int c_new = sn[i * FTU + fl], c_old = so[i * FTU + fl];
                if (c_new < C) { const int g = gn[c_new * n + i]; if (g >= 0 && pos[g] >= 0) atomicAdd(&hist[(pos[g] * FTU + fl) * S + x[j]], 1); }
                if (c_old < C) { const int g = go[c_old * n + i]; if (g >= 0 && pos[g] >= 0) atomicAdd(&hist[(pos[g] * FTU + fl) * S + x[j]], -1); }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_touched * FTU * S; e += kBlock) {
        const int t = e / (FTU * S), r = e % (FTU * S), ff = f0 + r / S;
        if (ff < F) out[((int64_t)t * F + ff) * S + r % S] = (float)hist[e];
    }
    // the caller waits for the difference only: the flag goes out before the following slot is brought up to date (the
    // next operation of the stream is ordered behind this kernel's end anyway; nothing below reads the mapped block)
    signal_done(done);
    if (follow.src) {                                    // the subset's new source ids, this block's 16 columns
        for (int t = threadIdx.x; t < n * FTU; t += kBlock) {
            const int i = t / FTU, tf = t % FTU;
            if (f0 + tf < F) follow.src[(int64_t)obj[i] * Fp + f0 + tf] = sn[i * FTU + tf];
        }
    }
    follow_tile_rows(follow, hist, tgl, n_touched, f0, F, S, Gtot);
}

// The following slot of sbe_counts_delta_apply behind the GENERAL difference kernel (subsets beyond the tile form): the
// difference rows [T][F][S] (device memory) are added to the slot's counts, the probability rows rebuilt.  One thread per
// (touched group, feature); rare path.
static __global__ void k_add_count_rows(const float* __restrict__ diff /* [T][F][S] */, const int32_t* __restrict__ touched, int n_touched,
                                 int F, int S, int Gtot, DeltaFollow follow) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_touched * F) return;
    const int i = (int)(t / F), f = (int)(t % F);
    const int g = touched[i];
    const int64_t at = ((int64_t)g * F + f) * S;
    const float* d = diff + t * S;
    for (int k = 0; k < S; ++k) follow.counts[at + k] += (int32_t)d[k];
    if (follow.probs) {
        float* out_row = follow.probs + at;
        float* out_t = follow.probs_t + (((int64_t)(f / follow.ft) * (Gtot + 1) + g) * S) * follow.ft + f % follow.ft;   // (k_probs' tile layout)
        probs_row([&](int k) { return (float)follow.counts[at + k]; }, follow.conc + at, nullptr, S, 0.0, 0.0, follow.status,
                  [&](int k, float v) { out_row[k] = v; out_t[(int64_t)k * follow.ft] = v; });
    }
}

static __global__ void k_set_source_ids(const uint8_t* __restrict__ ids /* [n][F] */, const int32_t* __restrict__ objects, int n, int F, int Fp,
                                 uint8_t* __restrict__ src /* slot's [N][Fp] */) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * F) return;
    src[(int64_t)objects[t / F] * Fp + t % F] = ids[t];
}

// float32 count rows of listed groups -> the slot's resident int32 counts (Engine.set_counts_rows: the bind cache
// sends only the groups whose rows differ from what the slot holds)
struct CountRowsArgs {
    const float* rows; const int32_t* group_idx; int32_t* counts; const double* conc; float* probs; float* probs_t;
    int n, F, S, Gtot, ft; int* status;
};
__device__ __forceinline__ void set_count_rows_item(const CountRowsArgs& a, int64_t i) {
    const int64_t fs = (int64_t)a.F * a.S;
    if (i >= (int64_t)a.n * fs) return;
    a.counts[(int64_t)a.group_idx[i / fs] * fs + i % fs] = (int32_t)a.rows[i];
}
static __global__ void k_set_count_rows(const float* __restrict__ rows /* [n][F][S] */, const int32_t* __restrict__ group_idx,
                                 int32_t* __restrict__ counts /* slot's [Gtot][F][S] */, int n, int64_t fs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * fs) return;
    counts[(int64_t)group_idx[i / fs] * fs + i % fs] = (int32_t)rows[i];
}

// The same patch with the probability rows of the patched groups rebuilt in the same launch (sbe_set_counts_rows_probs:
// the bind cache sends the rows of the groups whose counts changed, and the tables of exactly those groups are what is
// stale afterwards -- one launch instead of this one and a k_probs over the whole component).  One thread per (row,
// feature): the S float32 counts go to the resident int32 table, probs_row (k_probs' arithmetic, untempered) writes the
// slot's probability row and its tile-transposed copy.
__device__ __forceinline__ void set_count_rows_probs_item(const CountRowsArgs& a, int64_t t) {
    if (t >= (int64_t)a.n * a.F) return;
    const int F = a.F, S = a.S, ft = a.ft;
    const int i = (int)(t / F), f = (int)(t % F);
    const int g = a.group_idx[i], tile = f / ft, tl = f % ft;
    const float* in = a.rows + t * S;
    const int64_t base = ((int64_t)g * F + f) * S;
    int32_t* cnt = a.counts + base;
    for (int s = 0; s < S; ++s) cnt[s] = (int32_t)in[s];
    float* out_row = a.probs + base;
    float* out_t = a.probs_t + (((int64_t)tile * (a.Gtot + 1) + g) * S) * ft + tl;                 // (k_probs' tile layout)
    probs_row([&](int s) { return (float)(int32_t)in[s]; }, a.conc + base, nullptr, S, 0.0, 0.0, a.status,
              [&](int s, float v) { out_row[s] = v; out_t[(int64_t)s * ft] = v; });
}
static __global__ void k_set_count_rows_probs(const float* __restrict__ rows /* [n][F][S] */, const int32_t* __restrict__ group_idx,
                                       int32_t* __restrict__ counts /* slot's [Gtot][F][S] */, const double* __restrict__ conc,
                                       float* __restrict__ probs, float* __restrict__ probs_t, int n, int F, int S, int Gtot, int ft,
                                       int* __restrict__ status) {
    set_count_rows_probs_item(CountRowsArgs{rows, group_idx, counts, conc, probs, probs_t, n, F, S, Gtot, ft, status},
                              (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// The same for S <= W (8 or 16): W lanes per (row, feature), lane j <-> state j (probs_row_x16) -- the staged float rows are
// read coalesced (they sit in host-mapped memory: S reads at a 4 S-byte stride per thread were S PCIe requests each), one
// division per lane.  (Every thread of a wave calls: the row form shuffles.)
template <int W>
__device__ __forceinline__ void set_count_rows_probs_x_item(const CountRowsArgs& a, int64_t t) {
    const int F = a.F, S = a.S, ft = a.ft;
    const int64_t grp = t / W;
    const int j = (int)(t % W);
    const bool row_on = grp < (int64_t)a.n * F;
    const int i = row_on ? (int)(grp / F) : 0, f = row_on ? (int)(grp % F) : 0;
    const int g = row_on ? a.group_idx[i] : 0, tile = f / ft, tl = f % ft;
    const float* in = a.rows + grp * S;
    const int64_t base = ((int64_t)g * F + f) * S;
    float cj = 0.0f;
    if (row_on && j < S) { cj = (float)(int32_t)in[j]; a.counts[base + j] = (int32_t)in[j]; }
    float* out_row = a.probs + base;
    float* out_t = a.probs_t + (((int64_t)tile * (a.Gtot + 1) + g) * S) * ft + tl;                 // (k_probs' tile layout)
    probs_row_x16<W>(j, row_on, [&](int) { return cj; }, a.conc + base, nullptr, S, 0.0, 0.0, a.status,
                     [&](int s, float v) { out_row[s] = v; out_t[(int64_t)s * ft] = v; });
}
template <int W>
__global__ void k_set_count_rows_probs_x(const float* __restrict__ rows /* [n][F][S] */, const int32_t* __restrict__ group_idx,
                                         int32_t* __restrict__ counts, const double* __restrict__ conc, float* __restrict__ probs,
                                         float* __restrict__ probs_t, int n, int F, int S, int Gtot, int ft, int* __restrict__ status) {
    set_count_rows_probs_x_item<W>(CountRowsArgs{rows, group_idx, counts, conc, probs, probs_t, n, F, S, Gtot, ft, status},
                                   (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Several state-setting calls of ONE bind in one launch (sbe_set_slot_delta: the bind cache's revert after a rejected step is
// "old group ids + old count rows", its forward step "new group ids + new source rows"): block ranges [0, n_group_blocks) run
// k_scatter_weight_patterns' grid (x = b % group_gx, y = b / group_gx), the next n_rows_blocks the count-row patch in the form
// `rows_kind` says (0: counts only, 1: + probability rows, 8 / 16: + probability rows by lane groups), the rest the source ingest.
// The three jobs touch disjoint resident arrays.
struct SetterJobs {
    const uint8_t* group_base; ScatterSegs sg; WeightPatternArgs wp; int has_wp; unsigned group_gx, n_group_blocks;
    CountRowsArgs rows; int rows_kind; unsigned n_rows_blocks;
    const uint8_t* src_rows; const int32_t* src_objects; uint8_t* src_id; int src_n, src_F, src_C, src_Fp; int* src_status; unsigned n_src_blocks;
};
static __global__ __launch_bounds__(256) void k_apply_setters(SetterJobs j) {
    unsigned b = blockIdx.x;
    if (b < j.n_group_blocks) {
        const unsigned bx = b % j.group_gx;
        const int y = (int)(b / j.group_gx);
        if (y < j.sg.n) { scatter_segment(j.group_base, j.sg, y, bx, j.group_gx); return; }
        const int i = bx * blockDim.x + threadIdx.x;
        if (j.has_wp && i < j.wp.P * j.wp.F) weight_patterns_item(i, j.wp);
        return;
    }
    b -= j.n_group_blocks;
    if (b < j.n_rows_blocks) {
        const int64_t t = (int64_t)b * blockDim.x + threadIdx.x;
        if (j.rows_kind == 0) set_count_rows_item(j.rows, t);
        else if (j.rows_kind == 8) set_count_rows_probs_x_item<8>(j.rows, t);
        else if (j.rows_kind == 16) set_count_rows_probs_x_item<16>(j.rows, t);
        else set_count_rows_probs_item(j.rows, t);
        return;
    }
    b -= j.n_rows_blocks;
    if (b < j.n_src_blocks) ingest_source_block(j.src_rows, j.src_objects, j.src_id, j.src_n, j.src_F, j.src_C, j.src_Fp, j.src_status, b);
}

// ------------------------------------------------------------------------------------------
// component_likelihood_given_unchanged (operators.py:863-928), count part, from RESIDENT data: the float32 count tables
// the reference builds from the observations that are NOT being resampled,
//   row 0            = sum over members of cluster `i_cluster` outside the subset of [source == 0] * one-hot   (:876-883)
//   row 1 + (gg - K) = counts[gg] - sum over subset objects in confounder group gg of [source == c] * one-hot   (:896-901)
// from the slot's group ids and source (the bound candidate: new clusters, source not yet resampled) and its resident
// counts (still the old state's, which is what sample.feature_counts holds at that point).  `in_subset` [N] bytes.
// Block = (row, 16-feature tile), 16 features x 64 object lanes (1024 threads), LDS histogram [16][S].
// ------------------------------------------------------------------------------------------
constexpr int kUnchangedBlock = 1024;
static __global__ __launch_bounds__(kUnchangedBlock) void k_unchanged_counts(
    const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid /* slot's [C][Np] */,
    const uint8_t* __restrict__ src /* slot's [N][Fp] */, const int32_t* __restrict__ counts /* slot's [Gtot][F][S] */,
    const int32_t* __restrict__ objects /* [n_sub]; may be host-mapped: read ONCE per block */, int n_sub,
    const int32_t* __restrict__ comp_of_group /* [Gtot] */, int i_cluster, int K, int N, int Np, int F, int S, int Fp,
    const double* __restrict__ conc /* [Gtot][F][S] */, const double* __restrict__ unif /* [F][S] */, double temperature,
    double prior_temperature, int* __restrict__ status, float* __restrict__ out /* [1 + Gtot - K][F][S] probability tables */,
    int list_in_lds /* 0: the object list is too long for LDS and is read in place */) {
    // 16 features x 64 object lanes: row 0 walks ALL objects (the cluster's members are found by id), so the object
    // axis gets the lanes; a wave reads four 16-byte runs of four state rows per step
    // The subset's object list lives in host-mapped memory: the block copies it into LDS with one coalesced pass (a
    // walk over it in place would pay a PCIe round trip per step) and, for row 0, turns it into a bitmap over all objects.
    constexpr int FTU = 16, OL = kUnchangedBlock / FTU;
    extern __shared__ int32_t hist[];                                   // [16][S] | object list [n_sub] | bitmap [(N + 31) / 32]
    int32_t* sub_lds = hist + FTU * S;
    const int32_t* sub = list_in_lds ? sub_lds : objects;
    uint32_t* in_subset = reinterpret_cast<uint32_t*>(sub_lds + (list_in_lds ? n_sub : 0));
    const int r = blockIdx.x, f0 = blockIdx.y * FTU;
    for (int i = threadIdx.x; i < FTU * S; i += kUnchangedBlock) hist[i] = 0;
    if (list_in_lds) for (int i = threadIdx.x; i < n_sub; i += kUnchangedBlock) sub_lds[i] = objects[i];
    if (r == 0) for (int i = threadIdx.x; i < (N + 31) / 32; i += kUnchangedBlock) in_subset[i] = 0u;
    __syncthreads();
    if (r == 0) {
        for (int i = threadIdx.x; i < n_sub; i += kUnchangedBlock) atomicOr(&in_subset[sub[i] >> 5], 1u << (sub[i] & 31));
        __syncthreads();
    }
    const int fl = threadIdx.x & (FTU - 1), ol = threadIdx.x / FTU;
    const int f = f0 + fl;
    if (f < F) {
        if (r == 0) {
            const uint16_t want = (uint16_t)i_cluster;                  // component 0: global index = cluster index
            for (int n = ol; n < N; n += 8 * OL) {                      // eight objects per lane and step: their loads overlap
                bool take[8];
                uint8_t x[8], sc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int nn = n + j * OL;
                    take[j] = nn < N && gid[nn] == want && !((in_subset[nn >> 5] >> (nn & 31)) & 1u);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t at = (int64_t)(n + j * OL) * Fp + f;
                    x[j] = take[j] ? state[at] : kNA;
                    sc[j] = take[j] ? src[at] : kNA;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (x[j] != kNA && sc[j] == 0) atomicAdd(&hist[fl * S + x[j]], 1);
            }
        } else {
            const int gg = K + r - 1, c = comp_of_group[gg];
            for (int i = ol; i < n_sub; i += OL) {
                const int n = sub[i];
                if (gid[(int64_t)c * Np + n] != (uint16_t)gg) continue;
                const uint8_t x = state[(int64_t)n * Fp + f];
                if (x != kNA && src[(int64_t)n * Fp + f] == c) atomicAdd(&hist[fl * S + x], -1);
            }
        }
    }
    __syncthreads();
    // conditional_effect_mean (conditionals.py:105-122) of the kept counts, row by row: the cluster's row with the
    // cluster's prior, a confounder group's row with its own (k_probs' arithmetic: probs_row)
    if (threadIdx.x < FTU && f0 + threadIdx.x < F) {
        const int ff = f0 + threadIdx.x;
        const int gg = r == 0 ? i_cluster : K + r - 1;
        const int32_t* base = counts + ((int64_t)gg * F + ff) * S;
        const int32_t* h = hist + threadIdx.x * S;
        float* out_row = out + ((int64_t)r * F + ff) * S;
        probs_row([&](int s) { return (float)((r == 0 ? 0 : base[s]) + h[s]); }, conc + ((int64_t)gg * F + ff) * S,
                  unif + (int64_t)ff * S, S, temperature, prior_temperature, status, [&](int s, float v) { out_row[s] = v; });
    }
}

// ------------------------------------------------------------------------------------------
// component_likelihood_given_unchanged / ClusterOperator.gibbs_sample_source in ONE launch: a block owns
// a 16-feature tile, builds the kept-observations tables of ALL R rows for its features in LDS -- k_unchanged_counts'
// histograms (row 0 over the cluster's members outside the subset, the confounder rows in one pass over the subset) and
// probs_row's normalisation, same operations in the same order -- and then serves the subset's observations of its features
// out of LDS: k_subset_lh's gather (kGibbs = false) or k_given_unchanged_gibbs' resampling (kGibbs = true).  No table in
// global memory, no second launch waiting for the first.  The call's host-mapped input block (object list, table rows,
// has_components rows) is staged into LDS in one PCIe round trip per block.  LDS: tables [R][16][S] (int32 histogram, then
// float32 in place) | staged input block | bitmap.
// ------------------------------------------------------------------------------------------
struct GuFusedArgs {
    const uint8_t* state; const uint16_t* gid; const uint8_t* src; const int32_t* counts;
    const uint32_t* mapped_in;       // the call's host-mapped input block: object list [n_sub] | group_idx [C][n_sub] (group of the
    int in_words;                    //   object within its component, -1 none) | ... | has_components rows (Gibbs form); `in_words`
    int objects_word, group_idx_word, hc_new_word, hc_old_word;   //   32-bit words in all, the arrays at these word offsets
    int table_offsets[kMaxComponents];   // first table row of component c (0 for the cluster, 1 + goff[c] - K)
    const double* conc; const double* unif;
    double temperature, prior_temperature;
    int* status;
    int n_sub, i_cluster, K, N, Np, F, S, C, Fp, R;
    float* out;                      // kGibbs = false: [n_sub][F][C]
    float inv_t; int use_pow;
    // Gibbs form with the count delta of the proposal (sbe_given_unchanged_gibbs_counts; n_touched = 0: not asked for): the
    // subset's GLOBAL group ids in both samples sit in the staged block ([C][n_sub] each, -1 none); the rows of new counts -
    // old counts of the `touched` groups (ascending, host-mapped) go to rows_out [n_touched][F][S]
    int gid_old_word, gid_new_word, n_touched, Gtot;
    const int32_t* touched;
    float* rows_out;
    // ... and the slot itself FOLLOWS the proposal (sbe_given_unchanged_gibbs_apply; follow.counts == nullptr: not asked for):
    // behind the completion flag its counts take the delta, the touched groups' probability rows are rebuilt (follow.probs)
    // and the subset's source rows become the drawn ids (follow.src).  LDS: + touched groups [n_touched] + drawn ids [n_sub][16].
    DeltaFollow follow;
};

template <bool kGibbs>
__global__ __launch_bounds__(kUnchangedBlock) void k_given_unchanged_fused(GuFusedArgs a, GuGibbsArgs gb, uint8_t* __restrict__ src_new,
                                                                           float* __restrict__ sel_new, float* __restrict__ sel_back,
                                                                           DoneSig done = DoneSig{}) {
    constexpr int FTU = 16, OL = kUnchangedBlock / FTU;
    extern __shared__ int32_t lds[];
    const int S = a.S, R = a.R, n_sub = a.n_sub, C = a.C;
    int32_t* hist = lds;                                                 // [R][FTU][S]
    uint32_t* stage = reinterpret_cast<uint32_t*>(hist + R * FTU * S);   // the call's host-mapped input block, word for word
    uint32_t* in_subset = stage + a.in_words;                            // [(N + 31) / 32]
    int32_t* dhist = reinterpret_cast<int32_t*>(in_subset + (a.N + 31) / 32);   // count delta [n_touched][FTU][S] (Gibbs form, if asked)
    int32_t* dpos = dhist + a.n_touched * FTU * S;                       // touched index of a group, -1 [Gtot]
    int32_t* tgl = dpos + a.Gtot;                                        // (following slot) the touched groups [n_touched]
    uint8_t* knew = reinterpret_cast<uint8_t*>(tgl + a.n_touched);       // (following slot) drawn component [n_sub][FTU], 0xFF none
    const bool following = kGibbs && a.n_touched > 0 && a.follow.counts != nullptr;
    const int32_t* sub = reinterpret_cast<const int32_t*>(stage + a.objects_word);       // [n_sub]
    const int32_t* gidx = reinterpret_cast<const int32_t*>(stage + a.group_idx_word);    // [C][n_sub]
    const int f0 = blockIdx.x * FTU;
    // the whole input block (object list, table rows, has_components rows) crosses PCIe in ONE round trip: every thread's
    // loads are issued before the first LDS store waits for one (two loops over two mapped arrays were two round trips)
    for (int i0 = threadIdx.x; i0 < a.in_words; i0 += 4 * kUnchangedBlock) {
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = i0 + j * kUnchangedBlock; v[j] = i < a.in_words ? a.mapped_in[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = i0 + j * kUnchangedBlock; if (i < a.in_words) stage[i] = v[j]; }
    }
    // Gibbs form: the uniform and the old source id of the observation this thread serves first (the consumer loop below:
    // t = threadIdx.x) are asked for now -- they sit in host-mapped staging, and their round trip hides behind the rest
    double z_first = 0.0;
    int id_old_first = 0xFF;
    if constexpr (kGibbs) {
        const int r = threadIdx.x / FTU, ff = f0 + (threadIdx.x & (FTU - 1));
        if (r < n_sub && ff < a.F) { z_first = gb.z[(int64_t)r * a.F + ff]; id_old_first = gb.src_old[(int64_t)r * a.F + ff]; }
    }
    int32_t touched_mine = -1;
    if constexpr (kGibbs) {
        if ((int)threadIdx.x < a.n_touched) touched_mine = a.touched[threadIdx.x];              // (host-mapped: asked for now)
        for (int i = threadIdx.x; i < a.n_touched * FTU * S; i += kUnchangedBlock) dhist[i] = 0;
        if (a.n_touched > 0) for (int i = threadIdx.x; i < a.Gtot; i += kUnchangedBlock) dpos[i] = -1;
    }
    for (int i = threadIdx.x; i < R * FTU * S; i += kUnchangedBlock) hist[i] = 0;
    for (int i = threadIdx.x; i < (a.N + 31) / 32; i += kUnchangedBlock) in_subset[i] = 0u;
    __syncthreads();
    if constexpr (kGibbs) {
        if (touched_mine >= 0) { dpos[touched_mine] = threadIdx.x; if (following) tgl[threadIdx.x] = touched_mine; }
        for (int t = threadIdx.x + kUnchangedBlock; t < a.n_touched; t += kUnchangedBlock) {
            const int g = a.touched[t];
            dpos[g] = t;
            if (following) tgl[t] = g;
        }
    }
    for (int i = threadIdx.x; i < n_sub; i += kUnchangedBlock) atomicOr(&in_subset[sub[i] >> 5], 1u << (sub[i] & 31));
    __syncthreads();
    const int fl = threadIdx.x & (FTU - 1), ol = threadIdx.x / FTU;
    const int f = f0 + fl;
    if (f < a.F) {
        const uint16_t want = (uint16_t)a.i_cluster;
        constexpr int U = 16;                                            // objects per lane and pass: two load levels per pass
        for (int n = ol; n < a.N; n += U * OL) {
            // row 0: members of the cluster outside the subset whose source is the cluster component (operators.py:876-883)
            uint16_t gd[U];
#pragma unroll
            for (int j = 0; j < U; ++j) { const int nn = n + j * OL; gd[j] = nn < a.N ? a.gid[nn] : kNoGroup; }
            if (n == ol) {
                // confounder rows, under the ids' flight: what the subset's objects contribute to their groups' counts comes
                // off (operators.py:896-901)
                for (int i = ol; i < n_sub; i += OL) {
                    const int64_t at = (int64_t)sub[i] * a.Fp + f;
                    const uint8_t x = a.state[at], sc = a.src[at];
                    if (x == kNA || sc == 0 || sc >= C) continue;
                    const int g = gidx[sc * n_sub + i];
                    if (g >= 0) atomicAdd(&hist[((a.table_offsets[sc] + g) * FTU + fl) * S + x], -1);
                }
            }
            uint8_t x[U], sc[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int nn = n + j * OL;
                const bool take = nn < a.N && gd[j] == want && !((in_subset[nn >> 5] >> (nn & 31)) & 1u);
                const int64_t at = (int64_t)nn * a.Fp + f;
                x[j] = take ? a.state[at] : kNA;
                sc[j] = take ? a.src[at] : kNA;
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (x[j] != kNA && sc[j] == 0) atomicAdd(&hist[fl * S + x[j]], 1);
        }
    }
    __syncthreads();
    // conditional_effect_mean (conditionals.py:105-122) of the kept counts, in place: sixteen lanes per (row, feature) when
    // S <= 16 (probs_row_x16), else one thread per row
    auto rows_by_lane_groups = [&](auto width) {
        constexpr int W = decltype(width)::value;
        for (int row0 = 0; row0 < R * FTU; row0 += kUnchangedBlock / W) {
            const int row = row0 + (int)threadIdx.x / W, j = threadIdx.x & (W - 1);
            const int r = row / FTU, tf = row % FTU, ff = f0 + tf;
            const bool row_on = row < R * FTU && ff < a.F;
            const int gg = r == 0 ? a.i_cluster : a.K + r - 1;       // rows 1.. are the confounder groups in global order
            const int64_t at = row_on ? ((int64_t)gg * a.F + ff) * S : 0;
            int32_t* h = hist + (row_on ? (r * FTU + tf) * S : 0);
            probs_row_x16<W>(j, row_on, [&](int s) { return (float)((r == 0 ? 0 : a.counts[at + s]) + h[s]); }, a.conc + at,
                             a.unif + (row_on ? (int64_t)ff * S : 0), S, a.temperature, a.prior_temperature, a.status,
                             [&](int s, float v) { h[s] = __float_as_int(v); });
        }
    };
    if (S <= 8) rows_by_lane_groups(std::integral_constant<int, 8>{});
    else if (S <= 16) rows_by_lane_groups(std::integral_constant<int, 16>{});
    else {
        for (int t = threadIdx.x; t < R * FTU; t += kUnchangedBlock) {
            const int r = t / FTU, tf = t % FTU, ff = f0 + tf;
            if (ff >= a.F) continue;
            const int gg = r == 0 ? a.i_cluster : a.K + r - 1;
            const int32_t* base = a.counts + ((int64_t)gg * a.F + ff) * S;
            int32_t* h = hist + (r * FTU + tf) * S;
            probs_row([&](int s) { return (float)((r == 0 ? 0 : base[s]) + h[s]); }, a.conc + ((int64_t)gg * a.F + ff) * S,
                      a.unif + (int64_t)ff * S, S, a.temperature, a.prior_temperature, a.status,
                      [&](int s, float v) { h[s] = __float_as_int(v); });
        }
    }
    __syncthreads();
    const float* tab = reinterpret_cast<const float*>(hist);
    GuGibbsArgs g2 = gb;
    if constexpr (kGibbs) {                                              // (the has_components rows: out of the staged block)
        g2.hc_new = reinterpret_cast<const uint8_t*>(stage + a.hc_new_word);
        g2.hc_old = reinterpret_cast<const uint8_t*>(stage + a.hc_old_word);
    }
    for (int t = threadIdx.x; t < n_sub * FTU; t += kUnchangedBlock) {
        const int r = t / FTU, tf = t % FTU, ff = f0 + tf;
        if (ff >= a.F) continue;
        const uint8_t x = a.state[(int64_t)sub[r] * a.Fp + ff];
        const int64_t i = (int64_t)r * a.F + ff;
        if constexpr (kGibbs) {
            const bool first = t == (int)threadIdx.x;
            const int id_old = first ? id_old_first : (int)gb.src_old[i];
            const int k = gu_gibbs_obs(g2, i, r, ff, x, first ? z_first : gb.z[i], id_old,
                                       [&](int c) { return gidx[c * n_sub + r]; },
                                       [&](int c, int g) { return tab[((a.table_offsets[c] + g) * FTU + tf) * S + x]; },
                                       src_new, sel_new, sel_back, a.status);
            if (following) knew[r * FTU + tf] = k >= 0 ? (uint8_t)k : (uint8_t)0xFF;
            if (a.n_touched > 0 && k >= 0) {
                // update_feature_counts (counts.py:55-95) of the proposal, this observation's share: one more in its NEW group of
                // the drawn component, one less in its OLD group of the old source component
                const int32_t* gid_new = reinterpret_cast<const int32_t*>(stage + a.gid_new_word);
                const int32_t* gid_old = reinterpret_cast<const int32_t*>(stage + a.gid_old_word);
                const int gn = gid_new[k * n_sub + r];
                if (gn >= 0 && dpos[gn] >= 0) atomicAdd(&dhist[(dpos[gn] * FTU + tf) * S + x], 1);
                if (id_old < C) {
                    const int go = gid_old[id_old * n_sub + r];
                    if (go >= 0 && dpos[go] >= 0) atomicAdd(&dhist[(dpos[go] * FTU + tf) * S + x], -1);
                }
            }
        } else {
            float* o = a.out + i * C;
            for (int c = 0; c < C; ++c) {
                float v = 1.0f;
                if (x != kNA) {
                    const int g = gidx[c * n_sub + r];
                    v = g < 0 ? 0.0f : tab[((a.table_offsets[c] + g) * FTU + tf) * S + x];
                }
                o[c] = a.use_pow ? lib_powf(v, a.inv_t) : v;
            }
        }
    }
    if constexpr (kGibbs) {
        if (a.n_touched > 0) {
            __syncthreads();
            for (int e = threadIdx.x; e < a.n_touched * FTU * S; e += kUnchangedBlock) {
                const int t = e / (FTU * S), q = e % (FTU * S), ff = f0 + q / S;
                if (ff < a.F) a.rows_out[((int64_t)t * a.F + ff) * S + q % S] = (float)dhist[e];
            }
        }
    }
    signal_done(done);
    if constexpr (kGibbs) {
        if (following) {                                  // (block-uniform; nothing below reads the mapped block)
            if (!done.flag) __syncthreads();              // (signal_done's barrier made knew / dhist complete otherwise)
            if (a.follow.src) {
                for (int t = threadIdx.x; t < n_sub * FTU; t += kUnchangedBlock) {
                    const int r = t / FTU, tf = t % FTU;
                    if (f0 + tf < a.F) a.follow.src[(int64_t)sub[r] * a.Fp + f0 + tf] = knew[t];
                }
            }
            follow_tile_rows(a.follow, dhist, tgl, a.n_touched, f0, a.F, S, a.Gtot);
        }
    }
}

}  // namespace sbe
