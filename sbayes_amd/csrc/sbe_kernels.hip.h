// sbe_kernels.hip.h -- gfx950 (MI355X / CDNA4) device code of the sBayes likelihood engine.
//
// Everything here is a categorical gather-multiply-reduce over the objects x features x states
// block: HBM/L2/LDS-bound byte and table traffic, no MFMA (SURVEY.md 8(d) "Bound").
// Wave = 64 lanes; blocks are 256 threads (4 waves, one per SIMD of a CU).
//
// Numerics contract: probability tables and normalised weights are float32 and
// are produced with NumPy's exact operation order (pairwise reduction order of
// numpy/_core/src/umath/loops_utils.h.src, IEEE division, one rounding to float32), so they are
// bit-identical to the reference's; everything downstream of them is float64.  The file is
// compiled with -ffp-contract=off: NumPy never fuses a multiply into an add.
#pragma once
#include "sbe_device_common.hip.h"
#include "sbe_mixture.hip.h"

namespace sbe {

// ---- self-test kernels of the device routines (sbe_test_lgamma / _fast_log / _tab_log) ----
static __global__ void k_test_lgamma(const double* __restrict__ in, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sbe_lgamma_pos(in[i]);
}

static __global__ void k_test_fast_log(const double* __restrict__ in, double* __restrict__ out_fast,
                                double* __restrict__ out_lib, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_fast[i] = fast_log_pos(in[i]);
    out_lib[i] = log(in[i]);
}

// Round trip of a host-synchronous call with nothing to do (sbe_test_roundtrip: the floor every drop-in call pays):
// mode bit 0: read one word of the host-mapped input block, bit 1: store one double to the host-mapped result block.
static __global__ void k_test_roundtrip(const int32_t* __restrict__ mapped_in, double* __restrict__ mapped_out, int mode, DoneSig done) {
    int v = 0;
    if ((mode & 1) && threadIdx.x == 0) v = mapped_in[0];
    if ((mode & 2) && threadIdx.x == 0) mapped_out[blockIdx.x] = (double)v;
    signal_done(done);
}

static __global__ void k_test_tab_log(const double* __restrict__ in, const double2* __restrict__ logtab,
                               double* __restrict__ out, int n) {
    __shared__ double2 tab[kLogTabEntries];
    if (threadIdx.x < kLogTabEntries) tab[threadIdx.x] = logtab[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = tab_log_pos(in[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double2*)tab);
}

// ------------------------------------------------------------------------------------------
// K0: one-hot ingest.  raw [N][F*S] (any non-zero byte = True) -> normalised 0/1 copy with a
// 16-byte-aligned row pitch, packed state index [N][Fp] (0xFF = NA), validation counters.
// ------------------------------------------------------------------------------------------
static __global__ void k_ingest_onehot(const uint8_t* __restrict__ raw, uint8_t* __restrict__ onehot,
                                uint8_t* __restrict__ state, uint8_t* __restrict__ state_q,
                                uint16_t* __restrict__ state_h /* or null */, int N, int F,
                                int S, int rs_pitch, int Fp, int Fq, int* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // observation index
    int multi = 0, na = 0;
    if (i < (int64_t)N * F) {
        const int n = (int)(i / F), f = (int)(i % F);
        const uint8_t* src = raw + (int64_t)i * S;
        uint8_t* dst = onehot + (int64_t)n * rs_pitch + (int64_t)f * S;
        int x = -1, cnt = 0;
        for (int s = 0; s < S; ++s) {
            const uint8_t b = src[s] != 0;
            dst[s] = b;
            if (b) { x = s; ++cnt; }
        }
        const uint8_t xb = cnt == 0 ? kNA : (uint8_t)x;
        state[(int64_t)n * Fp + f] = xb;
        // object-quad interleaved copy: dword (n/4, f) holds objects 4*(n/4) .. +3 of feature f
        // (NA and padding byte here is S, not 0xFF: it indexes the zero row of the group-tuple log table
        //  directly; every reader treats any byte >= S as NA)
        state_q[((int64_t)(n >> 2) * Fq + f) * 4 + (n & 3)] = cnt == 0 ? (uint8_t)S : xb;
        // k_mixture_tuple64's stream: byte offset of the observation inside a tuple's [S+1][64] f64 block
        if (state_h) state_h[((int64_t)(n >> 2) * Fq + f) * 4 + (n & 3)] = (uint16_t)((cnt == 0 ? S : x) * 512 + (f & 63) * 8);
        multi = cnt > 1;
        na = cnt == 0;
    }
    const int m = __popcll(__ballot(multi)), a = __popcll(__ballot(na));
    if ((threadIdx.x & 63) == 0) {
        if (m) atomicAdd(&status[ST_MULTI_STATE], m);
        if (a) atomicAdd(&status[ST_NA_COUNT], a);
    }
}

// NA / padding image of the prepared stream: every entry points at the zero row (x = S) of its column.
static __global__ void k_init_state_h(uint16_t* __restrict__ state_h, int64_t n_entries /* NQ*Fq */, int Fq, int S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const uint16_t v = (uint16_t)(S * 512 + (int)((i % Fq) & 63) * 8);
    uint16_t* o = state_h + i * 4;
    o[0] = v; o[1] = v; o[2] = v; o[3] = v;
}

// Several small host arrays to their resident places in ONE launch: the arrays are staged back to back in the mapped
// ring (`base`, read over PCIe element-parallel), segment y goes to sg.dst[y].  Words when everything is 4-byte aligned.
struct ScatterSegs { uint8_t* dst[8]; uint32_t off[8]; uint32_t bytes[8]; int n; };
__device__ __forceinline__ void scatter_segment(const uint8_t* __restrict__ base, const ScatterSegs& sg, int y, uint32_t bx, uint32_t gx) {
    const uint8_t* src = base + sg.off[y];
    uint8_t* dst = sg.dst[y];
    const uint32_t nb = sg.bytes[y], stride = gx * blockDim.x, i0 = bx * blockDim.x + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src | nb) & 3) == 0) {
        for (uint32_t j = i0; j < nb / 4; j += stride) reinterpret_cast<uint32_t*>(dst)[j] = reinterpret_cast<const uint32_t*>(src)[j];
    } else {
        for (uint32_t j = i0; j < nb; j += stride) dst[j] = src[j];
    }
}
static __global__ void k_scatter_bytes(const uint8_t* __restrict__ base, ScatterSegs sg) {
    if ((int)blockIdx.y < sg.n) scatter_segment(base, sg, blockIdx.y, blockIdx.x, gridDim.x);
}

// K0b: source ingest.  bool rows [rows][F][C] -> component id per observation (0xFF = none).
// `objects` == nullptr: row r is object r.
__device__ __forceinline__ void ingest_source_block(const uint8_t* __restrict__ rows, const int32_t* __restrict__ objects,
                                                    uint8_t* __restrict__ src_id, int n_rows, int F, int C, int Fp,
                                                    int* __restrict__ status, int64_t block) {
    const int64_t i = block * blockDim.x + threadIdx.x;
    int multi = 0;
    if (i < (int64_t)n_rows * F) {
        const int r = (int)(i / F), f = (int)(i % F);
        const int n = objects ? objects[r] : r;
        const uint8_t* p = rows + (int64_t)i * C;
        int id = kNA, cnt = 0;
        for (int c = 0; c < C; ++c)
            if (p[c]) { id = c; ++cnt; }
        src_id[(int64_t)n * Fp + f] = (uint8_t)id;
        multi = cnt > 1;
    }
    const int m = __popcll(__ballot(multi));
    if ((threadIdx.x & 63) == 0 && m) raise_status(status, ST_MULTI_SOURCE, m);
}
static __global__ void k_ingest_source(const uint8_t* __restrict__ rows, const int32_t* __restrict__ objects,
                                uint8_t* __restrict__ src_id, int n_rows, int F, int C, int Fp,
                                int* __restrict__ status) {
    ingest_source_block(rows, objects, src_id, n_rows, F, C, Fp, status, blockIdx.x);
}

// Inverse of K0b for the listed objects: component id -> bool row [F][C] (all False for 0xFF).
static __global__ void k_expand_source(const uint8_t* __restrict__ src_id, const int32_t* __restrict__ objects,
                                uint8_t* __restrict__ rows, int n_rows, int F, int C, int Fp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rows * F) return;
    const int r = (int)(i / F), f = (int)(i % F);
    const int id = src_id[(int64_t)objects[r] * Fp + f];
    uint8_t* p = rows + i * C;
    for (int c = 0; c < C; ++c) p[c] = c == id;
}

// ------------------------------------------------------------------------------------------
// a9: feature counts (counts.py:10-52, 55-95).
// counts[gg][f][s] (int32, gg = global group index over all components) accumulates
//   sign_a * [object in A-state] + sign_b * [object in B-state]
// over the listed objects.  Block = feature tile (ft features) x all listed objects of its
// chunk; privatised LDS histogram hist[Gtot][ft][S]; lanes of a wave sit on different features
// so LDS atomics only collide across object lanes.  Non-zero cells are flushed with one global
// atomic each; `changed[gg]` is raised for groups with a non-zero net delta in this block.
// With a single object chunk (delta updates) that is exactly the reference's
// np.any(diff != 0) per group (state.py:349-350).
// ------------------------------------------------------------------------------------------
struct CountSide {
    const uint16_t* gid;   // [C][N] global group index or 0xFFFF
    const uint8_t* src;    // [N][Fp] source component id
    int sign;              // 0 = side unused
};

static __global__ __launch_bounds__(kBlock) void k_counts(
    const uint8_t* __restrict__ state, CountSide A, CountSide B, const int32_t* __restrict__ objects,
    int n_listed, int objs_per_chunk, int Np, int F, int S, int C, int Fp, int Gtot, int ft,
    int32_t* __restrict__ counts, uint8_t* __restrict__ changed) {
    extern __shared__ int32_t hist[];
    const int f0 = blockIdx.x * ft;
    const int cells = Gtot * ft * S;
    for (int i = threadIdx.x; i < cells; i += kBlock) hist[i] = 0;
    __syncthreads();

    const int fl = threadIdx.x % ft, ol = threadIdx.x / ft, olanes = kBlock / ft;
    const int f = f0 + fl;
    const int i0 = blockIdx.y * objs_per_chunk;
    const int i1 = min(n_listed, i0 + objs_per_chunk);
    if (f < F) {
        for (int i = i0 + ol; i < i1; i += olanes) {
            const int n = objects ? objects[i] : i;
            const uint8_t x = state[(int64_t)n * Fp + f];
            if (x == kNA) continue;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const CountSide& sd = side ? B : A;
                if (sd.sign == 0) continue;
                const uint8_t c = sd.src[(int64_t)n * Fp + f];
                if (c >= C) continue;
                const uint16_t gg = sd.gid[(int64_t)c * Np + n];
                if (gg == kNoGroup) continue;
                atomicAdd(&hist[((int)gg * ft + fl) * S + x], sd.sign);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += kBlock) {
        const int h = hist[i];
        if (h != 0) {
            const int gg = i / (ft * S), r = i % (ft * S);
            const int ff = f0 + r / S, s = r % S;
            atomicAdd(&counts[((int64_t)gg * F + ff) * S + s], h);
            if (changed) changed[gg] = 1;
        }
    }
}

// Stateless compute_effect_counts (counts.py:10-32): arbitrary bool group matrix [G][N]
// (overlapping groups count the object in each, as the reference's per-group loop does).
static __global__ __launch_bounds__(kBlock) void k_effect_counts(
    const uint8_t* __restrict__ state, const uint8_t* __restrict__ groups, const uint8_t* __restrict__ mask,
    const int32_t* __restrict__ objects, int n_listed, int objs_per_chunk, int N, int F, int S, int Fp, int G,
    int ft, int32_t* __restrict__ counts) {
    extern __shared__ int32_t hist[];
    const int f0 = blockIdx.x * ft;
    const int cells = G * ft * S;
    for (int i = threadIdx.x; i < cells; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int fl = threadIdx.x % ft, ol = threadIdx.x / ft, olanes = kBlock / ft;
    const int f = f0 + fl;
    const int i0 = blockIdx.y * objs_per_chunk;
    const int i1 = min(n_listed, i0 + objs_per_chunk);
    if (f < F) {
        for (int i = i0 + ol; i < i1; i += olanes) {
            const int n = objects ? objects[i] : i;
            const uint8_t x = state[(int64_t)n * Fp + f];
            if (x == kNA || !mask[(int64_t)n * F + f]) continue;
            for (int g = 0; g < G; ++g)
                if (groups[(int64_t)g * N + n]) atomicAdd(&hist[(g * ft + fl) * S + x], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += kBlock) {
        const int h = hist[i];
        if (h != 0) {
            const int g = i / (ft * S), r = i % (ft * S);
            atomicAdd(&counts[((int64_t)g * F + f0 + r / S) * S + r % S], h);
        }
    }
}

// Fallback for histograms that do not fit in LDS: one global atomic per observation.
static __global__ void k_counts_global(const uint8_t* __restrict__ state, CountSide A, CountSide B,
                                const int32_t* __restrict__ objects, int n_listed, int Np, int F, int S,
                                int C, int Fp, int32_t* __restrict__ counts, uint8_t* __restrict__ changed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_listed * F) return;
    const int r = (int)(i / F), f = (int)(i % F);
    const int n = objects ? objects[r] : r;
    const uint8_t x = state[(int64_t)n * Fp + f];
    if (x == kNA) return;
    for (int side = 0; side < 2; ++side) {
        const CountSide& sd = side ? B : A;
        if (sd.sign == 0) continue;
        const uint8_t c = sd.src[(int64_t)n * Fp + f];
        if (c >= C) continue;
        const uint16_t gg = sd.gid[(int64_t)c * Np + n];
        if (gg == kNoGroup) continue;
        atomicAdd(&counts[((int64_t)gg * F + f) * S + x], sd.sign);
        if (changed) changed[gg] = 1;   // over-approximation (no cancellation visible here)
    }
}

// source_is_component bool [N][F] -> src id 0 / 0xFF in the [N][Fp] layout (stateless a9 call)
static __global__ void k_mask_to_src(const uint8_t* __restrict__ mask, uint8_t* __restrict__ src, int N, int F, int Fp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * F) return;
    const int n = (int)(i / F), f = (int)(i % F);
    src[(int64_t)n * Fp + f] = mask[i] ? 0 : kNA;
}

static __global__ void k_i32_to_f32(const int32_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
// the same conversion as a call's LAST kernel: any grid (the blocks stride over the array), completion by flag
static __global__ void k_i32_to_f32_done(const int32_t* __restrict__ in, float* __restrict__ out, int64_t n, DoneSig done) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
    signal_done(done);
}
static __global__ void k_f32_to_i32(const float* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

// ------------------------------------------------------------------------------------------
// a4 / a10: probs = normalize(counts/T + prior') -> float32   (util.py:990-1007,
// conditionals.py:105-122, 175-179).  One thread per (group, feature) row of S states.
//   counts are float32 in the reference (counts.py:20): counts / T is a float32 division
//   (NumPy: float32 array / Python float), then + float64 prior -> float64.
// ------------------------------------------------------------------------------------------
// One table row of normalize(counts / T + prior') -> float32 (util.py:990-1007, conditionals.py:105-122): the S states
// summed in NumPy's pairwise order.  `cnt(s)` = the count as the reference's float32; `unif_row` null: prior untempered.
// The two divisions that temper a row are exact no-ops at temperature 1 (x / 1 = x in IEEE arithmetic) and are skipped
// there -- a correctly rounded fp64 division is a chain of a dozen dependent instructions, and a row has 2 S + S of them
// otherwise; every table kernel is a latency chain of such rows.  The additions stay (u + (a - u) is not always a).
// S <= CAP: the row's S posterior values are computed once into registers -- the 3 S loads issued together, the tempering
// divisions independent of each other -- summed by np_sum_regs and divided; beyond CAP the general form (values recomputed
// per pass, NumPy's recursion for S > 128).  Same operations on the same operands in the same order either way.
template <int CAP = 16, class GetCount, class Emit>
__device__ __forceinline__ void probs_row(GetCount cnt, const double* __restrict__ conc_row, const double* __restrict__ unif_row,
                                          int S, double temperature, double prior_temperature, int* __restrict__ status, Emit emit) {
    const bool tempered = temperature > 0.0 && temperature != 1.0;
    const bool prior_shaped = prior_temperature > 0.0 && unif_row != nullptr;
    const bool prior_tempered = prior_shaped && prior_temperature != 1.0;
    const float t32 = (float)temperature;
    if (S <= CAP) {
        float c[CAP];
        double a[CAP], u[CAP], pv[CAP];
#pragma unroll
        for (int s = 0; s < CAP; ++s) {
            const bool on = s < S;
            c[s] = on ? cnt(s) : 0.0f;
            a[s] = on ? conc_row[s] : 0.0;
            u[s] = (on && prior_shaped) ? unif_row[s] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < CAP; ++s) {
            float cs = c[s];
            if (tempered) cs = cs / t32;
            double as = a[s];
            if (prior_shaped) {
                const double d = as - u[s];
                as = u[s] + (prior_tempered ? d / prior_temperature : d);
            }
            pv[s] = (double)cs + as;
        }
        const double total = np_sum_regs<double, CAP>(pv, S);
        if (status && !(total > 0.0)) raise_status(status, ST_BAD_NORMALIZE, 1);     // (null: another block reports the row)
#pragma unroll
        for (int s = 0; s < CAP; ++s) if (s < S) emit(s, (float)(pv[s] / total));
        return;
    }
    auto post = [&](int s) -> double {
        float c = cnt(s);
        if (tempered) c = c / t32;
        double a = conc_row[s];
        if (prior_shaped) {
            const double u = unif_row[s];
            const double d = a - u;
            a = u + (prior_tempered ? d / prior_temperature : d);
        }
        return (double)c + a;
    };
    const double total = np_pairwise_sum<double>(post, S);
    if (status && !(total > 0.0)) raise_status(status, ST_BAD_NORMALIZE, 1);
    for (int s = 0; s < S; ++s) emit(s, (float)(post(s) / total));
}

// The same row built by W = 16 (or 8) consecutive lanes together (S <= W): lane j of the group loads and tempers state j -- the
// group's loads are one coalesced access per operand instead of S strided ones per thread -- every lane gathers the sixteen
// posterior values by shuffles and sums them itself in NumPy's order (np_sum_regs: the same additions on the same values),
// and there is ONE division per lane instead of S in a row per thread (a correctly rounded fp64 division is a dependent
// chain of a dozen instructions: ten of them back to back were half of a 4 us row; profiles/r4/gu_kernel_clock.log).
// Convergent: every lane of the wave calls it, `row_on` says whether its group has a row.  Same bits as probs_row.
template <int W = 16, class GetCount, class Emit>
__device__ __forceinline__ void probs_row_x16(int j, bool row_on, GetCount cnt, const double* __restrict__ conc_row,
                                              const double* __restrict__ unif_row, int S, double temperature, double prior_temperature,
                                              int* __restrict__ status, Emit emit) {
    const bool tempered = temperature > 0.0 && temperature != 1.0;
    const bool prior_shaped = prior_temperature > 0.0 && unif_row != nullptr;
    const bool prior_tempered = prior_shaped && prior_temperature != 1.0;
    const bool on = row_on && j < S;
    float c = on ? cnt(j) : 0.0f;
    double a = on ? conc_row[j] : 0.0;
    const double u = (on && prior_shaped) ? unif_row[j] : 0.0;
    if (tempered) c = c / (float)temperature;
    if (prior_shaped) {
        const double d = a - u;
        a = u + (prior_tempered ? d / prior_temperature : d);
    }
    const double pv = (double)c + a;
    double v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = __shfl(pv, k, W);
    const double total = np_sum_regs<double, W>(v, S);
    if (row_on && j == 0 && status && !(total > 0.0)) raise_status(status, ST_BAD_NORMALIZE, 1);
    if (on) emit(j, (float)(pv / total));
}

// The inputs of one tempered table row, resident in device memory: what a consumer kernel needs to build the rows it
// reads itself -- once per block, into LDS -- instead of waiting for a k_probs launch in front of it (one launch per
// drop-in call: VERDICT r3 item 4).
struct RowSource {
    const int32_t* counts;           // [F][S] integer counts of the row's group (the slot's resident table row)
    const double* conc;              // [F][S] concentration of that group
};

template <class TC>
__global__ void k_probs(const TC* __restrict__ counts, const double* __restrict__ conc,
                        const double* __restrict__ unif /* [F][S] or null */, float* __restrict__ probs,
                        int g_lo, int g_hi, int F, int S, double temperature, double prior_temperature,
                        int conc_per_group, int* __restrict__ status, int64_t out_shift = 0,
                        float* __restrict__ probs_t = nullptr /* tile-transposed copy [n_ftiles][Gtot+1][S][ft], or null */,
                        int Gtot = 0, int ft = 1) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_rows = (int64_t)(g_hi - g_lo) * F;
    if (row >= n_rows) return;
    const int64_t base = ((int64_t)g_lo * F + row) * S;
    const int f = (int)(row % F);
    const int g = g_lo + (int)(row / F), tile = f / ft, tl = f % ft;
    float* out_row = probs + base + out_shift;                                 // (out_shift: rows of a scratch table)
    float* out_t = probs_t ? probs_t + (((int64_t)tile * (Gtot + 1) + g) * S) * ft + tl : nullptr;   // (k_tile_probs' layout)
    probs_row([&](int s) { return (float)counts[base + s]; }, conc_per_group ? conc + base : conc + (int64_t)f * S,
              unif ? unif + (int64_t)f * S : nullptr, S, temperature, prior_temperature, status,
              [&](int s, float v) { out_row[s] = v; if (out_t) out_t[(int64_t)s * ft] = v; });
}

// ------------------------------------------------------------------------------------------
// a5: per-pattern normalised weights (likelihood.py:171-190).  One thread per (pattern, f).
//   w = pattern * weights (bool x float32 -> float32);  w /= sum_c w   (float32, NumPy order)
// ------------------------------------------------------------------------------------------
// Optional extras of the slot form (one launch per sbe_set_weights): `weights_keep` = the slot's resident [F][C] copy
// of `weights` (which may then be host-mapped staging memory), `wpat_t` = the tile-transposed float64 copy
// [n_ftiles][Pmax][C][ft], exact widening (what the fused kernels read; padding features stay zero from creation).
struct WeightPatternArgs {
    const float* weights;          // [F][C]
    const uint32_t* pattern_bits;  // [P]
    float* wpat;                   // [P][F][C]
    float* weights_keep;           // or nullptr
    double* wpat_t;                // or nullptr
    int P, F, C, Pmax, ft;
};
__device__ __forceinline__ void weight_patterns_item(int i, const WeightPatternArgs& a) {
    const int F = a.F, C = a.C;
    const int p = i / F, f = i % F;
    const uint32_t bits = a.pattern_bits[p];
    float w[kMaxComponents];
    for (int c = 0; c < C; ++c) w[c] = a.weights[(int64_t)f * C + c];
    if (a.weights_keep && p == 0) for (int c = 0; c < C; ++c) a.weights_keep[(int64_t)f * C + c] = w[c];
    auto masked = [&](int c) -> float { return ((bits >> c) & 1u) ? w[c] : 0.0f * w[c]; };
    const float total = np_pairwise_sum<float>(masked, C);
    float* out = a.wpat + ((int64_t)p * F + f) * C;
    double* out_t = a.wpat_t ? a.wpat_t + (((int64_t)(f / a.ft) * a.Pmax + p) * C) * a.ft + f % a.ft : nullptr;
    for (int c = 0; c < C; ++c) {
        const float v = masked(c) / total;
        out[c] = v;
        if (out_t) out_t[(int64_t)c * a.ft] = (double)v;
    }
}
static __global__ void k_weight_patterns(const float* __restrict__ weights /* [F][C] */,
                                  const uint32_t* __restrict__ pattern_bits /* [P] */,
                                  float* __restrict__ wpat /* [P][F][C] */, int P, int F, int C,
                                  float* __restrict__ weights_keep = nullptr, double* __restrict__ wpat_t = nullptr,
                                  int Pmax = 0, int ft = 1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * F) return;
    weight_patterns_item(i, WeightPatternArgs{weights, pattern_bits, wpat, weights_keep, wpat_t, P, F, C, Pmax, ft});
}

// A slot's new group ids in ONE launch (sbe_set_groups and friends): rows y < sg.n of the grid put the staged arrays (the
// ids, the pattern id per object, the pattern bits, the group-tuple tables) in their resident places, row sg.n computes
// the per-pattern normalised weights -- k_weight_patterns' arithmetic -- with the pattern bits read from the STAGED copy,
// so the two halves do not depend on each other.
static __global__ void k_scatter_weight_patterns(const uint8_t* __restrict__ base, ScatterSegs sg, WeightPatternArgs wp) {
    const int y = blockIdx.y;
    if (y < sg.n) { scatter_segment(base, sg, y, blockIdx.x, gridDim.x); return; }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < wp.P * wp.F) weight_patterns_item(i, wp);
}

// normalize_weights (likelihood.py:171-190), stateless, ROW form: out[n][f][:] = has_components[n] * weights[f] / sum --
// k_weight_patterns' arithmetic evaluated per row instead of per distinct pattern (the same operations on the same
// operands: the same bits), so nobody sorts patterns and the call is ONE launch.  Block = 8 rows; the [F][C] weights
// and the block's has_components rows are read once into LDS (both may live in host-mapped staging memory).
constexpr int kNwRows = 8;
static __global__ void k_normalize_weight_rows(const float* __restrict__ weights /* [F][C] */,
                                        const uint8_t* __restrict__ has_components /* [N][C] */,
                                        float* __restrict__ out /* [N][F][C] */, int N, int F, int C, int stage_weights,
                                        DoneSig done = DoneSig{}) {
    extern __shared__ __align__(16) float nw_lds[];          // [F][C] weights (stage_weights == 0: read in place from
    __shared__ uint32_t bits[kNwRows];                       //  device memory -- tables beyond the LDS budget)
    const int n0 = blockIdx.x * kNwRows;
    // Both inputs may live in host-mapped staging memory: every read of the block is asked for before the first one is
    // waited for (the rows' flags, then the weights as 16-byte words, four per thread and pass), one trip over the link.
    uint8_t hb[kMaxComponents];
    {
        const int n = n0 + (int)threadIdx.x;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c)
            hb[c] = (threadIdx.x < kNwRows && n < N && c < C) ? has_components[(int64_t)n * C + c] : (uint8_t)0;
    }
    if (stage_weights) {
        const int nw = F * C, n4 = nw >> 2;
        const float4* w4 = reinterpret_cast<const float4*>(weights);   // (ring slots and the scratch base are 64-byte aligned)
        const float tail = (int)threadIdx.x < (nw & 3) ? weights[(n4 << 2) + threadIdx.x] : 0.0f;
        for (int base = 0; base < n4; base += 4 * (int)blockDim.x) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * (int)blockDim.x + (int)threadIdx.x; if (i < n4) v[u] = w4[i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * (int)blockDim.x + (int)threadIdx.x; if (i < n4) reinterpret_cast<float4*>(nw_lds)[i] = v[u]; }
        }
        if ((int)threadIdx.x < (nw & 3)) nw_lds[(n4 << 2) + threadIdx.x] = tail;
    }
    if (threadIdx.x < kNwRows) {
        uint32_t b = 0;
#pragma unroll
        for (int c = 0; c < kMaxComponents; ++c) if (hb[c]) b |= 1u << c;
        bits[threadIdx.x] = b;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kNwRows * F; i += blockDim.x) {
        const int r = i / F, f = i - r * F, n = n0 + r;
        if (n >= N) break;
        const uint32_t b = bits[r];
        const float* w = (stage_weights ? nw_lds : weights) + f * C;
        auto masked = [&](int c) -> float { return ((b >> c) & 1u) ? w[c] : 0.0f * w[c]; };
        const float total = np_pairwise_sum<float>(masked, C);
        float* o = out + ((int64_t)n * F + f) * C;
        for (int c = 0; c < C; ++c) o[c] = masked(c) / total;
    }
    signal_done(done);
}

static __global__ void k_expand_weights(const float* __restrict__ wpat, const uint8_t* __restrict__ pid,
                                 float* __restrict__ out /* [N][F][C] */, int N, int F, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * F * C) return;
    const int n = (int)(i / ((int64_t)F * C));
    const int64_t r = i % ((int64_t)F * C);
    out[i] = wpat[(int64_t)pid[n] * F * C + r];
}

// ------------------------------------------------------------------------------------------
// a1: compute_component_likelihood (likelihood.py:104-133), dense device side.
//   sel[n] >= 0 : row n takes table sel[n];  -1 : row <- 0 (object in no group);
//   -2 : row untouched (member of an unchanged group) -- the host scatter skips it.
// One-hot => the float32 sum over states has exactly one non-zero term: value = probs[.][f][x]
// converted exactly to float64; NA rows sum to 0.
// ------------------------------------------------------------------------------------------
template <class TP>
__device__ __forceinline__ double component_lh_value(const uint8_t* __restrict__ state, const TP* __restrict__ probs,
                                                    const int32_t* __restrict__ sel, int64_t i, int F, int S, int Fp, double na_value) {
    const int n = (int)(i / F), f = (int)(i % F);
    const int g = sel[n];
    const uint8_t x = state[(int64_t)n * Fp + f];
    if (x == kNA) return na_value;        // 0 for the literal a1 contract; 1 when serving likelihood_per_component
    return g >= 0 ? (double)probs[((int64_t)g * F + f) * S + x] : 0.0;
}

// two consecutive output elements per thread, stored as one 16-byte word (tools/d2h_probe.hip: 16-byte stores into
// host-mapped memory move 41-47 GB/s, 8-byte stores 36-43)
template <class TP>
__global__ void k_component_lh(const uint8_t* __restrict__ state, const TP* __restrict__ probs,
                               const int32_t* __restrict__ sel, double* __restrict__ out /* [N][F] */,
                               int N, int F, int S, int Fp, double na_value, ChunkSig chunks = ChunkSig{}) {
    const int64_t total = (int64_t)N * F;
    auto store_pair = [&](int64_t i) {
        if (i + 1 < total) {
            double2 v;
            v.x = component_lh_value(state, probs, sel, i, F, S, Fp, na_value);
            v.y = component_lh_value(state, probs, sel, i + 1, F, S, Fp, na_value);
            *reinterpret_cast<double2*>(out + i) = v;
        } else if (i < total) {
            out[i] = component_lh_value(state, probs, sel, i, F, S, Fp, na_value);
        }
    };
    if (chunks.chunk_elems > 0) {         // ordered form: the grid walks the chunks one after another
        for (unsigned c = 0; c < chunks.n_chunks; ++c) {
            const int64_t lo = (int64_t)c * chunks.chunk_elems, hi = min(total, lo + (int64_t)chunks.chunk_elems);
            for (int64_t i = lo + 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); i < hi; i += 2 * (int64_t)gridDim.x * blockDim.x)
                store_pair(i);
            signal_chunk_ordered(chunks, c);
        }
        return;
    }
    store_pair(2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x));
    signal_chunk(chunks);                 // (`out` in host-mapped memory: completion chunk by chunk)
}

// ------------------------------------------------------------------------------------------
// a3: likelihood_per_component (conditionals.py:152-223): dense [N][F][C] float64 from the
// slot's group ids and tables; NA <- 1 (conditionals.py:216), no group <- 0 (likelihood.py:122).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double lh_dense_value(const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid,
                                                const float* __restrict__ probs, int64_t j, int Np, int F, int S, int C, int Fp) {
    const int64_t i = j / C;
    const int c = (int)(j - i * C);
    const int n = (int)(i / F), f = (int)(i % F);
    const uint8_t x = state[(int64_t)n * Fp + f];
    if (x == kNA) return 1.0;
    const uint16_t gg = gid[(int64_t)c * Np + n];
    return gg == kNoGroup ? 0.0 : (double)probs[((int64_t)gg * F + f) * S + x];
}

static __global__ void k_lh_dense(const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid,
                           const float* __restrict__ probs, double* __restrict__ out, int N, int Np, int F,
                           int S, int C, int Fp, ChunkSig chunks = ChunkSig{}) {
    // consecutive lanes store consecutive OUTPUT elements (n, f, c), two per thread as one 16-byte word: full lines
    // whether `out` is in HBM or, streamed, in host-mapped memory (a thread per observation storing C values at stride
    // C left every store instruction with half-filled lines: 3.2 MB crossed PCIe in 300 us instead of 70)
    const int64_t total = (int64_t)N * F * C;
    auto store_pair = [&](int64_t j) {
        if (j + 1 < total) {
            double2 v;
            v.x = lh_dense_value(state, gid, probs, j, Np, F, S, C, Fp);
            v.y = lh_dense_value(state, gid, probs, j + 1, Np, F, S, C, Fp);
            *reinterpret_cast<double2*>(out + j) = v;
        } else if (j < total) {
            out[j] = lh_dense_value(state, gid, probs, j, Np, F, S, C, Fp);
        }
    };
    if (chunks.chunk_elems > 0) {         // ordered form: the grid walks the chunks one after another
        for (unsigned c = 0; c < chunks.n_chunks; ++c) {
            const int64_t lo = (int64_t)c * chunks.chunk_elems, hi = min(total, lo + (int64_t)chunks.chunk_elems);
            for (int64_t j = lo + 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); j < hi; j += 2 * (int64_t)gridDim.x * blockDim.x)
                store_pair(j);
            signal_chunk_ordered(chunks, c);
        }
        return;
    }
    store_pair(2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x));
    signal_chunk(chunks);                 // (`out` in host-mapped memory: completion chunk by chunk)
}

// a2: likelihood_per_component_exact (conditionals.py:300-367): leave-one-out tables.
// For observation (n, f) in group g of component c the table row is
//   normalize(counts[g,f,:] + prior[g,f,:] - onehot(n,f,:) * source[n,f,c])   (float32)
// With `wpat` != nullptr the kernel instead writes obs[n][f] = sum_c w[pat(n)][f][c] * lh_exact (the row the
// reference's LikelihoodLogger stores, loggers.py:354-359), NumPy order, no FMA.
static __global__ void k_lh_exact(const uint8_t* __restrict__ state, const uint8_t* __restrict__ src,
                           const uint16_t* __restrict__ gid, const int32_t* __restrict__ counts,
                           const double* __restrict__ conc, double* __restrict__ out, int N, int Np, int F,
                           int S, int C, int Fp, int* __restrict__ status,
                           const float* __restrict__ wpat = nullptr, const uint8_t* __restrict__ pid = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * F) return;
    const int n = (int)(i / F), f = (int)(i % F);
    const uint8_t x = state[(int64_t)n * Fp + f];
    const uint8_t sc = src[(int64_t)n * Fp + f];
    double* o = out + i * C;
    const float* w = wpat ? wpat + ((int64_t)pid[n] * F + f) * C : nullptr;
    double vals[kMaxComponents];
    for (int c = 0; c < C; ++c) {
        double v = 1.0;
        if (x != kNA) {
            const uint16_t gg = gid[(int64_t)c * Np + n];
            if (gg == kNoGroup) v = 0.0;
            else {
                const int64_t base = ((int64_t)gg * F + f) * S;
                const double own = sc == c ? 1.0 : 0.0;
                auto post = [&](int s) -> double {
                    const double a = (double)(float)counts[base + s] + conc[base + s];
                    return s == x ? a - own : a - 0.0;
                };
                const double total = np_pairwise_sum<double>(post, S);
                if (!(total > 0.0)) raise_status(status, ST_BAD_NORMALIZE, 1);
                v = (double)(float)(post(x) / total);
            }
        }
        if (w) vals[c] = (double)w[c] * v;
        else o[c] = v;
    }
    if (w) {
        auto term = [&](int c) -> double { return vals[c]; };
        out[i] = np_pairwise_sum<double>(term, C);
    }
}

// ------------------------------------------------------------------------------------------
// a7 + a8: collapsed Dirichlet-categorical log-pdf (util.py:1373-1394, likelihood.py:65-101).
// k_dcl: one thread per (group, feature):
//   float32( lgamma(sum a) - lgamma(n + sum a) + sum_{s: a>0} (lgamma(c + a) - lgamma(a)) )
// k_group_sum_f32: one thread per group: float32 NumPy-order sum over features -> float64 cache.
// ------------------------------------------------------------------------------------------
template <class TC>
__global__ void k_dcl(const TC* __restrict__ counts, const double* __restrict__ conc_in,
                      float* __restrict__ per_feature, int g_lo, int g_hi, int F, int S, int conc_per_group) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (int64_t)(g_hi - g_lo) * F) return;
    const int64_t base = ((int64_t)g_lo * F + row) * S;
    // concentration either per group [G][F][S] or one [F][S] table broadcast over groups
    const double* conc = conc_per_group ? conc_in : conc_in - base + (row % F) * (int64_t)S;
    auto cnt = [&](int s) -> float { return (float)counts[base + s]; };
    auto a = [&](int s) -> double { return conc[base + s]; };
    const float n = np_pairwise_sum<float>(cnt, S);
    const double sum_a = np_pairwise_sum<double>(a, S);
    const double cst = sbe_lgamma_pos(sum_a) - sbe_lgamma_pos((double)n + sum_a);
    auto series = [&](int s) -> double {
        const double as = conc[base + s];
        return as > 0.0 ? sbe_lgamma_pos((double)(float)counts[base + s] + as) - sbe_lgamma_pos(as) : 0.0;
    };
    per_feature[row] = (float)(cst + np_pairwise_sum<double>(series, S));
}

static __global__ void k_group_sum_f32(const float* __restrict__ per_feature, double* __restrict__ per_group,
                                int n_groups, int F) {
    // one group per octet of lanes (np_pairwise_sum_f32_x8: NumPy's order, an eighth of the dependent chain)
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (g >= n_groups) return;
    const float* p = per_feature + (int64_t)g * F;
    auto get = [&](int i) -> float { return p[i]; };
    const float total = np_pairwise_sum_f32_x8(get, F, (int)(threadIdx.x & 7));
    if ((threadIdx.x & 7) == 0) per_group[g] = (double)total;
}

// a7 + a8 in ONE launch (round 3: the drop-in Likelihood.__call__ asks for this once or twice per MCMC step, and a
// launch costs more than the arithmetic): block = one group.  Phase 1, one thread per table element: the lgamma term
// of the element, with k_conc_lgamma's count-independent tables (ONE lgamma per element -- same function, same
// arguments as k_dcl's two, so the same value); phase 2, one thread per feature: the ordered sums over the S states,
// the row constant, the float32 cast (k_dcl's expression); phase 3, eight lanes: the float32 NumPy-order sum over the
// features (k_group_sum_f32's).  Dynamic LDS: F*S doubles + F floats.  `per_feature` (may be null) gets the rows.
template <class TC>
__device__ __forceinline__ void collapsed_group_block(
    const TC* __restrict__ counts, const double* __restrict__ conc, const double* __restrict__ lg_conc,
    const double* __restrict__ sum_a, const double* __restrict__ lg_sum_a, float* __restrict__ per_feature,
    double* __restrict__ per_group, int g_lo, int F, int S, int block /* = group - g_lo */, unsigned char* cg_lds) {
    double* ser = reinterpret_cast<double*>(cg_lds);                       // [F][S]
    float* pf = reinterpret_cast<float*>(cg_lds + (size_t)F * S * sizeof(double));   // [F]
    const int g = g_lo + block;
    const int64_t gbase = (int64_t)g * F * S;
    for (int e = threadIdx.x; e < F * S; e += blockDim.x) {
        const double as = conc[gbase + e];
        ser[e] = as > 0.0 ? sbe_lgamma_pos((double)(float)counts[gbase + e] + as) - lg_conc[gbase + e] : 0.0;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const int64_t base = gbase + (int64_t)f * S;
        auto cnt = [&](int k) -> float { return (float)counts[base + k]; };
        auto ser_at = [&](int k) -> double { return ser[f * S + k]; };
        const float n = np_pairwise_sum<float>(cnt, S);
        const double sa = sum_a[(int64_t)g * F + f];
        const double cst = lg_sum_a[(int64_t)g * F + f] - sbe_lgamma_pos((double)n + sa);
        const float v = (float)(cst + np_pairwise_sum<double>(ser_at, S));
        pf[f] = v;
        if (per_feature) per_feature[(int64_t)block * F + f] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        auto get = [&](int i) -> float { return pf[i]; };
        const float total = np_pairwise_sum_f32_x8(get, F, (int)threadIdx.x);
        if (threadIdx.x == 0) per_group[block] = (double)total;
    }
}
template <class TC>
__global__ __launch_bounds__(1024) void k_collapsed_groups(
    const TC* __restrict__ counts, const double* __restrict__ conc, const double* __restrict__ lg_conc,
    const double* __restrict__ sum_a, const double* __restrict__ lg_sum_a, float* __restrict__ per_feature,
    double* __restrict__ per_group, int g_lo, int F, int S, DoneSig done = DoneSig{}) {
    extern __shared__ __align__(16) unsigned char cg_lds[];
    collapsed_group_block<TC>(counts, conc, lg_conc, sum_a, lg_sum_a, per_feature, per_group, g_lo, F, S, (int)blockIdx.x, cg_lds);
    signal_done(done);
}

// ------------------------------------------------------------------------------------------
// a6, per-observation form: obs[n][f] = sum_c w[pat(n)][f][c] * lh_c(n, f)   (loggers.py:355-357,
// operators.py:568-574, 1060-1061) with lh = 1 for NA observations (the reference multiplies the
// weights by likelihood_per_component's NA value 1) and lh = 0 for "no group".  Dense output
// kernel (N*F doubles), float64, NumPy order ((0 + t0) + t1) + ..., no FMA: bit-exact.
// ------------------------------------------------------------------------------------------
static __global__ void k_observation_lh(const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid,
                                 const uint8_t* __restrict__ pid, const float* __restrict__ probs,
                                 const float* __restrict__ wpat, double* __restrict__ obs, int N, int Np, int F,
                                 int S, int C, int Fp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * F) return;
    const int n = (int)(i / F), f = (int)(i % F);
    const uint8_t x = state[(int64_t)n * Fp + f];
    const float* w = wpat + ((int64_t)pid[n] * F + f) * C;
    auto term = [&](int c) -> double {
        double lh = 1.0;
        if (x != kNA) {
            const uint16_t gg = gid[(int64_t)c * Np + n];
            lh = gg == kNoGroup ? 0.0 : (double)probs[((int64_t)gg * F + f) * S + x];
        }
        return (double)w[c] * lh;
    };
    obs[i] = np_pairwise_sum<double>(term, C);       // NumPy's order, also for C >= 8 (8-way unrolled block)
}


// Epilogue of the one-call MCMC step (sbe_step), run by ONE extra block of k_reduce_partials: per-group collapsed
// log-likelihood (a7: float32 NumPy-order sum of the per-feature values, likelihood.py:74-77), the changed-group
// flags and the data-check words, all written straight into host-mapped pinned memory -- no D2H copies.
struct StepFinish {
    const float* per_feature;      // [Gtot][F] (k_step_tables); nullptr = no epilogue
    double* group_out;             // mapped [Gtot]
    const uint8_t* changed;        // [Gtot] flags, or nullptr when `stamp` is used
    const uint32_t* stamp;         // [Gtot] group changed in step `step_id` iff stamp[g] == step_id (k_step_core)
    uint32_t step_id;
    uint8_t* changed_out;          // mapped [Gtot]
    const int* status;             // [ST_WORDS]
    int* status_out;               // mapped [ST_WORDS]
    int Gtot, F;
    // one-call Gibbs step: the two transition log-probabilities = fixed-order sums of k_sum_log_f32's partials
    const double* lq_partials[2]; int lq_n[2];
    double* lq_out;                // mapped [2] (nullptr: none)
};

// Final fixed-order reduction of the per-block partials: one block per slot (+ one for the step epilogue).
// Batched steps (sbe_step_batch): `slot_list` names the slots, `fins` holds one epilogue per chain (n_fin blocks).
static __global__ __launch_bounds__(kBlock) void k_reduce_partials(const double* __restrict__ partials,
                                                           int64_t partials_stride, int n_blocks,
                                                           double* __restrict__ results, int first_slot,
                                                           int n_slots, StepFinish fin_single,
                                                           const int32_t* __restrict__ slot_list = nullptr,
                                                           const StepFinish* __restrict__ fins = nullptr,
                                                           DoneSig done = DoneSig{}) {
    __shared__ double red4[4];
    if ((int)blockIdx.x >= n_slots) {                                   // step epilogue block
        const StepFinish fin = fins ? fins[(int)blockIdx.x - n_slots] : fin_single;
        // per-feature values staged through LDS in chunks of whole groups (coalesced loads; the ordered sums
        // then run at LDS latency instead of one L2 round trip per 8 elements)
        __shared__ float stage[8192];
        const int gpc = max(1, 8192 / fin.F);                            // groups per chunk
        for (int g0 = 0; g0 < fin.Gtot; g0 += gpc) {
            const int ng = min(gpc, fin.Gtot - g0);
            const bool fits = fin.F <= 8192;
            if (fits) {
                for (int i = threadIdx.x; i < ng * fin.F; i += kBlock) stage[i] = fin.per_feature[(int64_t)g0 * fin.F + i];
                __syncthreads();
            }
            // one group per octet of lanes: NumPy's float32 pairwise sum over the F features with its eight accumulators
            // on eight lanes (np_pairwise_sum_f32_x8) -- a thread per group walked 200 dependent LDS reads + adds (6 us)
            for (int g = threadIdx.x >> 3; g < ng; g += kBlock / 8) {
                const float* p = fits ? stage + g * fin.F : fin.per_feature + (int64_t)(g0 + g) * fin.F;
                auto get = [&](int i) -> float { return p[i]; };
                const float total = np_pairwise_sum_f32_x8(get, fin.F, (int)(threadIdx.x & 7));
                if ((threadIdx.x & 7) == 0) {
                    fin.group_out[g0 + g] = (double)total;
                    fin.changed_out[g0 + g] = fin.stamp ? (uint8_t)(fin.stamp[g0 + g] == fin.step_id) : fin.changed[g0 + g];
                }
            }
            __syncthreads();
        }
        if (threadIdx.x < ST_WORDS) fin.status_out[threadIdx.x] = fin.status[threadIdx.x];
        if (fin.lq_out) {                                                // kernel-uniform; fixed order: deterministic
            for (int t = 0; t < 2; ++t) {
                double acc = 0.0;
                for (int i = threadIdx.x; i < fin.lq_n[t]; i += kBlock) acc += fin.lq_partials[t][i];
                __syncthreads();                                         // (red4 reuse)
                const double total = block_sum(acc, red4);
                if (threadIdx.x == 0) fin.lq_out[t] = total;
            }
        }
        signal_done(done);
        return;
    }
    const int slot = slot_list ? slot_list[blockIdx.x] : first_slot + (int)blockIdx.x;
    const double* p = partials + (int64_t)slot * partials_stride;
    double v = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += kBlock) v += p[i];
    const double total = block_sum(v, red4);
    if (threadIdx.x == 0) results[slot] = total;
    signal_done(done);
}

// Per-object LDS byte offsets of k_mixture_rows, rebuilt whenever a slot's group ids change; quad-major so that the
// offsets a wave step needs (its object quads x (C+1) x 4 objects) are one contiguous run of dwords:
//   out[q][c][j]  (c < C) = row(g_c(4q+j)) * (S+1) * FT * 4      row = global group index, Gtot for "no group"
//   out[q][C][j]          = pattern(4q+j) * ceil(C/2) * FT * 16  (weight planes of the object's has_components pattern)
// objects n >= N of the last quad get the "no group" row and pattern 0 (their state bytes are NA).
static __global__ void k_rowoff(const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid, uint32_t* __restrict__ out,
                         int64_t gid_stride, int64_t pid_stride, int64_t out_stride, int first_slot,
                         const int32_t* __restrict__ slot_list, int N, int Np,
                         int C, int Gtot, uint32_t row_bytes, uint32_t pat_bytes) {
    const int slot = slot_list ? slot_list[blockIdx.y] : first_slot + (int)blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (c, n)
    if (i >= (C + 1) * Np) return;
    const int c = i / Np, n = i - c * Np;
    const int o = ((n >> 2) * (C + 1) + c) * 4 + (n & 3);
    uint32_t v;
    if (c < C) {
        const uint32_t g = n < N ? gid[(int64_t)slot * gid_stride + (int64_t)c * Np + n] : (uint32_t)kNoGroup;
        v = (g < (uint32_t)Gtot ? g : (uint32_t)Gtot) * row_bytes;
    } else {
        v = (n < N ? (uint32_t)pid[(int64_t)slot * pid_stride + n] : 0u) * pat_bytes;
    }
    out[(int64_t)slot * out_stride + o] = v;
}


}  // namespace sbe

#include "sbe_kernels_operators.hip.h"
#include "sbe_kernels_sampling.hip.h"
