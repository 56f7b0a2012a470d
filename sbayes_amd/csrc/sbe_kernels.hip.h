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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbe {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kMaxComponents = 8;
constexpr int kMaxTuples = 64;
constexpr uint16_t kNoGroup = 0xFFFF;
constexpr uint8_t kNA = 0xFF;

// status words written by kernels (d_status[...])
enum StatusWord : int {
    ST_MULTI_STATE = 0,      // (object, feature) rows with more than one set state
    ST_NA_COUNT = 1,         // NA observations
    ST_BAD_NORMALIZE = 2,    // normalize(): a row sum was not > 0 (util.py:1006 assert)
    ST_MULTI_SOURCE = 3,     // source rows with more than one component set
    ST_FLAG_PTR = 4,         // words 4..5: 64-bit address of the engine's host-mapped flag words, or 0 (lane-private status
                             // arrays of the batched steps: their words travel in the step epilogue's mapped block)
    ST_WORDS = 8
};

// A kernel-raised data check: the count goes to the device word, and -- error path only -- a plain store marks the
// engine's host-mapped flag word, so that the host learns "nothing was raised" from its own memory after the
// synchronisation it performs anyway, without a status read-back per call.
__device__ __forceinline__ void raise_status(int* status, int word, int count) {
    atomicAdd(&status[word], count);
    int* flag = *reinterpret_cast<int* const*>(status + ST_FLAG_PTR);
    if (flag) __hip_atomic_store(flag + word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Completion by flag (latency-bound calls): the LAST block of a call's final kernel stores the call's sequence number
// into a host-mapped word once every block's results are on their way, so the host can spin on its own memory instead of
// going through the runtime's stream wait.  Every block calls signal_done() as its last action, all threads of the block
// together.  `flag` == nullptr: no signalling asked for.
// Ordering.  Every wave waits until its own stores are acknowledged (s_waitcnt vmcnt(0): on gfx9 stores count in vmcnt;
// the acknowledgement comes from the XCD's L2), then ONE thread of the block issues a system-scope release fence -- it
// pushes that L2's pending writes out to the fabric and waits for them -- before the block takes its ticket (a device-wide
// atomic).  When the last ticket is taken every block's results are therefore globally visible, and the flag follows.
// Two cheaper forms were measured and dropped: a system-scope fence in EVERY thread (+15 us on a 64-chain sweep whose
// step cores had just dirtied megabytes of L2: 32 k fences), and no fence at all, relying on the store acknowledgement
// alone -- the flag then overtook a result written by the same thread (tests/test_gpu_c_abi.py caught it at once).
struct DoneSig { unsigned* ticket; unsigned long long* flag; unsigned long long seq; unsigned n_blocks; };
__device__ __forceinline__ void signal_done(const DoneSig& d) {
    if (!d.flag) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's result stores are acknowledged by the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: the L2's pending writes are out and confirmed
        const unsigned t = __hip_atomic_fetch_add(d.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == d.n_blocks - 1) {                          // every other block fenced before its ticket
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (pairs with their release fences through the ticket chain)
            __hip_atomic_store(d.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// A LARGE result written by the kernel itself into host-mapped pinned memory (posted PCIe writes from every CU: as fast
// as the copy engine on this platform, and no copy operation, event or second synchronisation behind the kernel), with
// completion reported CHUNK BY CHUNK: consecutive blocks form a chunk; each block fences like signal_done and takes a
// ticket of its chunk, the chunk's last block stores the call's sequence number into the chunk's host-mapped flag.  The
// host copies a chunk out of the staging buffer as soon as its flag shows the sequence number, while the later chunks
// are still crossing PCIe (sbe_engine.hip: stream_result).  Every thread of the block must reach the call.
struct ChunkSig { unsigned* tickets; unsigned long long* flags; unsigned long long seq; unsigned blocks_per_chunk; unsigned n_blocks; };
__device__ __forceinline__ void signal_chunk(const ChunkSig& c) {
    if (!c.flags) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's result stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: the pending writes are out and confirmed
        const unsigned chunk = blockIdx.x / c.blocks_per_chunk;
        const unsigned first = chunk * c.blocks_per_chunk;
        const unsigned in_chunk = min(c.blocks_per_chunk, c.n_blocks - first);
        const unsigned t = __hip_atomic_fetch_add(c.tickets + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == in_chunk - 1) {                            // every other block of the chunk fenced before its ticket
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(c.tickets + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.flags + chunk, c.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------
// NumPy reduction order (pairwise sum, PW_BLOCKSIZE = 128, 8-way unrolled block).
// `get(i)` returns element i as T.  Matches @TYPE@_pairwise_sum for any n.
// ------------------------------------------------------------------------------------------
template <class T, class Get>
__device__ __forceinline__ T np_block_sum(Get get, int lo, int n) {
    if (n < 8) {
        T res = T(0);
        for (int i = 0; i < n; ++i) res = res + get(lo + i);
        return res;
    }
    T r0 = get(lo + 0), r1 = get(lo + 1), r2 = get(lo + 2), r3 = get(lo + 3);
    T r4 = get(lo + 4), r5 = get(lo + 5), r6 = get(lo + 6), r7 = get(lo + 7);
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
        r0 = r0 + get(lo + i + 0); r1 = r1 + get(lo + i + 1);
        r2 = r2 + get(lo + i + 2); r3 = r3 + get(lo + i + 3);
        r4 = r4 + get(lo + i + 4); r5 = r5 + get(lo + i + 5);
        r6 = r6 + get(lo + i + 6); r7 = r7 + get(lo + i + 7);
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + get(lo + i);
    return res;
}

template <class T, class Get>
__device__ T np_pairwise_sum(Get get, int n) {
    if (n <= 128) return np_block_sum<T>(get, 0, n);
    // iterative post-order walk of NumPy's recursion: split n2 = n/2 - (n/2)%8
    struct Frame { int lo, n, stage; T left; };
    Frame st[28];
    int sp = 0;
    T ret = T(0);
    st[sp++] = Frame{0, n, 0, T(0)};
    while (sp > 0) {
        Frame& f = st[sp - 1];
        if (f.stage == 0) {
            if (f.n <= 128) { ret = np_block_sum<T>(get, f.lo, f.n); --sp; }
            else { int n2 = f.n / 2; n2 -= n2 % 8; f.stage = 1; st[sp++] = Frame{f.lo, n2, 0, T(0)}; }
        } else if (f.stage == 1) {
            f.left = ret; f.stage = 2;
            int n2 = f.n / 2; n2 -= n2 % 8;
            st[sp++] = Frame{f.lo + n2, f.n - n2, 0, T(0)};
        } else {
            ret = f.left + ret; --sp;
        }
    }
    return ret;
}

// The same sum computed by EIGHT consecutive lanes together (lane j of the octet owns NumPy's accumulator r[j] of every
// 128-element leaf; the leaf's r[] tree is three shuffle steps): identical operations in identical order, so the result
// is bit for bit np_pairwise_sum's, at an eighth of the dependent-add chain.  Every lane of the octet must call it with
// the same n; the result is valid on all eight lanes.  (round 3: the step epilogue's per-group sums over F features)
template <class Get>
__device__ float np_pairwise_sum_f32_x8(Get get, int n, int j) {
    auto leaf = [&](int lo, int m) -> float {
        float res;
        if (m < 8) {
            res = 0.0f;
            for (int i = 0; i < m; ++i) res = res + get(lo + i);
        } else {
            float r = get(lo + j);
            const int lim = m - (m % 8);
            for (int i = 8; i < lim; i += 8) r = r + get(lo + i + j);
            r = r + __shfl_down(r, 1, 8);                    // lanes 0, 2, 4, 6: r0+r1, r2+r3, r4+r5, r6+r7
            r = r + __shfl_down(r, 2, 8);                    // lanes 0, 4: (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
            r = r + __shfl_down(r, 4, 8);                    // lane 0
            res = __shfl(r, 0, 8);
            for (int i = lim; i < m; ++i) res = res + get(lo + i);
        }
        return res;
    };
    if (n <= 128) return leaf(0, n);
    // NumPy's recursion (split n2 = n/2 - (n/2) % 8) unrolled four levels deep: no frame stack -- the generic walk below
    // keeps its frames in scratch memory, and a dozen scratch round trips were most of a 200-feature sum (5 of the 9 us of
    // the step epilogue and of k_collapsed_groups).  Four levels reach leaves of <= 128 for every n <= 1000.
    auto cut = [](int m) { const int h = m / 2; return h - h % 8; };
    if (n <= 1000) {
        auto s0 = [&](int lo, int m) -> float { return leaf(lo, m); };                      // (m <= 128 here)
        auto s1 = [&](int lo, int m) -> float { if (m <= 128) return leaf(lo, m); const int c = cut(m); const float l = s0(lo, c); return l + s0(lo + c, m - c); };
        auto s2 = [&](int lo, int m) -> float { if (m <= 128) return leaf(lo, m); const int c = cut(m); const float l = s1(lo, c); return l + s1(lo + c, m - c); };
        auto s3 = [&](int lo, int m) -> float { if (m <= 128) return leaf(lo, m); const int c = cut(m); const float l = s2(lo, c); return l + s2(lo + c, m - c); };
        const int c = cut(n);
        const float l = s3(0, c);
        return l + s3(c, n - c);
    }
    struct Frame { int lo, n, stage; float left; };
    Frame st[28];
    int sp = 0;
    float ret = 0.0f;
    st[sp++] = Frame{0, n, 0, 0.0f};
    while (sp > 0) {
        Frame& f = st[sp - 1];
        if (f.stage == 0) {
            if (f.n <= 128) { ret = leaf(f.lo, f.n); --sp; }
            else { int n2 = f.n / 2; n2 -= n2 % 8; f.stage = 1; st[sp++] = Frame{f.lo, n2, 0, 0.0f}; }
        } else if (f.stage == 1) {
            f.left = ret; f.stage = 2;
            int n2 = f.n / 2; n2 -= n2 % 8;
            st[sp++] = Frame{f.lo + n2, f.n - n2, 0, 0.0f};
        } else {
            ret = f.left + ret; --sp;
        }
    }
    return ret;
}

// ------------------------------------------------------------------------------------------
// wave64 / block reductions (fixed order => run-to-run deterministic)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double* lds4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    if (lane == 0) lds4[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    return r;   // valid in thread 0
}

// ------------------------------------------------------------------------------------------
// fp64 natural log for the table build of the group-tuple kernel.  The device library's log() costs
// ~100+ instructions (~520 cycles per wave measured); this one is the classic argument reduction
// v = 2^k * m, m in [sqrt(1/2), sqrt(2)), s = f/(2+f), f = m-1, with the degree-14 odd minimax series of
// log((1+s)/(1-s)) (fdlibm e_log.c coefficients) and a Newton-refined reciprocal: ~40 instructions,
// < 1 ulp over positive normal doubles (tests/test_gpu_engine.py::test_fast_log_accuracy).  Anything
// else (0, subnormal, negative, inf, NaN) takes the library log so -inf / NaN behave like NumPy's.
// ------------------------------------------------------------------------------------------
// (the library routines behind the rare paths are NOT inlined: inlined, their coefficient tables are hoisted out of the
//  callers' loops and held in ~150 VGPRs for the whole kernel -- k_step_core stood at 255 VGPRs + scratch for it)
__device__ __attribute__((noinline)) double lib_log(double v) { return log(v); }
__device__ __attribute__((noinline)) double lib_lgamma(double v) { return lgamma(v); }

__device__ __forceinline__ double fast_log_pos(double v) {
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const uint32_t ex = (uint32_t)(bits >> 52);                  // sign + exponent
    if (__builtin_expect(ex - 1u >= 0x7FEu, 0)) return lib_log(v);   // not a positive normal finite double
    int k = (int)ex - 1023;
    double m = __longlong_as_double((long long)((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));   // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; ++k; }               // [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);                          // ~2^-26 relative; two Newton steps -> full fp64
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double s = f * r;
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // log(v) = k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f)
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// ------------------------------------------------------------------------------------------
// lgamma for the Dirichlet-categorical terms (a8: util.py:1373-1394, arguments = concentrations and counts +
// concentrations: x > 0).  The device library's lgamma keeps ~150 VGPRs of polynomial coefficients live across the
// table loops of k_step_core (255 VGPRs, 688 bytes of scratch, two blocks per CU).  This one: shift x up to y >= 8 with
// the recurrence (at most 8 multiplications), Stirling's series at y to 1/y^13 (truncation < 2e-15 relative at y = 8)
// and two logs (fast_log_pos).  Absolute error <= 1e-14 * max(1, |lgamma(x)|) (observed 6.3e-15): the results are differenced, summed
// per feature and cast to float32 (the reference's own arithmetic there is float32 and numba fastmath),
// tests/test_gpu_engine.py::test_lgamma_accuracy pins it against SciPy's gammaln.  x <= 0, inf, NaN: library lgamma.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double sbe_lgamma_pos(double x) {
    if (__builtin_expect(!(x > 0.0) || x > 1e300, 0)) return lib_lgamma(x);
    double p = 1.0, y = x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                 // branch-free: lanes differ in how far they are from 8
        const bool s = y < 8.0;
        p = s ? p * y : p;
        y = s ? y + 1.0 : y;
    }
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    r = fma(fma(-y, r, 1.0), r, r);
    const double r2 = r * r;
    double q = 1.0 / 156.0;                                       // B_{2k} / (2k (2k-1)), k = 7 .. 1
    q = fma(q, r2, -691.0 / 360360.0);
    q = fma(q, r2, 1.0 / 1188.0);
    q = fma(q, r2, -1.0 / 1680.0);
    q = fma(q, r2, 1.0 / 1260.0);
    q = fma(q, r2, -1.0 / 360.0);
    q = fma(q, r2, 1.0 / 12.0);
    const double lg = fma(y - 0.5, fast_log_pos(y), -y) + (fma(q, r, 0.91893853320467274178));
    return x < 8.0 ? lg - fast_log_pos(p) : lg;
}

__global__ void k_test_lgamma(const double* __restrict__ in, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sbe_lgamma_pos(in[i]);
}

__global__ void k_test_fast_log(const double* __restrict__ in, double* __restrict__ out_fast,
                                double* __restrict__ out_lib, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_fast[i] = fast_log_pos(in[i]);
    out_lib[i] = log(in[i]);
}

// ------------------------------------------------------------------------------------------
// Table-driven fp64 log of k_mixture_tuple64's table build (the build is VALU-bound on the log: ~20 vector
// instructions here against ~48 in fast_log_pos).  v = 2^k * m, m in [1, 2); the top 7 mantissa bits pick
// an interval with centre c (c = 1 exactly for the first interval, so log(1) = 0 exactly); the table holds
// inv_c = RN(1/c) and logc = RN(-log(inv_c)) (computed on the host in long double), so that
// log(v) = k*ln2 + logc + log1p(r) with r = fma(m, inv_c, -1) EXACT up to one rounding and |r| <= 2^-7:
// degree-8 Taylor of log1p (truncation < 2^-63).  Error: <= 1 ulp of the result + 2^-53 absolute (the
// rounding of logc; visible only where k*ln2 + logc cancels, i.e. for v just below 1) -- six orders of
// magnitude inside what the 1e-10 relative tolerance of the summed log-likelihood needs
// (tests/test_gpu_engine.py::test_fast_log_accuracy).  Not a positive normal double: library log.
// `tab`: absolute LDS byte address of the 128 {inv_c, logc} pairs.
// ------------------------------------------------------------------------------------------
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f64x2_t lds_cf64x2_t;
constexpr int kLogTabEntries = 128;

// (tab_log_core: the straight-line part, garbage for arguments that are not positive normal doubles --
//  tab_log_special tells; callers that interleave several logs test the specials once, after the batch)
__device__ __forceinline__ bool tab_log_special(double v) {
    return ((uint32_t)__double2hiint(v) >> 20) - 1u >= 0x7FEu;
}
__device__ __forceinline__ double tab_log_core(double v, uint32_t tab) {
    const uint32_t hi = (uint32_t)__double2hiint(v);
    const uint32_t ex = hi >> 20;                                 // sign + exponent
    const f64x2_t e = *(lds_cf64x2_t*)(uintptr_t)(tab + ((hi >> 9) & 0x7F0u));        // entry (hi >> 13) & 127
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), __double2loint(v));
    const double r = fma(m, e.x, -1.0);
    double q = fma(-0.125, r, 1.0 / 7.0);
    q = fma(q, r, -1.0 / 6.0);
    q = fma(q, r, 0.2);
    q = fma(q, r, -0.25);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -0.5);
    const double lp = fma(r * r, q, r);                           // log1p(r)
    const double kd = (double)((int)ex - 1023);
    return fma(kd, 6.93147180369123816490e-01, e.y) + fma(kd, 1.90821492927058770002e-10, lp);
}
__device__ __forceinline__ double tab_log_pos(double v, uint32_t tab) {
    if (__builtin_expect(tab_log_special(v), 0)) return lib_log(v);
    return tab_log_core(v, tab);
}
// G logs at once, written stage by stage so that the G dependent chains are interleaved in program order (the
// compiler keeps a chain-by-chain source order chain by chain: one vector instruction per dependent-issue latency)
template <int G>
__device__ __forceinline__ void tab_log_core_n(const double (&v)[G], double (&out)[G], uint32_t tab) {
    f64x2_t e[G];
    double m[G], r[G], q[G], kd[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t hi = (uint32_t)__double2hiint(v[g]);
        e[g] = *(lds_cf64x2_t*)(uintptr_t)(tab + ((hi >> 9) & 0x7F0u));
        m[g] = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), __double2loint(v[g]));
        kd[g] = (double)((int)(hi >> 20) - 1023);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) r[g] = fma(m[g], e[g].x, -1.0);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(-0.125, r[g], 1.0 / 7.0);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(q[g], r[g], -1.0 / 6.0);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(q[g], r[g], 0.2);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(q[g], r[g], -0.25);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(q[g], r[g], 1.0 / 3.0);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(q[g], r[g], -0.5);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(r[g] * r[g], q[g], r[g]);          // log1p(r)
#pragma unroll
    for (int g = 0; g < G; ++g)
        out[g] = fma(kd[g], 6.93147180369123816490e-01, e[g].y) + fma(kd[g], 1.90821492927058770002e-10, q[g]);
}

__global__ void k_test_tab_log(const double* __restrict__ in, const double2* __restrict__ logtab,
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
__global__ void k_ingest_onehot(const uint8_t* __restrict__ raw, uint8_t* __restrict__ onehot,
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
__global__ void k_init_state_h(uint16_t* __restrict__ state_h, int64_t n_entries /* NQ*Fq */, int Fq, int S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const uint16_t v = (uint16_t)(S * 512 + (int)((i % Fq) & 63) * 8);
    uint16_t* o = state_h + i * 4;
    o[0] = v; o[1] = v; o[2] = v; o[3] = v;
}

// Several small host arrays to their resident places in ONE launch: the arrays are staged back to back in the mapped
// ring (`base`, read over PCIe element-parallel), segment y goes to sg.dst[y].  Words when everything is 4-byte aligned.
struct ScatterSegs { uint8_t* dst[8]; uint32_t off[8]; uint32_t bytes[8]; int n; };
__global__ void k_scatter_bytes(const uint8_t* __restrict__ base, ScatterSegs sg) {
    const int y = blockIdx.y;
    if (y >= sg.n) return;
    const uint8_t* src = base + sg.off[y];
    uint8_t* dst = sg.dst[y];
    const uint32_t nb = sg.bytes[y], stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src | nb) & 3) == 0) {
        for (uint32_t j = i0; j < nb / 4; j += stride) reinterpret_cast<uint32_t*>(dst)[j] = reinterpret_cast<const uint32_t*>(src)[j];
    } else {
        for (uint32_t j = i0; j < nb; j += stride) dst[j] = src[j];
    }
}

// K0b: source ingest.  bool rows [rows][F][C] -> component id per observation (0xFF = none).
// `objects` == nullptr: row r is object r.
__global__ void k_ingest_source(const uint8_t* __restrict__ rows, const int32_t* __restrict__ objects,
                                uint8_t* __restrict__ src_id, int n_rows, int F, int C, int Fp,
                                int* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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

// Inverse of K0b for the listed objects: component id -> bool row [F][C] (all False for 0xFF).
__global__ void k_expand_source(const uint8_t* __restrict__ src_id, const int32_t* __restrict__ objects,
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

__global__ __launch_bounds__(kBlock) void k_counts(
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
__global__ __launch_bounds__(kBlock) void k_effect_counts(
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
__global__ void k_counts_global(const uint8_t* __restrict__ state, CountSide A, CountSide B,
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
__global__ void k_mask_to_src(const uint8_t* __restrict__ mask, uint8_t* __restrict__ src, int N, int F, int Fp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * F) return;
    const int n = (int)(i / F), f = (int)(i % F);
    src[(int64_t)n * Fp + f] = mask[i] ? 0 : kNA;
}

__global__ void k_i32_to_f32(const int32_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
__global__ void k_f32_to_i32(const float* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
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
template <class GetCount, class Emit>
__device__ __forceinline__ void probs_row(GetCount cnt, const double* __restrict__ conc_row, const double* __restrict__ unif_row,
                                          int S, double temperature, double prior_temperature, int* __restrict__ status, Emit emit) {
    const bool tempered = temperature > 0.0;
    const bool prior_tempered = prior_temperature > 0.0 && unif_row != nullptr;
    const float t32 = (float)temperature;
    auto post = [&](int s) -> double {
        float c = cnt(s);
        if (tempered) c = c / t32;
        double a = conc_row[s];
        if (prior_tempered) {
            const double u = unif_row[s];
            a = u + (a - u) / prior_temperature;
        }
        return (double)c + a;
    };
    const double total = np_pairwise_sum<double>(post, S);
    if (!(total > 0.0)) raise_status(status, ST_BAD_NORMALIZE, 1);
    for (int s = 0; s < S; ++s) emit(s, (float)(post(s) / total));
}

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
__global__ void k_weight_patterns(const float* __restrict__ weights /* [F][C] */,
                                  const uint32_t* __restrict__ pattern_bits /* [P] */,
                                  float* __restrict__ wpat /* [P][F][C] */, int P, int F, int C,
                                  float* __restrict__ weights_keep = nullptr, double* __restrict__ wpat_t = nullptr,
                                  int Pmax = 0, int ft = 1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * F) return;
    const int p = i / F, f = i % F;
    const uint32_t bits = pattern_bits[p];
    float w[kMaxComponents];
    for (int c = 0; c < C; ++c) w[c] = weights[(int64_t)f * C + c];
    if (weights_keep && p == 0) for (int c = 0; c < C; ++c) weights_keep[(int64_t)f * C + c] = w[c];
    auto masked = [&](int c) -> float { return ((bits >> c) & 1u) ? w[c] : 0.0f * w[c]; };
    const float total = np_pairwise_sum<float>(masked, C);
    float* out = wpat + ((int64_t)p * F + f) * C;
    double* out_t = wpat_t ? wpat_t + (((int64_t)(f / ft) * Pmax + p) * C) * ft + f % ft : nullptr;
    for (int c = 0; c < C; ++c) {
        const float v = masked(c) / total;
        out[c] = v;
        if (out_t) out_t[(int64_t)c * ft] = (double)v;
    }
}

// normalize_weights (likelihood.py:171-190), stateless, ROW form: out[n][f][:] = has_components[n] * weights[f] / sum --
// k_weight_patterns' arithmetic evaluated per row instead of per distinct pattern (the same operations on the same
// operands: the same bits), so nobody sorts patterns and the call is ONE launch.  Block = 8 rows; the [F][C] weights
// and the block's has_components rows are read once into LDS (both may live in host-mapped staging memory).
constexpr int kNwRows = 8;
__global__ void k_normalize_weight_rows(const float* __restrict__ weights /* [F][C] */,
                                        const uint8_t* __restrict__ has_components /* [N][C] */,
                                        float* __restrict__ out /* [N][F][C] */, int N, int F, int C, int stage_weights,
                                        DoneSig done = DoneSig{}) {
    extern __shared__ float nw_lds[];                        // [F][C] weights (stage_weights == 0: read in place from
    __shared__ uint32_t bits[kNwRows];                       //  device memory -- tables beyond the LDS budget)
    if (stage_weights) for (int i = threadIdx.x; i < F * C; i += blockDim.x) nw_lds[i] = weights[i];
    const int n0 = blockIdx.x * kNwRows;
    if (threadIdx.x < kNwRows) {
        uint32_t b = 0;
        const int n = n0 + threadIdx.x;
        if (n < N) for (int c = 0; c < C; ++c) if (has_components[(int64_t)n * C + c]) b |= 1u << c;
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

__global__ void k_expand_weights(const float* __restrict__ wpat, const uint8_t* __restrict__ pid,
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
    const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    const int64_t total = (int64_t)N * F;
    if (i + 1 < total) {
        double2 v;
        v.x = component_lh_value(state, probs, sel, i, F, S, Fp, na_value);
        v.y = component_lh_value(state, probs, sel, i + 1, F, S, Fp, na_value);
        *reinterpret_cast<double2*>(out + i) = v;
    } else if (i < total) {
        out[i] = component_lh_value(state, probs, sel, i, F, S, Fp, na_value);
    }
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

__global__ void k_lh_dense(const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid,
                           const float* __restrict__ probs, double* __restrict__ out, int N, int Np, int F,
                           int S, int C, int Fp, ChunkSig chunks = ChunkSig{}) {
    // consecutive lanes store consecutive OUTPUT elements (n, f, c), two per thread as one 16-byte word: full lines
    // whether `out` is in HBM or, streamed, in host-mapped memory (a thread per observation storing C values at stride
    // C left every store instruction with half-filled lines: 3.2 MB crossed PCIe in 300 us instead of 70)
    const int64_t j = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    const int64_t total = (int64_t)N * F * C;
    if (j + 1 < total) {
        double2 v;
        v.x = lh_dense_value(state, gid, probs, j, Np, F, S, C, Fp);
        v.y = lh_dense_value(state, gid, probs, j + 1, Np, F, S, C, Fp);
        *reinterpret_cast<double2*>(out + j) = v;
    } else if (j < total) {
        out[j] = lh_dense_value(state, gid, probs, j, Np, F, S, C, Fp);
    }
    signal_chunk(chunks);                 // (`out` in host-mapped memory: completion chunk by chunk)
}

// a2: likelihood_per_component_exact (conditionals.py:300-367): leave-one-out tables.
// For observation (n, f) in group g of component c the table row is
//   normalize(counts[g,f,:] + prior[g,f,:] - onehot(n,f,:) * source[n,f,c])   (float32)
// With `wpat` != nullptr the kernel instead writes obs[n][f] = sum_c w[pat(n)][f][c] * lh_exact (the row the
// reference's LikelihoodLogger stores, loggers.py:354-359), NumPy order, no FMA.
__global__ void k_lh_exact(const uint8_t* __restrict__ state, const uint8_t* __restrict__ src,
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

__global__ void k_group_sum_f32(const float* __restrict__ per_feature, double* __restrict__ per_group,
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
__global__ __launch_bounds__(1024) void k_collapsed_groups(
    const TC* __restrict__ counts, const double* __restrict__ conc, const double* __restrict__ lg_conc,
    const double* __restrict__ sum_a, const double* __restrict__ lg_sum_a, float* __restrict__ per_feature,
    double* __restrict__ per_group, int g_lo, int F, int S, DoneSig done = DoneSig{}) {
    extern __shared__ __align__(16) unsigned char cg_lds[];
    double* ser = reinterpret_cast<double*>(cg_lds);                       // [F][S]
    float* pf = reinterpret_cast<float*>(cg_lds + (size_t)F * S * sizeof(double));   // [F]
    const int g = g_lo + blockIdx.x;
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
        if (per_feature) per_feature[(int64_t)blockIdx.x * F + f] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        auto get = [&](int i) -> float { return pf[i]; };
        const float total = np_pairwise_sum_f32_x8(get, F, (int)threadIdx.x);
        if (threadIdx.x == 0) per_group[blockIdx.x] = (double)total;
    }
    signal_done(done);
}

// ------------------------------------------------------------------------------------------
// a6, per-observation form: obs[n][f] = sum_c w[pat(n)][f][c] * lh_c(n, f)   (loggers.py:355-357,
// operators.py:568-574, 1060-1061) with lh = 1 for NA observations (the reference multiplies the
// weights by likelihood_per_component's NA value 1) and lh = 0 for "no group".  Dense output
// kernel (N*F doubles), float64, NumPy order ((0 + t0) + t1) + ..., no FMA: bit-exact.
// ------------------------------------------------------------------------------------------
__global__ void k_observation_lh(const uint8_t* __restrict__ state, const uint16_t* __restrict__ gid,
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

// Log-accumulation modes of the fused kernels:
//   LOG_PER_OBS : fp64 log per observation, fp64 sum
//   LOG_PRODUCT : the observation likelihoods of a step are multiplied into a running mantissa whose
//                 binary exponent is stripped with integer ops; one fp64 log per thread (see ProdAcc)
enum MixMode : int { LOG_PER_OBS = 0, LOG_PRODUCT = 1, WRITE_OBS = 2 };

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
__global__ __launch_bounds__(kBlock) void k_reduce_partials(const double* __restrict__ partials,
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
struct Mix2Params {
    int N, NQ, Np, F, Fq, S, C, Gtot, P;
    int n_ftiles, quads_per_chunk;
    int n_work, n_batch;                           // (tile, chunk) work items per slot; slots in this launch
    int slot_groups, slots_per_group;              // XCD-aware block order (see k_mixture_v2)
    // group-tuple form (k_mixture_combo): per slot the distinct (g_0..g_{C-1}) tuples of its objects
    const uint8_t* tid;      int64_t tid_stride;       // [Np] tuple index per object
    const uint16_t* tuple_g; int64_t tuple_g_stride;   // [kMaxTuples][kMaxComponents] global group index (Gtot = none)
    const uint8_t* tuple_p;  int64_t tuple_p_stride;   // [kMaxTuples] pattern id of the tuple
    int KT;                                            // tuples used by the slots of this launch (max)
    int combo_w_off;                                   // byte offset of the weight tile in the combo kernel's LDS
    int combo_tab_off;                                 // byte offset of the one-hot byte -> (state, feature) table
    const uint32_t* state_q;                       // [NQ][Fq]
    const uint2* state_h;                          // [NQ][Fq] 4 x u16 prepared LDS offsets (k_mixture_tuple64), or null
    const uint32_t* toff;  int64_t toff_stride;    // per slot [Np] byte offset of the object's tuple block
    const double2* logtab;                         // [128] {1/c, log c} of tab_log_pos
    int gen_slots;                                 // k_mixture_tuple64 block order: slots per XCD and generation
    int rows_cum[5];                               // k_mixture_rows: cumulative per-mille shares of a block's steps by wave age class
    int ragged_w;                                  // valid features of the last tile if it runs in sub-row mode (<= 32), else 0
    uint64_t* stamps;                              // diagnostic builds (-DSBE_STAMPS): [blocks][4 waves][8] cycle stamps
    const uint8_t* onehot; int rs_pitch;           // [N][rs_pitch] (one-hot variant)
    const uint16_t* gid;   int64_t gid_stride;     // per slot [C][Np]
    const uint8_t* pid;    int64_t pid_stride;     // per slot [Np]
    const float* probs_t;  int64_t probs_t_stride; // per slot [n_ftiles][(Gtot+1)*S*FT]
    const double* wpat_t;  int64_t wpat_t_stride;  // per slot [n_ftiles][Pmax*C*FT]
    int wpat_tile_stride;                          // Pmax*C*FT
    double* partials;      int64_t partials_stride;
    int first_slot;
    const int32_t* slot_list;                      // slots of this launch (n_batch entries), or null: first_slot + i
    // rows kernel (k_mixture_rows): engine tile width of probs_t, canonical per-pattern weights, per-object row offsets
    int eft;                                       // tile width of probs_t (64 / 32 / 16)
    const float* wpat;     int64_t wpat_stride;    // per slot [Pmax][F][C] float32 normalised weights (a5)
    const uint32_t* rowoff; int64_t rowoff_stride; // per slot [C+1][Np]: LDS byte offsets (see k_rowoff)
};

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
    if (threadIdx.x == 0) p.partials[(int64_t)slot * p.partials_stride + work] = total;
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
    if (threadIdx.x == 0) p.partials[(int64_t)slot * p.partials_stride + work] = total;
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
constexpr int kRowsBlock = 1024;
constexpr int kRowsWaves = kRowsBlock / kWave;

// Per-object LDS byte offsets of k_mixture_rows, rebuilt whenever a slot's group ids change; quad-major so that the
// offsets a wave step needs (its object quads x (C+1) x 4 objects) are one contiguous run of dwords:
//   out[q][c][j]  (c < C) = row(g_c(4q+j)) * (S+1) * FT * 4      row = global group index, Gtot for "no group"
//   out[q][C][j]          = pattern(4q+j) * ceil(C/2) * FT * 16  (weight planes of the object's has_components pattern)
// objects n >= N of the last quad get the "no group" row and pattern 0 (their state bytes are NA).
__global__ void k_rowoff(const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid, uint32_t* __restrict__ out,
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

template <int MODE, int FT, int CT>
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
#ifdef SBE_STAMPS
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 48 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t state_bytes = FT * 4;                                  // one state row of a group block
    const uint32_t row_bytes = (uint32_t)S1 * state_bytes;
    const uint32_t tab_bytes = (uint32_t)(p.Gtot + 1) * row_bytes;
    const int q0 = chunk * p.quads_per_chunk;
    const int q1 = min(p.NQ, q0 + p.quads_per_chunk);
    const int nq = q1 - q0;                                               // >= 1

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
#ifdef SBE_STAMPS
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 48 + 1] = __builtin_amdgcn_s_memrealtime();
#endif

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
    double thread_ll;
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
#ifdef SBE_STAMPS
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 48 + 2] = __builtin_amdgcn_s_memrealtime();
        if (p.stamps && lane == 0) p.stamps[(size_t)blockIdx.x * 48 + 4 + wave] = __builtin_amdgcn_s_memrealtime();       // per wave:
        if (p.stamps && lane == 0) p.stamps[(size_t)blockIdx.x * 48 + 20 + wave] = __builtin_amdgcn_ballot_w64(pa.bad != 0u);   // loop end, bad lanes
#endif
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
    const double wsum = wave_sum(thread_ll);
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kRowsWaves; ++w) total += red[w];            // fixed order: run-to-run deterministic
        p.partials[(int64_t)slot * p.partials_stride + work] = total;
#ifdef SBE_STAMPS
        if (p.stamps) p.stamps[(size_t)blockIdx.x * 48 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
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
    if (threadIdx.x == 0) p.partials[(int64_t)slot * p.partials_stride + work] = total;
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
#ifdef SBE_STAMPS
    uint64_t stamp[8]; int n_stamp = 0;
    const uint64_t rt0 = wall_clock64();
#define SBE_STAMP() do { stamp[n_stamp++] = __builtin_readcyclecounter(); } while (0)
#else
#define SBE_STAMP() do {} while (0)
#endif
    SBE_STAMP();
    if ((uint32_t)(uintptr_t)(lds_uchar_t*)lds_raw != 0u) {       // absolute LDS addressing needs base 0
        if (threadIdx.x == 0) p.partials[(int64_t)slot * p.partials_stride + work] = __longlong_as_double(0x7FF8000000000000ll);
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
    SBE_STAMP();
    __syncthreads();
    SBE_STAMP();

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
#ifdef SBE_ABL_NOLOG
                        double lg[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) lg[g] = vv[g];
#else
                        double lg[G];
                        bool special = false;
#pragma unroll
                        for (int g = 0; g < G; ++g) special |= tab_log_special(vv[g]);
                        tab_log_core_n<G>(vv, lg, tab_off);
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0ull, 0)) {     // rare: library log
#pragma unroll
                            for (int g = 0; g < G; ++g) if (tab_log_special(vv[g])) lg[g] = lib_log(vv[g]);
                        }
#endif
#pragma unroll
                        for (int g = 0; g < G; ++g)
                            if (ok[g]) *(lds_double_t*)(uintptr_t)dd[g] = lg[g];
                    }
                }
            }
        }
        SBE_STAMP();
        __syncthreads();
        SBE_STAMP();

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
#ifdef SBE_ABL_NOGATHER
        if (p.N < 0)
#endif
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
#ifdef SBE_ABL_NOGATHER
        if (p.N < 0)
#endif
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
        SBE_STAMP();
        __syncthreads();
        SBE_STAMP();
#ifdef SBE_ABL_NOGATHER
        if (p.N < 0)
#endif
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
    SBE_STAMP();
    {   // fixed-order block reduction over NW waves
        const double wsum = wave_sum((a0 + a2) + (a1 + a3));
        if (lane == 0) red4[w] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            double total = 0.0;
            if (NW == 4) total = (red4[0] + red4[1]) + (red4[2] + red4[3]);
            else {
#pragma unroll
                for (int i = 0; i < NW; ++i) total += red4[i];
            }
            p.partials[(int64_t)slot * p.partials_stride + work] = total;
        }
    }
#ifdef SBE_STAMPS
    SBE_STAMP();
    if (p.stamps && lane == 0) {
        uint64_t* o = p.stamps + ((int64_t)blockIdx.x * 4 + (w & 3)) * 12;
        for (int i = 0; i < n_stamp; ++i) o[i] = stamp[i];
        o[7] = (uint64_t)ragged; o[8] = rt0; o[9] = wall_clock64();
    }
#endif
}
#undef SBE_STAMP
#undef SBE_SDWA_ADD

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

// One BLOCK per available object, thread <-> feature: the with/without weight row of the object's pattern is
// computed in place (no table kernel in front), a lane's loads (state byte, weights, table entries of every
// component) are all in flight together and there is one dependent-load chain per object instead of one per
// 64 features; the two fp64 logs per observation are table-driven (tab_log_pos; the sums of logs carry far more
// accuracy than the reference's linear-space products).  Fixed-order block reduction: deterministic.
__global__ __launch_bounds__(kBlock) void k_cluster_marginals(
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
        float wc[kMaxComponents], wf[kMaxComponents];
        weight_tables_z_row(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, wc, wf);   // this object's pattern
        double v0 = 0.0, v1 = 0.0;
        for (int c = 0; c < C; ++c) {
            double lh = 1.0;
            if (x != kNA) {
                if (c == 0) lh = (double)table0[(int64_t)f * S + x];
                else {
                    const uint16_t gg = gid[(int64_t)c * Np + n];
                    lh = gg == kNoGroup ? 0.0 : (double)probs[((int64_t)gg * F + f) * S + x];
                }
            }
            const double a = (double)wc[c], b = (double)wf[c];
            v1 = v1 + lh * (inside ? a : b);         // z = 1: the object is (or becomes) a cluster member
            v0 = v0 + lh * (inside ? b : a);
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
__global__ __launch_bounds__(kBlock) void k_jump_lh(
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
        float wc[kMaxComponents], wf[kMaxComponents];
        weight_tables_z_row(weights + (int64_t)f * C, bits, C, inv_tp, use_pow, wc, wf);   // wc = weights_heated row
        float pc = 0.0f;
        for (int c = 1; c < C; ++c) {
            const uint16_t gg = gid[(int64_t)c * Np + n];
            if (gg != kNoGroup) pc = pc + wc[c] * pconf[((int64_t)(gg - G0) * F + f) * S + x];
        }
        const float ps = pc + wc[0] * p_source[(int64_t)f * S + x];
        const float pt = pc + wc[0] * p_target[(int64_t)f * S + x];
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
__global__ __launch_bounds__(1024) void k_source_lh_by_feature(
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

__global__ void k_test_philox(const uint32_t* __restrict__ ctr_key, int n, uint32_t* __restrict__ out) {
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
__device__ inline bool source_posterior_row(const SrcPostArgs& a, int n, int f, float* p) {
    const uint8_t x = a.state[(int64_t)n * a.Fp + f];
    const float* w = a.wpat + ((int64_t)a.pid[n] * a.F + f) * a.C;
    if (a.from_prior) {
        auto term = [&](int c) -> float { return a.pow_w ? powf(w[c], a.inv_tp) : w[c]; };
        const float total = np_pairwise_sum<float>(term, a.C);
        for (int c = 0; c < a.C; ++c) p[c] = term(c) / total;
        return total > 0.0f;
    }
    auto term = [&](int c) -> double {
        double lh = 1.0;
        if (x != kNA) {
            const uint16_t gg = a.gid[(int64_t)c * a.Np + n];
            lh = gg == kNoGroup ? 0.0 : (double)a.probs[((int64_t)gg * a.F + f) * a.S + x];
        }
        if (a.pow_lh) lh = pow(lh, a.inv_t);
        const float wc = a.pow_w ? powf(w[c], a.inv_tp) : w[c];
        return lh * (double)wc;
    };
    const double total = np_pairwise_sum<double>(term, a.C);
    for (int c = 0; c < a.C; ++c) p[c] = (float)(term(c) / total);
    return total > 0.0;
}

__global__ void k_source_posterior(SrcPostArgs a, float* __restrict__ out, int* __restrict__ status, DoneSig done = DoneSig{}) {
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

__global__ __launch_bounds__(kBlock) void k_sample_source(SrcPostArgs a, const double* __restrict__ z, uint64_t seed,
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
__global__ __launch_bounds__(kBlock) void k_source_logprob(SrcPostArgs a, const uint8_t* __restrict__ src,
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
__global__ __launch_bounds__(kBlock) void k_sum_log_f32(const float* __restrict__ v, int64_t n,
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
__global__ void k_subset_lh(const uint8_t* __restrict__ state, const float* __restrict__ tables,
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

__global__ __launch_bounds__(kBlock) void k_given_unchanged_gibbs(GuGibbsArgs a, uint8_t* __restrict__ src_new,
                                                                 float* __restrict__ sel_new, float* __restrict__ sel_back,
                                                                 int* __restrict__ status, DoneSig done = DoneSig{}) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)a.n_sub * a.F) {
        const int r = (int)(i / a.F), f = (int)(i % a.F);
        const uint8_t x = a.state[(int64_t)a.objects[r] * a.Fp + f];
        const bool na = x == kNA;
        float lh[kMaxComponents];
        for (int c = 0; c < a.C; ++c) {
            float v = 1.0f;
            if (!na) {
                const int g = a.group_idx[(int64_t)c * a.n_sub + r];
                v = g < 0 ? 0.0f : a.tables[((int64_t)(a.table_offsets[c] + g) * a.F + f) * a.S + x];
            }
            lh[c] = a.pow_lh ? powf(v, a.inv_t) : v;
        }
        const float* w = a.weights + (int64_t)f * a.C;
        float p[2][kMaxComponents];
        bool ok = true;
        for (int side = 0; side < 2; ++side) {
            const uint8_t* hc = (side == 0 ? a.hc_new : a.hc_old) + (int64_t)r * a.C;
            auto masked = [&](int c) -> float { return hc[c] ? w[c] : 0.0f * w[c]; };
            const float wtot = np_pairwise_sum<float>(masked, a.C);
            float t[kMaxComponents];
            for (int c = 0; c < a.C; ++c) {
                float wc = masked(c) / wtot;                            // normalize_weights (likelihood.py:171-190)
                if (a.pow_w) wc = powf(wc, a.inv_tp);
                t[c] = a.from_prior ? wc : wc * lh[c];
            }
            if (a.from_prior) { for (int c = 0; c < a.C; ++c) p[side][c] = t[c]; continue; }
            auto term = [&](int c) -> float { return t[c]; };
            const float tot = np_pairwise_sum<float>(term, a.C);
            ok = ok && tot > 0.0f;                                      // normalize's assert (util.py:1006)
            for (int c = 0; c < a.C; ++c) p[side][c] = t[c] / tot;
        }
        if (!ok) raise_status(status, ST_BAD_NORMALIZE, 1);
        // sample_categorical (preprocessing.py:224-256): float32 cumulative sums, divided by the last, first c with z < cdf[c]
        float cdf[kMaxComponents];
        float run = p[0][0];
        cdf[0] = run;
        for (int c = 1; c < a.C; ++c) { run = run + p[0][c]; cdf[c] = run; }
        const float last = cdf[a.C - 1];
        const double zz = a.z[i];
        int k = 0;
        for (int c = a.C - 1; c >= 0; --c)
            if (zz < (double)(cdf[c] / last)) k = c;
        src_new[i] = na ? (uint8_t)kNA : (uint8_t)k;
        float sn = 1.0f, sb = 1.0f;
        const int id_old = a.src_old[i];
        for (int c = 0; c < a.C; ++c) {
            sn = (!na && c == k) ? p[0][c] : sn;
            sb = (c == id_old) ? p[1][c] : sb;
        }
        sel_new[i] = sn;
        sel_back[i] = sb;
    }
    signal_done(done);
}

// SourcePrior.__call__ (prior.py:573-611), per-object values:
//   sp[n] = float32( sum_{f valid} log( w[pat(n)][f][source(n,f)] ) )     (float32 logs)
// One wave per object, lanes over features, wave64 shuffle reduce.  An observation whose source has no
// component set contributes log(0) = -inf like the reference's sum(w * s) = 0.
__global__ __launch_bounds__(1024) void k_source_prior(const uint8_t* __restrict__ state,
                                                        const uint8_t* __restrict__ src,
                                                        const uint8_t* __restrict__ pid,
                                                        const float* __restrict__ wpat, double* __restrict__ out,
                                                        int N, int F, int C, int Fp, DoneSig done = DoneSig{}) {
    const int lane = threadIdx.x & (kWave - 1);
    const int n = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);      // (16 objects per 1024-thread block: a block's
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
    signal_done(done);
}

// One-launch copy of all per-slot arrays (sbe_copy_slot): up to 16 dword-granular segments.
struct CopySegs {
    const uint32_t* src[16];
    uint32_t* dst[16];
    uint32_t end[16];        // exclusive prefix end of each segment, in dwords
    int n;
};

__global__ void k_multi_copy(CopySegs cs) {
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
__global__ void k_conc_lgamma(const double* __restrict__ conc, double* __restrict__ lg_conc, double* __restrict__ sum_a,
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

__global__ __launch_bounds__(kBlock) void k_step_core(StepCore a) {
    extern __shared__ __align__(16) unsigned char core_lds[];
    step_core_body(a, core_lds, (int)blockIdx.x);
}

// sbe_step_batch: one chain per blockIdx.y, each with its own StepCore (current / candidate slot, payload, counts ...)
// (the single-chain kernel takes what registers it wants -- 255, its lgamma expansions are wide -- at two blocks per CU;
//  a batch has thousands of blocks and is better off at 128 VGPRs / four blocks per CU despite more spills: 225 against
//  231 us per 64-chain sweep; 64 VGPRs: 268 us)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
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
__global__ void k_tile_probs(const float* __restrict__ probs, float* __restrict__ probs_t, int g_lo,
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
__global__ __launch_bounds__(kBlock) void k_counts_delta(
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

// float32 count rows of listed groups -> the slot's resident int32 counts (Engine.set_counts_rows: the bind cache
// sends only the groups whose rows differ from what the slot holds)
__global__ void k_set_count_rows(const float* __restrict__ rows /* [n][F][S] */, const int32_t* __restrict__ group_idx,
                                 int32_t* __restrict__ counts /* slot's [Gtot][F][S] */, int n, int64_t fs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * fs) return;
    counts[(int64_t)group_idx[i / fs] * fs + i % fs] = (int32_t)rows[i];
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
__global__ __launch_bounds__(kUnchangedBlock) void k_unchanged_counts(
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

}  // namespace sbe
