// sbe_device_common.hip.h -- constants, completion signals and device helpers shared by every kernel of the engine
// (NumPy-order sums, wave / block reductions, the fp64 log and lgamma routines).  No __global__ function lives here: the
// header is included by both translation units (sbe_engine.hip, sbe_mixture.hip).
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
struct ChunkSig { unsigned* tickets; unsigned long long* flags; unsigned long long seq; unsigned blocks_per_chunk; unsigned n_blocks;
                  unsigned n_chunks; long long chunk_elems; };     // chunk_elems > 0: ORDERED form (every block walks the chunks in turn)
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

// ORDERED form: all blocks of the (small) grid work on chunk 0, then on chunk 1, ...: a chunk is complete -- and its flag
// raised by the last block to get there -- while the later chunks have not been started, so the host copy of chunk k runs
// under the transfer of chunk k+1 (with one block per 2 x 256 output elements and every block resident at once, as in the
// plain form, all chunks complete together at the end of the kernel).
__device__ __forceinline__ void signal_chunk_ordered(const ChunkSig& c, unsigned chunk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        const unsigned t = __hip_atomic_fetch_add(c.tickets + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(c.tickets + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.flags + chunk, c.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
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

// np_block_sum over a REGISTER array of capacity CAP (a multiple of 8, <= 128; n <= CAP): the same additions in the same
// order, written as fully unrolled, guarded code so that `v` is never indexed dynamically.  Why it exists: a run-time trip
// count makes every get(i) of the general form a load the loop waits for before the next iteration (and a dynamically
// indexed array lives in scratch memory); the small table kernels of the drop-in path are chains of such waits -- a
// 10-state row cost 9 us.  With the values loaded up front (all loads in flight together) the sum is arithmetic only.
template <class T, int CAP>
__device__ __forceinline__ T np_sum_regs(const T (&v)[CAP], int n) {
    static_assert(CAP % 8 == 0 && CAP <= 128, "one NumPy leaf");
    if (n < 8) {
        T res = T(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) if (i < n) res = res + v[i];
        return res;
    }
    T r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int k = 1; k < CAP / 8; ++k)
        if (8 * k + 8 <= n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + v[8 * k + j];
        }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const int lim = n - (n % 8);
#pragma unroll
    for (int i = 8; i < CAP; ++i) if (i >= lim && i < n) res = res + v[i];
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
// powf / pow of the tempered paths (a temperature other than 1: heated MC3 chains only).  Inlined at every use they were
// most of the 19 800 instructions of the fused marginals kernel -- code a latency-bound kernel has to fetch cold.
__device__ __attribute__((noinline)) float lib_powf(float a, float b) { return powf(a, b); }
__device__ __attribute__((noinline)) double lib_pow(double a, double b) { return pow(a, b); }

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

}  // namespace sbe
