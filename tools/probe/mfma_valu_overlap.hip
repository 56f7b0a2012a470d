// Does fp64 vector work of a wave run in the shadow of its own i8 MFMAs on gfx950?  One block per CU, W waves per SIMD;
// every loop iteration issues NM v_mfma_i32_32x32x32_i8 (three independent accumulators) and NV fp64 FMAs (four
// independent chains), interleaved by sched_group_barrier.  Printed: time per iteration per wave for vector-only,
// matrix-only and both, in nanoseconds and in 2.4 GHz cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NV, int NM>
__global__ void k(const v4i* ab, double* out, int iters, double seed) {
    v16i acc[3] = {};
    const v4i a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];
    double x[4] = {seed, seed + 1.0, seed + 2.0, seed + 3.0};
    const double m = 1.0 + 1e-9 * seed, c = 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i % 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i % 3], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i & 3] = fma(x[i & 3], m, c);
        if constexpr (NM > 0 && NV > 0) {
            constexpr int PER = NV / (NM > 0 ? NM : 1);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
            }
        }
    }
    double s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += (double)acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// inter-wave form: waves 0-3 of the block (one per SIMD) issue only MFMAs, waves 4-7 (their SIMD partners) only FMAs
__global__ void k_split(const v4i* ab, double* out, int iters_m, int iters_v, double seed) {
    v16i acc[3] = {};
    const v4i a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];
    double x[4] = {seed, seed + 1.0, seed + 2.0, seed + 3.0};
    const double m = 1.0 + 1e-9 * seed, c = 1e-3;
    if ((threadIdx.x >> 6) < 4) {
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int i = 0; i < 42; ++i) x[i & 3] = fma(x[i & 3], m, c);
        }
    }
    double s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += (double)acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float run_split(int iters_m, int iters_v, const v4i* ab, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_split<<<256, 512>>>(ab, out, 10, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_split<<<256, 512>>>(ab, out, iters_m, iters_v, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int NV, int NM>
static float run(int threads, int iters, const v4i* ab, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NM><<<256, threads>>>(ab, out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, NM><<<256, threads>>>(ab, out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    v4i* ab; double* out;
    hipMalloc(&ab, 128 * 16); hipMemset(ab, 1, 128 * 16);
    hipMalloc(&out, 256 * 1024 * 8);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        const int wps = threads / 256;
        const float tv = run<42, 0>(threads, iters, ab, out), tm = run<0, 3>(threads, iters, ab, out), tb = run<42, 3>(threads, iters, ab, out);
        const float tv2 = run<84, 0>(threads, iters, ab, out), tb2 = run<84, 3>(threads, iters, ab, out);
        auto per = [&](float ms) { return ms * 1e6 / iters; };
        printf("%d waves per SIMD, per iteration of ONE wave (ns; cycles at 2.4 GHz per SIMD = ns * 2.4):\n", wps);
        printf("   42 fp64 FMA only        %7.1f ns  (%5.1f cycles per FMA per SIMD)\n", per(tv), per(tv) * 2.4 / (42 * wps));
        printf("   3 MFMA only             %7.1f ns  (%5.1f cycles per MFMA per SIMD)\n", per(tm), per(tm) * 2.4 / (3 * wps));
        printf("   42 FMA + 3 MFMA         %7.1f ns  (sum of the two alone %7.1f, max %7.1f)\n", per(tb), per(tv) + per(tm), per(tv) > per(tm) ? per(tv) : per(tm));
        printf("   84 fp64 FMA only        %7.1f ns\n", per(tv2));
        printf("   84 FMA + 3 MFMA         %7.1f ns  (sum %7.1f, max %7.1f)\n", per(tb2), per(tv2) + per(tm), per(tv2) > per(tm) ? per(tv2) : per(tm));
    }
    // inter-wave: the MFMA wave runs 2x the iterations so that both halves take about as long alone
    const float sm = run_split(2 * iters, 0, ab, out), sv = run_split(0, iters, ab, out), sb = run_split(2 * iters, iters, ab, out);
    printf("inter-wave (one MFMA-only wave and one FMA-only wave per SIMD): MFMA wave alone %.3f ms, FMA wave alone %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n",
           sm, sv, sb, sm + sv, sm > sv ? sm : sv);
    return 0;
}
