// Probe for v_mfma_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands on gfx950, as a carrier of EXACT 0/1 count contractions
// (VERDICT r5 next #3): 0 = 0x0 and 1.0 = 0x2 are exact in e2m1, sums of products <= 2^24 are exact in the f32 accumulator.
//   (1) layout: D[row][col] = sum over the 64 (lane half, nibble) positions of A-nibble x B-nibble with row / col = lane & 31 and
//       the standard 32x32 C/D map; both operands take k from the same (lane half, nibble) position, so the instruction's
//       internal k order does not matter -- checked with asymmetric random 0/1 data, scale operands 0 (the compiler then
//       selects the unscaled form: no scale registers);
//   (2) issue rate: cycles per instruction of back-to-back independent MFMAs, one wave per SIMD, against v_mfma_i32_32x32x32_i8.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_fp4_probe.hip -o /tmp/mfma_fp4_probe && /tmp/mfma_fp4_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_layout(const v4i* a, const v4i* b, v16f* d) {
    const v4i a4 = a[threadIdx.x], b4 = b[threadIdx.x];
    const v8i a8 = {a4.x, a4.y, a4.z, a4.w, 0, 0, 0, 0}, b8 = {b4.x, b4.y, b4.z, b4.w, 0, 0, 0, 0};
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 4, 4, 0, 0, 0, 0);
    d[threadIdx.x] = acc;
}

template <int FP4>
__global__ void k_rate(const v4i* a, const v4i* b, float* out, long long* cycles, int iters) {
    const v4i a4 = a[threadIdx.x & 63], b4 = b[threadIdx.x & 63];
    const v8i a8 = {a4.x, a4.y, a4.z, a4.w, 0, 0, 0, 0}, b8 = {b4.x, b4.y, b4.z, b4.w, 0, 0, 0, 0};
    v16f f[4] = {};
    v16i q[4] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (FP4) f[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f[u], 4, 4, 0, 0, 0, 0);
            else q[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, q[u], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) s += FP4 ? f[u][i] : (float)q[u][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    std::vector<uint8_t> A(32 * 64), B(64 * 32);          // A[row][k], B[k][col], values 0 / 1
    srand(11);
    for (auto& v : A) v = (uint8_t)(rand() % 3 == 0);
    for (auto& v : B) v = (uint8_t)(rand() % 2);
    std::vector<int> D(32 * 32, 0);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { int s = 0; for (int kk = 0; kk < 64; ++kk) s += A[r * 64 + kk] * B[kk * 32 + c]; D[r * 32 + c] = s; }
    // lane l (r = l & 31, h = l >> 5), nibble j = 0..31 (byte j >> 1, low nibble first) <- k = 32 h + j; 1.0 = 0x2 in e2m1
    std::vector<uint8_t> fa(64 * 16, 0), fb(64 * 16, 0);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
        const int kk = 32 * (l >> 5) + j;
        if (A[(l & 31) * 64 + kk]) fa[l * 16 + (j >> 1)] |= (uint8_t)(0x2 << (4 * (j & 1)));
        if (B[kk * 32 + (l & 31)]) fb[l * 16 + (j >> 1)] |= (uint8_t)(0x2 << (4 * (j & 1)));
    }
    void *da, *db, *dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 64 * 64);
    hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>((const v4i*)da, (const v4i*)db, (v16f*)dd);
    std::vector<float> out(64 * 16);
    hipMemcpy(out.data(), dd, 64 * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 16; ++reg) {
        const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
        if (out[l * 16 + reg] != (float)D[row * 32 + col]) { if (bad < 4) printf("  lane %d reg %d: got %g want %d\n", l, reg, out[l * 16 + reg], D[row * 32 + col]); ++bad; }
    }
    printf("fp4 32x32x64, scale operands 0: %d of 1024 outputs differ from the exact 0/1 contraction (row=(reg&3)+8*(reg>>2)+4*(lane>>5), col=lane&31)\n", bad);
    // issue rate: 256 blocks of 256 threads (one wave per SIMD on every CU)
    float* dout; long long* dcyc;
    hipMalloc(&dout, 256 * 256 * sizeof(float)); hipMalloc(&dcyc, sizeof(long long));
    const int iters = 2000;
    for (int fp4 = 0; fp4 < 2; ++fp4) {
        for (int rep = 0; rep < 2; ++rep) {
            if (fp4) k_rate<1><<<256, 256>>>((const v4i*)da, (const v4i*)db, dout, dcyc, iters);
            else k_rate<0><<<256, 256>>>((const v4i*)da, (const v4i*)db, dout, dcyc, iters);
            hipDeviceSynchronize();
        }
        long long cyc = 0;
        hipMemcpy(&cyc, dcyc, sizeof(cyc), hipMemcpyDeviceToHost);
        printf("%s: %.2f counter ticks per MFMA (4 independent accumulators, %d x 4 instructions, one wave per SIMD)\n",
               fp4 ? "v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4)" : "v_mfma_i32_32x32x32_i8            ", (double)cyc / (iters * 4.0), iters);
    }
    return bad != 0;
}
