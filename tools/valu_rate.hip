// Micro-benchmark: VALU issue rate of gfx950 per SIMD for wave64 instructions, by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate && ./tools/valu_rate
// Each wave runs ITER iterations of 32 independent instructions of one kind; cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(unsigned long long* out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    unsigned u0 = (unsigned)seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP8(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 2) { REP8(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (KIND == 3) { REP8(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (KIND == 4) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (KIND == 5) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) == 1234.5f) out[0] = 0;
}

template <int KIND>
void run(const char* name, unsigned long long* d_out) {
    const int iters = 2000;                         // 32 instructions per iteration
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd), blocks_per_cu = waves_per_simd > 4 ? 2 : 1;
        const int n_blocks = 256 * blocks_per_cu;
        k<KIND><<<n_blocks, threads>>>(d_out, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)n_blocks * threads / 64);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
        const double per_instr_wave = mean / (iters * 32.0);
        printf("%-14s waves/SIMD %d: %.2f cycles per instruction per wave  =>  %.2f cycles per instruction per SIMD\n", name,
               waves_per_simd, per_instr_wave, per_instr_wave / waves_per_simd);
    }
}

int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, 1 << 20);
    run<0>("v_add_u32", d_out); run<1>("v_fma_f32", d_out); run<2>("v_fma_f64", d_out); run<3>("v_add_f64", d_out);
    run<4>("v_cvt_f64_f32", d_out); run<5>("v_pk_fma_f32", d_out);
    return 0;
}
