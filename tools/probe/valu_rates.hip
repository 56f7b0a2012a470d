// Issue cost of the vector instructions the fp64 epilogues are made of, on gfx950: cycles per wave64 instruction per SIMD at
// 1 / 2 / 4 waves per SIMD (one block per CU).  Each kernel runs 32 copies of ONE instruction per loop iteration over eight
// independent register sets (inline asm, so the compiler neither folds nor reorders them).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rates.hip -o build/probe/valu_rates && build/probe/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// D = destination class, S = source class: 'd' = 64-bit VGPR pair, 'f' = 32-bit VGPR
#define KERNEL(NAME, ASM, DT, ST)                                                                          \
    __global__ void NAME(double* out, int iters, double seed) {                                            \
        DT d[8]; ST a[8], b[8];                                                                            \
        for (int i = 0; i < 8; ++i) { d[i] = (DT)(seed + i); a[i] = (ST)(seed * 1.5 + i); b[i] = (ST)(1.0 + 1e-3 * i); } \
        for (int it = 0; it < iters; ++it) {                                                               \
            REP32(ASM)                                                                                     \
        }                                                                                                  \
        double s = 0.0;                                                                                    \
        for (int i = 0; i < 8; ++i) s += (double)d[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                    \
    }

#define A_FMA64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_MUL64(i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_ADD64(i) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_CVT64_32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_CVT64_I32(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_CVT32_64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_FMA32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_ADD32I(i) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_AND32(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_LSHR32(i) asm volatile("v_lshrrev_b32 %0, 9, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %1, %2, %1" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_CLASS64(i) asm volatile("v_cmp_class_f64 vcc, %1, %2" : "+v"(d[i]) : "v"(a[i]), "v"(ib[i]) : "vcc");
#define A_MOV32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
#define A_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(d[i]) : "v"(a[i]), "v"(ib[i]));
#define A_FREXPM64(i) asm volatile("v_frexp_mant_f64 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_LOG32(i) asm volatile("v_log_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_RCP64(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define A_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));

KERNEL(k_fma64, A_FMA64, double, double)
KERNEL(k_mul64, A_MUL64, double, double)
KERNEL(k_add64, A_ADD64, double, double)
KERNEL(k_cvt64_32, A_CVT64_32, double, float)
KERNEL(k_cvt64_i32, A_CVT64_I32, double, int)
KERNEL(k_cvt32_64, A_CVT32_64, float, double)
KERNEL(k_fma32, A_FMA32, float, float)
KERNEL(k_add32i, A_ADD32I, int, int)
KERNEL(k_and32, A_AND32, int, int)
KERNEL(k_lshr32, A_LSHR32, int, int)
KERNEL(k_andor, A_ANDOR, int, int)
KERNEL(k_cndmask, A_CNDMASK, int, int)
KERNEL(k_mov32, A_MOV32, int, int)
KERNEL(k_mad24, A_MAD24, int, int)
KERNEL(k_frexpm64, A_FREXPM64, double, double)
KERNEL(k_log32, A_LOG32, float, float)
KERNEL(k_rcp64, A_RCP64, double, double)
KERNEL(k_pkfma32, A_PKFMA32, double, double)

__global__ void k_class64(double* out, int iters, double seed) {
    double d[8], a[8]; int ib[8];
    for (int i = 0; i < 8; ++i) { d[i] = seed + i; a[i] = seed * 1.5 + i; ib[i] = 0x2FF; }
    for (int it = 0; it < iters; ++it) { REP32(A_CLASS64) }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_ldexp64(double* out, int iters, double seed) {
    double d[8], a[8]; int ib[8];
    for (int i = 0; i < 8; ++i) { d[i] = seed + i; a[i] = seed * 1.5 + i; ib[i] = i; }
    for (int it = 0; it < iters; ++it) { REP32(A_LDEXP64) }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(double*, int, double);

static float run(kern_t k, int threads, int iters, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, threads>>>(out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, threads>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 256 * 1024 * 8);
    const int iters = 4000;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_cvt_f64_f32", k_cvt64_32}, {"v_cvt_f64_i32", k_cvt64_i32},
        {"v_cvt_f32_f64", k_cvt32_64}, {"v_ldexp_f64", k_ldexp64}, {"v_frexp_mant_f64", k_frexpm64}, {"v_cmp_class_f64", k_class64}, {"v_rcp_f64", k_rcp64},
        {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma32}, {"v_log_f32", k_log32}, {"v_add_u32", k_add32i}, {"v_and_b32", k_and32}, {"v_lshrrev_b32", k_lshr32},
        {"v_and_or_b32", k_andor}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov32}, {"v_mad_u32_u24", k_mad24},
    };
    printf("cycles per wave64 instruction per SIMD at 2.4 GHz (one block per CU)\n%-20s %10s %10s %10s\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (auto& e : ks) {
        printf("%-20s", e.name);
        for (int threads : {256, 512, 1024}) {
            const float ms = run(e.k, threads, iters, out);
            const double per = ms * 1e-3 / ((double)iters * 32 * (threads / 256)) * 2.4e9;
            printf(" %10.2f", per);
        }
        printf("\n");
    }
    return 0;
}
