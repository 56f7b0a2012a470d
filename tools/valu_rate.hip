// Micro-benchmark: issue cost of gfx950 vector instructions per SIMD, by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/valu_rate.hip -o tools/valu_rate && ./tools/valu_rate
// Each wave runs ITERS iterations of 32 instructions of one kind (4 independent chains); the wall-clock time of the
// whole launch (HIP events) gives ns -- and cycles at the nominal 2.4 GHz -- per instruction per SIMD.
// (s_memtime does NOT tick at the shader clock on this part: the first version of this tool reported its ticks.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
// u0..u3: uint chains, u4..u7 read-only; a*: float; d*: double
#define BODY_U(txt) REP8(asm volatile(txt : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u4), "v"(u5), "v"(u6), "v"(u7) : "vcc", "s40", "s41", "s42", "s43");)
#define BODY_D(txt) REP8(asm volatile(txt : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4), "v"(d5), "v"(d6), "v"(d7), "v"(u4), "v"(a4) : "vcc", "s40", "s41");)
#define BODY_A(txt) REP8(asm volatile(txt : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(u4) : "vcc");)

template <int KIND>
__global__ void k(unsigned long long* out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    __shared__ unsigned lds_buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds_buf[i] = i;
    __syncthreads();
    const unsigned lane_ = threadIdx.x & 63u;
    const unsigned la = (unsigned)(size_t)lds_buf + lane_ * 4u, la8 = (unsigned)(size_t)lds_buf + lane_ * 8u, la16 = (unsigned)(size_t)lds_buf + lane_ * 16u;
    const unsigned lb = (unsigned)(size_t)lds_buf + (lane_ >> 5) * 64u;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    const double dk = seed + 0.5;
    unsigned u0 = (unsigned)seed + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { BODY_U("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7") }
        if (KIND == 1) { BODY_A("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5") }
        if (KIND == 2) { BODY_D("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5") }
        if (KIND == 3) { BODY_D("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4") }
        if (KIND == 4) { BODY_D("v_cvt_f64_f32 %0, %9\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %9\n v_cvt_f64_f32 %3, %9") }
        if (KIND == 5) { BODY_D("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5") }
        if (KIND == 6) { BODY_D("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7") }
        if (KIND == 7) { BODY_U("v_lshl_add_u32 %0, %0, 7, %4\n v_lshl_add_u32 %1, %1, 7, %5\n v_lshl_add_u32 %2, %2, 7, %6\n v_lshl_add_u32 %3, %3, 7, %7") }
        if (KIND == 8) { BODY_U("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc") }
        if (KIND == 9) { BODY_D("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4") }
        if (KIND == 10) { BODY_U("v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_cndmask_b32_e64 %1, %1, %5, s[40:41]\n v_cndmask_b32_e64 %2, %2, %6, s[40:41]\n v_cndmask_b32_e64 %3, %3, %7, s[40:41]") }
        if (KIND == 11) { BODY_U("v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %5\n v_cmp_gt_u32 vcc, %2, %6\n v_cmp_gt_u32 vcc, %3, %7") }
        if (KIND == 12) { BODY_U("v_cmp_gt_u32_e64 s[40:41], %0, %4\n v_cmp_gt_u32_e64 s[42:43], %1, %5\n v_cmp_gt_u32_e64 s[40:41], %2, %6\n v_cmp_gt_u32_e64 s[42:43], %3, %7") }
        if (KIND == 13) { BODY_U("v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_u32 vcc, %1, %5\n v_cndmask_b32 %1, %1, %5, vcc") }
        if (KIND == 14) { BODY_U("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8") }
        if (KIND == 15) { BODY_U("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %7") }
        if (KIND == 16) { BODY_U("v_lshrrev_b32 %0, 8, %0\n v_lshrrev_b32 %1, 8, %1\n v_lshrrev_b32 %2, 8, %2\n v_lshrrev_b32 %3, 8, %3") }
        if (KIND == 17) { BODY_U("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %5\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %7") }
        if (KIND == 18) { BODY_U("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %2, %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %3, %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0") }
        if (KIND == 19) { BODY_U("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %5, %6\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %7, %4") }
        if (KIND == 20) { BODY_U("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7") }
        if (KIND == 21) { BODY_U("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %5\n v_or_b32 %2, %2, %6\n v_or_b32 %3, %3, %7") }
        if (KIND == 22) { BODY_U("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4") }
        if (KIND == 23) { BODY_U("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7") }
        if (KIND == 24) { BODY_U("v_readlane_b32 s40, %0, 3\n v_readlane_b32 s41, %1, 3\n v_readlane_b32 s42, %2, 3\n v_readlane_b32 s43, %3, 3") }
        if (KIND == 25) { BODY_D("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8") }
        if (KIND == 26) { BODY_D("v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1\n v_frexp_mant_f64 %2, %2\n v_frexp_mant_f64 %3, %3") }
        if (KIND == 27) { BODY_D("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4") }
        if (KIND == 28) { BODY_U("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %5, %1, %6\n v_bfi_b32 %2, %6, %2, %7\n v_bfi_b32 %3, %7, %3, %4") }
        if (KIND == 29) { BODY_U("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %5, %6\n v_perm_b32 %2, %2, %6, %7\n v_perm_b32 %3, %3, %7, %4") }
        if (KIND == 30) { BODY_D("v_cmp_gt_f64 vcc, %0, %4\n v_cmp_gt_f64 vcc, %1, %4\n v_cmp_gt_f64 vcc, %2, %4\n v_cmp_gt_f64 vcc, %3, %4") }
        if (KIND == 31) { BODY_U("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %5\n v_add_co_u32 %2, vcc, %2, %6\n v_add_co_u32 %3, vcc, %3, %7") }
        if (KIND == 32) { BODY_U("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %7") }
        if (KIND == 33) { BODY_U("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3") }
        if (KIND == 34) { BODY_U("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %6\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %7, %4") }
        if (KIND == 35) { BODY_A("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3") }
        if (KIND == 36) { BODY_D("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3") }
        if (KIND == 37) { BODY_A("v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %8\n v_cvt_f32_u32 %2, %8\n v_cvt_f32_u32 %3, %8") }
        if (KIND == 38) { BODY_U("v_cmp_eq_u32 vcc, %0, %4\n v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_cmp_eq_u32 vcc, %1, %5\n v_addc_co_u32 %1, vcc, %1, %5, vcc") }
        // --- mixes (32 instructions per iteration as well) ---
        if (KIND == 40) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %4, %4, %10\n v_add_u32 %5, %5, %10" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        if (KIND == 41) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        if (KIND == 42) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        if (KIND == 43) { REP8(asm volatile("v_add_u32 %4, %4, %10\n v_add_u32 %4, %4, %10\n v_add_u32 %4, %4, %10\n v_add_u32 %4, %4, %10" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        if (KIND == 44) { REP8(asm volatile("v_cvt_f64_f32 %0, %11\n v_fma_f64 %1, %0, %8, %1\n v_cvt_f64_f32 %2, %11\n v_fma_f64 %1, %2, %8, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4), "v"(a4));) }
        if (KIND == 45) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_lshl_add_u32 %4, %4, 7, %10\n v_fma_f64 %1, %1, %8, %9\n v_lshl_add_u32 %5, %5, 7, %10" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        if (KIND == 46) { REP8(asm volatile("v_add_u32 %4, %4, %10\n v_lshl_add_u32 %5, %5, 7, %10\n v_add_u32 %6, %6, %10\n v_lshl_add_u32 %7, %7, 7, %10" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        if (KIND == 47) { REP8(asm volatile("v_add_u32 %4, %4, %10\n v_add_u32 %5, %4, %10\n v_add_u32 %6, %5, %10\n v_add_u32 %7, %6, %10" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d4), "v"(d5), "v"(u4));) }
        // --- LDS reads (conflict-free, lane-contiguous) alone and mixed with VALU; 32 instructions per iteration ---
        if (KIND == 50) { REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(la));) }
        if (KIND == 51) { REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(la8));) }
        if (KIND == 52) { REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(la16));) }
        // 16 LDS reads + 16 VALU per iteration (time is reported per instruction over all 32)
        if (KIND == 53) { REP8(asm volatile("ds_read_b32 %0, %8\n v_fma_f64 %4, %4, %9, %9\n ds_read_b32 %1, %8 offset:256\n v_fma_f64 %5, %5, %9, %9\n s_waitcnt lgkmcnt(0)" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(la), "v"(d4));) }
        if (KIND == 54) { REP8(asm volatile("ds_read_b128 %0, %8\n v_fma_f64 %4, %4, %9, %9\n ds_read_b128 %1, %8 offset:1024\n v_fma_f64 %5, %5, %9, %9\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(la16), "v"(d4));) }
        if (KIND == 55) { REP8(asm volatile("ds_read_b64 %0, %8\n v_fma_f64 %4, %4, %9, %9\n ds_read_b64 %1, %8 offset:512\n v_fma_f64 %5, %5, %9, %9\n s_waitcnt lgkmcnt(0)" : "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(la8), "v"(dk));) }
        // broadcast b128 (all lanes of a half-wave read the same 16 bytes: the rows kernel's offset reads)
        if (KIND == 56) { REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(lb));) }
    }
    if (q0.x + q1.x + q2.x + q3.x == 77u) out[1] = 0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) == 1234.5f) out[0] = 0;
}

template <int KIND>
void run(const char* name, unsigned long long* d_out) {
    const int iters = 10000;                        // 32 instructions per iteration
    printf("%-28s", name);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd), blocks_per_cu = waves_per_simd > 4 ? 2 : 1;
        const int n_blocks = 256 * blocks_per_cu;
        k<KIND><<<n_blocks, threads>>>(d_out, iters, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<KIND><<<n_blocks, threads>>>(d_out, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double ns = (double)ms * 1e6 / ((double)waves_per_simd * iters * 32.0);     // per instruction per SIMD
        printf("  w%d %6.3f ns %5.2f cyc", waves_per_simd, ns, ns * 2.4);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, 1 << 20);
    printf("ns / cycles (at 2.4 GHz) per wave64 instruction per SIMD, by waves per SIMD\n");
    run<0>("v_add_u32", d_out); run<32>("v_sub_u32", d_out); run<15>("v_and_b32", d_out); run<21>("v_or_b32", d_out);
    run<16>("v_lshrrev_b32", d_out); run<33>("v_lshlrev_b32", d_out); run<17>("v_min_u32", d_out); run<23>("v_mov_b32", d_out);
    run<18>("v_add_u32_sdwa", d_out); run<31>("v_add_co_u32 (vcc)", d_out);
    run<7>("v_lshl_add_u32", d_out); run<14>("v_bfe_u32", d_out); run<22>("v_add3_u32", d_out); run<34>("v_and_or_b32", d_out);
    run<28>("v_bfi_b32", d_out); run<29>("v_perm_b32", d_out); run<19>("v_mad_u32_u24", d_out); run<20>("v_mul_lo_u32", d_out);
    run<11>("v_cmp_gt_u32 -> vcc", d_out); run<12>("v_cmp_gt_u32_e64 -> sgpr", d_out); run<8>("v_cndmask_b32 (vcc)", d_out);
    run<10>("v_cndmask_b32_e64 (sgpr)", d_out); run<13>("v_cmp + v_cndmask pairs", d_out); run<38>("v_cmp + v_addc pairs", d_out);
    run<24>("v_readlane_b32", d_out);
    run<1>("v_fma_f32", d_out); run<5>("v_pk_fma_f32", d_out); run<35>("v_log_f32", d_out); run<37>("v_cvt_f32_u32", d_out);
    run<2>("v_fma_f64", d_out); run<3>("v_add_f64", d_out); run<9>("v_mul_f64", d_out); run<4>("v_cvt_f64_f32", d_out);
    run<6>("v_mov_b64", d_out); run<27>("v_max_f64", d_out); run<25>("v_ldexp_f64", d_out); run<26>("v_frexp_mant_f64", d_out);
    run<30>("v_cmp_gt_f64 -> vcc", d_out); run<36>("v_rcp_f64", d_out);
    printf("mixes (per instruction, 32 per iteration):\n");
    run<40>("2 fma_f64 + 2 add_u32 grouped", d_out); run<41>("fma_f64 / add_u32 alternating", d_out);
    run<45>("fma_f64 / lshl_add alternating", d_out); run<46>("add_u32 / lshl_add alternating", d_out);
    run<42>("fma_f64 dependent chain", d_out); run<43>("add_u32 dependent chain", d_out); run<47>("add_u32 chain via 4 regs", d_out);
    run<44>("cvt_f64_f32 -> fma_f64 chain", d_out);
    printf("LDS (per instruction; mixes: 16 reads + 16 fma_f64 per iteration, per instruction over all 32):\n");
    run<50>("ds_read_b32", d_out); run<51>("ds_read_b64", d_out); run<52>("ds_read_b128", d_out); run<56>("ds_read_b128 broadcast", d_out);
    run<53>("ds_read_b32 / fma_f64", d_out); run<55>("ds_read_b64 / fma_f64", d_out); run<54>("ds_read_b128 / fma_f64", d_out);
    return 0;
}
