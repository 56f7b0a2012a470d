// d2h_probe.hip -- how fast can a few MB get from HBM to host memory on this box?  (the literal a1 / a3 surfaces return
// 1.6 / 3.2 MB per call: DESIGN 5.)  hipcc --offload-arch=gfx950 -O3 tools/d2h_probe.hip -o tools/d2h_probe
//   copy engine (hipMemcpyAsync, one stream / two streams) against a kernel that stores straight into host-mapped
//   pinned memory (posted PCIe writes from all CUs), for 0.4 .. 12.8 MB.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void k_store(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_store_x2(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const size_t max_bytes = (size_t)16 << 20;
    double *d, *h, *hm_dev;
    CK(hipMalloc(&d, max_bytes));
    CK(hipMemset(d, 1, max_bytes));
    CK(hipHostMalloc((void**)&h, max_bytes, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&hm_dev, h, 0));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    std::vector<char> pageable(max_bytes);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (size_t bytes : {(size_t)400 << 10, (size_t)1600 << 10, (size_t)3200 << 10, (size_t)12800 << 10}) {
        const int reps = 200;
        double t_one = 0, t_two = 0, t_k = 0, t_k2 = 0, t_cpy = 0, t_four = 0;
        for (int r = -20; r < reps; ++r) {
            auto a = now();
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
            auto b = now();
            CK(hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s0));
            CK(hipMemcpyAsync((char*)h + bytes / 2, (char*)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s1));
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
            auto c = now();
            k_store<<<1024, 256, 0, s0>>>(d, hm_dev, bytes / 8);
            CK(hipStreamSynchronize(s0));
            auto e = now();
            k_store_x2<<<1024, 256, 0, s0>>>((const double2*)d, (double2*)hm_dev, bytes / 16);
            CK(hipStreamSynchronize(s0));
            auto f = now();
            memcpy(pageable.data(), h, bytes);
            auto g = now();
            for (int q = 0; q < 4; ++q) CK(hipMemcpyAsync((char*)h + q * (bytes / 4), (char*)d + q * (bytes / 4), bytes / 4, hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
            auto hh = now();
            if (r >= 0) { t_one += us(a, b); t_two += us(b, c); t_k += us(c, e); t_k2 += us(e, f); t_cpy += us(f, g); t_four += us(g, hh); }
        }
        auto gbs = [&](double t) { return bytes / (t / reps) / 1e3; };
        std::printf("%6zu KB: memcpyAsync %.1f us (%.1f GB/s) | 4 pieces one stream %.1f us | two streams %.1f us (%.1f GB/s) | kernel->mapped 8B %.1f us (%.1f GB/s) | "
                    "16B %.1f us (%.1f GB/s) | host memcpy out of pinned %.1f us (%.1f GB/s)\n",
                    bytes >> 10, t_one / reps, gbs(t_one), t_four / reps, t_two / reps, gbs(t_two), t_k / reps, gbs(t_k), t_k2 / reps, gbs(t_k2), t_cpy / reps, gbs(t_cpy));
    }
    return 0;
}
