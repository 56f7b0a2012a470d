// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the two access widths the
// engine uses (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a 16-B/lane stream; other
// widths must be calibrated on a known byte count).  Streams a 1 GiB buffer (4x the Infinity
// Cache) once with 4-byte and once with 16-byte lane loads; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read_dword(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_dwordx4(const uint4* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    void* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        read_dword<<<2048, 256>>>((const uint32_t*)buf, bytes / 4, out);
        read_dwordx4<<<2048, 256>>>((const uint4*)buf, bytes / 16, out);
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel launch\n", bytes);
    return 0;
}
