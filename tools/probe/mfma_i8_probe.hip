// Layout probe for v_mfma_i32_32x32x32_i8 on gfx950: which (lane, byte) holds which A[row][k] / B[k][col], and which
// (lane, register) holds which D[row][col].  Exact integer data, asymmetric operands.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_i8_probe.hip -o /tmp/mfma_i8_probe && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const v4i* a, const v4i* b, v16i* d) {
    v16i acc = {};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}

int main() {
    std::vector<int8_t> A(32 * 32), B(32 * 32);          // A[row][k], B[k][col]
    srand(7);
    for (auto& v : A) v = (int8_t)(rand() % 7 - 3);
    for (auto& v : B) v = (int8_t)(rand() % 5 - 2);
    std::vector<int> D(32 * 32, 0);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += A[r * 32 + kk] * B[kk * 32 + c]; D[r * 32 + c] = s; }
    // candidate k maps: byte j of lane half h -> k
    auto kmap = [](int variant, int h, int j) { return variant == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3); };
    for (int va = 0; va < 2; ++va) for (int vb = 0; vb < 2; ++vb) {
        std::vector<int8_t> fa(64 * 16), fb(64 * 16);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
            fa[l * 16 + j] = A[(l & 31) * 32 + kmap(va, l >> 5, j)];
            fb[l * 16 + j] = B[kmap(vb, l >> 5, j) * 32 + (l & 31)];
        }
        void *da, *db, *dd;
        hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 64 * 64);
        hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>((const v4i*)da, (const v4i*)db, (v16i*)dd);
        std::vector<int> out(64 * 16);
        hipMemcpy(out.data(), dd, 64 * 64, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 16; ++reg) {
            const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
            if (out[l * 16 + reg] != D[row * 32 + col]) ++bad;
        }
        printf("A k-map %d, B k-map %d: %d of 1024 outputs differ from row=(reg&3)+8*(reg>>2)+4*(lane>>5), col=lane&31\n", va, vb, bad);
        hipFree(da); hipFree(db); hipFree(dd);
    }
    return 0;
}
