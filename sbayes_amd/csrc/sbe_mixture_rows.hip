// sbe_mixture_rows.hip -- the rows form of the fused mixture log-likelihood (k_mixture_rows: 1024-thread blocks over one LDS image
// of a 32- / 16-feature tile, per-object row offsets), its pattern-sorted variant, and the kernels that prepare the sorted
// variant's inputs (k_rowsort, k_state_s), with their launchers.
#include <cstdlib>

#include "sbe_kernels_mixture.hip.h"

namespace sbe {

template <int MODE, int FT, bool SORTED>
static void launch_rows_ft(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_rows<MODE, FT, 1, SORTED><<<grid, kRowsBlock, lds, st>>>(p); break;
        case 2: k_mixture_rows<MODE, FT, 2, SORTED><<<grid, kRowsBlock, lds, st>>>(p); break;
        case 3: k_mixture_rows<MODE, FT, 3, SORTED><<<grid, kRowsBlock, lds, st>>>(p); break;
        default: k_mixture_rows<MODE, FT, 4, SORTED><<<grid, kRowsBlock, lds, st>>>(p); break;
    }
}

template <int MODE>
static void launch_rows_t(int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool sorted) {
    if (ft == 32 && sorted) launch_rows_ft<MODE, 32, true>(C, p, grid, lds, st);       // (the sorted form exists at 32-feature tiles)
    else if (ft == 32) launch_rows_ft<MODE, 32, false>(C, p, grid, lds, st);
    else launch_rows_ft<MODE, 16, false>(C, p, grid, lds, st);
}

void launch_rows(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool sorted) {
    if (mode == LOG_PRODUCT) launch_rows_t<LOG_PRODUCT>(ft, C, p, grid, lds, st, sorted);
    else launch_rows_t<LOG_PER_OBS>(ft, C, p, grid, lds, st, sorted);
}

// ---- inputs of the pattern-sorted rows form -----------------------------------------------------------------------------
// One 1024-thread block per slot: a counting sort of the slot's objects by has_components pattern id.  STABLE -- ascending
// object index inside a pattern -- so the order, and with it the summation order of the kernel, is a function of the ids
// alone (same ids, same bits).  Every pattern's run is padded to a multiple of `step` objects (one wave step of the rows
// kernel) with NULL objects: table rows = the "no group" zero row of every component, state row = row N of state_s (all
// NA) -- they evaluate to exactly 1.  Wave w owns the objects [w * per, (w + 1) * per): per-wave pattern counts, a prefix over
// (pattern, wave), then every wave places its objects chunk by chunk (ballot ranks).
// out[q][c][j] (c < C): LDS byte offset of the object's group row of component c;
// out[q][C][j]: byte offset of the object's state row (| pattern << 24 in object 0 of a quad: a quad holds one pattern).
constexpr int kSortWaves = 16;
__global__ __launch_bounds__(kSortWaves * kWave) void k_rowsort(
        const uint16_t* __restrict__ gid, const uint8_t* __restrict__ pid, uint32_t* __restrict__ out,
        int32_t* __restrict__ nq_out, int64_t gid_stride, int64_t pid_stride, int64_t out_stride, int first_slot,
        const int32_t* __restrict__ slot_list, int N, int Np, int C, int Gtot, int Pmax, uint32_t row_bytes,
        uint32_t state_pitch, int step) {
    __shared__ int cnt[kSortWaves][64];          // objects of pattern q in wave w's range; then: first position of that block
    __shared__ int base[65], tot[64];
    const int slot = slot_list ? slot_list[blockIdx.x] : first_slot + (int)blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const uint8_t* pids = pid + (int64_t)slot * pid_stride;
    uint32_t* o = out + (int64_t)slot * out_stride;
    const int per = ((N + kSortWaves - 1) / kSortWaves + 63) / 64 * 64;          // objects per wave: whole chunks of 64
    const int n_lo = min(N, w * per), n_hi = min(N, n_lo + per);
    if (lane < Pmax) cnt[w][lane] = 0;
    __syncthreads();
    for (int n0 = n_lo; n0 < n_hi; n0 += 64) {                                   // (wave-private rows of cnt: no atomics)
        const int n = n0 + lane;
        const int pt = n < n_hi ? (int)pids[n] : -1;
        for (int q = 0; q < Pmax; ++q) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64(pt == q);
            if (lane == 0 && b) cnt[w][q] += __popcll(b);
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)Pmax) {                                          // run lengths
        int t = 0;
        for (int ww = 0; ww < kSortWaves; ++ww) t += cnt[ww][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int q = 0; q < Pmax; ++q) { base[q] = acc; acc += (tot[q] + step - 1) / step * step; }
        base[Pmax] = acc;
        nq_out[slot] = acc / 4;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)Pmax) {                                          // counts -> first position of (pattern, wave)
        int acc = base[threadIdx.x];
        for (int ww = 0; ww < kSortWaves; ++ww) { const int c0 = cnt[ww][threadIdx.x]; cnt[ww][threadIdx.x] = acc; acc += c0; }
    }
    __syncthreads();
    auto put = [&](int pos, int n, int pt) {                                     // entry of object n (n < 0: a null object) at position pos
        const int q4 = pos >> 2, j = pos & 3;
        for (int c = 0; c < C; ++c) {
            const uint32_t g = n >= 0 ? gid[(int64_t)slot * gid_stride + (int64_t)c * Np + n] : (uint32_t)kNoGroup;
            o[(q4 * (C + 1) + c) * 4 + j] = (g < (uint32_t)Gtot ? g : (uint32_t)Gtot) * row_bytes;
        }
        o[(q4 * (C + 1) + C) * 4 + j] = (j == 0 ? ((uint32_t)pt << 24) : 0u) | ((uint32_t)(n >= 0 ? n : N) * state_pitch);
    };
    for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
        const int n = n0 + lane;
        const int pt = n < n_hi ? (int)pids[n] : -1;
        for (int q = 0; q < Pmax; ++q) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64(pt == q);
            if (!b) continue;                                                    // (wave-uniform)
            const int first = cnt[w][q];                                         // (every lane reads before lane 0 advances it:
            if (pt == q) put(first + __popcll(b & ((1ull << lane) - 1ull)), n, q);   //  LDS operations of a wave are in order)
            if (lane == 0) cnt[w][q] = first + __popcll(b);
        }
    }
    __syncthreads();
    for (int q = w; q < Pmax; q += kSortWaves)                                   // padding of every run
        for (int pos = base[q] + tot[q] + lane; pos < base[q + 1]; pos += 64) put(pos, -1, q);
}

void launch_rowsort(const uint16_t* gid, const uint8_t* pid, uint32_t* out, int32_t* nq_out, int64_t gid_stride, int64_t pid_stride,
                    int64_t out_stride, int first_slot, const int32_t* slot_list, int n_slots, int N, int Np, int C, int Gtot, int Pmax,
                    uint32_t row_bytes, uint32_t state_pitch, int step_objects, hipStream_t st) {
    k_rowsort<<<n_slots, kSortWaves * kWave, 0, st>>>(gid, pid, out, nq_out, gid_stride, pid_stride, out_stride, first_slot, slot_list, N, Np,
                                                      C, Gtot, Pmax, row_bytes, state_pitch, step_objects);
}

// state index per observation with NA = S (the packed block the engine keeps has NA = 0xFF), plus the null object's row N
// (row pitch `pitch` >= every feature index a lane of the last tile can have: F rounded up to 64)
__global__ void k_state_s(const uint8_t* __restrict__ state, uint8_t* __restrict__ state_s, int N, int F, int Fp, int pitch, int S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)(N + 1) * pitch) return;
    const int n = (int)(i / pitch), f = (int)(i % pitch);
    const uint8_t x = (n < N && f < F) ? state[(int64_t)n * Fp + f] : (uint8_t)0xFF;
    state_s[i] = x >= (uint8_t)S ? (uint8_t)S : x;
}

void launch_state_s(const uint8_t* state, uint8_t* state_s, int N, int F, int Fp, int pitch, int S, hipStream_t st) {
    const int64_t n = (int64_t)(N + 1) * pitch;
    k_state_s<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(state, state_s, N, F, Fp, pitch, S);
}

}  // namespace sbe
