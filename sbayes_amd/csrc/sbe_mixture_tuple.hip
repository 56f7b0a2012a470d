// sbe_mixture_tuple.hip -- the group-tuple forms of the fused mixture log-likelihood on the vector pipe (k_mixture_combo: tuple
// metadata in LDS, any tile width, both streams; k_mixture_tuple64: the scalar-unit form at 64-feature tiles) and their launchers.
#include <cstdlib>

#include "sbe_kernels_mixture.hip.h"

namespace sbe {

template <int FT, bool ONEHOT>
static void launch_combo_ft(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_combo<FT, 1, ONEHOT><<<grid, kBlock, lds, st>>>(p); break;
        case 2: k_mixture_combo<FT, 2, ONEHOT><<<grid, kBlock, lds, st>>>(p); break;
        case 3: k_mixture_combo<FT, 3, ONEHOT><<<grid, kBlock, lds, st>>>(p); break;
        case 4: k_mixture_combo<FT, 4, ONEHOT><<<grid, kBlock, lds, st>>>(p); break;
        default: k_mixture_combo<FT, 0, ONEHOT><<<grid, kBlock, lds, st>>>(p); break;
    }
}

template <bool OFF16, int NW>
static void launch_tuple64_o(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_tuple64<1, OFF16, NW><<<grid, NW * kWave, lds, st>>>(p); break;
        case 2: k_mixture_tuple64<2, OFF16, NW><<<grid, NW * kWave, lds, st>>>(p); break;
        case 3: k_mixture_tuple64<3, OFF16, NW><<<grid, NW * kWave, lds, st>>>(p); break;
        case 4: k_mixture_tuple64<4, OFF16, NW><<<grid, NW * kWave, lds, st>>>(p); break;
        default: k_mixture_tuple64<0, OFF16, NW><<<grid, NW * kWave, lds, st>>>(p); break;
    }
}

void launch_tuple64(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    if (const char* env = getenv("SBE_T64_LDS_PAD")) lds += (size_t)atoi(env);        // experiments: fewer blocks per CU
    // 16-bit tuple-block offsets when the whole log table sits below 64 KiB
    const bool off16 = (int64_t)p.KT * (p.S + 1) * 512 <= 65536;
    if (off16) launch_tuple64_o<true, 4>(C, p, grid, lds, st); else launch_tuple64_o<false, 4>(C, p, grid, lds, st);
}

template <bool ONEHOT>
static void launch_combo_t(int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    if (ft == 64) launch_combo_ft<64, ONEHOT>(C, p, grid, lds, st);
    else if (ft == 32) launch_combo_ft<32, ONEHOT>(C, p, grid, lds, st);
    else launch_combo_ft<16, ONEHOT>(C, p, grid, lds, st);
}

void launch_combo(bool onehot, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    if (onehot) launch_combo_t<true>(ft, C, p, grid, lds, st);
    else launch_combo_t<false>(ft, C, p, grid, lds, st);
}

}  // namespace sbe
