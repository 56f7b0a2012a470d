// sbe_mixture.hip -- translation unit of the fused mixture log-likelihood kernels (sbe_kernels_mixture.hip.h) and their
// launchers (declared in sbe_mixture.hip.h).  Compiled separately from sbe_engine.hip and linked into the same
// libsbe_engine.so: these templates are most of the library's compile time, and nothing else depends on their bodies.
#include <cstdlib>

#include "sbe_kernels_mixture.hip.h"

namespace sbe {

template <int MODE, int FT, bool DIRECT = false>
static void launch_v2_ft(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_v2<MODE, FT, 1, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        case 2: k_mixture_v2<MODE, FT, 2, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        case 3: k_mixture_v2<MODE, FT, 3, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        case 4: k_mixture_v2<MODE, FT, 4, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        default: k_mixture_v2<MODE, FT, 0, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
    }
}

template <int MODE, int FT, bool DIRECT = false>
static void launch_oh2_ft(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_onehot_v2<MODE, FT, 1, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        case 2: k_mixture_onehot_v2<MODE, FT, 2, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        case 3: k_mixture_onehot_v2<MODE, FT, 3, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        case 4: k_mixture_onehot_v2<MODE, FT, 4, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
        default: k_mixture_onehot_v2<MODE, FT, 0, DIRECT><<<grid, kBlock, lds, st>>>(p); break;
    }
}

template <int MODE>
static void launch_oh2_t(int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (direct) launch_oh2_ft<MODE, 16, true>(C, p, grid, lds, st);
    else if (ft == 64) launch_oh2_ft<MODE, 64>(C, p, grid, lds, st);
    else if (ft == 32) launch_oh2_ft<MODE, 32>(C, p, grid, lds, st);
    else launch_oh2_ft<MODE, 16>(C, p, grid, lds, st);
}

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

template <int MODE>
static void launch_v2_t(int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (direct) launch_v2_ft<MODE, 16, true>(C, p, grid, lds, st);
    else if (ft == 64) launch_v2_ft<MODE, 64>(C, p, grid, lds, st);
    else if (ft == 32) launch_v2_ft<MODE, 32>(C, p, grid, lds, st);
    else launch_v2_ft<MODE, 16>(C, p, grid, lds, st);
}

template <int MODE, int FT>
static void launch_rows_ft(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    switch (C) {
        case 1: k_mixture_rows<MODE, FT, 1><<<grid, kRowsBlock, lds, st>>>(p); break;
        case 2: k_mixture_rows<MODE, FT, 2><<<grid, kRowsBlock, lds, st>>>(p); break;
        case 3: k_mixture_rows<MODE, FT, 3><<<grid, kRowsBlock, lds, st>>>(p); break;
        default: k_mixture_rows<MODE, FT, 4><<<grid, kRowsBlock, lds, st>>>(p); break;
    }
}

template <int MODE>
static void launch_rows_t(int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    if (ft == 32) launch_rows_ft<MODE, 32>(C, p, grid, lds, st);
    else launch_rows_ft<MODE, 16>(C, p, grid, lds, st);
}

void launch_v2(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (mode == LOG_PRODUCT) launch_v2_t<LOG_PRODUCT>(ft, C, p, grid, lds, st, direct);
    else launch_v2_t<LOG_PER_OBS>(ft, C, p, grid, lds, st, direct);
}

void launch_oh2(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (mode == LOG_PRODUCT) launch_oh2_t<LOG_PRODUCT>(ft, C, p, grid, lds, st, direct);
    else launch_oh2_t<LOG_PER_OBS>(ft, C, p, grid, lds, st, direct);
}

void launch_combo(bool onehot, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    if (onehot) launch_combo_t<true>(ft, C, p, grid, lds, st);
    else launch_combo_t<false>(ft, C, p, grid, lds, st);
}

void launch_rows(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st) {
    if (mode == LOG_PRODUCT) launch_rows_t<LOG_PRODUCT>(ft, C, p, grid, lds, st);
    else launch_rows_t<LOG_PER_OBS>(ft, C, p, grid, lds, st);
}

}  // namespace sbe
