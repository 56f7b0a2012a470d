// sbe_mixture.hip -- the GENERAL forms of the fused mixture log-likelihood kernels (sbe_kernels_mixture.hip.h: k_mixture_v2, its
// one-hot stream form) and their launchers (declared in sbe_mixture.hip.h).  The fused kernels are templates over (log mode, tile
// width, component count) and most of the library's compile time, so they are spread over units that build in parallel --
// this one, sbe_mixture_tuple.hip (group-tuple forms), sbe_mixture_rows.hip (rows form), sbe_mixture_mfma.hip (matrix-pipe
// form) -- and nothing else depends on their bodies: all are linked into libsbe_engine.so.
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

template <int MODE>
static void launch_v2_t(int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (direct) launch_v2_ft<MODE, 16, true>(C, p, grid, lds, st);
    else if (ft == 64) launch_v2_ft<MODE, 64>(C, p, grid, lds, st);
    else if (ft == 32) launch_v2_ft<MODE, 32>(C, p, grid, lds, st);
    else launch_v2_ft<MODE, 16>(C, p, grid, lds, st);
}

void launch_v2(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (mode == LOG_PRODUCT) launch_v2_t<LOG_PRODUCT>(ft, C, p, grid, lds, st, direct);
    else launch_v2_t<LOG_PER_OBS>(ft, C, p, grid, lds, st, direct);
}

void launch_oh2(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct) {
    if (mode == LOG_PRODUCT) launch_oh2_t<LOG_PRODUCT>(ft, C, p, grid, lds, st, direct);
    else launch_oh2_t<LOG_PER_OBS>(ft, C, p, grid, lds, st, direct);
}

}  // namespace sbe
