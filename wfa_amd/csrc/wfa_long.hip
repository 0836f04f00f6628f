// wfa_long.hip -- translation unit of the long-pair kernels (wfa_long.hpp).
#define WFA_NO_AUX_KERNELS 1
#include "wfa_generic.hpp"
#include "wfa_team.hpp"
#include "wfa_teamc.hpp"
#include "wfa_long.hpp"

namespace wfa {

namespace {
template <int WAVES, int MODE>
hipError_t launch_one(const KParams &P, uint32_t slots, size_t lds_bytes, hipStream_t st) {
    auto kfn = wfa_generic_kernel<WAVES, MODE>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(slots), dim3(64 * WAVES), lds_bytes, st, P);
    return hipGetLastError();
}
}  // namespace

hipError_t wfa_launch_generic(const KParams &P, int waves, int mode, uint32_t slots, size_t lds_bytes, hipStream_t st) {
    switch (waves * 2 + mode) {
    case 1 * 2 + 0: return launch_one<1, 0>(P, slots, lds_bytes, st);
    case 1 * 2 + 1: return launch_one<1, 1>(P, slots, lds_bytes, st);
    case 4 * 2 + 0: return launch_one<4, 0>(P, slots, lds_bytes, st);
    case 4 * 2 + 1: return launch_one<4, 1>(P, slots, lds_bytes, st);
    case 16 * 2 + 0: return launch_one<16, 0>(P, slots, lds_bytes, st);
    case 16 * 2 + 1: return launch_one<16, 1>(P, slots, lds_bytes, st);
    }
    return hipErrorInvalidValue;
}

hipError_t wfa_launch_team(const KParams &P, int mode, uint32_t grid, size_t lds_bytes, hipStream_t st, uint32_t *team_ctl, uint32_t T,
                           uint32_t solo_max, uint32_t wave_rows, uint32_t strict) {
    auto kfn = mode == 0 ? wfa_team_kernel<0> : wfa_team_kernel<1>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(TEAM_THREADS), lds_bytes, st, P, team_ctl, T, solo_max, wave_rows, strict);
    return hipGetLastError();
}

hipError_t wfa_launch_teamc(const KParams &P, const TcArgs &X, int mode, uint32_t grid, size_t lds_bytes, hipStream_t st) {
    auto kfn = mode == 0 ? wfa_teamc_kernel<0> : wfa_teamc_kernel<1>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(TC_THREADS), lds_bytes, st, P, X);
    return hipGetLastError();
}

}  // namespace wfa
