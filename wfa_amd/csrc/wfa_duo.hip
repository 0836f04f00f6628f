// wfa_duo.hip -- translation unit of wfa_duo_kernel: built with LLVM's atomic optimizer off (wfa_duo_cfg.hpp, Makefile).
// (the non-template kernels of the shared headers -- wfa_packed_kernel, wfa_backtrace_kernel, wfa_prepack_kernel -- are launched from
// wfa_host.hip only: this unit takes the device functions and leaves those kernels out)
#define WFA_NO_AUX_KERNELS 1
#include "wfa_duo.hpp"

namespace wfa {

hipError_t wfa_launch_duo(const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st, bool census) {
    if (census)
        hipLaunchKernelGGL((wfa_duo_kernel<true>), dim3(grid), dim3(64), lds_bytes, st, P);
    else
        hipLaunchKernelGGL((wfa_duo_kernel<false>), dim3(grid), dim3(64), lds_bytes, st, P);
    return hipGetLastError();
}

}  // namespace wfa
