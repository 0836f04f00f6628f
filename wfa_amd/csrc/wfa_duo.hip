// wfa_duo.hip -- translation unit of wfa_duo_kernel: built with LLVM's atomic optimizer off (wfa_duo_cfg.hpp, Makefile).
// (the non-template kernels of the shared headers -- wfa_packed_kernel, wfa_backtrace_kernel, wfa_prepack_kernel -- are launched from
// wfa_host.hip only: this unit takes the device functions and leaves those kernels out)
#define WFA_NO_AUX_KERNELS 1
#include "wfa_duo.hpp"
#include "wfa_fwd.hpp"

namespace wfa {

namespace {
template <int DX, int DOE>
hipError_t go_duo(const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st, bool census) {
    if (census)
        hipLaunchKernelGGL((wfa_duo_kernel<true, DX, DOE>), dim3(grid), dim3(64), lds_bytes, st, P);
    else
        hipLaunchKernelGGL((wfa_duo_kernel<false, DX, DOE>), dim3(grid), dim3(64), lds_bytes, st, P);
    return hipGetLastError();
}
}  // namespace

// shape: index of the penalty shape (wfa_fwd.hpp: fwd_shape())
hipError_t wfa_launch_duo(int shape, const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st, bool census) {
    switch (shape) {
    case 0: return go_duo<2, 4>(P, grid, lds_bytes, st, census);
    case 1: return go_duo<1, 3>(P, grid, lds_bytes, st, census);
    case 2: return go_duo<1, 2>(P, grid, lds_bytes, st, census);
    case 3: return go_duo<2, 3>(P, grid, lds_bytes, st, census);
    case 4: return go_duo<2, 2>(P, grid, lds_bytes, st, census);
    case 5: return go_duo<3, 3>(P, grid, lds_bytes, st, census);
    }
    return hipErrorInvalidValue;
}

}  // namespace wfa
