// wfa_long.hpp -- launchers of the long-pair kernels (wfa_generic_kernel: a workgroup per pair; wfa_team_kernel: a team of
// workgroups per pair), which live in a translation unit of their own (wfa_long.hip) so that they compile beside the
// sub-wave kernels' units.
#pragma once
#include "wfa_common.hpp"

namespace wfa {

// waves: 1, 4 or 16 per pair; mode 0 = 2-bit packed sequences in LDS, 1 = bytes in global memory
hipError_t wfa_launch_generic(const KParams &P, int waves, int mode, uint32_t slots, size_t lds_bytes, hipStream_t st);
// grid workgroups of 1 024 threads; the remaining arguments are the kernel's (wfa_team.hpp)
hipError_t wfa_launch_team(const KParams &P, int mode, uint32_t grid, size_t lds_bytes, hipStream_t st, uint32_t *team_ctl, uint32_t T,
                           uint32_t solo_max, uint32_t wave_rows, uint32_t strict);

struct TcArgs;
// wfa_teamc_kernel (round 5, wfa_teamc.hpp): grid workgroups of 1 024 threads
hipError_t wfa_launch_teamc(const KParams &P, const TcArgs &X, int mode, uint32_t grid, size_t lds_bytes, hipStream_t st);

}  // namespace wfa
