// wfa_fwd.hpp -- launchers of the sub-wave forward kernels (wfa_blk_kernel, wfa_lane_kernel, wfa_duo_kernel), one
// translation unit per PENALTY SHAPE.
//
// The reference takes any penalties (wfa.go:32-36).  What the register-ring kernels are built around is not the penalties
// but their shape in units of g = gcd(x, o+e, e): DX = x/g and DOE = (o+e)/g say how many score steps back M[s-x] and
// M[s-o-e] lie (wfa.go:557-560), and with e/g == 1 the I and D sources are the previous row.  Rounds 1-4 had the one
// shape of the default penalties, 2 : 4 (4/6/2 and its multiples, 2/3/1); everything else ran on the LDS-ring kernel
// (wfa_packed_kernel, any shape, ring depth a run-time value) or the one-workgroup-per-pair kernel.  Round 5 instantiates the
// same kernels per shape (template arguments DX, DOE: the M ring holds max(DX, DOE) rows and the step loop is unrolled
// that many times), each shape in a translation unit of its own (wfa_fwd_s<DX><DOE>.hip includes wfa_fwd_shape.inc) so
// that they compile side by side:
//     2 : 4   4/6/2 (default), 2/3/1, 8/12/4      1 : 3   2/4/2
//     1 : 2   1/1/1, 2/2/2                          2 : 3   4/4/2, 2/2/1
//     2 : 2   4/2/2, 2/1/1                          3 : 3   6/4/2 (x == o+e)
// Shapes without an instance (e/g != 1, a ring deeper than four rows) keep the LDS-ring kernel.
#pragma once
#include "wfa_common.hpp"

namespace wfa {

constexpr int FWD_N_SHAPES = 6;
// index of the shape's instances, -1: none
inline int fwd_shape(uint32_t dx, uint32_t doe, uint32_t de) {
    if (de != 1u) return -1;
    if (dx == 2u && doe == 4u) return 0;
    if (dx == 1u && doe == 3u) return 1;
    if (dx == 1u && doe == 2u) return 2;
    if (dx == 2u && doe == 3u) return 3;
    if (dx == 2u && doe == 2u) return 4;
    if (dx == 3u && doe == 3u) return 5;
    return -1;
}

// flags of wfa_launch_fwd
enum : uint32_t {
    FWD_CENSUS   = 1u,  // count the stored wavefront words (REC_CELLS)
    FWD_STREAM   = 2u,  // kind 3 only, shape 2 : 4 only: the launch also walks the backtrace of finished pairs
    FWD_BATCH    = 4u,  // kinds 3 / 6: a group stages BLK_BATCH queue entries per refill (short reads)
    FWD_ADAPTIVE = 8u   // kind 10: the instance that tracks wf-adaptive's distances
};

// Forward kernel of `kind` (the host's numbering: 3 = wfa_blk_kernel<16>, 4 = <8>, 5 = <64> (256 diagonals), 6 = <8, .., 4> (32
// diagonals), 8 = wfa_duo_kernel, 9 = <32> (128 diagonals), 10 = wfa_lane_kernel, 11 / 12 / 13 = 3 / 9 / 5 with sliding sequence
// windows, 14 / 15 = a wave per pair with one / two diagonals per lane) for penalty shape `shape`.  hipErrorInvalidValue: no
// such instance.
hipError_t wfa_launch_fwd(int shape, int kind, uint32_t flags, const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st);

// wfahip_align_pair's lone-pair instance (a lane per diagonal, the wave walks its own backtrace): lds_arena = the rows in LDS
hipError_t wfa_launch_pair(int shape, bool lds_arena, const KParams &P, size_t lds_bytes, hipStream_t st);

// per-shape entry points (wfa_fwd_s*.hip)
#define WFA_FWD_DECL(tag)                                                                                                   \
    hipError_t wfa_launch_fwd_##tag(int kind, uint32_t flags, const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st); \
    hipError_t wfa_launch_pair_##tag(bool lds_arena, const KParams &P, size_t lds_bytes, hipStream_t st);
WFA_FWD_DECL(s24)
WFA_FWD_DECL(s13)
WFA_FWD_DECL(s12)
WFA_FWD_DECL(s23)
WFA_FWD_DECL(s22)
WFA_FWD_DECL(s33)
#undef WFA_FWD_DECL

}  // namespace wfa
