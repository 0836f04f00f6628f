// wfa_duo_cfg.hpp -- sizes of wfa_duo_kernel (wfa_duo.hpp) that the host needs too, and its launcher.
// The kernel lives in a translation unit of its own (wfa_duo.hip) because it is compiled with
// -mllvm -amdgpu-atomic-optimizer-strategy=None: its prefetch claims queue entries with an atomicAdd whose result must
// NOT be waited for on the spot, and LLVM's atomic optimizer turns a uniform atomicAdd into a wave-aggregated atomic
// followed at once by s_waitcnt + v_readfirstlane.  The other kernels want the optimizer: the lane-per-pair backtrace
// reserves its CIGAR op slots with one atomicAdd per lane, which the optimizer folds into one per wave (with the flag on
// the whole library that kernel went from 1.2 to 19 ms).
#pragma once
#include "wfa_common.hpp"

namespace wfa {

#ifndef WFA_DUO_WAVES
#define WFA_DUO_WAVES 4  // waves per SIMD the kernel is compiled for
#endif
#ifndef WFA_DUO_PARK
#define WFA_DUO_PARK 4
#endif
#ifndef WFA_DUO_WIDEN_AT
#define WFA_DUO_WIDEN_AT 26
#endif
#ifndef WFA_DUO_NARROW_AT
#define WFA_DUO_NARROW_AT 21
#endif
#ifndef WFA_DUO_GROUP_MAJOR
#define WFA_DUO_GROUP_MAJOR 2
#endif
// Arena layout of wfa_duo_kernel (16-bit words; a lane's four diagonals of a score are one 8-byte store, and 8 scores of them
// one 64-byte piece, in every one of the three):
//   0  CompactView fmt 7 (round 3) -- tiles of 8 scores x 64 diagonals (1 KB), inside a tile [diagonal / 4][score & 7][diagonal & 3]:
//      the row pointer needs "next score, and the next tile after every eighth" -- seven vector instructions a step;
//   1  fmt 9 (round 6) -- group-major, [diagonal / 4 & 15][score][diagonal & 3]: a constant stride, and the backtrace's walk,
//      which stays within a few groups, reads from a handful of DRAM pages instead of one per tile.  But ONE lane fills a
//      128-byte line, over 16 steps: twice the dirty lines in L2, evicted half-written and written again -- WRITE_SIZE 26 GB
//      per 1e6 x 1 kbp pairs instead of 10.8;
//   2  fmt 10 (round 6, the default) -- the groups two by two: [diagonal / 8 & 7][score / 8][diagonal / 4 & 1][score & 7][diagonal & 3].
//      A 128-byte line is filled by two neighbouring lanes in 8 steps as in the tiles, a pair of groups is contiguous over
//      all the scores as in fmt 9, and the address comes from the score index (and, add, shift-add) with no pointer to carry.
constexpr uint32_t DUO_ARENA_FMT = WFA_DUO_GROUP_MAJOR == 2 ? 10u : WFA_DUO_GROUP_MAJOR == 1 ? 9u : 7u;
constexpr int DUO_PARK       = WFA_DUO_PARK;           // park records per wave
constexpr int DUO_PARK_WORDS = 8 * 12 + 16;            // rings of 8 lanes, two 16-bit offsets per word (reads under 2 048 bases) + 16 scalars
constexpr int DUO_BUFS       = 8 + 1 + DUO_PARK;       // sequence buffers per wave: running pairs, staging, parked pairs (one pair per fetch)
constexpr int DUO_FETCH_MAX  = 8;                      // short reads: pairs per fetch (as many prepacked slots as one 16-byte load per lane covers)
constexpr int DUO_WIDEN_AT   = WFA_DUO_WIDEN_AT;       // a narrow pair whose band spans more diagonals than this widens (a recentred band must fit 32 - 6)
constexpr int DUO_NARROW_AT  = WFA_DUO_NARROW_AT;      // a wide pair whose band spans at most this many narrows
constexpr int DUO_WIDE_MAX   = 56;                     // a wide pair whose band spans more is handed on

// pairs one fetch brings in: the slots of consecutive queue entries are contiguous in the prepack buffer, and one
// 16-byte load per lane covers 256 words -- eight slots of a 150-base pair, one of a 1 kbp pair
__host__ __device__ inline uint32_t duo_fetch_pairs(uint32_t prepack_words) {
    const uint32_t f = 256u / prepack_words;
    return f < 1u ? 1u : (f > (uint32_t)DUO_FETCH_MAX ? (uint32_t)DUO_FETCH_MAX : f);
}
__host__ __device__ inline uint32_t duo_bufs(uint32_t prepack_words) { return 8u + duo_fetch_pairs(prepack_words) + (uint32_t)DUO_PARK; }
__host__ __device__ inline uint32_t duo_lds_words(uint32_t prepack_words) {
    return duo_bufs(prepack_words) * prepack_words + (uint32_t)DUO_PARK * DUO_PARK_WORDS;
}

// launches wfa_duo_kernel<census, DX, DOE> of penalty shape `shape` (wfa_duo.hip; wfa_fwd.hpp: fwd_shape())
hipError_t wfa_launch_duo(int shape, const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st, bool census);

}  // namespace wfa
