// wfa_blk.hpp -- kernel D: blocked register-window forward kernel (64/G pairs per wave64).
//
// A group of G lanes (G = 16: one DPP row, or G = 8: half a row) owns one pair; lane j of the group holds the
// PP = 64/G CONSECUTIVE diagonals kb + PP*j + p (p = 0..PP-1) of a 64-diagonal window that follows the band one
// lane (PP diagonals) at a time.  Consequences of the blocked layout:
//   * the k-1 / k+1 sources of WF_NEXT (wfa.go:579,580,614,615) are the lane's own neighbouring registers for
//     all but one position, so a score step needs four DPP lane shifts in total (Mo/I below, Mo/D above);
//   * a lane's PP finished cells are adjacent in the arena: one 16-byte store per lane and row;
//   * every per-pair reduction (band range, wf-adaptive) is a 3- or 4-stage DPP butterfly inside the group.
// The register rings hold bare offsets (0 = absent), not offset<<3|tag words: WF_NEXT only reads offsets
// (wfa.go:579-655 strip the tag), and the tags go straight into the compact backtrace word that is stored.
// The ring of the last four M rows is indexed by (step & 3) with the step loop unrolled four times, so the ring
// advances by renaming, not by register moves: with penalties shaped 2:4:1 (x : o+e : e in units of g) the row
// written at step i replaces M[s-o-e], the row it was computed from, and M[s-x] is slot (i+2)&3.
//
// Fast path / exact path.  The bounds rejections of next() (offset > m, offset-k > n; wfa.go:581-588,616-623,
// 651-654), the k-range clamp (wfa.go:562-563) and the termination test (wfa.go:235-239) can only matter once
// some cell of the pair has reached a sequence end (h >= m or v >= n).  Until then -- almost the whole
// alignment -- the wave runs a rejection-free WF_NEXT in which backTrace's unbounded recomputation of the
// pre-extension offset (wfa.go:766-817) equals the offset just computed.  A sticky per-pair flag, set by the
// first cell that hits an end, switches the wave to the exact code (bit-for-bit the rules of next_cell()).
//
// Output: one backtrace word per diagonal (blk_word(): the pre-extension offset and the four decisions of next(), not
// the reference's tag values) in a directory-free arena layout (CompactView
// fmt 3: tiles of 8 scores x 64 diagonals; fmt 1 / 4: 64 / 256 words per score, diagonal k at slot k & (W-1); the
// window base stays a multiple of PP so a lane's PP words are aligned 16-byte stores), and pair_meta for
// wfa_backtrace_kernel.
//
// G = 64 (one pair per wave, PP = 4, a 256-diagonal window) is the retry rung for the pairs whose band outgrows
// the 64-diagonal window: same code, wave-wide DPP shifts and readlane-combined reductions.
#pragma once
#include "wfa_device.hpp"
#include "wfa_packed.hpp"
#include <type_traits>

namespace wfa {

#ifndef WFA_BLK_TILED
#define WFA_BLK_TILED 1  // 1: 64-diagonal arenas are tiled 8 scores x 64 diagonals (CompactView fmt 3); 0: plain rows (fmt 1)
#endif

template <int G>
struct BlkOps;

template <>
struct BlkOps<16> {
    // value held by lane j-1 / j+1 of the group; 0 beyond the group edge (bound_ctrl)
    static WFA_DEV uint32_t dn1(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); }
    static WFA_DEV uint32_t up1(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true); }
    // the window moves by WFA_BLK_SHIFT_LANES lanes: registers move towards higher (shr) / lower (shl) lanes
    // (measured on 1e6 x 1 kbp: 4 lanes -> 3 316 pairs outgrow the window, 2 lanes -> 1 294, 1 lane -> 595, same speed)
#ifndef WFA_BLK_SHIFT_LANES
#define WFA_BLK_SHIFT_LANES 1
#endif
    static constexpr int SHIFT_D = 4 * WFA_BLK_SHIFT_LANES;  // diagonals per window shift
    static WFA_DEV uint32_t shr(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + WFA_BLK_SHIFT_LANES, 0xf, 0xf, true); }
    static WFA_DEV uint32_t shl(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + WFA_BLK_SHIFT_LANES, 0xf, 0xf, true); }
};

template <>
struct BlkOps<8> {
    static WFA_DEV uint32_t dn1(uint32_t x, int j) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
        return j == 0 ? 0u : r;
    }
    static WFA_DEV uint32_t up1(uint32_t x, int j) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true);
        return j == 7 ? 0u : r;
    }
    static constexpr int SHIFT_D = 16;
    static WFA_DEV uint32_t shr(uint32_t x, int j) {  // 16 diagonals = 2 lanes
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
        return j < 2 ? 0u : r;
    }
    static WFA_DEV uint32_t shl(uint32_t x, int j) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x102, 0xf, 0xf, true);
        return j >= 6 ? 0u : r;
    }
};

// G = 8 with FOUR diagonals per lane: a 32-diagonal window, eight pairs per wave (short reads); the window moves by
// one lane.
struct BlkOps8n {
    static WFA_DEV uint32_t dn1(uint32_t x, int j) { return BlkOps<8>::dn1(x, j); }
    static WFA_DEV uint32_t up1(uint32_t x, int j) { return BlkOps<8>::up1(x, j); }
    static constexpr int SHIFT_D = 4;
    static WFA_DEV uint32_t shr(uint32_t x, int j) { return dn1(x, j); }
    static WFA_DEV uint32_t shl(uint32_t x, int j) { return up1(x, j); }
};

// G = 64: the whole wave owns one pair (a 256-diagonal window; the retry rung for pairs whose band outgrew 64
// diagonals).  Lane neighbours come from the wave-wide DPP shifts, which shift zeros in at lanes 0 / 63.
template <>
struct BlkOps<64> {
    static WFA_DEV uint32_t dn1(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true); }  // wave_shr:1
    static WFA_DEV uint32_t up1(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true); }  // wave_shl:1
    static constexpr int SHIFT_D = 4;
    static WFA_DEV uint32_t shr(uint32_t x, int j) { return dn1(x, j); }
    static WFA_DEV uint32_t shl(uint32_t x, int j) { return up1(x, j); }
};

// G = 32: two pairs per wave, a 128-diagonal window each (round 3: the rung between the 64- and the 256-diagonal
// windows -- 1 kbp pairs at 10-20 % error have bands of 60-110 diagonals).  Wave-wide shifts like G = 64, zeros
// forced at the boundary between the two halves of the wave.
template <>
struct BlkOps<32> {
    static WFA_DEV uint32_t dn1(uint32_t x, int j) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true);  // wave_shr:1
        return j == 0 ? 0u : r;
    }
    static WFA_DEV uint32_t up1(uint32_t x, int j) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true);  // wave_shl:1
        return j == 31 ? 0u : r;
    }
    static constexpr int SHIFT_D = 4;
    static WFA_DEV uint32_t shr(uint32_t x, int j) { return dn1(x, j); }
    static WFA_DEV uint32_t shl(uint32_t x, int j) { return up1(x, j); }
};

// Butterfly reductions inside a group, written as DPP-fused VOP2 instructions (one instruction per stage and
// value).  hipcc lowers the same butterfly from __builtin_amdgcn_update_dpp to mov + mov_dpp + op per stage.
// A DPP operand may be read two wait states after the VALU instruction that wrote it: the partner value's
// instruction plus one s_nop fill them.
#define WFA_DPP_CTL_XOR1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define WFA_DPP_CTL_XOR2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#define WFA_DPP_CTL_HMIR "row_half_mirror row_mask:0xf bank_mask:0xf"
#define WFA_DPP_CTL_MIR "row_mirror row_mask:0xf bank_mask:0xf"
#define WFA_DPP_ST2(opa, opb, ctl) opa " %0, %0, %0 " ctl "\n\t" opb " %1, %1, %1 " ctl "\n\t"
#define WFA_DPP_ST1(opa, ctl) opa " %0, %0, %0 " ctl "\n\t"

template <int G>
struct BlkRed {
    // a <- reduce(opa) over the group, b <- reduce(opb) over the group
#define WFA_RED2(name, opa, opb)                                                                          \
    static WFA_DEV void name(int &a, int &b) {                                                            \
        if constexpr (G == 16)                                                                            \
            asm("s_nop 1\n\t" WFA_DPP_ST2(opa, opb, WFA_DPP_CTL_XOR1) "s_nop 0\n\t" WFA_DPP_ST2(          \
                    opa, opb, WFA_DPP_CTL_XOR2) "s_nop 0\n\t" WFA_DPP_ST2(opa, opb, WFA_DPP_CTL_HMIR)     \
                    "s_nop 0\n\t" WFA_DPP_ST2(opa, opb, WFA_DPP_CTL_MIR)                                  \
                : "+v"(a), "+v"(b));                                                                      \
        else                                                                                              \
            asm("s_nop 1\n\t" WFA_DPP_ST2(opa, opb, WFA_DPP_CTL_XOR1) "s_nop 0\n\t" WFA_DPP_ST2(          \
                    opa, opb, WFA_DPP_CTL_XOR2) "s_nop 0\n\t" WFA_DPP_ST2(opa, opb, WFA_DPP_CTL_HMIR)     \
                : "+v"(a), "+v"(b));                                                                      \
    }
#define WFA_RED1(name, opa)                                                                               \
    static WFA_DEV int name(int a) {                                                                      \
        if constexpr (G == 16)                                                                            \
            asm("s_nop 1\n\t" WFA_DPP_ST1(opa, WFA_DPP_CTL_XOR1) "s_nop 1\n\t" WFA_DPP_ST1(               \
                    opa, WFA_DPP_CTL_XOR2) "s_nop 1\n\t" WFA_DPP_ST1(opa, WFA_DPP_CTL_HMIR)               \
                    "s_nop 1\n\t" WFA_DPP_ST1(opa, WFA_DPP_CTL_MIR)                                       \
                : "+v"(a));                                                                               \
        else                                                                                              \
            asm("s_nop 1\n\t" WFA_DPP_ST1(opa, WFA_DPP_CTL_XOR1) "s_nop 1\n\t" WFA_DPP_ST1(               \
                    opa, WFA_DPP_CTL_XOR2) "s_nop 1\n\t" WFA_DPP_ST1(opa, WFA_DPP_CTL_HMIR)               \
                : "+v"(a));                                                                               \
        return a;                                                                                         \
    }
    // a <- min, b <- max, c <- min: three independent chains fill each other's DPP wait states, no s_nop
    static WFA_DEV void min_max_min(int &a, int &b, int &c) {
#define WFA_DPP_ST3(ctl)                                                                               \
    "v_min_i32_dpp %0, %0, %0 " ctl "\n\tv_max_i32_dpp %1, %1, %1 " ctl "\n\tv_min_i32_dpp %2, %2, %2 " ctl "\n\t"
        if constexpr (G == 16)
            asm("s_nop 1\n\t" WFA_DPP_ST3(WFA_DPP_CTL_XOR1) WFA_DPP_ST3(WFA_DPP_CTL_XOR2) WFA_DPP_ST3(WFA_DPP_CTL_HMIR)
                    WFA_DPP_ST3(WFA_DPP_CTL_MIR)
                : "+v"(a), "+v"(b), "+v"(c));
        else
            asm("s_nop 1\n\t" WFA_DPP_ST3(WFA_DPP_CTL_XOR1) WFA_DPP_ST3(WFA_DPP_CTL_XOR2) WFA_DPP_ST3(WFA_DPP_CTL_HMIR)
                : "+v"(a), "+v"(b), "+v"(c));
#undef WFA_DPP_ST3
    }
    WFA_RED2(min_max, "v_min_i32_dpp", "v_max_i32_dpp")
    WFA_RED2(max_add, "v_max_i32_dpp", "v_add_u32_dpp")
    WFA_RED1(max1, "v_max_i32_dpp")
    WFA_RED1(or1, "v_or_b32_dpp")
#undef WFA_RED2
#undef WFA_RED1
};

// 64 lanes: the four row results (BlkRed<16>) are combined through readlane + scalar ops; every result is
// wave-uniform (it lives in an SGPR).
template <>
struct BlkRed<64> {
    static WFA_DEV int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
    static WFA_DEV int mn4(int v) { return imin2(imin2(rl(v, 0), rl(v, 16)), imin2(rl(v, 32), rl(v, 48))); }
    static WFA_DEV int mx4(int v) { return imax2(imax2(rl(v, 0), rl(v, 16)), imax2(rl(v, 32), rl(v, 48))); }
    static WFA_DEV void min_max(int &a, int &b) {
        BlkRed<16>::min_max(a, b);
        a = mn4(a), b = mx4(b);
    }
    static WFA_DEV void min_max_min(int &a, int &b, int &c) {
        BlkRed<16>::min_max_min(a, b, c);
        a = mn4(a), b = mx4(b), c = mn4(c);
    }
    static WFA_DEV void max_add(int &a, int &b) {
        BlkRed<16>::max_add(a, b);
        a = mx4(a), b = rl(b, 0) + rl(b, 16) + rl(b, 32) + rl(b, 48);
    }
    static WFA_DEV int max1(int a) { return mx4(BlkRed<16>::max1(a)); }
    static WFA_DEV int or1(int a) {
        a = BlkRed<16>::or1(a);
        return rl(a, 0) | rl(a, 16) | rl(a, 32) | rl(a, 48);
    }
};

// 32 lanes: the two row results of each half of the wave are combined through readlane + scalar ops and handed back
// to the half they belong to.
template <>
struct BlkRed<32> {
    static WFA_DEV int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
    static WFA_DEV int pick(int lo, int hi) { return (threadIdx.x & 32u) ? hi : lo; }
    static WFA_DEV int mn2(int v) { return pick(imin2(rl(v, 0), rl(v, 16)), imin2(rl(v, 32), rl(v, 48))); }
    static WFA_DEV int mx2(int v) { return pick(imax2(rl(v, 0), rl(v, 16)), imax2(rl(v, 32), rl(v, 48))); }
    static WFA_DEV void min_max(int &a, int &b) {
        BlkRed<16>::min_max(a, b);
        a = mn2(a), b = mx2(b);
    }
    static WFA_DEV void min_max_min(int &a, int &b, int &c) {
        BlkRed<16>::min_max_min(a, b, c);
        a = mn2(a), b = mx2(b), c = mn2(c);
    }
    static WFA_DEV void max_add(int &a, int &b) {
        BlkRed<16>::max_add(a, b);
        a = mx2(a), b = pick(rl(b, 0) + rl(b, 16), rl(b, 32) + rl(b, 48));
    }
    static WFA_DEV int max1(int a) { return mx2(BlkRed<16>::max1(a)); }
    static WFA_DEV int or1(int a) {
        a = BlkRed<16>::or1(a);
        return pick(rl(a, 0) | rl(a, 16), rl(a, 32) | rl(a, 48));
    }
};

WFA_DEV uint32_t ffbl_raw(uint32_t x) {  // index of the lowest set bit; 0xFFFFFFFF for x == 0
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

WFA_DEV uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return umax2(umax2(a, b), c); }

// blk_word() of a cell computed without rejections, in eight instructions: four compares into scalar registers, then
// four add-with-carry (w = 2 w + bit).  t = max(Isk, Dsk): the mismatch wins iff x1 >= t, else the insertion iff
// Isk >= Dsk (bit 0 means nothing next to a set bit 1).  The compares stand three instructions before their consumers
// (a scalar register written by a vector instruction may be read as an operand two wait states later).
WFA_DEV uint32_t blk_word_asm(uint32_t Msk, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t x1, uint32_t t, uint32_t Isk,
                              uint32_t Dsk) {
    // (the first add-with-carry reads Msk and writes the word: no copy of Msk in front of the block -- round 6)
    uint32_t w;
    asm("v_cmp_lt_u32_e64 s[40:41], %1, %2\n\t"
        "v_cmp_lt_u32_e64 s[42:43], %3, %4\n\t"
        "v_cmp_ge_u32_e64 s[44:45], %5, %6\n\t"
        "v_cmp_ge_u32_e64 s[46:47], %7, %8\n\t"
        "v_addc_co_u32_e64 %0, vcc, %9, %9, s[40:41]\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, s[42:43]\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, s[44:45]\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, s[46:47]"
        : "=&v"(w)
        : "v"(a), "v"(b), "v"(c), "v"(d), "v"(x1), "v"(t), "v"(Isk), "v"(Dsk), "v"(Msk)
        : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    return w;
}

constexpr int BK_BIG = 0x3FFFFFFF;
#ifndef WFA_BLK8_WAVES
#define WFA_BLK8_WAVES 2
#endif
#ifndef WFA_BLK_STREAM_WAVES5
#define WFA_BLK_STREAM_WAVES5 0  // 1: the streaming instance is compiled for WFA_BLK_WAVES too (experiment)
#endif
#ifndef WFA_BLK_WAVES
#define WFA_BLK_WAVES 5  // waves per SIMD the main instance is compiled for: 96 VGPRs, ~30 spilled values in the window and refill code;
                         // 20.2 ms per 1e6 x 1 kbp pairs against 20.9 at 4 waves (114 VGPRs, no spill): the issue rate of a SIMD grows with its waves
#endif

// BATCH > 1 (short reads: both sequences of a pair fit one staging pass of the group's own lanes): a group takes
// BATCH consecutive queue entries at a time and stages all of them into its LDS slots, so the refill chain (queue
// atomic -> lengths / offsets -> sequence bytes: three dependent memory round trips that stall all pairs of the
// wave) is paid once per BATCH pairs; the following pairs of the batch start from LDS.
//
// STREAM: finished pairs are handed to wfa_backtrace_stream_kernel while this kernel is still running.  The arena
// rows are stored write-through (sc1); a finished pair's done_q entry {index + 1, score, end offset, cells} is
// stored -- one 16-byte write-through store -- in the following refill, after an agent-scope release: the vmcnt of a
// write-through store returns before the store is at the memory side (measured in the team kernel, DESIGN.md section
// 6), so "every earlier store has been acknowledged" does not order the rows before the entry.
typedef uint32_t blk_u32x4 __attribute__((ext_vector_type(4)));

WFA_DEV void blk_store_sc1(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const blk_u32x4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// a pair that ends without a backtrace (rejected, handed on): its done_q entry only keeps the entry count complete
WFA_DEV void blk_push_not_ok(const KParams &P, uint32_t pidx) {
    const uint32_t t = atomicAdd(P.done_ctl, 1u);
    blk_store_sc1(P.done_q + t, (pidx + 1u) | DONE_NOT_OK, 0u, 0u, 0u);
}

// CENSUS: count the wavefront words a pair stores (REC_CELLS: the roofline accounting of bench.py and the tests' cross-
// check between kernels).  It is instrumentation, not part of the alignment, and costs 4 % of the forward pass
// (23.3 vs 24.3 ms per 1e6 x 1 kbp pairs): off unless the context's option "census" asks for it.
//
// LONG (round 4): reads of any length.  The plain instances keep BOTH whole sequences of a pair in LDS, which stops them at
// ~10 kbp (and at two waves per SIMD well before that); here a pair keeps a WINDOW of P.lds_seq_words packed words (3 840
// bases at the default 240 words) of each sequence around the band -- the band only moves forward, and under wf-adaptive the cells of a
// row lie within ~120 bases of each other -- and refills it from the pair's pre-packed slot (wfa_prepack_kernel) as the
// band advances.  Window word i of the query holds packed word qb + i, of the target word tb + i, the two bases tied
// together by the window's diagonal centre (qb16 = tb16 - kc16), so that ONE test on a cell's offset h (lo <= h <= hi) says
// whether both 16-base windows WF_EXTEND reads for it (wfa.go:408-454) are resident.  A cell outside -- or a match run
// that reaches the window's end -- takes the slow form of WF_EXTEND (long_extend): reposition the window at the lowest
// pending cell (all G lanes of the pair load the two windows in one round of 16-byte loads), extend what is inside, repeat
// until no cell is pending.  It always makes progress, so nothing is ever handed on for its length; results are those of
// the plain instances because WF_EXTEND computes the same full LCP either way.
// LDSA (round 4, wfahip_align_pair only): the rows of the pair's arena live in LDS behind the sequences (P.lds_arena_off words
// in, P.arena_words of them) instead of global memory -- the wave that walks the backtrace afterwards reads them where they
// are: no region refills, no global traffic at all between the sequences coming in and the record going out.  160 KB of LDS
// hold 620 score indices (scores up to 1 240 at g = 2); a pair that needs more reports ST_REDO_ARENA and the host runs the
// global-memory instance.
// DX / DOE (round 5): the penalty shape x/g : (o+e)/g the instance is built for, with e/g == 1 (wfa.go:32-36 takes any
// penalties; 4/6/2 and its multiples are 2 : 4, 2/4/2 is 1 : 3, 1/1/1 is 1 : 2, 4/4/2 is 2 : 3).  The M ring holds the last
// R = max(DX, DOE) rows, indexed by (step mod R) with the step loop unrolled R times; the row of step i goes to slot
// i mod R (the row of step i - R, which nothing reads any more), M[s-o-e] is slot (i - DOE) mod R and M[s-x] is slot
// (i - DX) mod R (wfa.go:557-560).  The I and D rings are the one previous row (e/g == 1).
template <int G, int BATCH, bool STREAM = false, int PPT = 0, bool CENSUS = true, bool LONG = false, bool LDSA = false, int DX = 2, int DOE = 4>
__global__ __launch_bounds__(64, (G == 8 && PPT == 0 ? WFA_BLK8_WAVES : (G == 16 && BATCH == 1 && !CENSUS && (!STREAM || WFA_BLK_STREAM_WAVES5) ? WFA_BLK_WAVES : 4))) void wfa_blk_kernel(const KParams P) {
    static_assert(!STREAM || (G == 16 && BATCH == 1), "streamed backtrace: 16 lanes per pair, unbatched refill");
    static_assert(PPT == 0 || (G == 8 && PPT == 4) || (G == 64 && (PPT == 1 || PPT == 2)),
                  "diagonals per lane can only be overridden for the 8-lane narrow instance and the lone-pair instances");
    static_assert(!LONG || (BATCH == 1 && !STREAM && (PPT == 0 || G == 64) && G >= 16), "sliding sequence windows: unbatched, pre-packed input");
    static_assert(!LDSA || (G == 64 && PPT == 1 && BATCH == 1 && !STREAM && !LONG && !CENSUS), "LDS-resident arena: the lone-pair instance");
    static_assert(DX >= 1 && DOE >= 1 && DX <= 4 && DOE <= 4, "ring depths of one to four score steps");
    constexpr int R = DX > DOE ? DX : DOE;  // rows of the M ring
    constexpr int PP  = PPT ? PPT : (G >= 32 ? 4 : 64 / G);  // diagonals per lane
    constexpr int NG  = 64 / G;                // pairs per wave
#ifdef WFA_BLK_W
    constexpr int W = WFA_BLK_W;  // experiment: pretend the window is narrower
#else
    constexpr int W   = G * PP;   // window width in diagonals: 64 (G = 16, 8), 128 (G = 32) or 256 (G = 64); also the arena's row pitch
#endif
    constexpr bool TILED = WFA_BLK_TILED != 0 && W == 64;  // arena layout: CompactView fmt 3 (else fmt 1 / 4: plain rows)
    // G = 64 with ONE or TWO diagonals per lane (round 4): the whole wave on one pair with a 64- / 128-diagonal window -- a quarter /
    // half of the instructions of a step of the four-diagonals-per-lane instances.  For pairs that are alone on their SIMD
    // anyway (one Align call; a few hundred long reads): there the step time is the latency of one wave's instruction
    // stream, not the GPU's throughput.
    using Ops         = typename std::conditional<(G == 8 && PP == 4), BlkOps8n, BlkOps<G>>::type;
    constexpr int SHD = G >= 32 ? PP : Ops::SHIFT_D;  // diagonals per window shift (G >= 32: the shifts move one lane)
#ifdef WFA_MARKS
#define WFA_MARK(i) asm volatile("; ##MARK " #i)
#else
#define WFA_MARK(i) do {} while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x, j = lane & (G - 1), grp = lane / G;
    if constexpr (STREAM) {
        if (blockIdx.x < P.n_stream_wgs) {  // (dispatched first: resident from the start)
            stream_backtrace(P);
            return;
        }
    }

    const uint32_t  SW = P.lds_seq_words;
    constexpr int   BM = 8;                                    // meta words per batch slot: wi, pr, nq, mt, q_off, t_off
    const uint32_t  GW = BATCH > 1 ? BATCH * (2 * SW + BM) : 2 * SW;  // LDS words of one group
    const uint32_t *lq = lds + grp * GW;                       // packed query / target of the group's current pair
    const uint32_t *lt = lq + SW;
    int             bslot = 0, bcnt = 0;                       // batch mode: next staged slot / staged slots
    bool            dry   = false;                             // batch mode: this wave has seen the end of the queue
    const uint64_t        cap      = P.arena_words;
    const int             mdd      = (int)P.max_dist_diff;
    const int             minwf    = (int)P.min_wf_len;
    const bool            adaptive = P.adaptive != 0;
    const uint32_t        seed_si  = P.dx;  // the mismatch seed M[x][0] belongs to step x/g

    using Red = BlkRed<G>;

    // per-pair state (identical in the G lanes of a group)
    int        st = 0;  // 0 = needs a pair, 1 = running, 2 = queue exhausted
    uint32_t   pidx = 0, si = 0, cells = 0;  // (pidx: the pair's index in the chunk = its arena slot; pair_of() is its id)
    int        n = 0, m = 0, kb = 0, k0 = 0;
    int        pend_h = 0;  // STREAM: end offset of the pair that waits to be pushed
    bool       slow = false, first_eq = false;
    bool       pend = false;  // STREAM: the pair just finished still has to be pushed (its score index, end offset and
                              // cell count wait in si, Ak and cells, which are dead until the next pair starts)
    uint32_t  *rowp = nullptr;  // row of the current score in the pair's arena slot (64 words per score)
    uint32_t   lone_taken = 0u;  // LDSA: the launch's one pair has been started
    if constexpr (LDSA) {
        if (lane == 0) lds[P.lds_arena_off - 4] = ST_PENDING;  // (the walk's start words: ST_OK once the forward pass has written them)
    }
    // LONG: the pair's sequence windows.  lqo / lto: LDS index of packed word 0 of the query / target (window base minus the
    // window's first word: lds[lqo + w] is packed word w while it is resident); [lo16, hi16]: offsets h whose two 16-base
    // windows are resident on every diagonal of the diagonal window; kc16: the diagonal the two bases are tied by
    int        lqo = 0, lto = 0, lo16 = INT32_MIN, hi16 = INT32_MIN, kc16 = 0;  // (lo16 = hi16 = INT32_MIN: no window yet -- no offset is inside)
    const int  CW  = (int)P.lds_seq_words;                       // LONG: words per sequence window (a multiple of 4 G)
    const int  SWp = LONG ? (int)((P.prepack_words - 4u) / 2u) : 0;  // LONG: words per sequence in a pre-packed slot (a multiple of 4)
    constexpr int LDM = (G * (PPT ? PPT : (G >= 32 ? 4 : 64 / G))) / 2 + 128;  // LONG: half the diagonal window + the drift of its centre a window tolerates
    constexpr int LMARGIN = 128;                                 // LONG: bases kept below the cell a window is positioned at

    uint32_t M[R][PP], I[PP], D[PP];  // offsets, 0 = absent; M[i mod R] = row of step i
    int      rlo[R], rhi[R];          // band of each kept M row (absolute k); empty = (BIG, -BIG)
    int      lim[PP], lmx[PP];        // max(1, min(n + k, m)) and max(n + k, m) of the lane's diagonals
#pragma unroll
    for (int d = 0; d < R; d++) {
        rlo[d] = BK_BIG, rhi[d] = -BK_BIG;
#pragma unroll
        for (int p = 0; p < PP; p++) M[d][p] = 0u;
    }
#pragma unroll
    for (int p = 0; p < PP; p++) I[p] = D[p] = 0u, lim[p] = 0, lmx[p] = 0;

    const auto pair_of = [&](uint32_t idx) { return P.work ? P.work[idx] : P.chunk_first + idx; };
    const int  rows_cap = (int)(cap / W);  // rows (score indices) a pair's arena slot holds
    const auto set_window = [&]() {
        k0 = kb + PP * j;
#pragma unroll
        for (int p = 0; p < PP; p++) lim[p] = imax2(1, imin2(n + k0 + p, m)), lmx[p] = imax2(n + k0 + p, m);
    };
    const auto clear_rings = [&]() {
#pragma unroll
        for (int d = 0; d < R; d++) {
            rlo[d] = BK_BIG, rhi[d] = -BK_BIG;
#pragma unroll
            for (int p = 0; p < PP; p++) M[d][p] = 0u;
        }
#pragma unroll
        for (int p = 0; p < PP; p++) I[p] = D[p] = 0u;
    };

    // 16-base windows of the query at base v / the target at base h (LONG: through the sliding windows)
    const auto winq = [&](int v) -> uint32_t {
        if constexpr (LONG) {
            const int w = lqo + (v >> 4);
            return __funnelshift_r(lds[w], lds[w + 1], (uint32_t)(v & 15) * 2u);
        } else {
            return SeqView<0>::win16(lq, v);
        }
    };
    const auto wint = [&](int h) -> uint32_t {
        if constexpr (LONG) {
            const int w = lto + (h >> 4);
            return __funnelshift_r(lds[w], lds[w + 1], (uint32_t)(h & 15) * 2u);
        } else {
            return SeqView<0>::win16(lt, h);
        }
    };
    // LONG: the groups with `mine` set load their two sequence windows so that offset hm lies LMARGIN bases above the lower
    // end of what is resident (hm, kb, pidx: the same in all lanes of a group).  Bases may be negative (the first window of a
    // pair starts below base 0) or reach past the packed words: those words read as zero and are never compared -- a cell's
    // room ends at the sequence ends.
    const auto reposition = [&](bool mine, int hm) __attribute__((always_inline)) {
        if constexpr (LONG) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (mine) {
                kc16            = (kb + W / 2) & ~63;
                const int tb16  = (hm - LDM - LMARGIN) & ~63, qb16 = tb16 - kc16;
                const int tbw   = tb16 >> 4, qbw = qb16 >> 4;  // multiples of 4 words
                const uint32_t *const slot = P.prepack + (uint64_t)pidx * P.prepack_words + 4u;
                uint32_t *const dq = lds + grp * GW, *const dt = dq + CW;
                for (int i = 4 * j; i < CW; i += 4 * G) {
                    // (always a load from inside the slot, the words outside it zeroed afterwards: a select between "the slot" and
                    // "a zero vector" makes the compiler park that vector in scratch memory and load through a flat pointer)
                    const int  wq = qbw + i, wt = tbw + i;
                    const bool oq = wq >= 0 && wq < SWp, ot = wt >= 0 && wt < SWp;
                    uint4 a = *reinterpret_cast<const uint4 *>(slot + (oq ? wq : 0));
                    uint4 b = *reinterpret_cast<const uint4 *>(slot + SWp + (ot ? wt : 0));
                    a.x = oq ? a.x : 0u, a.y = oq ? a.y : 0u, a.z = oq ? a.z : 0u, a.w = oq ? a.w : 0u;
                    b.x = ot ? b.x : 0u, b.y = ot ? b.y : 0u, b.z = ot ? b.z : 0u, b.w = ot ? b.w : 0u;
                    *reinterpret_cast<uint4 *>(dq + i) = a;
                    *reinterpret_cast<uint4 *>(dt + i) = b;
                }
                lqo  = (int)(grp * GW) - qbw;
                lto  = (int)(grp * GW) + CW - tbw;
                lo16 = tb16 + LDM;
                hi16 = tb16 + 16 * CW - 32 - LDM;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };

#ifdef WFA_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    unsigned long long evt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-steps, slow steps, hit steps, reduce steps, found steps,
                                                            // found pair-steps, continuation iterations, running pair-steps
#define WFA_EVT(i, v) (evt[i] += (v))
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#else
#define WFA_EVT(i, v) ((void)0)
#endif
    // refill of idle groups; returns true when the queue is exhausted and no pair is left
    const auto refill = [&]() __attribute__((always_inline)) -> bool {
        {
            // ------------------------------------------------------------ refill: all groups that need a pair take
            // consecutive queue entries with ONE atomic, load their pair's lengths/offsets together, and stage +
            // 2-bit pack the sequences: short pairs (one pass of the group's own lanes) all groups at once, long
            // pairs one group after the other with all 64 lanes (coalesced dword loads).
            const unsigned long long need = __ballot(st == 0);
            if constexpr (BATCH > 1) {
                if (WFA_RARE(need != 0ull)) {
                    uint32_t *const gbase = lds + grp * GW;
                    uint32_t *const gmeta = gbase + BATCH * 2 * SW;
                    // ---- groups whose batch is used up take BATCH queue entries each (one atomic for all of them)
                    const bool               fetch = st == 0 && bslot >= bcnt;
                    const unsigned long long fneed = __ballot(fetch);
                    if (fneed != 0ull && dry) {  // nothing left to take: no round trip to find that out again
                        if (fetch) st = 2;
                    } else if (fneed != 0ull) {
                        uint32_t gbits = 0u;
#pragma unroll
                        for (int r = 0; r < NG; r++) gbits |= (uint32_t)((fneed >> (G * r)) & 1ull) << r;
                        uint32_t base = 0;
                        // (bn <= BATCH entries per grab: the host picks it so that a small chunk spreads evenly over
                        // the resident groups instead of leaving a quarter of them without a batch)
                        const int bn = (int)P.blk_batch_n;
                        if (lane == 0) base = atomicAdd(P.queue_head, (uint32_t)(bn * __builtin_popcount(gbits)));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        dry  = base + (uint32_t)(bn * __builtin_popcount(gbits)) >= P.chunk_n;
                        const uint32_t wi0 = base + (uint32_t)(bn * __builtin_popcount(gbits & ((1u << grp) - 1u)));
                        // lane j < bn of a fetching group owns entry wi0 + j
                        const uint32_t wi  = wi0 + (uint32_t)j;
                        const bool     own = fetch && j < bn;
                        const bool     got = own && wi < P.chunk_n;
                        uint32_t pr = 0, nq = 0, mt = 0;
                        uint64_t qo = 0, to = 0;
                        if (got) {
                            pr = P.work ? P.work[wi] : P.chunk_first + wi;
                            nq = P.q_len[pr], mt = P.t_len[pr], qo = P.q_off[pr], to = P.t_off[pr];
                        }
                        uint32_t status = ST_PENDING;
                        if (nq == 0 || mt == 0)
                            status = ST_EMPTY;  // wfa.go:204-206
                        else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
                            status = ST_TOO_LONG;  // wfa.go:207-209
                        else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW || ((nq > mt ? nq : mt) + 15u) / 16u + 1u > 16u)
                            status = ST_REDO_LDS;  // (batch mode is for reads of at most 240 bases: 1-2 staging passes of the group)
                        if (got && status != ST_PENDING) {
                            P.pair_meta[wi] = make_uint4(status, 0u, 0u, 0u);
                            if (status >= ST_REDO_BYTES) push_redo(P, pr, status);
                        }
                        if (own) {
                            uint32_t *const mw = gmeta + BM * j;
                            mw[0] = wi, mw[1] = pr, mw[2] = (got && status == ST_PENDING) ? nq : 0u, mw[3] = mt;
                            mw[4] = (uint32_t)qo, mw[5] = (uint32_t)(qo >> 32), mw[6] = (uint32_t)to, mw[7] = (uint32_t)(to >> 32);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        // ---- staging: the fetching groups one after the other, each with ALL 64 lanes in one round --
                        // eight lanes per batch slot, four per sequence, up to four packed words per lane (a sequence
                        // of this mode has at most 16 words), all loads of a round independent.  Staged slot by slot
                        // with a group's own lanes, the dependent load rounds of a batch (8 slots x 2 sequences x 1-2
                        // passes) were half of a short-read wave's time.
                        // When most groups of the wave fetch at once (the start of the kernel) the groups stage their
                        // own batches side by side instead: the same number of load rounds, no serialisation by group.
                        static_assert(BATCH <= 8, "one staging round covers eight slots");
                        uint32_t   badmask = 0u;
                        const bool side_by_side = 2 * __builtin_popcount(gbits) > NG;
                        if (side_by_side) {
#pragma unroll 1
                            for (int sl = 0; sl < bn; sl++) {
                                const uint32_t *const mw = gmeta + BM * sl;
                                const uint32_t nq_s = fetch ? mw[2] : 0u, mt_s = mw[3];
                                if (nq_s != 0u) {
                                    const uint64_t qo_s = (uint64_t)mw[4] | ((uint64_t)mw[5] << 32);
                                    const uint64_t to_s = (uint64_t)mw[6] | ((uint64_t)mw[7] << 32);
                                    uint32_t *const sq  = gbase + sl * 2 * SW;
                                    bool b = stage_pack<G>(P.blob, qo_s, nq_s, sq, j);
                                    b |= stage_pack<G>(P.blob, to_s, mt_s, sq + SW, j);
                                    badmask |= b ? (1u << sl) : 0u;
                                }
                            }
                            badmask = (uint32_t)Red::or1((int)badmask);
                        }
#pragma unroll 1
                        for (uint32_t gb = side_by_side ? 0u : gbits; gb != 0u; gb &= gb - 1u) {
                            const int             r     = __builtin_ctz(gb);  // wave-uniform
                            uint32_t *const       rbase = lds + r * GW;
                            const uint32_t *const rmeta = rbase + BATCH * 2 * SW;
                            uint32_t              badl  = 0u;
                            const int             sl = lane >> 3, sq = (lane >> 2) & 1;
                            if (sl < bn) {
                                const uint32_t *const mw = rmeta + BM * sl;
                                const uint32_t nq_s = mw[2];
                                const uint32_t len  = sq ? mw[3] : nq_s;
                                const uint32_t nw   = (len + 15u) >> 4;
                                const uint64_t off  = (uint64_t)mw[4 + 2 * sq] | ((uint64_t)mw[5 + 2 * sq] << 32);
                                bool           b    = false;
                                if (nq_s != 0u) {
#pragma unroll 1
                                    for (uint32_t jw = (uint32_t)lane & 3u; jw <= nw; jw += 4u)
                                        rbase[sl * 2 * SW + sq * SW + jw] = stage_word(P.blob, off, len, jw, b);
                                }
                                badl |= b ? (1u << sl) : 0u;
                            }
                            uint32_t wb = 0u;
#pragma unroll
                            for (int sl = 0; sl < BATCH; sl++) wb |= (__ballot((badl >> sl) & 1u) != 0ull ? 1u : 0u) << sl;
                            if (grp == r) badmask = wb;
                        }
                        if (own && ((badmask >> j) & 1u) != 0u) {  // a byte outside ACGT: the byte-compare path takes it
                            P.pair_meta[wi] = make_uint4(ST_REDO_BYTES, 0u, 0u, 0u);
                            push_redo(P, pr, ST_REDO_BYTES);
                            gmeta[BM * j + 2] = 0u;
                        }
                        if (fetch) {
                            bslot = 0;
                            bcnt  = wi0 >= P.chunk_n ? 0 : (int)imin2(bn, (int)(P.chunk_n - wi0));
                            if (bcnt == 0) st = 2;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                    // ---- idle groups with a staged pair start it (no global memory on this path)
                    if (st == 0 && bslot < bcnt) {
                        const uint32_t *const mw = gmeta + BM * bslot;
                        const uint32_t nq = mw[2], mt = mw[3];
                        if (nq != 0u) {
                            pidx = mw[0];
                            lq = gbase + bslot * 2 * SW, lt = lq + SW;
                            n = (int)nq, m = (int)mt;
                            const int Ak = m - n;
                            si = 0, cells = 0, slow = false;
                            kb   = -(W / 2) + PP * imax2(-(3 * W / 8) / PP, imin2((3 * W / 8) / PP, Ak / (2 * PP)));
                            rowp = P.arena + (uint64_t)pidx * cap;
                            first_eq = ((lq[0] ^ lt[0]) & 3u) == 0u;  // q[0] == t[0] (wfa.go:155)
                            set_window();
                            clear_rings();
                            st = 1;
                        }  // (a rejected entry: the group takes the next slot in the next round)
                        bslot += 1;
                    }
                }
            } else if (WFA_RARE(need != 0ull)) {
                uint32_t gbits = 0u;  // bit r: group r needs a pair (wave-uniform)
#pragma unroll
                for (int r = 0; r < NG; r++) gbits |= (uint32_t)((need >> (G * r)) & 1ull) << r;
                uint32_t base = 0;
                if constexpr (STREAM) {
                    // lane 0 takes the queue entries, lane 1 reserves the done_q entries of the groups that finished in
                    // the step before: one atomic instruction, one round trip
                    uint32_t pbits = 0u;
                    const unsigned long long pb = __ballot(pend);
#pragma unroll
                    for (int r = 0; r < NG; r++) pbits |= (uint32_t)((pb >> (G * r)) & 1ull) << r;
                    uint32_t *const aptr = lane == 0 ? P.queue_head : P.done_ctl;
                    const uint32_t  aval = (uint32_t)__builtin_popcount(lane == 0 ? gbits : pbits);
                    uint32_t        r2   = 0u;
                    if (lane < 2 && aval != 0u) r2 = atomicAdd(aptr, aval);
                    base              = (uint32_t)__builtin_amdgcn_readlane((int)r2, 0);
                    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)r2, 1);
                    if (pb != 0ull) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the rows before their entry
                    if (pend && j == 0)
                        blk_store_sc1(P.done_q + (t0 + (uint32_t)__builtin_popcount(pbits & ((1u << grp) - 1u))), pidx + 1u,
                                      si * P.g, (uint32_t)pend_h, cells);
                    pend = false;
                } else if constexpr (LDSA) {
                    base = lone_taken, lone_taken = 1u;  // (one pair, one wave: no queue to ask -- two round trips to the L2 less per Align)
                } else {
                    if (lane == 0) base = atomicAdd(P.queue_head, (uint32_t)__builtin_popcount(gbits));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                }
                const bool     want = st == 0;
                const uint32_t wi   = base + (uint32_t)__builtin_popcount(gbits & ((1u << grp) - 1u));
                const bool     got  = want && wi < P.chunk_n;
                if (want && !got) st = 2;
                uint32_t pr = 0, nq = 0, mt = 0;
                uint64_t qo = 0, to = 0;
                uint32_t status = ST_PENDING;
                const uint32_t *slot = nullptr;  // the pair's pre-packed slot
                if (got) {
                    pr = P.work ? P.work[wi] : P.chunk_first + wi;
                    if (P.prepack) {
                        // lengths, status and the packed words were prepared by wfa_prepack_kernel: one load round after the
                        // queue atomic instead of two (lengths / offsets, then the bytes) and no packing arithmetic here
                        slot = P.prepack + (uint64_t)wi * P.prepack_words;
                        nq = slot[0], mt = slot[1], status = slot[2];
                    } else if constexpr (LDSA) {
                        nq = P.one_n, mt = P.one_m, qo = 0ull, to = ((uint64_t)P.one_n + 15ull) & ~15ull;
                    } else {
                        nq = P.q_len[pr], mt = P.t_len[pr], qo = P.q_off[pr], to = P.t_off[pr];
                    }
                }
                if (!P.prepack) {
                    if (nq == 0 || mt == 0)
                        status = ST_EMPTY;  // wfa.go:204-206
                    else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
                        status = ST_TOO_LONG;  // wfa.go:207-209
                    else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW)
                        status = ST_REDO_LDS;
                }
                const bool stage = got && status == ST_PENDING;
                bool       bad   = false;
                if constexpr (LONG) {
                    // (nothing to copy here: the windows are loaded below, once the pair's window base is known)
                } else if (P.prepack) {
                    if (stage) {  // every refilling group copies its own 2 SW words, all groups side by side
                        uint32_t *const dst = const_cast<uint32_t *>(lq);
                        for (uint32_t w = (uint32_t)j; w < 2u * SW; w += (uint32_t)G) dst[w] = slot[4u + w];
                    }
                } else if (__ballot(stage && ((nq > mt ? nq : mt) + 15u) / 16u + 1u > (uint32_t)G) == 0ull) {
                    if (stage) {
                        bad = stage_pack<G>(P.blob, qo, nq, const_cast<uint32_t *>(lq), j);
                        bad |= stage_pack<G>(P.blob, to, mt, const_cast<uint32_t *>(lt), j);
                    }
                    bad = Red::or1(bad ? 1 : 0) != 0;
                } else {
                    for (unsigned long long todo = __ballot(stage); todo != 0ull;) {
                        const int r = __builtin_ctzll(todo) / G;  // wave-uniform group index
                        todo &= ~((G == 64 ? ~0ull : (1ull << (G & 63)) - 1ull) << ((G * r) & 63));
                        const uint64_t qo_r = ((uint64_t)__shfl((uint32_t)(qo >> 32), G * r, 64) << 32) | __shfl((uint32_t)qo, G * r, 64);
                        const uint64_t to_r = ((uint64_t)__shfl((uint32_t)(to >> 32), G * r, 64) << 32) | __shfl((uint32_t)to, G * r, 64);
                        const uint32_t nq_r = __shfl(nq, G * r, 64), mt_r = __shfl(mt, G * r, 64);
                        uint32_t *const rq  = lds + r * 2 * SW;
                        bool b = stage_pack<64>(P.blob, qo_r, nq_r, rq, lane);
                        b |= stage_pack<64>(P.blob, to_r, mt_r, rq + SW, lane);
                        if (__ballot(b) != 0ull && grp == r) bad = true;
                    }
                }
                if (stage && bad) status = ST_REDO_BYTES;
                if (got && status != ST_PENDING && j == 0) {
                    P.pair_meta[wi] = make_uint4(status, 0u, 0u, 0u);
                    if (status >= ST_REDO_BYTES) push_redo(P, pr, status);
                    if constexpr (STREAM) blk_push_not_ok(P, wi);
                }  // (the group stays in state 0 and pulls another pair in the next round)
                if (stage && !bad) {
                    pidx = wi;
                    n = (int)nq, m = (int)mt;
                    const int Ak = m - n;
                    si = 0, cells = 0, slow = false;
                    kb   = -(W / 2) + PP * imax2(-(3 * W / 8) / PP, imin2((3 * W / 8) / PP, Ak / (2 * PP)));  // k = 0 (the seed) inside, biased towards Ak
                    if constexpr (LDSA) rowp = lds + P.lds_arena_off;
                    else rowp = P.arena + (uint64_t)pidx * cap;
                    if constexpr (LONG) first_eq = ((slot[4] ^ slot[4 + SWp]) & 3u) == 0u;
                    else first_eq = ((lq[0] ^ lt[0]) & 3u) == 0u;  // q[0] == t[0] (wfa.go:155)
                    set_window();
                    clear_rings();
                    st = 1;
                }
                if constexpr (LONG) {
                    if (__ballot(stage && !bad) != 0ull) reposition(stage && !bad, 1);  // the seed cell's offset is 1
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            return __ballot(st != 2) == 0ull;
        }
    };
    // one score step of every running pair; PH = step & 3 selects the ring slots at compile time
    const auto step = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_c)::value;
        {
            const bool run = (st == 1);
            WFA_STAMP(0); WFA_MARK(0);  // refill

            uint32_t(&Mo)[PP] = M[(ph + R - DOE) % R];  // M[s-o-e]
            uint32_t(&Mx)[PP] = M[(ph + R - DX) % R];   // M[s-x]
            uint32_t(&Mn)[PP] = M[ph];                  // the slot of the row being computed: the oldest row (= M[s-o-e] when DOE >= DX), read before it is replaced

            // ------------------------------------------------------------ WF_NEXT (wfa.go:549-700)
            uint32_t nM[PP], nI[PP], nD[PP], wd[PP], cc[PP];
            const uint32_t a_edge = Ops::dn1(Mo[PP - 1], j), b_edge = Ops::dn1(I[PP - 1], j);
            const uint32_t c_edge = Ops::up1(Mo[0], j), d_edge = Ops::up1(D[0], j);
            const bool     slow_any = __ballot(run && slow) != 0ull;
            WFA_EVT(0, 1), WFA_EVT(1, slow_any ? 1 : 0), WFA_EVT(7, __builtin_popcountll(__ballot(run)) / G);
            if (WFA_OFTEN(!slow_any)) {
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const uint32_t a = p ? Mo[p - 1] : a_edge, b = p ? I[p - 1] : b_edge;
                    const uint32_t c = p < PP - 1 ? Mo[p + 1] : c_edge, d = p < PP - 1 ? D[p + 1] : d_edge;
                    const uint32_t x = Mx[p];
                    const uint32_t mi = umax2(a, b), Isk = mi + (mi != 0u ? 1u : 0u);  // wfa.go:579-609
                    const uint32_t Dsk = umax2(c, d);                                   // wfa.go:614-645
                    const uint32_t x1  = x + (x != 0u ? 1u : 0u);
                    const uint32_t Msk = umax3(Isk, Dsk, x1);                           // wfa.go:655
                    // the four decisions of wfa.go:590-600,626-636,657-693, one bit each, shifted in under the offset
                    // (blk_word(): what the backtrace needs of this diagonal and score; no rejections on this path,
                    // so backTrace's recomputed pre-extension offset, wfa.go:766-817, is Msk itself)
                    nM[p] = Msk, nI[p] = Isk, nD[p] = Dsk;
#ifndef WFA_BLK_NO_ASM_BITS  // (hand-placed: 21.07 vs 21.77 ms per 1e6 x 1 kbp pairs against what hipcc makes of blk_word())
                    wd[p] = blk_word_asm(Msk, a, b, c, d, x1, umax2(Isk, Dsk), Isk, Dsk);
#else
                    wd[p] = blk_word(Msk, a < b, c < d, Msk == x1, Msk == Isk);
#endif
                    cc[p] = CENSUS ? (mi != 0u ? 1u : 0u) + (Dsk != 0u ? 1u : 0u) + (Msk != 0u ? 1u : 0u) : 0u;
                }
            } else {
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const int      k  = k0 + p;
                    const uint32_t a0 = p ? Mo[p - 1] : a_edge, b0 = p ? I[p - 1] : b_edge;
                    const uint32_t c0 = p < PP - 1 ? Mo[p + 1] : c_edge, d0 = p < PP - 1 ? D[p + 1] : d_edge;
                    const uint32_t x0 = Mx[p];
                    // rejections: > m (not >=) for I and X sources, offset - k > n for D and X sources
                    const uint32_t a = (int)a0 > m ? 0u : a0, b = (int)b0 > m ? 0u : b0;
                    const uint32_t c = (int)c0 - k > n ? 0u : c0, d = (int)d0 - k > n ? 0u : d0;
                    const uint32_t x = ((int)x0 > m || (int)x0 - k > n) ? 0u : x0;
                    const uint32_t mi = umax2(a, b), tI = umin2(mi, 1u), Isk = mi + tI;
                    const uint32_t Dsk = umax2(c, d), tD = umin2(Dsk, 1u);
                    const uint32_t x1  = x + umin2(x, 1u);
                    const uint32_t Msk = umax3(Isk, Dsk, x1);
                    // wfa.go:657-693: the mismatch wins a tie when its source exists, then the insertion
                    const bool fromX = x != 0u && Msk == x1;
                    const bool fromI = !fromX && Msk == Isk;
                    // backTrace recomputes the pre-extension offset from the un-rejected sources (wfa.go:766-817)
                    const uint32_t mu = umax2(a0, b0), Iu = mu + umin2(mu, 1u), Du = umax2(c0, d0);
                    const uint32_t Xu = x0 + umin2(x0, 1u);
                    const bool     iext = a < b, dext = c < d;
                    const uint32_t o0   = (fromI && iext) ? Iu : ((!fromX && !fromI && dext) ? Du : umax3(Iu, Du, Xu));
                    const bool     kin  = k >= -(n - 1) && k <= m - 1;  // wfa.go:562-563
                    nM[p] = kin ? Msk : 0u, nI[p] = kin ? Isk : 0u, nD[p] = kin ? Dsk : 0u;
                    wd[p] = blk_word(o0, iext, dext, fromX, fromI);
                    cc[p] = (CENSUS && kin) ? tI + tD + umin2(Msk, 1u) : 0u;
                }
            }
            // seeds of initComponents (wfa.go:155-160): M[0][0] = 1/Match or M[x][0] = 1/Mismatch
            if (WFA_RARE(__ballot(si <= seed_si) != 0ull)) {  // (ONE compare on the step's path; `want` below is the exact test)
                const bool want = run && ((si == 0u && first_eq) || (si == seed_si && !first_eq));
#pragma unroll
                for (int p = 0; p < PP; p++)
                    if (want && k0 + p == 0 && nM[p] == 0u)
                        nM[p] = 1u, wd[p] = first_eq ? BLK_SEED_MATCH : BLK_SEED_MISMATCH, cc[p] = CENSUS ? 1u : 0u;
            }
            // The lone-wave instances (one / two diagonals per lane) issue the LDS reads of WF_EXTEND's first windows HERE, ahead of
            // the store of the row: a wave that is alone on its SIMD has nobody to hide that round trip behind but its own
            // address arithmetic and the store's.
            constexpr bool EARLY = G == 64 && PP <= 2;
            uint32_t       ew[PP][4];
            if constexpr (EARLY) {
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const int h = (int)nM[p], v = h - (k0 + p);
                    if constexpr (LONG) {
                        const int iq = lqo + (v >> 4), it = lto + (h >> 4);
                        ew[p][0] = lds[iq], ew[p][1] = lds[iq + 1], ew[p][2] = lds[it], ew[p][3] = lds[it + 1];
                    } else {
                        ew[p][0] = lq[v >> 4], ew[p][1] = lq[(v >> 4) + 1], ew[p][2] = lt[h >> 4], ew[p][3] = lt[(h >> 4) + 1];
                    }
                }
            }
            // ------------------------------------------------------------ store the row's words
            // Right away: the words are complete (they hold PRE-extension offsets), and their registers are free for the
            // rest of the step.  A lane stores when one of its cells exists -- 3/4 of the lanes have none, their lines stay
            // untouched; what wf-adaptive deletes below is stored too, nothing ever reads it.
            const bool no_room = run && (int)si >= rows_cap;
            {
                uint32_t anyc = 0u;
#pragma unroll
                for (int p = 0; p < PP; p++) anyc |= nM[p];
                if (run && !no_room && anyc != 0u) {
                    if constexpr (PP < 4) {
                        // one or two diagonals per lane: a 4- or 8-byte store (the same tiles / plain rows)
                        uint32_t *const row = TILED ? rowp + (((uint32_t)k0 & 60u) << 3) + ((uint32_t)k0 & 3u) : rowp + ((uint32_t)k0 & (uint32_t)(W - 1));
                        if constexpr (PP == 1) *row = wd[0];
                        else *reinterpret_cast<uint2 *>(row) = make_uint2(wd[0], wd[PP - 1]);
                    } else if constexpr (TILED) {
                        // tile of 8 scores x 64 diagonals: [diagonal / 4][score & 7][diagonal & 3] (CompactView fmt 3)
                        uint32_t *const row = rowp + (((uint32_t)k0 & 63u) << 3);
                        if constexpr (STREAM)
                            blk_store_sc1(row, wd[0], wd[1], wd[2], wd[3]);
                        else
                            *reinterpret_cast<uint4 *>(row) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                        if constexpr (PP == 8) *reinterpret_cast<uint4 *>(row + 32) = make_uint4(wd[4], wd[5], wd[6], wd[7]);
                    } else {
                        uint32_t *const row = rowp + ((uint32_t)k0 & (uint32_t)(W - 1));
                        if constexpr (STREAM)
                            blk_store_sc1(row, wd[0], wd[1], wd[2], wd[3]);
                        else
                            *reinterpret_cast<uint4 *>(row) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                        if constexpr (PP == 8) *reinterpret_cast<uint4 *>(row + 4) = make_uint4(wd[4], wd[5], wd[6], wd[7]);
                    }
                }
            }
            WFA_STAMP(1); WFA_MARK(1);  // next

            // ------------------------------------------------------------ WF_EXTEND (wfa.go:381-458), first 16 bases
            uint32_t cmask = 0u;  // positions whose first window matched completely and may go on
            uint32_t lpend = 0u;  // LONG: positions whose extension waits for their sequence window
            bool     lslow = false;
            if constexpr (LONG) {
                // every cell that has room must find both its 16-base windows resident; one outside (the band has moved on,
                // or back) sends the wave through long_extend below with all its cells pending
                bool oob = false;
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const int h = (int)nM[p];
                    // (lo16 <= h <= hi16 as ONE unsigned compare; h > 0 and room left: lim > h)
                    oob |= h != 0 && lim[p] > h && (uint32_t)(h - lo16) > (uint32_t)(hi16 - lo16);
                }
                lslow = __ballot(run && oob) != 0ull;
            }
            if (WFA_OFTEN(!lslow)) {
#pragma unroll
            for (int p = 0; p < PP; p++) {
                const int      h    = (int)nM[p];
                const int      rem  = lim[p] - h;  // bases left on this diagonal; <= 0: at / past an end (wfa.go:404)
                const uint32_t room = h ? (uint32_t)imax2(rem, 0) : 0u;  // (nothing for an absent cell)
                const int      v    = h - (k0 + p);  // absent cells read a harmless word (LDS reads cannot fault)
                uint32_t       xr;
                if constexpr (EARLY)
                    xr = __funnelshift_r(ew[p][0], ew[p][1], (uint32_t)(v & 15) * 2u) ^ __funnelshift_r(ew[p][2], ew[p][3], (uint32_t)(h & 15) * 2u);
                else
                    xr = winq(v) ^ wint(h);
                // (v_ffbl_b32 of 0 is 0xFFFFFFFF: a window that matched completely runs to the end of the room)
                const uint32_t run  = umin2(ffbl_raw(xr) >> 1, room);
                nM[p] += umin2(run, 16u);
                if (run > 16u) cmask |= 1u << p;  // the whole window matched and bases remain
            }
            // the few cells (normally the one on the alignment path) that matched a whole window: each lane takes its
            // candidates one at a time and keeps comparing 16-base windows until a mismatch or a sequence end
            if (WFA_RARE(__ballot(cmask != 0u) != 0ull)) do {  // (out of line: the step without a candidate falls straight through)
                const int psel = (int)ffbl_raw(cmask);  // -1 in lanes without a candidate
                int       h = 0, lm = 0;
#pragma unroll
                for (int p = 0; p < PP; p++)
                    if (psel == p) h = (int)nM[p], lm = lim[p];
                const int kd = k0 + psel;
                bool      go = cmask != 0u;
                if constexpr (LONG) {  // (a run that has reached the end of the window waits for the next one)
                    if (go && h > hi16) lpend |= 1u << psel, go = false;
                }
                do {
                    WFA_EVT(6, 1);
                    const int      rem = lm - h;
                    const uint32_t xr  = winq(h - kd) ^ wint(h);
                    const uint32_t cnt = umin2(ffbl_raw(xr) >> 1, (uint32_t)imin2(imax2(rem, 0), 16));
                    h += go ? (int)cnt : 0;
                    go = go && xr == 0u && rem > 16;
                    if constexpr (LONG) {
                        if (go && h > hi16) lpend |= 1u << psel, go = false;
                    }
                } while (__ballot(go) != 0ull);
#pragma unroll
                for (int p = 0; p < PP; p++)
                    if (psel == p) nM[p] = (uint32_t)h;
                cmask &= cmask - 1u;
            } while (__ballot(cmask != 0u) != 0ull);
            }
            if constexpr (LONG) {
                if (WFA_RARE(lslow)) {  // nothing has been extended yet: every cell with room is pending
#pragma unroll
                    for (int p = 0; p < PP; p++)
                        if (run && nM[p] != 0u && lim[p] - (int)nM[p] > 0) lpend |= 1u << p;
                }
                // long_extend: until no cell is pending -- the pairs whose lowest pending cell is not inside their window
                // reposition it there; every pending cell inside its window is extended, 16 bases a round, until a
                // mismatch, the end of its room (done) or the end of the window (still pending).  The lowest pending
                // cell of a pair always moves on or finishes, so the loop ends; which pairs share the wave changes
                // nothing but the number of rounds.
                if (WFA_RARE(__ballot(lpend != 0u) != 0ull)) do {
                    int hm = BK_BIG;
#pragma unroll
                    for (int p = 0; p < PP; p++) hm = ((lpend >> p) & 1u) ? imin2(hm, (int)nM[p]) : hm;
                    hm = -Red::max1(-hm);
                    const bool move = hm != BK_BIG && (hm < lo16 || hm > hi16);
                    if (__ballot(move) != 0ull) reposition(move, hm);
#pragma unroll
                    for (int p = 0; p < PP; p++) {
                        int        h  = (int)nM[p];
                        const int  kd = k0 + p, lm = lim[p];
                        bool       go = ((lpend >> p) & 1u) != 0u && h >= lo16 && h <= hi16, fin = false;
                        while (__ballot(go) != 0ull) {
                            const int      rem  = lm - h;
                            const uint32_t xr   = winq(h - kd) ^ wint(h);
                            const uint32_t cnt  = umin2(ffbl_raw(xr) >> 1, (uint32_t)imin2(imax2(rem, 0), 16));
                            const bool     done = xr != 0u || rem <= 16;
                            h += go ? (int)cnt : 0;
                            fin = fin || (go && done);
                            go  = go && !done && h <= hi16;
                        }
                        nM[p] = (uint32_t)h;
                        if (fin) lpend &= ~(1u << p);
                    }
                } while (__ballot(lpend != 0u) != 0ull);
            }
            WFA_STAMP(2); WFA_MARK(2);  // extend

            // ------------------------------------------------------------ ends reached? termination (wfa.go:235-239)
            bool nz[PP], hit[PP], hitl = false;
#pragma unroll
            for (int p = 0; p < PP; p++) nz[p] = nM[p] != 0u, hit[p] = nM[p] >= (uint32_t)lim[p], hitl |= hit[p];
            bool       term    = false;
            const bool hit_any = __ballot(hitl) != 0ull;
            bool       ghit    = false;  // this pair has a cell at a sequence end in this step
            WFA_EVT(2, hit_any ? 1 : 0);
            if (WFA_RARE(hit_any)) {
                bool tl = false;
#pragma unroll
                for (int p = 0; p < PP; p++) tl |= (k0 + p == m - n && nz[p] && (int)nM[p] >= m);
                const int r = Red::or1((hitl ? 1 : 0) | (tl ? 2 : 0));
                ghit = (r & 1) != 0;
                slow |= ghit;
                term = run && (r & 2) != 0;
            }

            // ------------------------------------------------------------ band of the row + wf-adaptive (wfa.go:461-540)
            // remaining distance (wfa.go:488) = max(m-h, n-v) = max(m, n+k) - h; an entry is usable iff h < min(m, n+k)
            int      ilo = 0, ihi = -1;  // band to keep (window-relative)
            bool     anyM = false;
            uint32_t csum = 0u;
            if constexpr (G == 64 && PP <= 2) {
              if (WFA_OFTEN(!hit_any)) {
                // One or two diagonals per lane, lanes in diagonal order: the row's range and the band wf-adaptive keeps are the
                // first / last set bits of ballots (scalar code); the minimum distance is the one real reduction of the step
                int d[PP], dmin = BK_BIG;
#pragma unroll
                for (int p = 0; p < PP; p++) d[p] = lmx[p] - (int)nM[p], dmin = nz[p] ? imin2(dmin, d[p]) : dmin;
                const int mind = wave_min(dmin);
                int       glo = BK_BIG, ghi = -BK_BIG;
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const unsigned long long bm = __ballot(nz[p]);
                    if (bm != 0ull) glo = imin2(glo, PP * (int)__builtin_ctzll(bm) + p), ghi = imax2(ghi, PP * (63 - (int)__builtin_clzll(bm)) + p);
                }
                anyM            = ghi >= 0;
                const bool want = run && adaptive && anyM && (ghi - glo + 1) >= minwf;
                const int  thr  = want ? mind + mdd : BK_BIG;
                ilo = BK_BIG, ihi = -BK_BIG;
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const unsigned long long bo = __ballot(nz[p] && d[p] <= thr);
                    if (bo != 0ull) ilo = imin2(ilo, PP * (int)__builtin_ctzll(bo) + p), ihi = imax2(ihi, PP * (63 - (int)__builtin_clzll(bo)) + p);
                }
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const int  ix   = PP * j + p;
                    const bool keep = ix >= ilo && ix <= ihi;
                    nM[p] = keep ? nM[p] : 0u, nI[p] = keep ? nI[p] : 0u, nD[p] = keep ? nD[p] : 0u;
                    csum += keep ? cc[p] : 0u;
                }
              }
            }
            if (WFA_OFTEN(!hit_any) && !(G == 64 && PP <= 2)) {
                // No cell of the wave sits at a sequence end (97 % of the steps): every M cell is usable, so the
                // tight range of M (M.Lo/M.Hi, the wf-adaptive trigger of wfa.go:242) is [first, last usable entry],
                // and with the threshold at +infinity when wf-adaptive does not run, [first_ok, last_ok] IS the
                // band to keep in every case (nothing fails -> it is the tight range; wfa.go:509-524 otherwise: the
                // entries between the last leading failure and first_ok are holes).  Two reduction rounds, no
                // branch: (first M, last M, min distance), then (first_ok, last_ok).
                int glo = BK_BIG, ghi = -BK_BIG, mind = BK_BIG, dd[PP];
#pragma unroll
                for (int p = PP - 1; p >= 0; p--) glo = nz[p] ? PP * j + p : glo;
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    ghi   = nz[p] ? PP * j + p : ghi;
                    dd[p] = lmx[p] - (int)nM[p];
                    mind  = nz[p] ? imin2(mind, dd[p]) : mind;
                }
                Red::min_max_min(glo, ghi, mind);
                anyM = ghi >= 0;
                const bool want = run && adaptive && anyM && (ghi - glo + 1) >= minwf;
                const int  thr  = want ? mind + mdd : BK_BIG;
                int        first_ok = BK_BIG, last_ok = -BK_BIG;
#pragma unroll
                for (int p = PP - 1; p >= 0; p--) first_ok = (nz[p] && dd[p] <= thr) ? PP * j + p : first_ok;
#pragma unroll
                for (int p = 0; p < PP; p++) last_ok = (nz[p] && dd[p] <= thr) ? PP * j + p : last_ok;
                Red::min_max(first_ok, last_ok);
                ilo = first_ok, ihi = last_ok;
#pragma unroll
                for (int p = 0; p < PP; p++) {  // Delete of wfa.go:526-535: the words never exist
                    const int  ix   = PP * j + p;
                    const bool keep = ix >= ilo && ix <= ihi;
                    nM[p] = keep ? nM[p] : 0u, nI[p] = keep ? nI[p] : 0u, nD[p] = keep ? nD[p] : 0u;
                    csum += keep ? cc[p] : 0u;  // (the words of deleted cells stay as they are: nothing ever reads them)
                }
            } else if (WFA_RARE(hit_any)) {
                // ------------------------------------------------------------ tight range of the M cells = M.Lo/M.Hi
                int glo = BK_BIG, ghi = -BK_BIG;  // window-relative index of the lane's first / last M cell
#pragma unroll
                for (int p = PP - 1; p >= 0; p--) glo = nz[p] ? PP * j + p : glo;
#pragma unroll
                for (int p = 0; p < PP; p++) ghi = nz[p] ? PP * j + p : ghi;
                Red::min_max(glo, ghi);
                anyM = ghi >= 0;
                ilo = glo, ihi = ghi;
                csum = 0u;
#pragma unroll
                for (int p = 0; p < PP; p++) csum += cc[p];

                // ------------------------------------------------------------ wf-adaptive (wfa.go:461-540)
                // remaining distance (wfa.go:488) = max(m-h, n-v) = max(m, n+k) - h; an entry is usable iff h < min(m, n+k)
                const bool want_reduce = run && !term && adaptive && anyM && (ghi - glo + 1) >= minwf;
                if (__ballot(want_reduce) != 0ull) {
                    WFA_EVT(3, 1);
                    int  dd[PP], mind = BK_BIG, maxd = -BK_BIG;
                    bool vd[PP];
#pragma unroll
                    for (int p = 0; p < PP; p++) {
                        vd[p] = nz[p] && !hit[p];
                        dd[p] = lmx[p] - (int)nM[p];
                        mind  = vd[p] ? imin2(mind, dd[p]) : mind;
                        maxd  = vd[p] ? imax2(maxd, dd[p]) : maxd;
                    }
                    Red::min_max(mind, maxd);
                    const int  thr   = mind + mdd;
                    const bool found = want_reduce && mind != BK_BIG && maxd > thr;  // some distance fails (wfa.go:507)
                    if (__ballot(found) != 0ull) {
                        WFA_EVT(4, 1), WFA_EVT(5, __builtin_popcountll(__ballot(found)) / G);
                        int first_ok = BK_BIG, last_ok = -BK_BIG;
#pragma unroll
                        for (int p = PP - 1; p >= 0; p--) first_ok = (vd[p] && dd[p] <= thr) ? PP * j + p : first_ok;
#pragma unroll
                        for (int p = 0; p < PP; p++) last_ok = (vd[p] && dd[p] <= thr) ? PP * j + p : last_ok;
                        Red::min_max(first_ok, last_ok);
                        // wfa.go:509-511: _lo = one past the last failing entry before the first non-failing one.  The
                        // entries between that one and first_ok are unusable ones: holes (absent cells) or cells at a
                        // sequence end.  While no cell of the PAIR has reached an end they are all holes, and deleting
                        // or keeping a hole is the same thing: _lo = first_ok gives the identical row.  (The test is per
                        // pair, not per wave: a band that keeps leading holes because a neighbour pair sits at an end
                        // would make the window bookkeeping -- and with it the rare hand-over of a pair whose band
                        // touches the window edge -- depend on which pairs share a wave.)
                        int newlo = first_ok;
                        if (hit_any) {
                            int leadp = -1;
#pragma unroll
                            for (int p = 0; p < PP; p++) leadp = (vd[p] && PP * j + p < first_ok) ? PP * j + p : leadp;
                            leadp = Red::max1(leadp);
                            newlo = ghit ? (leadp >= 0 ? leadp + 1 : glo) : first_ok;
                        }
                        if (found) ilo = newlo, ihi = last_ok;  // wfa.go:517-524
                        csum = 0u;
#pragma unroll
                        for (int p = 0; p < PP; p++) {  // Delete of wfa.go:526-535: the words never exist
                            const int  ix   = PP * j + p;
                            const bool keep = ix >= ilo && ix <= ihi;
                            nM[p] = keep ? nM[p] : 0u, nI[p] = keep ? nI[p] : 0u, nD[p] = keep ? nD[p] : 0u;
                            csum += keep ? cc[p] : 0u;
                        }
                    }
                }
            }
            WFA_STAMP(3); WFA_MARK(3);  // ranges + wf-adaptive

            // ------------------------------------------------------------ the row's census and position
            // (groups that are not running hold all-zero rings: every cell above is absent)
            const bool keepl = anyM && ihi >= ilo && !no_room;
            cells += keepl ? csum : 0u;
            if constexpr (TILED) {
                rowp += 4;
                rowp += (((uint32_t)(uintptr_t)rowp & 0x70u) == 0u) ? 480 : 0;  // past the tile's 8th score: next tile
            } else {
                rowp += W;
            }
            WFA_STAMP(4); WFA_MARK(4);  // stores

            // ------------------------------------------------------------ the new row enters the rings
#pragma unroll
            for (int p = 0; p < PP; p++) Mn[p] = nM[p], I[p] = nI[p], D[p] = nD[p];
            rlo[ph] = keepl ? kb + ilo : BK_BIG;
            rhi[ph] = keepl ? kb + ihi : -BK_BIG;

            // ------------------------------------------------------------ finish / next score
            bool fin = run && (term || no_room);
            if (WFA_RARE(__ballot(fin) != 0ull)) {
                int ctot = (CENSUS && P.census) ? (int)cells : 0;
                int hf   = 0;  // extended offset of the end cell M[s][Ak]: where the backtrace starts
#pragma unroll
                for (int p = 0; p < PP; p++)
                    if (k0 + p == m - n) hf = (int)Mn[p];
                Red::max_add(hf, ctot);
                if (fin && j == 0) {
                    if (no_room) {
                        P.pair_meta[pidx] = make_uint4(ST_REDO_ARENA, 0u, 0u, 0u);
                        push_redo(P, pair_of(pidx), ST_REDO_ARENA);
                        if constexpr (STREAM) blk_push_not_ok(P, pidx);
                    } else if constexpr (!STREAM) {
                        P.pair_meta[pidx] = make_uint4(ST_OK, si * P.g, (uint32_t)hf, (uint32_t)ctot);
                        // (the walk of the LDS instance reads its start from the four words in front of the rows: nothing of it waits for global memory)
                        if constexpr (LDSA) *reinterpret_cast<uint4 *>(lds + P.lds_arena_off - 4) = make_uint4(ST_OK, si * P.g, (uint32_t)hf, (uint32_t)ctot);
                    }
                }
                if (fin) {
                    if constexpr (STREAM) {
                        pend  = !no_room;  // pushed by the next refill; until then si / Ak / cells keep the entry's fields
                        pend_h = hf;
                        cells = (uint32_t)ctot;
                    }
                    st = 0;
                    clear_rings();
                }
            }
            if (run && !fin) si += 1u;

            // ------------------------------------------------------------ keep every kept row inside [kb+1, kb+62]
            {
                // the older rows already lie inside: only the row just added can touch the window's edge
                const bool live    = keepl && !fin;
                const bool need_dn = live && ilo <= 0;
                const bool need_up = live && ihi >= W - 1;
                if (WFA_RARE(__ballot(need_dn || need_up) != 0ull)) {
                    int ulo = rlo[0], uhi = rhi[0];
#pragma unroll
                    for (int d = 1; d < R; d++) ulo = imin2(ulo, rlo[d]), uhi = imax2(uhi, rhi[d]);
                    const bool wide = (need_dn && (need_up || uhi >= kb - SHD + W - 1)) || (need_up && ulo <= kb + SHD);
                    const bool dn = need_dn && !wide, up = need_up && !wide;
#pragma unroll
                    for (int d = 0; d < R; d++)
#pragma unroll
                        for (int p = 0; p < PP; p++) {
                            const uint32_t a = Ops::shr(M[d][p], j), b = Ops::shl(M[d][p], j);
                            M[d][p]          = dn ? a : (up ? b : M[d][p]);
                        }
#pragma unroll
                    for (int p = 0; p < PP; p++) {
                        const uint32_t a = Ops::shr(I[p], j), b = Ops::shl(I[p], j);
                        const uint32_t c = Ops::shr(D[p], j), d = Ops::shl(D[p], j);
                        I[p]             = dn ? a : (up ? b : I[p]);
                        D[p]             = dn ? c : (up ? d : D[p]);
                    }
                    kb += dn ? -SHD : (up ? SHD : 0);
                    set_window();
                    if constexpr (LONG) {
                        // the sequence windows are tied together at diagonal kc16: once the diagonal window has drifted further
                        // from it than [lo16, hi16] allows for, they count as empty and the next WF_EXTEND positions them anew
                        const int dk = kb + W / 2 - kc16;
                        if (dk < -128 || dk > 128) lo16 = INT32_MIN, hi16 = INT32_MIN;
                    }
                    if (WFA_RARE(__ballot(wide) != 0ull)) {  // the band does not fit the window: hand the pair on
                        if (wide && j == 0) {
                            P.pair_meta[pidx] = make_uint4(ST_REDO_BAND, 0u, 0u, 0u);
                            push_redo(P, pair_of(pidx), ST_REDO_BAND);
                            if constexpr (STREAM) blk_push_not_ok(P, pidx);
                        }
                        if (wide) {
                            st = 0;
                            clear_rings();
                        }
                    }
                }
            }
            WFA_STAMP(5); WFA_MARK(5);  // ring + finish + window
        }
    };
#ifndef WFA_BLK_SHARED_REFILL
    for (;;) {  // R copies of refill + step (four at 2 : 4: 68 KB of code, but measured 6 % faster than the switch below)
        if (WFA_RARE(refill())) break;
        step(std::integral_constant<int, 0>{});
        if constexpr (R > 1) {
            if (WFA_RARE(refill())) break;
            step(std::integral_constant<int, 1 % R>{});
        }
        if constexpr (R > 2) {
            if (WFA_RARE(refill())) break;
            step(std::integral_constant<int, 2 % R>{});
        }
        if constexpr (R > 3) {
            if (WFA_RARE(refill())) break;
            step(std::integral_constant<int, 3 % R>{});
        }
    }
#else
    // experiment: one copy of the refill code, the ring phases of the step selected by a wave-uniform switch
    // (40 KB of code; slower: the merge after the switch costs register moves)
    for (int ph = 0;; ph = (ph + 1) % R) {
        if (WFA_RARE(refill())) break;
        switch (ph) {
        case 0: step(std::integral_constant<int, 0>{}); break;
        case 1: step(std::integral_constant<int, 1 % R>{}); break;
        case 2: step(std::integral_constant<int, 2 % R>{}); break;
        default: step(std::integral_constant<int, 3 % R>{}); break;
        }
    }
#endif
#ifdef WFA_STAMPS
    if (lane == 0 && P.debug_info) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(P.debug_info);
        for (int i = 0; i < 8; i++) atomicAdd(acc + i, stamp_acc[i]);
        for (int i = 0; i < 8; i++) atomicAdd(acc + 8 + i, evt[i]);
    }
#endif
    if constexpr (STREAM) {
        // The queue is exhausted and this wave's pairs are done (the last refill pushed their entries): help with the
        // backtrace of the pairs still in flight elsewhere.
        stream_backtrace(P);
    }
    if constexpr (G == 64 && PP == 1 && !LONG) {
        // wfahip_align_pair: the wave walks the backtrace of the pairs of its launch itself -- one launch for the whole Align.
        // The sequences in LDS are dead by now: their place is the walk's arena region (the host reserves 4 KB).  The arena
        // rows were stored by this very wave; the release / acquire pair makes them its loads' too.
        if (P.fuse_bt) {
            if constexpr (LDSA) {
                // (one pair: the rows are still where they were written, and so is the walk's start -- unless the pair was handed on:
                // then the start words still hold what the kernel began with, and the record comes from pair_meta)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (lds[P.lds_arena_off - 4] != ST_OK) __threadfence();
                backtrace_wave_one<true>(P, blockIdx.x, lds + P.lds_arena_off);
            } else {
                __threadfence();
            }
            if constexpr (LDSA) {
            } else
                for (uint32_t idx = blockIdx.x; idx < P.chunk_n; idx += gridDim.x) backtrace_wave_one(P, idx, lds);
            // (one wave, one pair: the control words go back to zero here, so the next Align starts without a memset of its own)
            if (lane == 0 && gridDim.x == 1u) *P.queue_head = 0u, *P.redo_count = 0u, *P.ops_cursor = 0ull;
        }
    }
}

// The sequences of a chunk, 2-bit packed once, by the whole GPU, before the forward kernel starts: 2 GB read, 0.5 GB
// written per 1e6 x 1 kbp pairs.  One workgroup per pair, one thread per packed word (16 bases): four or five aligned
// dword loads, a funnel shift, and per dword ten integer instructions -- the 2-bit codes are (byte >> 1) & 3, gathered
// with two shift-or steps; a byte outside ACGT shows when the code's canonical letter (one v_perm_b32 through "ACTG")
// differs from the byte.  (Round 2's version ran stage_word(), ~100 instructions per word, and was bound by them:
// 0.89 ms; the forward kernels' own refill still uses stage_word.)  slot = {n, m, status, 0, q words [SW], t words [SW]}.
// four bases (the bytes of w) as eight bits; bad |= a byte outside ACGT
WFA_DEV uint32_t prepack_dword(uint32_t w, bool &bad) {
    const uint32_t x = (w >> 1) & 0x03030303u;
    bad |= __builtin_amdgcn_perm(0u, 0x47544341u, x) != w;  // code -> 'A' 'C' 'T' 'G'
    const uint32_t y = x | (x >> 6);
    return (y & 0xFu) | ((y >> 12) & 0xF0u);
}
// The two halves of prepack_word(): the loads of word jw (into d; sh / nb: byte shift and number of bases; nb = 0: no such
// word), and the arithmetic on them.  Apart so that a caller can have the loads of several words in flight at once.
WFA_DEV void prepack_fetch(const uint8_t *blob, uint64_t off, uint32_t len, uint32_t jw, uint32_t (&d)[5], uint32_t &sh, uint32_t &nb) {
    const uint32_t nw = (len + 15u) >> 4;
    sh = 0u, nb = 0u;
#pragma unroll
    for (int i = 0; i < 5; i++) d[i] = 0u;
    if (jw >= nw) return;
    const uintptr_t a  = (uintptr_t)(blob + off) + 16ull * jw;
    const uint32_t *p  = (const uint32_t *)(a & ~(uintptr_t)3);
    sh = (uint32_t)(a & 3) * 8u;
    nb = (len - 16u * jw) < 16u ? (len - 16u * jw) : 16u;
    const uint32_t  nd = ((uint32_t)(a & 3) + nb + 3u) >> 2;  // dwords that hold valid bytes: 1..5
    if ((a & 15) == 0 && nb == 16u) {  // 16 whole bases at a 16-byte boundary (the usual layout): one 16-byte load, a KB per wave and instruction
        const uint4 v = *reinterpret_cast<const uint4 *>(a);
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w, d[4] = 0u;
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) d[i] = ((uint32_t)i < nd) ? p[i] : 0u;
    }
}
WFA_DEV uint32_t prepack_finish(const uint32_t (&d)[5], uint32_t sh, uint32_t nb, bool &bad) {
    if (nb == 0u) return 0u;
    uint32_t word = 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t w = __funnelshift_r(d[i], d[i + 1], sh);
        if (nb < 16u) {  // the last word of a sequence: bytes past its end count as 'A' (code 0)
            const uint32_t mb = nb > 4u * i ? (nb - 4u * i < 4u ? nb - 4u * i : 4u) : 0u;
            const uint32_t km = mb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * mb)) - 1u);
            w = (w & km) | (0x41414141u & ~km);
        }
        word |= prepack_dword(w, bad) << (8 * i);
    }
    return word;
}
WFA_DEV uint32_t prepack_word(const uint8_t *blob, uint64_t off, uint32_t len, uint32_t jw, bool &bad) {
    uint32_t d[5], sh, nb;
    prepack_fetch(blob, off, len, jw, d, sh, nb);
    return prepack_finish(d, sh, nb, bad);
}

// (a wave per pair, PREPACK_PAIRS pairs per wave: one workgroup per pair was bound by the dispatch of a million tiny
// workgroups -- 1.01 ms for 2.5 GB of traffic)
constexpr int PREPACK_PAIRS = 4;
#ifndef WFA_NO_AUX_KERNELS
// ppw: pairs a wave packs one after the other -- PREPACK_PAIRS, or 1 for a few long pairs (500 x 50 kbp: 125 waves of four pairs
// were 0.32 ms of latency; 500 waves: 0.08).
__global__ __launch_bounds__(256) void wfa_prepack_kernel(const KParams P, uint32_t *out, uint32_t SW, uint32_t PW, uint32_t ppw) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv   = blockIdx.x * 4u + (threadIdx.x >> 6);
    // short reads (a slot of at most 32 words): two pairs side by side, 32 lanes each
    const uint32_t LP = 2u * SW <= 32u ? 32u : 64u, side = lane / LP, v0 = lane % LP, n_side = 64u / LP;
    // (the lengths and offsets of the NEXT pair are loaded while this one is packed, and a lane has the loads of two words
    // in flight: the kernel was half the time waiting for a round trip with nothing else under way -- 3.2 TB/s)
    struct Hd { uint32_t nq, mt; uint64_t qo, to; };
    const auto load_hd = [&](uint32_t k) -> Hd {
        const uint32_t wi = wv * ppw + k + side;
        Hd h = {0u, 0u, 0ull, 0ull};
        if (k < ppw && wi < P.chunk_n) {
            const uint32_t pr = P.work ? P.work[wi] : P.chunk_first + wi;
            h.nq = P.q_len[pr], h.mt = P.t_len[pr], h.qo = P.q_off[pr], h.to = P.t_off[pr];
        }
        return h;
    };
    Hd nxt = load_hd(0);
    for (uint32_t k = 0; k < ppw; k += n_side) {
        const uint32_t wi = wv * ppw + k + side;
        if (wv * ppw + k >= P.chunk_n) return;
        const bool     have = wi < P.chunk_n;
        const Hd       cur  = nxt;
        nxt                 = load_hd(k + n_side);
        const uint32_t nq = cur.nq, mt = cur.mt;
        const uint64_t qo = cur.qo, to = cur.to;
        uint32_t       status = ST_PENDING;
        if (nq == 0 || mt == 0)
            status = ST_EMPTY;  // wfa.go:204-206
        else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
            status = ST_TOO_LONG;  // wfa.go:207-209
        else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW)
            status = ST_REDO_LDS;
        uint32_t *const slot = out + (uint64_t)wi * PW;
        bool            bad  = false;
        if (have && status == ST_PENDING) {
            for (uint32_t v = v0; v < 2u * SW; v += 2u * LP) {
                const uint32_t va = v, vb = v + LP;  // (vb >= 2 SW: past the slot -- fetched as "no such word", not stored)
                const bool     qa = va < SW, qb = vb < SW, hb = vb < 2u * SW;
                uint32_t       da[5], db[5], sha, shb, nba, nbb;
                prepack_fetch(P.blob, qa ? qo : to, qa ? nq : mt, qa ? va : va - SW, da, sha, nba);
                prepack_fetch(P.blob, qb ? qo : to, hb ? (qb ? nq : mt) : 0u, qb ? vb : vb - SW, db, shb, nbb);
                slot[4u + va] = prepack_finish(da, sha, nba, bad);
                if (hb) slot[4u + vb] = prepack_finish(db, shb, nbb, bad);
            }
        }
        const unsigned long long bm = __ballot(bad), mine = LP == 64u ? ~0ull : (0xFFFFFFFFull << (32u * side));
        if ((bm & mine) != 0ull) status = ST_REDO_BYTES;  // a byte outside ACGT: the byte-compare path takes the pair
        if (have && v0 == 0u) slot[0] = nq, slot[1] = mt, slot[2] = status, slot[3] = 0u;
    }
}

#endif  // WFA_NO_AUX_KERNELS

}  // namespace wfa
