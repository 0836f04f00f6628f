// wfa_wide.hpp -- kernel W (round 6): a WORKGROUP per pair with the last rows in 16-bit LDS rings of ANY width.
//
// For what the register-window kernels cannot hold: semi-global alignments of reads up to 2 047 bases (wfa.go:163-183 seeds every
// one of the n + m - 1 diagonals, at score 0 and again at score x, and wf-adaptive cuts nothing until the leading diagonal is
// MaxDistDiff bases ahead: a 1 kbp pair keeps ~2 000 diagonals for its first dozen score steps, eight times the cells of its global
// alignment), with wf-adaptive on or off.  Until round 6 those batches went to wfa_generic_kernel -- one workgroup per pair, the
// sources of WF_NEXT read back from the arena in global memory, three 32-bit words per diagonal stored: 2.5e6 pairs/s on
// 1e6 x 1 kbp, a 23-fold cliff behind the same pairs aligned globally.
//
// Here a workgroup owns a pair (the hardware dispatcher is the pair queue):
//   * both sequences 2-bit packed in LDS (packed by the workgroup itself, stage_pack), and the rows the next scores source as BARE
//     16-bit offsets in LDS rings indexed by diagonal: four M rows (slot = score index & 3), one I and one D row (penalties with
//     e / g == 1: the sub-wave kernels' shapes).  6 x (n + m) x 2 bytes -- 25 KB for a 1 kbp pair;
//   * wide rows (PHASE 0) are computed in rounds of 256 diagonals, FOUR CONSECUTIVE DIAGONALS PER LANE -- a lane's cells of a row are
//     one 8-byte LDS word, three of a cell's five sources its own registers -- by NW waves side by side (four when the rings take more
//     than 12 KB: with one, six workgroups a CU would leave a SIMD a wave and a half); narrow rows (PHASE 1) in tiles of 64, a lane
//     per diagonal.  Per cell the exact WF_NEXT (wfa.go:549-700, rejections included: a semi-global row always has cells at
//     sequence ends), the seeds (wfa.go:155-183), WF_EXTEND on the packed LDS words (wfa.go:381-458).  The new M row overwrites the
//     slot of M[s-o-e] and the I / D rows are updated in place: every read of a round goes out before its first write, and the one
//     cell a round overwrites that the next one sources (M[s-o-e][k-1], I[s-e][k-1] at its first diagonal) travels in a register
//     (several waves: through two LDS words in turn);
//   * wf-adaptive (wfa.go:461-540) as wave reductions over the rounds' partial results (several waves: + an exchange through LDS);
//     a row that cuts cells is swept once more to delete them from the rings.  A narrow row of one tile stays in registers: every
//     reduction on it is a ballot and a scalar bit scan;
//   * the semi-global end cell (wfa.go:270-375) is found while the rows are in LDS: the reference scans every score from
//     the last down to 0 and keeps the LOWEST score with a hit -- the first one in ascending order;
//   * per diagonal and score ONE 16-bit word goes to the arena -- the blocked kernels' blk_word(): the pre-extension
//     offset backTrace recomputes (wfa.go:766-817) over the four decisions of next() -- in rows laid back to back with a
//     16-byte directory entry {first halfword, lo, width} per score index growing down from the end of the pair's slot
//     (CompactView fmt 11); wfa_backtrace_kernel walks it with back_trace_compact().
// What the workgroup cannot hold (an arena that overflows) is handed to the generic ladder like every sub-wave kernel does.
#pragma once
#include "wfa_device.hpp"

namespace wfa {

constexpr uint32_t WIDE_MAX_LEN = 2047;  // 16-bit ring offsets and 12-bit arena offsets (blk_word() in a halfword)
constexpr uint32_t WIDE_FMT     = 11u;   // CompactView: rows of 16-bit blk_word()s + the 16-byte directory of fmt 0
// halfwords of one ring row: a slot per diagonal of [-(n-1), m-1] plus guards on both sides, a multiple of 64
__host__ __device__ inline uint32_t wide_row_hw(uint32_t max_len) { return (2u * max_len + 96u + 63u) & ~63u; }
// LDS words of a wave: the two packed sequences, then the six rows
__host__ __device__ inline uint32_t wide_lds_words(uint32_t seq_words, uint32_t max_len) {
    return ((2u * seq_words + 3u) & ~3u) + 6u * wide_row_hw(max_len) / 2u + 144u;
}
// pair_meta of a finished pair: {ST_OK, score of the walk's start, its extended offset | (its diagonal + WIDE_KBIAS) << 16, cells}
constexpr uint32_t WIDE_KBIAS = 32768u;
// Two launches per chunk.  PHASE 0 runs a pair from its seeds while its rows are wide -- 25 KB of rings per workgroup at 1 kbp: six
// per CU -- and leaves as soon as every row in the rings spans at most WIDE_NARROW diagonals (a score step or four after
// wf-adaptive's first cut: the older rows leave the rings) and score x has been seeded; it hands the pair on as a CHECKPOINT: the
// rings' live part, already in PHASE 1's layout, + the loop's state, WIDE_CKPT_WORDS words per pair.  PHASE 1 picks the pair up with
// rings of WIDE_RW diagonals indexed modulo (6 KB: the CU's full complement of waves) and runs it to its end; same code, same arena,
// same directory -- the backtrace sees one pair.  A pair that never narrows (wf-adaptive off) finishes in PHASE 0.
// (WIDE_RW 256 was too few: after the first cut a band is still 100-200 wide and widens by two a step until the leader pulls away.)
constexpr int      WIDE_RW = 512, WIDE_NARROW = 200;
constexpr uint32_t WIDE_SCR_WORDS = 144u;  // LDS words behind the rings: what the waves of a workgroup hand one another (two exchange buffers of 32, the boundary cells of up to 17 quarters)
constexpr uint32_t WIDE_CKPT_HDR = 24u, WIDE_CKPT_WORDS = WIDE_CKPT_HDR + 6u * WIDE_RW / 2u;
__host__ __device__ inline uint32_t wide_lds_words_narrow(uint32_t seq_words) { return ((2u * seq_words + 3u) & ~3u) + 6u * WIDE_RW / 2u + WIDE_SCR_WORDS; }

// DX / DOE: the penalty shape x/g : (o+e)/g (e/g == 1), as in the sub-wave kernels (wfa_fwd.hpp)
// Two 16-bit offsets per register (the wide rows' interior: v_pk_max_u16 / v_pk_add_u16 / v_pk_sub_u16 with clamp / v_pk_min_u16)
typedef unsigned short wide_us2 __attribute__((ext_vector_type(2)));
WFA_DEV wide_us2 wide_pk(uint32_t x) { return __builtin_bit_cast(wide_us2, x); }
WFA_DEV uint32_t wide_u32(wide_us2 x) { return __builtin_bit_cast(uint32_t, x); }
WFA_DEV wide_us2 wide_max(wide_us2 a, wide_us2 b) { return __builtin_elementwise_max(a, b); }
// (`one` = (1, 1) from a register the compiler cannot see through: it turns min(a, 1) into two compares, two selects and a permute otherwise)
WFA_DEV wide_us2 wide_ind(wide_us2 a, wide_us2 one) { return __builtin_elementwise_min(a, one); }                             // 1 where a != 0
WFA_DEV wide_us2 wide_lt(wide_us2 a, wide_us2 b, wide_us2 one) { return wide_ind(__builtin_elementwise_sub_sat(b, a), one); }  // 1 where a < b
// WF_NEXT of two neighbouring diagonals whose sources need no rejection (wfa.go:572-699; the decisions as blk_word_asm() takes them: the mismatch
// wins iff x1 >= max(Isk, Dsk), else the insertion iff Isk >= Dsk; backTrace's recomputed pre-extension offset is the M offset itself)
WFA_DEV void wide_next2(wide_us2 a, wide_us2 b, wide_us2 c, wide_us2 d, wide_us2 x, wide_us2 one, uint32_t &M2, uint32_t &I2, uint32_t &D2, uint32_t &W2) {
    const wide_us2 mi = wide_max(a, b), Isk = mi + wide_ind(mi, one), Dsk = wide_max(c, d), x1 = x + wide_ind(x, one);
    const wide_us2 t = wide_max(Isk, Dsk), Msk = wide_max(t, x1);
    const wide_us2 iext = wide_lt(a, b, one), dext = wide_lt(c, d, one), fx = one - wide_lt(x1, t, one), fi = one - wide_lt(Isk, Dsk, one);
    wide_us2 w = Msk + Msk + iext;
    w = w + w + dext, w = w + w + fx, w = w + w + fi;
    w &= (wide_us2)(0) - wide_ind(Msk, one);  // (no cell: no word)
    M2 = wide_u32(Msk), I2 = wide_u32(Isk), D2 = wide_u32(Dsk), W2 = wide_u32(w);
}

#ifndef WFA_WIDE_EU
#define WFA_WIDE_EU 5  // waves per SIMD the register allocation aims at (0: the compiler's choice -- 110 registers with the packed path: four waves, 8 % slower on g3)
#endif
#if WFA_WIDE_EU
#define WFA_WIDE_EU_ATTR __attribute__((amdgpu_waves_per_eu(WFA_WIDE_EU, WFA_WIDE_EU)))
#else
#define WFA_WIDE_EU_ATTR
#endif
// NW: waves of the pair's workgroup (PHASE 0: the rings of a 1 kbp pair are 25 KB, six workgroups a CU -- with one wave each the SIMDs would
// hold a wave and a half; the waves of a workgroup take a row's rounds side by side)
template <int DX = 2, int DOE = 4, int PHASE = 0, int NW = 1>
__global__ __launch_bounds__(64 * NW) WFA_WIDE_EU_ATTR void wfa_wide_kernel(const KParams P) {
    static_assert(DX >= 1 && DOE >= 1 && DX <= 4 && DOE <= 4, "ring depths of one to four score steps");
    static_assert(NW == 1 || (PHASE == 0 && (NW == 2 || NW == 4)), "the narrow phase is one wave");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int       tid  = (int)threadIdx.x;
    const int       lane = tid & 63;
    const int       wv   = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    const uint32_t  SW   = P.lds_seq_words;
    const uint32_t  WH   = PHASE ? (uint32_t)WIDE_RW : P.sub_lds_words;  // halfwords of a ring row (PHASE 0: wide_row_hw of the launch's longest pair)
    uint32_t *const lq   = lds;
    uint32_t *const lt   = lds + SW;
    uint16_t *const ring = reinterpret_cast<uint16_t *>(lds + ((2u * SW + 3u) & ~3u));
    uint16_t *const rowI = ring + 4u * WH, *const rowD = ring + 5u * WH;
    const auto      rowM = [&](uint32_t i) -> uint16_t * { return ring + (i & 3u) * WH; };
    const auto      rfl  = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    uint32_t *const scr  = reinterpret_cast<uint32_t *>(ring + 6u * WH);  // WIDE_SCR_WORDS: two exchange buffers of 32 words, then the quarters' boundary cells
    const auto      lds_sync = [] {  // what one lane of the workgroup stored, another lane reads: in order, and not from a stale register
        if constexpr (NW > 1) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    // every wave hands in CNT values; afterwards every wave holds all of them (all[w * CNT + i]).  Two buffers in turn: a wave that is
    // one exchange ahead writes the other one
    uint32_t   xcount = 0u;
    const auto xchg   = [&](const int *mine, int *all, const int CNT) {
        uint32_t *const b = scr + (xcount & 1u) * 32u;
        xcount++;
        if (lane == 0)
            for (int i = 0; i < CNT; i++) b[wv * CNT + i] = (uint32_t)mine[i];
        __syncthreads();
        for (int i = 0; i < NW * CNT; i++) all[i] = __builtin_amdgcn_readfirstlane((int)b[i]);
    };
    (void)xchg, (void)scr;

    const uint32_t x = P.x, g = P.g;
    const uint64_t cap      = P.arena_words;  // 32-bit words of a pair's slot
    const bool     glob     = P.global_alignment != 0;
    const bool     adaptive = P.adaptive != 0;
    const int      mdd = (int)P.max_dist_diff, minwf = (int)P.min_wf_len;
    constexpr int  BIG = 0x3FFFFFFF;

    {
        // ---------------------------------------------------------------- the workgroup's pair (one wave, one pair: the dispatcher is the queue --
        // a persistent wave's queue loop around a body of this size had the compiler spill the loop's exit mask to a VGPR lane
        // and, in one build, not bring it back: the wave never left)
        const uint32_t idx = blockIdx.x;
        if (idx >= P.chunk_n) return;
        uint32_t *const ck = P.wide_ckpt + (uint64_t)idx * WIDE_CKPT_WORDS;  // (only touched when the launch works in two phases)
        if constexpr (PHASE == 1) {
            if (rfl(ck[0]) != 1u) return;  // finished, or handed on, by the first launch
        } else {
            if (P.wide_ckpt_on != 0u && tid == 0) ck[0] = 0u;
        }
        const uint32_t pair = rfl(P.work ? P.work[idx] : P.chunk_first + idx);
        const uint32_t nq = rfl(P.q_len[pair]), mt = rfl(P.t_len[pair]);
        uint32_t       status = ST_PENDING;
        if (nq == 0 || mt == 0)
            status = ST_EMPTY;  // wfa.go:204-206
        else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
            status = ST_TOO_LONG;  // wfa.go:207-209
        else if ((nq > mt ? nq : mt) > WIDE_MAX_LEN || ((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW || (PHASE == 0 && wide_row_hw(nq > mt ? nq : mt) > WH))
            status = ST_REDO_LDS;
        if (status == ST_PENDING) {
            bool bad = stage_pack<64 * NW>(P.blob, P.q_off[pair], nq, lq, tid);
            bad |= stage_pack<64 * NW>(P.blob, P.t_off[pair], mt, lt, tid);
            bool anybad;
            if constexpr (NW > 1) anybad = __syncthreads_or(bad ? 1 : 0) != 0;
            else anybad = __ballot(bad) != 0ull;
            if (anybad) status = ST_REDO_BYTES;  // a byte outside ACGT: the byte-compare path takes the pair
        }
        if (status != ST_PENDING) {
            if (tid == 0) {
                P.pair_meta[idx] = make_uint4(status, 0u, 0u, 0u);
                if (status >= ST_REDO_BYTES) push_redo(P, pair, status);
            }
            return;
        }
        const int n = (int)nq, m = (int)mt, Ak = m - n;
        SeqView<0> sv;
        sv.q = lq, sv.t = lt, sv.n = n, sv.m = m;
        if constexpr (PHASE == 1) {  // the checkpoint's rings (already in this phase's layout: slot = diagonal modulo WIDE_RW)
            const uint4 *const c4 = reinterpret_cast<const uint4 *>(ck + WIDE_CKPT_HDR);
            uint4 *const       r4 = reinterpret_cast<uint4 *>(ring);
            for (uint32_t i = (uint32_t)lane; i < 6u * WIDE_RW / 8u; i += 64u) r4[i] = c4[i];
        } else {  // rings: all absent
            uint4 *const r4 = reinterpret_cast<uint4 *>(ring);
            for (uint32_t i = (uint32_t)tid; i < 6u * WH / 8u; i += 64u * NW) r4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        lds_sync();
        const int KOFF = n - 1 + 32;  // PHASE 0: ring index of diagonal k = k + KOFF (guards of 32 below and above); PHASE 1: k modulo WIDE_RW
        const auto RI = [&](int k) -> uint32_t { return PHASE ? ((uint32_t)k & (uint32_t)(WIDE_RW - 1)) : (uint32_t)(k + KOFF); };
        uint16_t *const arow = reinterpret_cast<uint16_t *>(P.arena + (uint64_t)idx * cap);  // rows, halfwords
        uint32_t *const adir = P.arena + (uint64_t)idx * cap + cap;                          // directory entry i: adir - 4 (i + 1)

        // bands (absolute diagonals, empty = (BIG, -BIG)) of the rows in the rings: M by slot, the I / D rows = the previous row's
        int      blo0 = BIG, blo1 = BIG, blo2 = BIG, blo3 = BIG, bhi0 = -BIG, bhi1 = -BIG, bhi2 = -BIG, bhi3 = -BIG;
        int      plo = BIG, phi = -BIG;
        // (selects, not arrays indexed at run time: those would live in scratch memory)
        const auto get_lo = [&](uint32_t i) { i &= 3u; return i == 0u ? blo0 : i == 1u ? blo1 : i == 2u ? blo2 : blo3; };
        const auto get_hi = [&](uint32_t i) { i &= 3u; return i == 0u ? bhi0 : i == 1u ? bhi1 : i == 2u ? bhi2 : bhi3; };
        uint32_t top = 0u;     // halfwords of rows laid down
        uint32_t cells = 0u;   // census of stored M / I / D words (REC_CELLS)
        bool     overflow = false;
        uint32_t ovf_si = 0u, ovf_why = 0u, ovf_span = 0u;  // (what WFAHIP_WIDE_TRACE prints)
        uint32_t s_final = 0u;
        // semi-global end cell: the first score (ascending) with a hit, the upward scan overriding the downward one
        bool     found = false;
        uint32_t fs = 0u;
        int      fk = 0, fh = 0;
        int      hf = 0;  // extended offset of M[s_final][Ak]
        uint32_t si0 = 0u;
        if constexpr (PHASE == 1) {
            si0 = rfl(ck[1]), top = rfl(ck[2]), cells = rfl(ck[3]), found = rfl(ck[4]) != 0u, fs = rfl(ck[5]), fk = (int)rfl(ck[6]), fh = (int)rfl(ck[7]);
            blo0 = (int)rfl(ck[8]), blo1 = (int)rfl(ck[9]), blo2 = (int)rfl(ck[10]), blo3 = (int)rfl(ck[11]);
            bhi0 = (int)rfl(ck[12]), bhi1 = (int)rfl(ck[13]), bhi2 = (int)rfl(ck[14]), bhi3 = (int)rfl(ck[15]);
            plo = (int)rfl(ck[16]), phi = (int)rfl(ck[17]);
        }

        for (uint32_t si = si0;; si++) {
            const uint32_t s = si * g;
            // ---- the range of next(s) (wfa.go:557-563) and of the seeds
            int lo = BIG, hi = -BIG;
            if (si != 0u) {
                const auto take = [&](int l, int h_) {
                    if (h_ >= l) lo = imin2(lo, l - 1), hi = imax2(hi, h_ + 1);
                };
                if (si >= (uint32_t)DX) take(get_lo(si - DX), get_hi(si - DX));
                if (si >= (uint32_t)DOE) take(get_lo(si - DOE), get_hi(si - DOE));
                take(plo, phi);
                lo = imax2(lo, -(n - 1)), hi = imin2(hi, m - 1);
            }
            const bool seeded = s == 0u || s == x;
            if (seeded) {
                if (glob) lo = imin2(lo, 0), hi = imax2(hi, 0);
                else lo = -(n - 1), hi = m - 1;
            }
            // what has to be rewritten: the range, and what the slots hold of older rows (M[s-4g] in the new M row's slot, the previous I / D rows)
            const uint32_t slot = si & 3u;
            const int ulo = imin2(lo, imin2(get_lo(slot), plo)), uhi = imax2(hi, imax2(get_hi(slot), phi));
            const int W = hi >= lo ? hi - lo + 1 : 0;
            if ((uint64_t)(top + (uint32_t)W + 4u) / 2u + 4ull * (si + 2u) > cap) {
                overflow = true, ovf_si = si, ovf_why = 1u, ovf_span = (uint32_t)W;
                break;
            }
            if (PHASE == 1 && uhi >= ulo && uhi - ulo + 1 > WIDE_RW - 4) {  // the band has outgrown this phase's rings: the ladder takes the pair
                overflow = true, ovf_si = si, ovf_why = 2u, ovf_span = (uint32_t)(uhi - ulo + 1);
                break;
            }
            uint16_t *const Mn = rowM(si), *const Mo = Mn /* M[s-o-e]: the same slot when DOE == 4 */, *const Mx = rowM(si - (uint32_t)DX);
            uint16_t *const Moe = rowM(si - (uint32_t)DOE);
            (void)Mo;
            const bool hasX = si >= (uint32_t)DX, hasO = si >= (uint32_t)DOE, hasE = si >= 1u;
            const bool inplace = (DOE & 3) == 0;  // the new row's slot IS M[s-o-e]'s: its k-1 cell travels in a register

            // ---- pass 1: next + seeds + extend, tile by tile
            int      mlo = BIG, mhi = -BIG, mind = BIG, maxd = -BIG;
            bool     term = false;
            uint32_t ncell = 0u;
            const bool census = P.census != 0u;  // (REC_CELLS: the stored words are only counted when somebody asks)
            uint32_t carryM = 0u, carryI = 0u;  // M[s-o-e][t0 - 1], I[s-e][t0 - 1] as they were before the previous tile overwrote them
            const bool ecs = !glob && !found;   // the end-cell search is on: the wide rows note its candidates while they are computed
            int        cdn = -BIG, cup = BIG;
            bool     single = false;            // the row is one tile of 64 diagonals (PHASE 1): sM / sI / sD = the lane's cells of it
            uint32_t sM = 0u, sI = 0u, sD = 0u;
            // one cell: WF_NEXT from the raw sources (0 = absent), the seeds, WF_EXTEND
            const auto cell = [&](const int k, const bool act, const uint32_t a0, const uint32_t b0, const uint32_t c0, const uint32_t d0, const uint32_t x0,
                                  uint32_t &nM, uint32_t &nI, uint32_t &nD, uint32_t &wd, const bool with_extend = true) {
                nM = nI = nD = wd = 0u;
                if (act && si != 0u) {
                    // rejections: > m (not >=) for I and X sources, offset - k > n for D and X sources (wfa.go:581-588,616-623,651-654)
                    const uint32_t a = (int)a0 > m ? 0u : a0, b = (int)b0 > m ? 0u : b0;
                    const uint32_t c = (int)c0 - k > n ? 0u : c0, d = (int)d0 - k > n ? 0u : d0;
                    const uint32_t xx = ((int)x0 > m || (int)x0 - k > n) ? 0u : x0;
                    const uint32_t mi = umax2(a, b), Isk = mi + umin2(mi, 1u);
                    const uint32_t Dsk = umax2(c, d);
                    const uint32_t x1  = xx + umin2(xx, 1u);
                    const uint32_t Msk = umax2(umax2(Isk, Dsk), x1);
                    const bool fromX = xx != 0u && Msk == x1;  // wfa.go:657-693: the mismatch wins a tie, then the insertion
                    const bool fromI = !fromX && Msk == Isk;
                    // backTrace recomputes the pre-extension offset from the un-rejected sources (wfa.go:766-817)
                    const uint32_t mu = umax2(a0, b0), Iu = mu + umin2(mu, 1u), Du = umax2(c0, d0);
                    const uint32_t Xu = x0 + umin2(x0, 1u);
                    const bool     iext = a < b, dext = c < d;
                    const uint32_t o0   = (fromI && iext) ? Iu : ((!fromX && !fromI && dext) ? Du : umax2(umax2(Iu, Du), Xu));
                    nM = Msk, nI = Isk, nD = Dsk;
                    wd = Msk != 0u ? blk_word(o0, iext, dext, fromX, fromI) : 0u;
                }
                if (act && seeded && nM == 0u) {  // seeds of initComponents that belong to this score (Set = last write wins: next()'s cell stays)
                    const uint32_t sw = seed_word<0>(sv, k, s, x, glob);
                    if (sw != 0u) nM = sw >> TAG_BITS, wd = (sw & TAG_MASK) == TAG_MATCH ? BLK_SEED_MATCH : BLK_SEED_MISMATCH;
                }
                // WF_EXTEND (wfa.go:394-455): only cells with 0 < v < n and h < m
                if (with_extend && nM != 0u) {
                    const int h = (int)nM, v = h - k;
                    if (v > 0 && v < n && h < m) nM += (uint32_t)sv.lcp(v, h);
                }
            };
            const auto cell_extend = [&](const int k, uint32_t &nM) {
                if (nM != 0u) {
                    const int h = (int)nM, v = h - k;
                    if (v > 0 && v < n && h < m) nM += (uint32_t)sv.lcp(v, h);
                }
            };
            // ... and what the row's reductions take from a cell (rows of more than one tile; a row of one tile answers them with ballots below)
            const auto cell_stats = [&](const int k, const uint32_t nM, const uint32_t nI, const uint32_t nD) {
                if (nM != 0u) {
                    mlo = imin2(mlo, k), mhi = imax2(mhi, k);
                    const int h = (int)nM, v = h - k;
                    if (k == Ak && h >= m) term = true, hf = h;  // wfa.go:235-239
                    if (!(v < 0 || v >= n || h >= m)) {          // wfa.go:474-494
                        const int dd = imax2(m - h, n - v);
                        mind = imin2(mind, dd), maxd = imax2(maxd, dd);
                    }
                    if (PHASE == 0 && ecs) {  // the cells the end-cell scans stop at or hit (wfa.go:270-375): the nearest one to diagonal Ak on either side
                        if (v <= 0 || v > n || h > m || (v == n && h >= n) || (h == m && v >= m)) {
                            if (k <= Ak) cdn = imax2(cdn, k);
                            else cup = imin2(cup, k);
                        }
                    }
                }
                if (census) ncell += (nM != 0u ? 1u : 0u) + (nI != 0u ? 1u : 0u) + (nD != 0u ? 1u : 0u);
            };
            // the lane's four consecutive cells of a wide row after WF_NEXT: WF_EXTEND and what the row's reductions take from them (cell_stats(), per lane
            // instead of per cell where the cells' order allows) -- shared by the packed and the exact path of the rounds
            const auto finish4 = [&](const int k0, uint32_t (&nM)[4]) {
                // WF_EXTEND (wfa.go:394-455), the first 16-base window of the lane's four cells side by side and without a branch; the
                // few cells whose window matches throughout go on in the loop
                uint32_t more = 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int  k = k0 + u, h = (int)nM[u], v = h - k;
                    const bool ext = h != 0 && v > 0 && v < n && h < m;
                    const int  vv = ext ? v : 0, hh = ext ? h : 0;  // (a cell that does not extend reads the sequences' first words)
                    const uint32_t xw  = SeqView<0>::win16(lq, vv) ^ SeqView<0>::win16(lt, hh);
                    const int      rem = imin2(n - vv, m - hh), tot = xw != 0u ? (int)(__builtin_ctz(xw) >> 1) : 16;
                    nM[u] = (uint32_t)(h + (ext ? imin2(tot, rem) : 0));
                    more |= (ext && xw == 0u && rem > 16) ? (1u << u) : 0u;
                }
                if (__ballot(more != 0u) != 0ull) {
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if ((more >> u) & 1u) {
                            const int h = (int)nM[u], v = h - (k0 + u);
                            nM[u] += (uint32_t)sv.lcp(v, h);
                        }
                }
                // what the row's reductions take from the four cells (cell_stats(), per lane instead of per cell where the cells' order allows)
                {
                    const bool z0 = nM[0] != 0u, z1 = nM[1] != 0u, z2 = nM[2] != 0u, z3 = nM[3] != 0u;
                    mlo = imin2(mlo, z0 ? k0 : (z1 ? k0 + 1 : (z2 ? k0 + 2 : (z3 ? k0 + 3 : BIG))));
                    mhi = imax2(mhi, z3 ? k0 + 3 : (z2 ? k0 + 2 : (z1 ? k0 + 1 : (z0 ? k0 : -BIG))));
                    const int ua = Ak - k0;  // the lane that holds the final diagonal (wfa.go:235-239)
                    if (ua >= 0 && ua < 4) {
                        const int hA = (int)(ua == 0 ? nM[0] : (ua == 1 ? nM[1] : (ua == 2 ? nM[2] : nM[3])));
                        if (hA >= m) term = true, hf = hA;
                    }
                    bool edge = false;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int  h = (int)nM[u], v = h - (k0 + u);
                        const bool valid = h != 0 && (uint32_t)v < (uint32_t)n && h < m;  // wfa.go:474-494 (v < 0 wraps)
                        const int  dd = imax2(m - h, n - v);
                        mind = imin2(mind, valid ? dd : BIG), maxd = imax2(maxd, valid ? dd : -BIG);
                        edge |= h != 0 && (!valid || v == 0);
                    }
                    // a stop or a hit of the end-cell scans is a cell at an end of a sequence: none in the interior of most rows
                    if (ecs && __ballot(edge) != 0ull) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int k = k0 + u, h = (int)nM[u], v = h - k;
                            if (h != 0 && (v <= 0 || v > n || h > m || (v == n && h >= n) || (h == m && v >= m))) {
                                if (k <= Ak) cdn = imax2(cdn, k);
                                else cup = imin2(cup, k);
                            }
                        }
                    }
                }
            };
            if (uhi >= ulo && PHASE == 0) {
                // ---- the wide rows: rounds of 256 diagonals, FOUR CONSECUTIVE DIAGONALS PER LANE.  A lane's cells of a row are one 8-byte LDS word
                // (the round starts on a ring index that is a multiple of four), three of a cell's five sources are the lane's own registers, and the
                // row's halfwords go to the arena eight bytes at a time (the row's first halfword is laid so that a lane's four are aligned)
                const int  t_first = ulo - (int)(RI(ulo) & 3u);
                const bool pk_ok   = !seeded && si != 0u && !census && P.wide_exact == 0u;
                uint32_t   pk_one  = 0x00010001u;
                asm volatile("" : "+v"(pk_one));
                {
                    const uint32_t want = (uint32_t)(lo - t_first) & 3u;  // top + (k0 - lo) a multiple of four for every lane's first diagonal k0
                    top += (want - top) & 3u;
                }
                carryM = hasO ? Moe[RI(t_first - 1)] : 0u;
                carryI = hasE ? rowI[RI(t_first - 1)] : 0u;
                const auto ld4 = [](const uint16_t *row, uint32_t r0, uint32_t (&v)[4]) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(row + r0);
                    v[0] = w.x & 0xFFFFu, v[1] = w.x >> 16, v[2] = w.y & 0xFFFFu, v[3] = w.y >> 16;
                };
                const auto st4 = [](uint16_t *row, uint32_t r0, const uint32_t (&v)[4]) {
                    *reinterpret_cast<uint2 *>(row + r0) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
                };
                // NW waves: each takes a quarter (256 diagonals) of every round.  The rows are updated in place, and the only cells one wave reads that another
                // overwrites are the two on either side of a quarter's edge: they are saved before the row, one barrier, and the waves run their quarters without
                // waiting for one another (with a barrier per round the row took as long as the SUM of every round's slowest wave -- the exact-path quarters at its two ends)
                uint32_t *const bnd = scr + 64;  // per quarter q: {M[s-o-e], I[s-e] at its first diagonal - 1; M[s-o-e], D[s-e] at its first diagonal}
                if constexpr (NW > 1) {
                    const int nq = (uhi - t_first) / 256 + 1;
                    if (tid <= nq) {
                        const int kf = t_first + 256 * tid;
                        bnd[4 * tid + 0] = (hasO && kf - 1 <= uhi) ? Moe[RI(kf - 1)] : 0u, bnd[4 * tid + 1] = (hasE && kf - 1 <= uhi) ? rowI[RI(kf - 1)] : 0u;
                        bnd[4 * tid + 2] = (hasO && kf <= uhi) ? Moe[RI(kf)] : 0u, bnd[4 * tid + 3] = (hasE && kf <= uhi) ? rowD[RI(kf)] : 0u;
                    }
                    lds_sync();
                }
                const auto wave_sync = [] {  // (the wave's own reads before its own writes)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                const int wr = wv;  // the quarter of a round this wave takes (rotating it with the score, so that the exact-path quarters at the row's two ends move from SIMD to SIMD, changed nothing: 159.5 against 158.3 ms)
                for (int t0 = t_first; t0 <= uhi; t0 += 256 * NW) {
                    const int      k0  = t0 + 256 * wr + 4 * lane;
                    const uint32_t r0  = RI(k0);
                    const bool     lin = k0 <= uhi;  // (diagonals of the lane's four beyond uhi, or below ulo in the first round: no cell in any row)
                    uint32_t oM[4] = {0u, 0u, 0u, 0u}, oI[4] = {0u, 0u, 0u, 0u}, oD[4] = {0u, 0u, 0u, 0u}, oX[4] = {0u, 0u, 0u, 0u};
                    uint32_t aL = 0u, bL = 0u, cR = 0u, dR = 0u;
                    // a round wholly outside the new row's range only clears what the new M row's slot still holds of M[s-4g] (the steps after
                    // wf-adaptive's first cut: the wide old rows leave the rings one slot a step) -- every source row is empty there
                    const bool ract = t0 + 256 * wr + 255 >= lo && t0 + 256 * wr <= hi;
                    if (lin && ract) {
                        if (hasO) ld4(Moe, r0, oM), aL = Moe[r0 - 1u], cR = Moe[r0 + 4u];
                        if (hasE) ld4(rowI, r0, oI), ld4(rowD, r0, oD), bL = rowI[r0 - 1u], dR = rowD[r0 + 4u];
                        if (hasX) ld4(Mx, r0, oX);
                    }
                    if constexpr (NW == 1) {
                        if (lane == 0) aL = carryM, bL = carryI;  // (what the previous round overwrote)
                        carryM = rfl((uint32_t)__builtin_amdgcn_readlane((int)oM[3], 63));
                        carryI = rfl((uint32_t)__builtin_amdgcn_readlane((int)oI[3], 63));
                    } else {
                        const int q = (t0 - t_first) / 256 + wr;
                        if (lane == 0) aL = bnd[4 * q], bL = bnd[4 * q + 1];
                        if (lane == 63) cR = bnd[4 * q + 6], dR = bnd[4 * q + 7];
                    }
                    wave_sync();  // (every read of the round before its first write: the rows are updated in place)
                    if (!ract) {
                        if (lin) *reinterpret_cast<uint2 *>(Mn + r0) = make_uint2(0u, 0u);
                        continue;
                    }
                    // ---- the interior of a wide row: every diagonal of the wave's round inside [lo, hi], no seeds, no source that next() would
                    // reject -- WF_NEXT on two diagonals per register; the I / D rows and the backtrace words are stored as they come out
                    if (pk_ok && t0 + 256 * wr >= lo && t0 + 256 * wr + 255 <= hi) {
                        const uint32_t pM0 = oM[0] | (oM[1] << 16), pM1 = oM[2] | (oM[3] << 16), pI0 = oI[0] | (oI[1] << 16), pI1 = oI[2] | (oI[3] << 16);
                        const uint32_t pD0 = oD[0] | (oD[1] << 16), pD1 = oD[2] | (oD[3] << 16), pX0 = oX[0] | (oX[1] << 16), pX1 = oX[2] | (oX[3] << 16);
                        // the sources of diagonals (k0, k0+1) and (k0+2, k0+3): k-1 of the M[s-o-e] / I rows, k+1 of the M[s-o-e] / D rows
                        const uint32_t a0p = (pM0 << 16) | aL, a1p = __builtin_amdgcn_alignbit(pM1, pM0, 16), c1p = (pM1 >> 16) | (cR << 16);
                        const uint32_t b0p = (pI0 << 16) | bL, b1p = __builtin_amdgcn_alignbit(pI1, pI0, 16);
                        const uint32_t d0p = __builtin_amdgcn_alignbit(pD1, pD0, 16), d1p = (pD1 >> 16) | (dR << 16);
                        // rejections (wfa.go:581-588,616-623,651-654): an offset > m, an offset - k > n -- none in the whole round, or the round takes the exact path
                        const wide_us2 hi_m = wide_max(wide_max(wide_max(wide_pk(a0p), wide_pk(a1p)), wide_max(wide_pk(b0p), wide_pk(b1p))), wide_max(wide_pk(pX0), wide_pk(pX1)));
                        const uint32_t nk0 = (uint32_t)(n + k0) | ((uint32_t)(n + k0 + 1) << 16), nk1 = (uint32_t)(n + k0 + 2) | ((uint32_t)(n + k0 + 3) << 16);
                        const wide_us2 over = __builtin_elementwise_sub_sat(hi_m, wide_pk((uint32_t)m * 0x10001u)) |
                                              __builtin_elementwise_sub_sat(wide_max(wide_max(wide_pk(a1p), wide_pk(d0p)), wide_pk(pX0)), wide_pk(nk0)) |
                                              __builtin_elementwise_sub_sat(wide_max(wide_max(wide_pk(c1p), wide_pk(d1p)), wide_pk(pX1)), wide_pk(nk1));
                        if (__ballot(lin && wide_u32(over) != 0u) == 0ull) {
                            uint32_t M0, M1, I0, I1, D0, D1, W0, W1;
                            wide_next2(wide_pk(a0p), wide_pk(b0p), wide_pk(a1p) /* = (M[k0+1], M[k0+2]) */, wide_pk(d0p), wide_pk(pX0), wide_pk(pk_one), M0, I0, D0, W0);
                            wide_next2(wide_pk(a1p), wide_pk(b1p), wide_pk(c1p), wide_pk(d1p), wide_pk(pX1), wide_pk(pk_one), M1, I1, D1, W1);
                            uint32_t nM[4] = {M0 & 0xFFFFu, M0 >> 16, M1 & 0xFFFFu, M1 >> 16};
                            finish4(k0, nM);
                            if (lin) {
                                st4(Mn, r0, nM);
                                *reinterpret_cast<uint2 *>(rowI + r0) = make_uint2(I0, I1);
                                *reinterpret_cast<uint2 *>(rowD + r0) = make_uint2(D0, D1);
                                *reinterpret_cast<uint2 *>(arow + ((int64_t)top + (int64_t)(k0 - lo))) = make_uint2(W0, W1);
                            }
                            if constexpr (NW == 1) lds_sync();
                            continue;
                        }
                    }
                    uint32_t nM[4], nI[4], nD[4], wd[4];
                    bool     act[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int k = k0 + u;
                        act[u] = k >= lo && k <= hi;
                        cell(k, act[u], u == 0 ? aL : oM[u - 1], u == 0 ? bL : oI[u - 1], u == 3 ? cR : oM[u + 1], u == 3 ? dR : oD[u + 1], oX[u], nM[u], nI[u], nD[u], wd[u], false);
                        if (census) ncell += (nM[u] != 0u ? 1u : 0u) + (nI[u] != 0u ? 1u : 0u) + (nD[u] != 0u ? 1u : 0u);
                    }
                    finish4(k0, nM);
                    if (lin) {
                        st4(Mn, r0, nM), st4(rowI, r0, nI), st4(rowD, r0, nD);
                        uint16_t *const ap = arow + ((int64_t)top + (int64_t)(k0 - lo));  // (only dereferenced where act)
                        if (act[0] && act[3]) {
                            *reinterpret_cast<uint2 *>(ap) = make_uint2(wd[0] | (wd[1] << 16), wd[2] | (wd[3] << 16));
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (act[u]) ap[u] = (uint16_t)wd[u];
                        }
                    }
                    if constexpr (NW == 1) lds_sync();  // (several waves: the next round touches other diagonals, and the carry has its own slots)
                }
                if constexpr (NW > 1) lds_sync();
            } else if (uhi >= ulo) {
                // ---- the narrow rows (PHASE 1): tiles of 64 diagonals, a lane per diagonal; a row of ONE tile -- nearly all of them -- stays in
                // registers for what follows
                single = uhi - ulo < 64;
                const int t_first = ulo;
                carryM = hasO ? Moe[RI(t_first - 1)] : 0u;
                carryI = hasE ? rowI[RI(t_first - 1)] : 0u;
                for (int t0 = t_first; t0 <= uhi; t0 += 64) {
                    const int      k  = t0 + lane;
                    const uint32_t ri = RI(k), rim = RI(k - 1), rip = RI(k + 1);
                    const bool     in = k <= uhi, act = in && k >= lo && k <= hi;
                    uint32_t       a0 = 0u, b0 = 0u, c0 = 0u, d0 = 0u, x0 = 0u, ownI = 0u, ownM = 0u;
                    if (in) {
                        // (the first lane: what the previous tile overwrote travels in carryM / carryI)
                        if (hasO) a0 = (lane == 0 && inplace) ? carryM : Moe[rim], c0 = Moe[rip], ownM = Moe[ri];
                        if (hasE) b0 = lane == 0 ? carryI : rowI[rim], d0 = rowD[rip], ownI = rowI[ri];
                        if (hasX) x0 = Mx[ri];
                    }
                    if (!single) {
                        carryM = rfl((uint32_t)__builtin_amdgcn_readlane((int)ownM, 63));
                        carryI = rfl((uint32_t)__builtin_amdgcn_readlane((int)ownI, 63));
                    }
                    lds_sync();  // (every read of the tile before its first write: the rows are updated in place)
                    uint32_t nM, nI, nD, wd;
                    cell(k, act, a0, b0, c0, d0, x0, nM, nI, nD, wd);
                    if (single) sM = nM, sI = nI, sD = nD;
                    else cell_stats(k, nM, nI, nD);
                    if (in) Mn[ri] = (uint16_t)nM, rowI[ri] = (uint16_t)nI, rowD[ri] = (uint16_t)nD;
                    if (act) arow[top + (uint32_t)(k - lo)] = (uint16_t)wd;
                    lds_sync();
                }
            }
            int nlo, nhi;  // the surviving band (I and D only hold cells where M does)
            if (single) {
                // ---- a row of one tile: the lane of diagonal ulo + lane holds its cells; every reduction is a ballot and a scalar bit scan
                const auto ctz64 = [](uint64_t b) { return (int)__builtin_ctzll(b); };
                const auto top64 = [](uint64_t b) { return 63 - (int)__builtin_clzll(b); };
                const int  k = ulo + lane;
                const int  h = (int)sM, v = h - k;
                const uint64_t bm = __ballot(sM != 0u);
                if (bm != 0ull) mlo = ulo + ctz64(bm), mhi = ulo + top64(bm);
                term = __ballot(sM != 0u && k == Ak && h >= m) != 0ull;  // wfa.go:235-239
                if (term) hf = __builtin_amdgcn_readlane((int)sM, (int)rfl((uint32_t)(Ak - ulo)));
                nlo = mlo, nhi = mhi;
                uint32_t cnt = 0u;
                if (census) cnt = (uint32_t)__popcll(bm) + (uint32_t)__popcll(__ballot(sI != 0u)) + (uint32_t)__popcll(__ballot(sD != 0u));
                const bool     valid = sM != 0u && !(v < 0 || v >= n || h >= m);  // wfa.go:474-494
                const int      dd    = imax2(m - h, n - v);
                const uint64_t bv    = __ballot(valid);
                if (!term && adaptive && bm != 0ull && (mhi - mlo + 1) >= minwf && bv != 0ull) {
                    const int      thr = wave_min(valid ? dd : BIG) + mdd;
                    const uint64_t bo  = __ballot(valid && dd <= thr);
                    if (bo != bv) {  // ---- reduce (wfa.go:496-537): some distance fails
                        const int      f     = ctz64(bo);
                        const uint64_t below = bv & ((1ull << f) - 1ull);  // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                        nlo = below != 0ull ? ulo + top64(below) + 1 : mlo;
                        nhi = ulo + top64(bo);  // wfa.go:517-524
                        const bool del = k >= mlo && k <= mhi && (k < nlo || k > nhi);  // wfa.go:526-535
                        if (census) cnt -= (uint32_t)__popcll(__ballot(del && sM != 0u)) + (uint32_t)__popcll(__ballot(del && sI != 0u)) + (uint32_t)__popcll(__ballot(del && sD != 0u));
                        if (del) {
                            const uint32_t ri = RI(k);
                            Mn[ri] = 0, rowI[ri] = 0, rowD[ri] = 0, sM = 0u;
                        }
                        lds_sync();
                    }
                }
                cells += cnt;
                // ---- semi-global end cell (wfa.go:270-375) of this score, on the row as it stays
                // (a stop or a hit is a cell at an end of a sequence: v <= 0, v >= n or h >= m -- none for most of a narrow phase's rows)
                if (!glob && !found && nhi >= nlo && __ballot(sM != 0u && (v <= 0 || v >= n || h >= m)) != 0ull) {
                    const bool     inb  = sM != 0u && k >= nlo && k <= nhi;
                    const bool     stop = v <= 0 || v > n || h > m;
                    const bool     hit  = !stop && ((v == n && h >= n) || (h == m && v >= m));
                    const uint64_t bs = __ballot(inb && (stop || hit)), bh = __ballot(inb && hit);
                    const int      a  = Ak - ulo;  // the lane of diagonal Ak: the scan down starts there, the scan up one above
                    const uint64_t mD = a >= 63 ? ~0ull : (a < 0 ? 0ull : ((2ull << a) - 1ull));
                    const uint64_t cd = bs & mD, cu = bs & ~mD;
                    if (cd != 0ull) {
                        const int l = top64(cd);
                        if ((bh >> l) & 1ull) found = true, fs = s, fk = ulo + l, fh = __builtin_amdgcn_readlane((int)sM, (int)rfl((uint32_t)l));
                    }
                    if (cu != 0ull) {
                        const int l = ctz64(cu);
                        if ((bh >> l) & 1ull) found = true, fs = s, fk = ulo + l, fh = __builtin_amdgcn_readlane((int)sM, (int)rfl((uint32_t)l));
                    }
                }
            } else {
            mlo = wave_min(mlo), mhi = wave_max(mhi), mind = wave_min(mind), maxd = wave_max(maxd);
            term = __ballot(term) != 0ull;
            if (term) hf = wave_max(hf);
            if (PHASE == 0 && ecs) cdn = wave_max(cdn), cup = wave_min(cup);
            if constexpr (NW > 1) {
                const int mine[7] = {mlo, mhi, mind, maxd, term ? hf : 0, cdn, cup};
                int       all[NW * 7];
                xchg(mine, all, 7);
                mlo = BIG, mhi = -BIG, mind = BIG, maxd = -BIG, hf = 0, cdn = -BIG, cup = BIG;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    mlo = imin2(mlo, all[7 * w]), mhi = imax2(mhi, all[7 * w + 1]), mind = imin2(mind, all[7 * w + 2]), maxd = imax2(maxd, all[7 * w + 3]), hf = imax2(hf, all[7 * w + 4]);
                    cdn = imax2(cdn, all[7 * w + 5]), cup = imin2(cup, all[7 * w + 6]);
                }
                term = hf != 0;
            }
            nlo = mlo, nhi = mhi;
            if (!term && adaptive && mhi >= mlo && (mhi - mlo + 1) >= minwf && mind != BIG && maxd - mind > mdd) {
                // ---- reduce (wfa.go:496-537): some distance fails
                const int thr = mind + mdd;
                int first_ok = BIG, last_ok = -BIG;
                for (int t0 = mlo + 64 * wv; t0 <= mhi; t0 += 64 * NW) {
                    const int k = t0 + lane;
                    if (k <= mhi) {
                        const int h = (int)Mn[RI(k)], v = h - k;
                        if (h != 0 && !(v < 0 || v >= n || h >= m) && imax2(m - h, n - v) <= thr) first_ok = imin2(first_ok, k), last_ok = imax2(last_ok, k);
                    }
                }
                first_ok = wave_min(first_ok), last_ok = wave_max(last_ok);
                if constexpr (NW > 1) {
                    const int mine[2] = {first_ok, last_ok};
                    int       all[NW * 2];
                    xchg(mine, all, 2);
#pragma unroll
                    for (int w = 0; w < NW; w++) first_ok = imin2(first_ok, all[2 * w]), last_ok = imax2(last_ok, all[2 * w + 1]);
                }
                int lead = -BIG;  // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                for (int t0 = mlo + 64 * wv; t0 < first_ok && t0 <= mhi; t0 += 64 * NW) {
                    const int k = t0 + lane;
                    if (k < first_ok && k <= mhi) {
                        const int h = (int)Mn[RI(k)], v = h - k;
                        if (h != 0 && !(v < 0 || v >= n || h >= m)) lead = imax2(lead, k);
                    }
                }
                lead = wave_max(lead);
                if constexpr (NW > 1) {
                    const int mine[1] = {lead};
                    int       all[NW];
                    xchg(mine, all, 1);
#pragma unroll
                    for (int w = 0; w < NW; w++) lead = imax2(lead, all[w]);
                }
                nlo  = lead != -BIG ? lead + 1 : mlo;
                nhi  = last_ok;  // wfa.go:517-524
                // wfa.go:526-535 deletes k outside [_lo, _hi] in M, I and D
                for (int t0 = mlo + 64 * wv; t0 <= mhi; t0 += 64 * NW) {
                    const int k = t0 + lane;
                    if (k <= mhi && (k < nlo || k > nhi)) {
                        const uint32_t ri = RI(k);
                        if (census) ncell -= (Mn[ri] != 0 ? 1u : 0u) + (rowI[ri] != 0 ? 1u : 0u) + (rowD[ri] != 0 ? 1u : 0u);
                        Mn[ri] = 0, rowI[ri] = 0, rowD[ri] = 0;
                    }
                }
                lds_sync();
            }
            // ---- semi-global end cell (wfa.go:270-375) of this score, on the row as it stays
            bool scan = !glob && !found && nhi >= nlo;
            if (PHASE == 0 && scan && cdn <= nhi && cup >= nlo) {
                // the wide rows noted the nearest candidate on either side of Ak while they were computed; unless wf-adaptive has just cut it off
                // (then the row is scanned as below) it is the cell the reference's scan ends at
                scan = false;
                const auto at = [&](const int k) {
                    const int h = (int)rfl((uint32_t)Mn[RI(k)]), v = h - k;
                    const bool stop = v <= 0 || v > n || h > m;
                    if (!stop && ((v == n && h >= n) || (h == m && v >= m))) found = true, fs = s, fk = k, fh = h;
                };
                if (cdn >= nlo) at(cdn);
                if (cup <= nhi) at(cup);  // (the scan up overrides the scan down)
            }
            if (scan) {
                uint32_t keyD = 0xFFFFFFFFu, keyU = 0xFFFFFFFFu;
                int      hD = 0, hU = 0;
                for (int t0 = nlo + 64 * wv; t0 <= nhi; t0 += 64 * NW) {
                    const int k = t0 + lane;
                    if (k <= nhi) {
                        const int h = (int)Mn[RI(k)], v = h - k;
                        if (h != 0) {
                            const bool stop = v <= 0 || v > n || h > m;
                            const bool hit  = !stop && ((v == n && h >= n) || (h == m && v >= m));
                            if (stop || hit) {
                                if (k <= Ak) {
                                    const uint32_t key = ((uint32_t)(Ak - k) << 1) | (hit ? 0u : 1u);
                                    if (key < keyD) keyD = key, hD = h;
                                } else {
                                    const uint32_t key = ((uint32_t)(k - Ak - 1) << 1) | (hit ? 0u : 1u);
                                    if (key < keyU) keyU = key, hU = h;
                                }
                            }
                        }
                    }
                }
                uint32_t kD = (uint32_t)wave_min((int)(keyD ^ 0x80000000u)) ^ 0x80000000u;  // unsigned min via signed min
                uint32_t kU = (uint32_t)wave_min((int)(keyU ^ 0x80000000u)) ^ 0x80000000u;
                // (the offset of the winning cell: the lane that holds the winning key)
                int wD = wave_max(keyD == kD && kD != 0xFFFFFFFFu ? hD : 0), wU = wave_max(keyU == kU && kU != 0xFFFFFFFFu ? hU : 0);
                if constexpr (NW > 1) {
                    const int mine[4] = {(int)kD, wD, (int)kU, wU};
                    int       all[NW * 4];
                    xchg(mine, all, 4);
                    kD = kU = 0xFFFFFFFFu, wD = wU = 0;
#pragma unroll
                    for (int w = 0; w < NW; w++) {  // (a key is a diagonal: no two waves hold the same one, save the empty key)
                        if ((uint32_t)all[4 * w] < kD) kD = (uint32_t)all[4 * w], wD = all[4 * w + 1];
                        if ((uint32_t)all[4 * w + 2] < kU) kU = (uint32_t)all[4 * w + 2], wU = all[4 * w + 3];
                    }
                }
                if (kD != 0xFFFFFFFFu && (kD & 1u) == 0u) found = true, fs = s, fk = Ak - (int)(kD >> 1), fh = wD;
                if (kU != 0xFFFFFFFFu && (kU & 1u) == 0u) found = true, fs = s, fk = Ak + 1 + (int)(kU >> 1), fh = wU;
            }
            if (census) {
                uint32_t c = ncell;  // (a wave sum: 64 lanes x at most 3 x tiles)
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
                if constexpr (NW > 1) {
                    const int mine[1] = {(int)c};
                    int       all[NW];
                    xchg(mine, all, 1);
                    c = 0u;
#pragma unroll
                    for (int w = 0; w < NW; w++) c += (uint32_t)all[w];
                }
                cells += c;
            }
            }  // (rows of more than one tile)
            // ---- the row's directory entry
            if (tid == 0) {
                const bool any = nhi >= nlo;
                *reinterpret_cast<uint4 *>(adir - 4u * (si + 1u)) = make_uint4(any ? top + (uint32_t)(nlo - lo) : 0u, any ? (uint32_t)nlo : 0u, any ? (uint32_t)(nhi - nlo + 1) : 0u, 0u);
            }
            top += ((uint32_t)W + 1u) & ~1u;
            plo = nhi >= nlo ? nlo : BIG, phi = nhi >= nlo ? nhi : -BIG;
            if (slot == 0u) blo0 = plo, bhi0 = phi;
            if (slot == 1u) blo1 = plo, bhi1 = phi;
            if (slot == 2u) blo2 = plo, bhi2 = phi;
            if (slot == 3u) blo3 = plo, bhi3 = phi;
            if (term) {
                s_final = s;
                break;
            }
            if constexpr (PHASE == 0) {
                if (P.wide_ckpt_on != 0u && s >= x) {  // (score x seeds every diagonal once more, wfa.go:163-183 -- however narrow an early cut left the rows)
                    const int clo = imin2(imin2(blo0, blo1), imin2(imin2(blo2, blo3), plo)), chi = imax2(imax2(bhi0, bhi1), imax2(imax2(bhi2, bhi3), phi));
                    if (chi >= clo && chi - clo + 1 <= WIDE_NARROW) {
                        // checkpoint: the rings' live part in PHASE 1's layout (slot = diagonal modulo WIDE_RW; a thread writes 16 bytes: the eight
                        // slots' diagonals of the live span, absent elsewhere), then the loop's state
                        uint4 *const c4 = reinterpret_cast<uint4 *>(ck + WIDE_CKPT_HDR);
                        for (uint32_t i = (uint32_t)tid; i < 6u * WIDE_RW / 8u; i += 64u * NW) {
                            const uint32_t r = i / (uint32_t)(WIDE_RW / 8), p0 = (i % (uint32_t)(WIDE_RW / 8)) * 8u;
                            uint32_t       hw[8];
#pragma unroll
                            for (uint32_t j = 0; j < 8u; j++) {
                                const int k = clo + (int)((p0 + j - (uint32_t)clo) & (uint32_t)(WIDE_RW - 1));  // the diagonal of slot p0 + j at or above clo
                                hw[j] = k <= chi ? (uint32_t)ring[r * WH + (uint32_t)(k + KOFF)] : 0u;
                            }
                            c4[i] = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
                        }
                        if (tid == 0) {
                            ck[1] = si + 1u, ck[2] = top, ck[3] = cells, ck[4] = found ? 1u : 0u, ck[5] = fs, ck[6] = (uint32_t)fk, ck[7] = (uint32_t)fh;
                            ck[8] = (uint32_t)blo0, ck[9] = (uint32_t)blo1, ck[10] = (uint32_t)blo2, ck[11] = (uint32_t)blo3;
                            ck[12] = (uint32_t)bhi0, ck[13] = (uint32_t)bhi1, ck[14] = (uint32_t)bhi2, ck[15] = (uint32_t)bhi3;
                            ck[16] = (uint32_t)plo, ck[17] = (uint32_t)phi;
                            ck[0] = 1u;
                            P.pair_meta[idx] = make_uint4(ST_PENDING, 0u, 0u, 0u);  // (the second launch writes the pair's record)
                        }
                        return;
                    }
                }
            }
        }
        if (overflow) {
            if (tid == 0) {
                P.pair_meta[idx] = make_uint4(ST_REDO_ARENA, ovf_si, ovf_why + 10u * (uint32_t)PHASE, ovf_span);
                push_redo(P, pair, ST_REDO_ARENA);
            }
            return;
        }
        if (tid == 0) {
            const uint32_t bs = (glob || !found) ? s_final : fs;
            const int      bk = (glob || !found) ? Ak : fk, bh = (glob || !found) ? hf : fh;
            P.pair_meta[idx] = make_uint4(ST_OK, bs, (uint32_t)bh | ((uint32_t)(bk + (int)WIDE_KBIAS) << 16), P.census ? cells : 0u);
        }
    }
}

}  // namespace wfa
