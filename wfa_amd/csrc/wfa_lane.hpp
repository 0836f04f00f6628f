// wfa_lane.hpp -- kernel F: ONE LANE PER PAIR, for short reads (at most 240 bases).
//
// The sub-wave kernels give a pair 8 lanes x 4 diagonals and make a wave step through eight pairs together; a 150-base
// pair at 2 % error lives eleven score steps with ~8 cells each, so most of a wave's time is the start and the end of
// pairs and the lanes of cells that do not exist (profiles/r03_c2_*: 713 vector instructions per pair, 60 % of the
// wave-cycles waiting, 2.1 wave-steps per pair against 1.25).  Here every lane runs its own pair from the first score to
// the last -- 64 pairs per wave, a "generation" -- cell by cell in ascending diagonal order:
//   * the last four M rows and the last I and D rows live in LDS as BYTES (an offset is at most 240; 0 = absent), one
//     32-slot ring per row (slot = diagonal & 31), 192 bytes per lane; the row of score step i replaces M[i-4] IN PLACE
//     and I / D replace themselves in place: a cell needs M[i-4][k-1], M[i-4][k+1], I[k-1] and D[k+1], so in ascending
//     order the only values overwritten too early are the two "k-1" ones, and those are carried in registers from the
//     cell before.  Every slot outside the kept band of the row it belongs to holds ZERO (the rings start zeroed, a step
//     writes every cell of its range, what wf-adaptive drops is zeroed, and a row's range covers the kept bands of its
//     sources +-1 -- so it overwrites all that an older occupant of its ring kept): the loads need no range checks;
//   * both sequences 2-bit packed in LDS (2 x lds_seq_words words per lane), packed by the lane itself from the bytes of
//     its pair (lane_pack_seq: all 16-byte chunks of a sequence in one round of loads; option lane_pack = 0: from the
//     slots of wfa_prepack_kernel).  A wave's first generation is entries 64 x its index, the queue hands out the rest
//     (one atomic per generation);
//   * every cell by the EXACT rules of next() -- rejections (> m, offset - k > n), the k-range clamp, mismatch-wins ties,
//     backTrace's unbounded recomputation of the pre-extension offset (wfa.go:549-700,766-817): the same formulas as the
//     exact path of wfa_blk_kernel -- then WF_EXTEND 32 bases a round (three words of each sequence in one round trip to
//     LDS; a cell with a longer run keeps its lane for another round while the other lanes go on to their next cells, so
//     the wave pays the longest SUM of rounds of a row, not every cell's longest run), termination, and the band /
//     wf-adaptive of reduce() (wfa.go:461-540): the distances are collected as the cells are stored, the leading / trailing
//     failures found in one more pass over the row when a distance fails;
//   * one 16-bit backtrace word per cell (blk_word: pre-extension offset + the four decisions) straight to the pair's
//     arena slot, rows of 32 halfwords (CompactView fmt 8), and pair_meta for wfa_backtrace_kernel -- unchanged behind it.
// A lane's stride in LDS is odd (49 + 2 x lds_seq_words words: 73 for 150-base reads -- eight waves per CU), so the 64
// lanes of an access at the same ring slot fall on different banks.
// A pair whose row would span more than 30 diagonals (32 slots less the k-1 and k+1 a cell reads), or which runs out of
// arena rows, is handed on (ST_REDO_BAND / ST_REDO_ARENA) to the sub-wave kernels like any pair that leaves a window.
// What bounds it (profiles/r03_c2_*; KERNELS.md 4b): instructions.  A generation is ~25 000 wave instructions, a wave
// issues one per 4.25 cycles (a lone wave: 45 us of issue + 39 % of its cycles in s_waitcnt = ~100 us for 64 x 150-base
// pairs, whether the launch holds a thousand pairs or a hundred thousand), and the two waves per SIMD that LDS allows share
// the SIMD's issue slots.  Measured and dropped: two cells per round side by side (more instructions per cell than the
// interleaved chains win back), the row collected in LDS and stored as 16-byte quarters, the backtrace inside the kernel.
#pragma once
#include "wfa_device.hpp"
#include "wfa_blk.hpp"  // prepack_word(), prepack_dword()

namespace wfa {

constexpr int LN_W          = 30;   // diagonals a row may span (32 ring slots)
constexpr int LN_SEQ_WORDS  = 16;   // most packed words per sequence incl. the pad word: reads of at most 240 bases
constexpr int LN_RING_WORDS = 48;   // 32 (M ring: 4 rows x 32 bytes) + 8 (I) + 8 (D)
// words of LDS a lane owns when a sequence takes sw (even) packed words: odd
constexpr uint32_t lane_stride_words(uint32_t sw) { return (uint32_t)LN_RING_WORDS + 2u * sw + 1u; }

// 2-bit packs the len (>= 16) bases at src (16-byte aligned) into dst[0 .. SW): every 16-byte chunk of the sequence in ONE
// round of loads (independent, so a lane pays their latency once), the bytes of a last partial chunk by the dwords that
// hold them -- nothing past the end of the sequence is read.  Returns true when it saw a byte outside ACGT.
WFA_DEV bool lane_pack_seq(const uint8_t *src, uint32_t len, uint32_t *dst, uint32_t SW) {
    const uint32_t nfull = len >> 4, nb = len & 15u, ntd = (nb + 3u) >> 2;
    // four bases as eight bits; diff collects (canonical letter ^ byte): nonzero = a byte outside ACGT
    const auto pk = [](uint32_t w, uint32_t &diff) -> uint32_t {
        const uint32_t x = (w >> 1) & 0x03030303u;
        diff |= __builtin_amdgcn_perm(0u, 0x47544341u, x) ^ w;  // code -> 'A' 'C' 'T' 'G'
        const uint32_t y = x | (x >> 6);
        return (y & 0xFu) | ((y >> 12) & 0xF0u);
    };
    const uint32_t *const tp = reinterpret_cast<const uint32_t *>(nb != 0u ? src + 16u * nfull : src);
    uint32_t              td[4];
#pragma unroll
    for (int i = 0; i < 4; i++) td[i] = tp[(uint32_t)i < ntd ? i : 0];
    uint32_t bad = 0u, bad_t = 0u, wt = 0u;  // wt: the partial chunk, bytes past the end count as 'A' (code 0)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t mb = nb > 4u * i ? (nb - 4u * i < 4u ? nb - 4u * i : 4u) : 0u;
        const uint32_t km = mb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * mb)) - 1u);
        wt |= pk((td[i] & km) | (0x41414141u & ~km), bad_t) << (8 * i);
    }
    bad = nb != 0u ? bad_t : 0u;
    // (eight chunks a round: 32 registers of loads in flight, two rounds for the longest read)
#pragma unroll 1
    for (int j0 = 0; j0 < LN_SEQ_WORDS; j0 += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = reinterpret_cast<const uint4 *>(src)[(uint32_t)(j0 + j) < nfull ? j0 + j : 0];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t       bj = 0u;
            const uint32_t wf = pk(v[j].x, bj) | (pk(v[j].y, bj) << 8) | (pk(v[j].z, bj) << 16) | (pk(v[j].w, bj) << 24);
            const bool     full = (uint32_t)(j0 + j) < nfull;
            bad |= full ? bj : 0u;
            const uint32_t w = full ? wf : (((uint32_t)(j0 + j) == nfull && nb != 0u) ? wt : 0u);
            if ((uint32_t)(j0 + j) < SW) dst[j0 + j] = w;
        }
        asm volatile("" ::: "memory");
    }
    return bad != 0u;
}

// DX / DOE (round 5): the penalty shape x/g : (o+e)/g with e/g == 1 and DX <= DOE <= 4.  The M ring holds DOE rows (slot =
// step mod DOE): the row of step i replaces M[i - DOE] = M[s-o-e] in place, which is what the in-place trick above needs;
// M[s-x] is slot (i - DX) mod DOE (with DX == DOE the very row being replaced: its cell at k is read before it is stored).
template <bool CENSUS, bool ADAPTIVE, int DX = 2, int DOE = 4>
__global__ __launch_bounds__(64) void wfa_lane_kernel(const KParams P) {
    static_assert(DX >= 1 && DX <= DOE && DOE <= 4, "the in-place ring replaces M[s-o-e]: x <= o+e, at most four rows");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const uint32_t        SW   = P.lds_seq_words;  // (even: a slot is a whole number of 16-byte units)
    uint32_t *const       base = lds + (uint32_t)lane * P.sub_lds_words;
    uint8_t *const        H    = reinterpret_cast<uint8_t *>(base);  // M rows at byte 32 r, I at 128, D at 160
    uint32_t *const       lq = base + LN_RING_WORDS, *const lt = lq + SW;
    const uint64_t        cap      = P.arena_words;
    const int             rows_cap = (int)(cap / 16);
    const int             mdd = (int)P.max_dist_diff, minwf = (int)P.min_wf_len;
    constexpr bool        adaptive = ADAPTIVE;  // (= P.adaptive: an instance each, the cell loop of the plain one does not track distances)
    const uint32_t        seed_si  = P.dx;
    constexpr int         BIG = 0x3FFFFFFF;

    bool first_gen = true;
    for (;;) {
        // ------------------------------------------------------------ a generation: 64 queue entries, one per lane
        // (the first generation of every wave is its own index: a thousand waves asking the one queue word at the start of the
        // launch are served one after the other -- the last of them some 10 us late; the queue hands out what follows)
        uint32_t gbase = blockIdx.x * 64u;
        if (!first_gen) {
            if (lane == 0) gbase = atomicAdd(P.queue_head, 64u);
            gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase) + gridDim.x * 64u;
        }
        first_gen = false;
        if (gbase >= P.chunk_n) break;
        const uint32_t wi = gbase + (uint32_t)lane;
        bool           active = wi < P.chunk_n;
        int            n = 0, m = 0;
        if (P.prepack) {  // slots of wfa_prepack_kernel
            if (active) {
                const uint4 *const slot = reinterpret_cast<const uint4 *>(P.prepack + (uint64_t)wi * P.prepack_words);
                const uint4        hdr  = slot[0];
                n = (int)hdr.x, m = (int)hdr.y;
                if (hdr.z != ST_PENDING) {  // empty / too long / longer than 240 bases / a byte outside ACGT: no alignment here
                    P.pair_meta[wi] = make_uint4(hdr.z, 0u, 0u, 0u);
                    if (hdr.z >= ST_REDO_BYTES) push_redo(P, P.work ? P.work[wi] : P.chunk_first + wi, hdr.z);
                    active = false;
                } else {
                    for (uint32_t i = 0; i < SW / 2u; i++) {
                        const uint4 v = slot[1 + i];
                        lq[4 * i] = v.x, lq[4 * i + 1] = v.y, lq[4 * i + 2] = v.z, lq[4 * i + 3] = v.w;  // (q words, then t words: contiguous)
                    }
                }
            }
        } else {
            // every lane packs the bytes of its own pair: a kernel of its own for that cost 21 us + a launch per 1e5 pairs.
            // (Sequences that start on a 16-byte boundary and hold at least 16 bases -- what the packing helpers of the
            // bindings produce; a wave that meets anything else packs word by word, a round trip per word.)
            uint32_t st0 = ST_PENDING, pair = 0u, nq = 0u, mt = 0u;
            uint64_t qo = 0ull, to = 0ull;
            if (active) {
                pair = P.work ? P.work[wi] : P.chunk_first + wi;
                nq = P.q_len[pair], mt = P.t_len[pair], qo = P.q_off[pair], to = P.t_off[pair];
                if (nq == 0u || mt == 0u)
                    st0 = ST_EMPTY;  // wfa.go:204-206
                else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
                    st0 = ST_TOO_LONG;  // wfa.go:207-209
                else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW)
                    st0 = ST_REDO_LDS;
            }
            const bool           pk = active && st0 == ST_PENDING;
            const uint8_t *const qs = P.blob + (pk ? qo : 0ull), *const tq = P.blob + (pk ? to : 0ull);
            const bool           fast = ((reinterpret_cast<uintptr_t>(qs) | reinterpret_cast<uintptr_t>(tq)) & 15u) == 0u && nq >= 16u && mt >= 16u;
            bool                 bad  = false;
            if (__ballot(pk && !fast) == 0ull) {
                if (pk) {
                    bad = lane_pack_seq(qs, nq, lq, SW);
                    asm volatile("" ::: "memory");
                    bad |= lane_pack_seq(tq, mt, lt, SW);
                }
            } else if (pk) {
                for (uint32_t j = 0; j < SW; j++) lq[j] = prepack_word(P.blob, qo, nq, j, bad), lt[j] = prepack_word(P.blob, to, mt, j, bad);
            }
            if (pk && bad) st0 = ST_REDO_BYTES;  // the byte-compare path takes the pair
            n = (int)nq, m = (int)mt;
            if (active && st0 != ST_PENDING) {
                P.pair_meta[wi] = make_uint4(st0, 0u, 0u, 0u);
                if (st0 >= ST_REDO_BYTES) push_redo(P, pair, st0);
                active = false;
            }
        }
#pragma unroll
        for (int i = 0; i < LN_RING_WORDS; i++) base[i] = 0u;
        const int   Ak = m - n;
        const bool  first_eq = ((lq[0] ^ lt[0]) & 3u) == 0u;  // q[0] == t[0] (wfa.go:155)
        SeqView<0>  sv;
        sv.q = lq, sv.t = lt, sv.n = n, sv.m = m;
        uint16_t *const A16 = reinterpret_cast<uint16_t *>(P.arena + (uint64_t)wi * cap);
        // kept bands of the M rows of the last DOE score steps (blo[0] / bhi[0]: the newest, which is also the I and D rows' band)
        int blo[DOE], bhi[DOE];
#pragma unroll
        for (int d = 0; d < DOE; d++) blo[d] = BIG, bhi[d] = -BIG;
        uint32_t fail = 0u;          // ST_REDO_* of a pair this kernel hands on
        uint32_t si = 0, cells = 0;  // (si: the same for every lane of the wave that is still running -- a generation starts together)
        int      r0 = 0;             // si mod DOE: ring slot of M[s-o-e] (replaced by this step's row)

        // ------------------------------------------------------------ score steps
        while (__ballot(active) != 0ull) {
            const int R0 = r0, R2 = r0 + (DOE - DX) >= DOE ? r0 - DX : r0 + (DOE - DX);  // ring slots of M[s-o-e] (replaced by this step's row) and M[s-x]
            // range of next() (wfa.go:557-563): sources' ranges +-1, clamped to the matrix
            int lo = imin2(imin2(blo[DOE - 1], blo[DX - 1]), blo[0]), hi = imax2(imax2(bhi[DOE - 1], bhi[DX - 1]), bhi[0]);
            const bool want_seed = active && ((si == 0u && first_eq) || (si == seed_si && !first_eq));
            bool       any_src = hi >= lo;
            lo = any_src ? lo - 1 : 0, hi = any_src ? hi + 1 : 0;
            if (!any_src && !want_seed) lo = 1, hi = 0;  // nothing to compute at this score
            lo = imax2(lo, -(n - 1)), hi = imin2(hi, m - 1);
            if (want_seed) lo = imin2(lo, 0), hi = imax2(hi, 0);
            // (handed on after the generation: a load or an atomic in this loop would make every step wait for the arena
            // stores of the step before -- one counter for loads and stores)
            if (active && (int)si >= rows_cap) fail = ST_REDO_ARENA, active = false;
            if (active && hi - lo + 1 > LN_W) fail = ST_REDO_BAND, active = false;
            uint8_t *const       Mo = H + 32 * R0;
            const uint8_t *const Mx = H + 32 * R2;
            uint8_t *const       Ir = H + 128, *const Dr = H + 160;
            uint16_t *const      arow = A16 + (size_t)si * 32u;
            uint32_t ghit_w = 0u;              // wf-adaptive: a cell of the row sits at a sequence end (nonzero)
            int      glo = BIG, ghi = -BIG;    // tight range of the M cells set (M.Lo / M.Hi of the new wavefront, wfa.go:242)
            int      mind = BIG, maxd = -BIG;  // wf-adaptive: distances of the usable entries (wfa.go:478-497)

            // ------------------------------------------------------------ the seed of initComponents (wfa.go:155-160)
            // Score 0 when the first bases agree, else score x: nothing exists at lower scores, so the row is this one cell
            // on diagonal 0 -- done here, and the cell loop below carries no test for it.
            if (__ballot(want_seed) != 0ull) {
                if (want_seed) {
                    uint32_t  h0   = 1u;
                    const int lim0 = imin2(n, m);
                    if ((int)h0 < lim0) h0 += (uint32_t)sv.lcp(1, 1);  // WF_EXTEND (wfa.go:381-458)
                    Mo[0] = (uint8_t)h0, Ir[0] = 0, Dr[0] = 0;
                    arow[0] = (uint16_t)(first_eq ? BLK_SEED_MATCH : BLK_SEED_MISMATCH);
                    glo = 0, ghi = 0;
                    if (ADAPTIVE) {
                        if ((int)h0 < lim0) mind = maxd = imax2(n, m) - (int)h0;
                        ghit_w = (int)h0 >= lim0 ? 1u : 0u;
                    }
                    lo = 1, hi = 0;  // (no other cell)
                }
            }

            // ------------------------------------------------------------ cells, ascending k (wfa.go:572-699)
            // A cell whose extension is longer than two 16-base windows keeps its lane for another round of the loop while the
            // other lanes go on to their next cells: the wave pays the longest SUM of windows of a row, not the longest run
            // of every cell.
            int      k = lo;
            uint32_t a0 = 0u, b0 = 0u, self_mo = 0u;  // M[s-o-e][k-1], I[s-e][k-1] and M[s-o-e][k] as they were before this step
            // (the four ring reads of a cell are issued when the cell before it is stored: a round trip to LDS less in the
            // chain of every round)
            uint32_t nc0 = 0u, nd0 = 0u, nx0 = 0u, nis = 0u;
            if (active && k <= hi) {
                self_mo = Mo[k & 31];  // (the slots of k - 1 hold zero: two below every kept band)
                nc0 = Mo[(k + 1) & 31], nd0 = Dr[(k + 1) & 31], nx0 = Mx[k & 31], nis = Ir[k & 31];
            }
            uint32_t h = 0u, wd = 0u, Isk = 0u, Dsk = 0u, c0s = 0u, is_s = 0u, pend = 0u;  // the cell being extended
            int      lim = 0;
            for (;;) {
                const bool go = active && k <= hi;
                if (__ballot(go) == 0ull) break;
                if (go && pend == 0u) {
                    const uint32_t c0 = nc0, d0 = nd0, x0 = nx0;
                    is_s = nis;  // I[s-e][k]: the next cell's k-1 source
                    c0s  = c0;
                    // rejections: > m (not >=) for I and X sources, offset - k > n for D and X sources
                    const uint32_t a = (int)a0 > m ? 0u : a0, b = (int)b0 > m ? 0u : b0;
                    const uint32_t c = (int)c0 - k > n ? 0u : c0, d = (int)d0 - k > n ? 0u : d0;
                    const uint32_t x = ((int)x0 > m || (int)x0 - k > n) ? 0u : x0;
                    const uint32_t mi = umax2(a, b);
                    Isk = mi + (mi != 0u ? 1u : 0u);
                    Dsk = umax2(c, d);
                    const uint32_t x1  = x + (x != 0u ? 1u : 0u);
                    uint32_t       Msk = umax2(umax2(Isk, Dsk), x1);
                    const bool fromX = x != 0u && Msk == x1;  // wfa.go:657-693: the mismatch wins a tie, then the insertion
                    const bool fromI = !fromX && Msk == Isk;
                    // backTrace recomputes the pre-extension offset from the un-rejected sources (wfa.go:766-817)
                    const uint32_t mu = umax2(a0, b0), Iu = mu + (mu != 0u ? 1u : 0u), Du = umax2(c0, d0);
                    const uint32_t Xu = x0 + (x0 != 0u ? 1u : 0u);
                    const bool     iext = a < b, dext = c < d;
                    const uint32_t o0   = (fromI && iext) ? Iu : ((!fromX && !fromI && dext) ? Du : umax2(umax2(Iu, Du), Xu));
                    wd = blk_word(o0, iext, dext, fromX, fromI);
                    // seeds of initComponents (wfa.go:155-160)
                    h = Msk, lim = imin2(n + k, m);  // (>= 1: k >= -(n - 1))
                    pend = 1u;
                }
                if (go) {
                    // (the ring reads of the NEXT cell, early: they have the whole extension to arrive.  None of them is the
                    // slot this cell is stored to, and a lane that stays on its cell reads the same slots again)
                    nc0 = Mo[(k + 2) & 31], nd0 = Dr[(k + 2) & 31], nx0 = Mx[(k + 1) & 31], nis = Ir[(k + 1) & 31];
                    // WF_EXTEND (wfa.go:381-458): 32 bases a round, from three words of each sequence read in one go
                    bool      done = true;
                    const int rem  = lim - (int)h;
                    if (h != 0u && rem > 0) {
                        const int      v = (int)h - k, wq = v >> 4, wt = (int)h >> 4;
                        uint32_t       q0 = lq[wq], q1 = lq[wq + 1], q2 = lq[wq + 2], t0 = lt[wt], t1 = lt[wt + 1], t2 = lt[wt + 2];
                        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(t0), "+v"(t1), "+v"(t2));  // (all six in one round trip)
                        const uint32_t sq = (uint32_t)(v & 15) * 2u, stt = (uint32_t)(h & 15u) * 2u;
                        const uint32_t xa = __funnelshift_r(q0, q1, sq) ^ __funnelshift_r(t0, t1, stt);
                        const uint32_t xb = __funnelshift_r(q1, q2, sq) ^ __funnelshift_r(t1, t2, stt);
                        const int      adv = xa != 0u ? (__builtin_ctz(xa) >> 1) : (xb != 0u ? 16 + (__builtin_ctz(xb) >> 1) : 32);
                        done = adv < 32 || adv >= rem;
                        h += (uint32_t)imin2(adv, rem);
                    }
                    if (done) {
                        const bool nz = h != 0u;
                        // (a word of a cell that does not exist is never read.  Measured instead of this 2-byte store per cell: two
                        // cells per 4-byte store, and the row collected in LDS and written as 16-byte quarters -- no gain, and the
                        // second costs a wave per CU)
                        arow[k & 31] = (uint16_t)wd;
                        glo = nz ? imin2(glo, k) : glo, ghi = nz ? k : ghi;
                        if (ADAPTIVE) {
                            if (nz && (int)h < lim) {  // a usable entry of wf-adaptive: inside both sequences
                                const int dd = imax2(n + k, m) - (int)h;
                                mind = imin2(mind, dd), maxd = imax2(maxd, dd);
                            }
                            ghit_w |= (nz && (int)h >= lim) ? 1u : 0u;
                        }
                        Mo[k & 31] = (uint8_t)h, Ir[k & 31] = (uint8_t)Isk, Dr[k & 31] = (uint8_t)Dsk;
                        a0 = self_mo, self_mo = c0s, b0 = is_s;
                        k += 1, pend = 0u;
                    }
                }
            }
            // the row has an M cell; the pair ends at this score when its cell on the final diagonal has reached the end of t
            // (wfa.go:228-236; a slot between glo and ghi holds this row's cell or zero)
            const bool anyM = ghi >= glo, ghit = ghit_w != 0u;
            const bool term = active && anyM && Ak >= glo && Ak <= ghi && (int)Mo[Ak & 31] >= m;

            // ------------------------------------------------------------ band of the row + wf-adaptive (wfa.go:461-540)
            int ilo = glo, ihi = ghi;
            if (__ballot(active && !term && adaptive && anyM && (ghi - glo + 1) >= minwf) != 0ull) {
                const bool want = active && !term && adaptive && anyM && (ghi - glo + 1) >= minwf;
                const int  thr   = mind + mdd;
                const bool found = want && mind != BIG && maxd > thr;  // some distance fails (wfa.go:507)
                if (__ballot(found) != 0ull) {
                    int first_ok = BIG, last_ok = -BIG, leadp = BIG + 1;  // leadp: last usable entry before first_ok (it fails)
                    bool have_lead = false;
                    for (int kk = glo; __ballot(found && kk <= ghi) != 0ull; kk++) {
                        if (found && kk <= ghi) {
                            const int h = (int)Mo[kk & 31], lim = imax2(1, imin2(n + kk, m));
                            if (h != 0 && h < lim) {
                                const int d = imax2(n + kk, m) - h;
                                if (d <= thr) {
                                    if (first_ok == BIG) first_ok = kk;
                                    last_ok = kk;
                                } else if (first_ok == BIG) {
                                    leadp = kk, have_lead = true;
                                }
                            }
                        }
                    }
                    // wfa.go:509-524: _lo = one past the last failing entry of the leading run (holes between it and the first
                    // non-failing entry stay when a cell of the pair sits at a sequence end; else they are holes either way)
                    if (found) ilo = ghit ? (have_lead ? leadp + 1 : glo) : first_ok, ihi = last_ok;
                }
            }
            const bool keepl = active && anyM && ihi >= ilo;
            // what wf-adaptive dropped holds zero from here on (cells outside [glo, ghi] never held anything: no M cell, so no I
            // or D cell either)
            if (__ballot(active && anyM && (ilo > glo || ihi < ghi)) != 0ull) {
                for (int kk = glo; __ballot(active && anyM && kk <= ghi) != 0ull; kk++)
                    if (active && anyM && kk <= ghi && (kk < ilo || kk > ihi)) Mo[kk & 31] = 0, Ir[kk & 31] = 0, Dr[kk & 31] = 0;
            }
            if (CENSUS && keepl) {
                for (int kk = ilo; kk <= ihi; kk++) cells += (Mo[kk & 31] != 0) + (Ir[kk & 31] != 0) + (Dr[kk & 31] != 0);
            }
            // (the I / D rows exist where wf-adaptive kept the M row: Delete of wfa.go:526-535; an I or D cell never exists
            // without the M cell of its diagonal -- M[s][k] >= I[s][k], D[s][k])
#pragma unroll
            for (int d = DOE - 1; d > 0; d--) blo[d] = blo[d - 1], bhi[d] = bhi[d - 1];
            blo[0] = keepl ? ilo : BIG, bhi[0] = keepl ? ihi : -BIG;
            if (active && term) {
                const uint32_t hf = Mo[Ak & 31];
                P.pair_meta[wi]   = make_uint4(ST_OK, si * P.g, hf, (CENSUS && P.census) ? cells : 0u);
                active            = false;
            }
            si += 1u;
            r0 = r0 + 1 == DOE ? 0 : r0 + 1;
        }
        if (fail != 0u) {
            P.pair_meta[wi] = make_uint4(fail, 0u, 0u, 0u);
            push_redo(P, P.work ? P.work[wi] : P.chunk_first + wi, fail);
        }
    }
}

}  // namespace wfa
