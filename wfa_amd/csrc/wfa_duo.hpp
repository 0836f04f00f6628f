// wfa_duo.hpp -- kernel E: blocked register-window forward kernel with a VARIABLE number of lanes per pair.
//
// wfa_blk_kernel<16,1> gives every pair a 64-diagonal window (16 lanes x 4 diagonals), because almost every 1 kbp pair
// needs that many diagonals at SOME score -- but the union of the rows a step reads spans 17 diagonals on average and
// 27 or fewer in 87 % of the pair-steps (oracle band traces, scripts/sim_rows.py): 72 % of the window slots idle.
// Here a 16-lane DPP row holds either ONE pair with a 64-diagonal window or TWO pairs with 32-diagonal windows (8 lanes
// x 4 diagonals each), and a pair changes between the two forms while it runs:
//   * widen   -- a narrow pair whose band outgrows 32 diagonals takes the other half of its row.  If another pair runs
//                there, that pair is PARKED: its rings (6 rows x 32 diagonals) and scalars go to an LDS record of the
//                wave, its packed sequences stay where they are, and it resumes in the next free half of the wave.
//   * narrow  -- a wide pair whose band has shrunk to <= 18 diagonals gives one half of its row back.
//   * recentre-- a pair whose newest row touches the edge of its window is moved so that its band sits in the middle.
// All three are ONE operation: every lane of the wave names the lane its ring registers come from (or none), and 24
// ds_bpermute_b32 move the rings (the LDS crossbar: no LDS memory, no bank conflicts).  It replaces the one-lane DPP
// window shifts of wfa_blk_kernel.
//
// Replaying the band traces through this scheme (scripts/sim_rows.py: four rows, park area of three, the policy
// above) gives 1.73x fewer wave-steps per pair, 1.9 parks per pair and 1.8 % of the pairs handed on because the park
// area is full when they need to widen (they go to the 256-diagonal rung like any pair whose band leaves the window).
//
// With twice the pairs per wave a refill is twice as frequent per wave step, and parked pairs resume ~2 times per
// pair: the refill chain of wfa_blk_kernel (queue atomic -> lengths / offsets -> bytes -> 2-bit packing; three dependent
// round trips during which all pairs of the wave stand still) would eat the gain.  So
//   * the chunk's sequences are 2-bit packed up front by wfa_prepack_kernel (slot = {n, m, status, 0, q words, t words}),
//   * every wave PREFETCHES its next pair one stage per loop iteration -- queue atomic; a step later the slot's words
//     into four registers per lane; a step later those registers into a spare LDS buffer -- so a result is only
//     consumed a whole score step after its request was issued, and a freed half starts its next pair from LDS,
//   * a parked pair resumes from LDS.
//
// Everything else -- the rejection-free WF_NEXT with the decisions shifted in under the offset, the first-window +
// candidate-at-a-time WF_EXTEND, the branch-free band / wf-adaptive reductions, the arena tiles of 8 scores x 64
// diagonals (here with 16-bit words, CompactView fmt 7) and pair_meta for wfa_backtrace_kernel -- is wfa_blk_kernel's,
// with the group size a per-lane value instead of a template argument.  Results do not depend on which pairs share a
// wave: the band of a row is a function of the pair alone as long as its window holds it, and a pair whose band cannot
// be held is handed on (ST_REDO_BAND) and recomputed from scratch by the retry rung.
#pragma once
#include "wfa_blk.hpp"
#include "wfa_duo_cfg.hpp"

namespace wfa {

// Group reductions: three butterfly stages inside the 8 lanes of a half row, and a fourth (row_mirror) that only the lanes
// of a 16-lane pair execute: EXEC is narrowed to the wave's wide pairs for it (wm = ballot of `wide`; both halves of a wide
// pair are in the mask, so every active lane's mirror partner is active too).  Round 6: before, the fourth stage ran on a
// copy in all lanes and a select kept it for the wide ones -- two more vector instructions per reduced value, ten per step.
// (wait states: a DPP source written by the previous stage is read at least two instructions later; EXEC is written by
// the scalar unit, which a DPP instruction need not wait for.)
struct DuoRed {
#define WFA_DUO_3(opa, opb, opc)                                                                                    \
    unsigned long long sv;                                                                                          \
    asm("s_nop 1\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_XOR1 "\n\t" opb " %1, %1, %1 " WFA_DPP_CTL_XOR1 "\n\t" opc      \
        " %2, %2, %2 " WFA_DPP_CTL_XOR1 "\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_XOR2 "\n\t" opb " %1, %1, %1 "        \
        WFA_DPP_CTL_XOR2 "\n\t" opc " %2, %2, %2 " WFA_DPP_CTL_XOR2 "\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_HMIR "\n\t" \
        opb " %1, %1, %1 " WFA_DPP_CTL_HMIR "\n\t" opc " %2, %2, %2 " WFA_DPP_CTL_HMIR "\n\t"                         \
        "s_and_saveexec_b64 %3, %4\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_MIR "\n\t" opb " %1, %1, %1 " WFA_DPP_CTL_MIR    \
        "\n\t" opc " %2, %2, %2 " WFA_DPP_CTL_MIR "\n\ts_mov_b64 exec, %3"                                            \
        : "+v"(a), "+v"(b), "+v"(c), "=&s"(sv)                                                                      \
        : "s"(wm));
#define WFA_DUO_2(opa, opb)                                                                                         \
    unsigned long long sv;                                                                                          \
    asm("s_nop 1\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_XOR1 "\n\t" opb " %1, %1, %1 " WFA_DPP_CTL_XOR1 "\n\ts_nop 0\n\t" \
        opa " %0, %0, %0 " WFA_DPP_CTL_XOR2 "\n\t" opb " %1, %1, %1 " WFA_DPP_CTL_XOR2 "\n\ts_nop 0\n\t" opa        \
        " %0, %0, %0 " WFA_DPP_CTL_HMIR "\n\t" opb " %1, %1, %1 " WFA_DPP_CTL_HMIR "\n\t"                            \
        "s_and_saveexec_b64 %2, %3\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_MIR "\n\t" opb " %1, %1, %1 " WFA_DPP_CTL_MIR    \
        "\n\ts_mov_b64 exec, %2"                                                                                    \
        : "+v"(a), "+v"(b), "=&s"(sv)                                                                               \
        : "s"(wm));
#define WFA_DUO_1(opa)                                                                                              \
    unsigned long long sv;                                                                                          \
    asm("s_nop 1\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_XOR1 "\n\ts_nop 1\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_XOR2      \
        "\n\ts_nop 1\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_HMIR "\n\t"                                                \
        "s_and_saveexec_b64 %1, %2\n\ts_nop 0\n\t" opa " %0, %0, %0 " WFA_DPP_CTL_MIR "\n\ts_mov_b64 exec, %1"       \
        : "+v"(a), "=&s"(sv)                                                                                        \
        : "s"(wm));
    static WFA_DEV void min_max_min(int &a, int &b, int &c, unsigned long long wm) { WFA_DUO_3("v_min_i32_dpp", "v_max_i32_dpp", "v_min_i32_dpp") }
    static WFA_DEV void min_max(int &a, int &b, unsigned long long wm) { WFA_DUO_2("v_min_i32_dpp", "v_max_i32_dpp") }
    static WFA_DEV void max_add(int &a, int &b, unsigned long long wm) { WFA_DUO_2("v_max_i32_dpp", "v_add_u32_dpp") }
    static WFA_DEV int  max1(int a, unsigned long long wm) {
        WFA_DUO_1("v_max_i32_dpp")
        return a;
    }
    static WFA_DEV int or1(int a, unsigned long long wm) {
        WFA_DUO_1("v_or_b32_dpp")
        return a;
    }
#undef WFA_DUO_3
#undef WFA_DUO_2
#undef WFA_DUO_1
};

// DX / DOE: the penalty shape, as in wfa_blk_kernel (R = max(DX, DOE) rows in the M ring)
template <bool CENSUS, int DX = 2, int DOE = 4>
#ifndef WFA_DUO_VGPRS
#define WFA_DUO_VGPRS 64  // (the attribute counts register PAIRS on gfx90a and later: 64 = no cap below the 128 of four waves per SIMD; 60 = 120 VGPRs)
#endif
__global__ __launch_bounds__(64, WFA_DUO_WAVES) __attribute__((amdgpu_num_vgpr(WFA_DUO_VGPRS))) void wfa_duo_kernel(const KParams P) {
    constexpr int PP = 4;
    static_assert(DX >= 1 && DOE >= 1 && DX <= 4 && DOE <= 4, "ring depths of one to four score steps");
    constexpr int R = DX > DOE ? DX : DOE;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int  lane = threadIdx.x, l7 = lane & 7, l15 = lane & 15;
    const bool hi_half = (lane & 8) != 0;

    const uint32_t  SW = P.lds_seq_words, PW = P.prepack_words;  // PW = 4 + 2 SW words per prepacked pair (a multiple of 4)
    const uint32_t  FP = duo_fetch_pairs(PW), NBUF = duo_bufs(PW);  // pairs per fetch, sequence buffers of the wave
    uint32_t *const park0 = lds + NBUF * PW;
    const uint64_t  cap      = P.arena_words;
    const int       mdd      = (int)P.max_dist_diff;
    const int       minwf    = (int)P.min_wf_len;
    const bool      adaptive = P.adaptive != 0;
    const uint32_t  seed_si  = P.dx;  // the mismatch seed M[x][0] belongs to step x/g
    const int       rows_cap = (int)(cap / 32);  // 16-bit words: a score's 64 diagonals are 32 words

    // ---- per-pair state (identical in the lanes of a pair)
    int       st = 0;  // 0 = free half, 1 = running
    bool      wide = false;
    uint32_t  pidx = 0, si = 0, cells = 0, sbuf = 0;
    int       n = 0, m = 0, kb = 0;
    bool      slow = false, first_eq = false;
    // ---- per-lane values derived from them
    int             j = l7, k0 = 0;
    uint32_t        mdn = 0u, mup = 0u;  // all-ones unless the lane is the first / last of its group
    const uint32_t *lq = lds + 4, *lt = lds + 4;
    uint32_t       *rowp = nullptr;

    uint32_t M[R][PP], I[PP], D[PP];  // offsets, 0 = absent; M[i mod R] = row of step i
    int      rlo[R], rhi[R];          // band of each kept M row (absolute k); empty = (BIG, -BIG)
    int      lim[PP], lmx[PP];
#pragma unroll
    for (int d = 0; d < R; d++) {
        rlo[d] = BK_BIG, rhi[d] = -BK_BIG;
#pragma unroll
        for (int p = 0; p < PP; p++) M[d][p] = 0u;
    }
#pragma unroll
    for (int p = 0; p < PP; p++) I[p] = D[p] = 0u, lim[p] = 0, lmx[p] = 0;

    // ---- wave-uniform state
    uint32_t buf_free  = (1u << NBUF) - 1u;      // sequence buffers nobody owns
    uint32_t park_used = 0u;                     // park records in use
    // prefetch pipeline (stages overlap: one new pair per iteration in steady state, three iterations of latency)
    // (one word of flags, not four bools: hipcc merges the stores to two bools into one store through a selected address,
    // which pins both in scratch memory -- two scratch loads per score step)
    constexpr uint32_t PF_TOK = 1u, PF_LD = 2u, PF_STAGED = 4u, PF_DRY = 8u;  // queue atomic in flight / slot loads in flight / a pair staged in LDS / queue exhausted
    uint32_t pf = 0u;
    uint32_t pf_tok = 0u, pf_ld_idx = 0u, pf_idx = 0u, pf_buf = 0u;
    uint32_t st_mask = 0u;  // buffers of the staged pairs, lowest bit = next pair (queue entry pf_idx)
    uint4    pf_w = make_uint4(0u, 0u, 0u, 0u);  // (a slot is at most 256 words: one 16-byte load per lane)

#ifdef WFA_STAMPS  // diagnostic build (scripts/stamps.sh): time per phase and event counts, summed over the waves
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    unsigned long long evt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-steps, restructuring rounds, rounds that move rings, parks, resumes,
                                                            // pairs started, running half-rows, pairs handed on
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    const auto pair_of = [&](uint32_t idx) { return P.work ? P.work[idx] : P.chunk_first + idx; };
    const auto dn1 = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true) & mdn; };
    const auto up1 = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true) & mup; };
    const auto other_half = [&](int x) { return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true); };  // row_ror:8
    const auto set_derived = [&]() {
        j   = wide ? l15 : l7;
        mdn = j == 0 ? 0u : ~0u;
        mup = j == (wide ? 15 : 7) ? 0u : ~0u;
        k0  = kb + PP * j;
#pragma unroll
        for (int p = 0; p < PP; p++) lim[p] = imax2(1, imin2(n + k0 + p, m)), lmx[p] = imax2(n + k0 + p, m);
        lq   = lds + sbuf * PW + 4u;
        lt   = lq + SW;
        if constexpr (DUO_ARENA_FMT == 10u)  // [pair of groups][score / 8][group & 1][score & 7]: the lane's base; the score's part is added at the store
            rowp = P.arena + (uint64_t)pidx * cap + (((uint32_t)k0 & 56u) >> 3) * (uint32_t)(rows_cap * 4) + (((uint32_t)k0 >> 2) & 1u) * 16u;
        else if constexpr (DUO_ARENA_FMT == 9u)  // [group of four diagonals][score]: the lane's group, then 8 bytes per score
            rowp = P.arena + (uint64_t)pidx * cap + (((uint32_t)k0 & 60u) >> 2) * (uint32_t)(rows_cap * 2) + si * 2u;
        else
            rowp = P.arena + (uint64_t)pidx * cap + (uint64_t)(si >> 3) * 256u + (si & 7u) * 2u;
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
    // OR over the wave of a value that is the same in the eight lanes of a half row
    const auto or_halves = [&](uint32_t v) {
        uint32_t r = 0u;
#pragma unroll
        for (int o = 0; o < 8; o++) r |= (uint32_t)__builtin_amdgcn_readlane((int)v, 8 * o);
        return r;
    };

    // ================================================================ between two score steps
    // prefetch pipeline, then -- only when something has to change -- recentre / widen / narrow / park / resume / start
    const auto refill = [&](auto ph_c) __attribute__((always_inline)) -> bool {
        constexpr int ph  = decltype(ph_c)::value;
        constexpr int NEW = (ph + R - 1) % R;  // ring slot of the newest row; M[ph] is the oldest (replaced by the next step)
        // ---------------------------------------------------------------- prefetch of the next pair, one stage per call
        // (wave-uniform by construction; said so explicitly, or hipcc keeps them in vector registers)
        pf        = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf);
        park_used = (uint32_t)__builtin_amdgcn_readfirstlane((int)park_used);
        buf_free  = (uint32_t)__builtin_amdgcn_readfirstlane((int)buf_free);
        st_mask   = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_mask);
        if ((pf & (PF_LD | PF_STAGED)) == PF_LD) {  // the words loaded an iteration ago go to spare LDS buffers, one per pair
            const uint32_t cnt = umin2(FP, P.chunk_n - pf_ld_idx);  // pairs this fetch really holds
            // the cnt lowest free buffers: slot s of the fetch goes to the s-th of them
            st_mask = 0u;
            {
                uint32_t f = buf_free;
                for (uint32_t c = 0; c < cnt; c++) st_mask |= f & (0u - f), f &= f - 1u;
            }
            buf_free &= ~st_mask;
            const uint32_t w0 = (uint32_t)lane * 4u, slot = w0 / PW;  // (a slot is a whole number of 16-byte pieces)
            if (slot < cnt) {
                uint32_t f = st_mask;
                for (uint32_t c = 0; c < slot; c++) f &= f - 1u;
                const uint32_t b = (uint32_t)__builtin_ctz(f);
                *reinterpret_cast<uint4 *>(lds + b * PW + (w0 - slot * PW)) = pf_w;
            }
            pf_idx = pf_ld_idx, pf = (pf | PF_STAGED) & ~PF_LD;
        }
        if ((pf & (PF_TOK | PF_LD)) == PF_TOK) {  // the queue entry claimed an iteration ago: its slot's words into registers
            pf_ld_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf_tok);
            const bool have = pf_ld_idx < P.chunk_n;
            if (have) {
                const uint32_t *const slot = P.prepack + (uint64_t)pf_ld_idx * PW;
                const uint32_t cnt = umin2(FP, P.chunk_n - pf_ld_idx);
                if ((uint32_t)lane * 4u < cnt * PW) pf_w = *reinterpret_cast<const uint4 *>(slot + lane * 4);
            }
            pf = (pf & ~PF_TOK) | (have ? PF_LD : PF_DRY);
        }
        if ((pf & (PF_TOK | PF_DRY)) == 0u) {
            // FP queue entries, by lane 0, result NOT awaited here: the library is built with
            // -mllvm -amdgpu-atomic-optimizer-strategy=None (Makefile).  With LLVM's atomic optimizer on, this atomicAdd()
            // becomes a wave-aggregated atomic followed AT ONCE by s_waitcnt + v_readfirstlane -- a memory round trip during
            // which the wave's pairs stand still, every four steps; without it the wait sits where the token is read, a
            // score step later.  (An earlier form issued the atomic as inline assembly: same timing, but the compiler then
            // does not know the result is outstanding, and a register copy or a spill of it before the wait would read
            // garbage.  Correct either way with this form; the flag only buys the overlap.)
            if (lane == 0) pf_tok = atomicAdd(P.queue_head, FP);
            pf |= PF_TOK;
        }
        // ---------------------------------------------------------------- does anything have to change?
        // (a loop: a parked pair that had asked to widen itself asks again as soon as it has resumed, before it steps)
        unsigned long long m_free;
        WFA_STAMP(0);  // prefetch stage
        for (;;) {
        const bool run   = st == 1;
        const int  Wc    = wide ? 64 : 32;
        const bool touch = run && rhi[NEW] >= rlo[NEW] && (rlo[NEW] <= kb || rhi[NEW] >= kb + Wc - 1);
        bool       narrowable = false;
        if constexpr (ph == 0) {  // (looked at every fourth step: a wide pair that could be narrow costs a half row, nothing else)
            int ulo = rlo[0], uhi = rhi[0];
#pragma unroll
            for (int d = 1; d < R; d++) ulo = imin2(ulo, rlo[d]), uhi = imax2(uhi, rhi[d]);
            narrowable    = run && wide && uhi - ulo + 1 <= DUO_NARROW_AT;
        }
        const unsigned long long m_ev = __ballot(touch || narrowable);
        m_free = __ballot(st == 0);
        // (expected: nothing to do -- tells the register allocator that what follows is the cold side of the loop)
        if (__builtin_expect(!(m_ev != 0ull || (m_free != 0ull && (park_used != 0u || (pf & PF_STAGED) != 0u))), 1)) break;
        WFA_EVT(1, 1);
        {
            // ------------------------------------------------------------ what each pair wants
            int ulo = rlo[0], uhi = rhi[0];
#pragma unroll
            for (int d = 1; d < R; d++) ulo = imin2(ulo, rlo[d]), uhi = imax2(uhi, rhi[d]);
            const int span = uhi - ulo + 1;
            int act = 0;  // 1 recentre, 2 widen, 3 narrow, 4 hand on (band wider than a whole row holds)
            if (touch) act = wide ? (span > DUO_WIDE_MAX ? 4 : 1) : (span > DUO_WIDEN_AT ? 2 : 1);
            else if (narrowable) act = 3;
            const int Wn  = act == 2 ? 64 : (act == 3 ? 32 : Wc);
            const int kbn = (((ulo + uhi + 1 - Wn) >> 1) + 2) & ~3;  // the band in the middle of the new window, base a multiple of 4
            // (cannot happen below the thresholds; if it ever did, the pair is handed on -- kbn was computed for Wn, so no other
            // move may be made of this request)
            if (act != 0 && act != 4 && !(ulo - kbn >= 1 && uhi - kbn <= Wn - 2)) act = 4;
            // ------------------------------------------------------------ widening: who takes whose half
            const int  o_act = other_half(act), o_kb = other_half(kb), o_kbn = other_half(kbn);
            const bool o_wide = other_half(wide ? 1 : 0) != 0;
            // both halves of a row ask: the lower one wins, the upper one is parked and asks again when it resumes
            bool widen  = act == 2 && !(hi_half && o_act == 2);
            bool taken  = !wide && !o_wide && o_act == 2 && !(!hi_half && act == 2);  // my half goes to the pair of the other half
            bool evict  = taken && st == 1;
            // ------------------------------------------------------------ park the evicted pairs (LDS records of the wave)
            unsigned long long m_evict = __ballot(evict);
            bool               lost    = false;  // a widening pair whose neighbour cannot be parked: it is handed on instead
            while (m_evict != 0ull) {
                const int      l0   = __builtin_ctzll(m_evict);  // first lane of an evicted half (wave-uniform)
                const uint32_t oct  = (uint32_t)l0 >> 3;
                m_evict &= ~(0xFFull << (8 * oct));
                const bool mine = evict && ((uint32_t)lane >> 3) == oct;
                if (park_used == (1u << DUO_PARK) - 1u) {  // no record left
                    if (((uint32_t)lane >> 3) == (oct ^ 1u)) lost = true;
                    if (mine) evict = false, taken = false;
                    continue;
                }
                const uint32_t rec = (uint32_t)__builtin_ctz(~park_used);
                park_used |= 1u << rec;
                WFA_EVT(3, 1);
                uint32_t *const pr = park0 + rec * DUO_PARK_WORDS;
                if (mine) {
                    // (record row r = the r-th oldest row of the ring, M[(ph + r) mod R]; a ring of fewer than four rows leaves the rest empty)
                    uint4 *const w = reinterpret_cast<uint4 *>(pr + 12 * l7);
                    const auto pk = [](uint32_t lo, uint32_t hi) { return lo | (hi << 16); };
                    uint32_t pw[8];
                    int      plo[4], phi[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        pw[2 * r]     = r < R ? pk(M[(ph + r) % R][0], M[(ph + r) % R][1]) : 0u;
                        pw[2 * r + 1] = r < R ? pk(M[(ph + r) % R][2], M[(ph + r) % R][3]) : 0u;
                        plo[r] = r < R ? rlo[(ph + r) % R] : BK_BIG, phi[r] = r < R ? rhi[(ph + r) % R] : -BK_BIG;
                    }
                    w[0] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
                    if constexpr (R > 2) w[1] = make_uint4(pw[4], pw[5], pw[6], pw[7]);
                    w[2] = make_uint4(pk(I[0], I[1]), pk(I[2], I[3]), pk(D[0], D[1]), pk(D[2], D[3]));
                    if (l7 == 0) {
                        uint4 *const s4 = reinterpret_cast<uint4 *>(pr + 96);
                        s4[0] = make_uint4(pidx, si, cells, sbuf);
                        s4[1] = make_uint4((uint32_t)n, (uint32_t)m, (uint32_t)kb, (slow ? 1u : 0u) | (first_eq ? 2u : 0u));
                        s4[2] = make_uint4((uint32_t)plo[0], (uint32_t)plo[1], (uint32_t)plo[2], (uint32_t)plo[3]);
                        s4[3] = make_uint4((uint32_t)phi[0], (uint32_t)phi[1], (uint32_t)phi[2], (uint32_t)phi[3]);
                    }
                }
            }
            if (lost) widen = false, act = 4;
            // a winner that was told "lost" leaves its neighbour alone; the neighbour's lanes must not join it
            {
                const bool o_lost = other_half(lost ? 1 : 0) != 0;
                if (o_lost) taken = false;
            }
            // ------------------------------------------------------------ pairs that are handed on
            if (__ballot(act == 4) != 0ull) {
                WFA_EVT(7, __builtin_popcountll(__ballot(act == 4 && j == 0)));
                if (act == 4 && j == 0) {
                    P.pair_meta[pidx] = make_uint4(ST_REDO_BAND, 0u, 0u, 0u);
                    push_redo(P, pair_of(pidx), ST_REDO_BAND);
                }
                buf_free |= or_halves(act == 4 ? (1u << sbuf) : 0u);
                if (act == 4) {
                    st = 0, wide = false;
                    clear_rings();
                }
            }
            // ------------------------------------------------------------ where every lane's rings come from
            // new group: base lane Bn, lanes Gn; old group of the pair the lane belongs to afterwards: base Bo, lanes Go;
            // lane Bn + jn takes the registers of lane Bo + jn + dl (dl = window move in lanes), zeros outside the old group
            const int  rb   = lane & ~15;
            const bool join = taken;  // (evicted or free half that becomes part of the neighbour's pair)
            int src = lane;  // -1: zeros
            bool moved = false, freed = false;
            {
                if (join) {
                    const int Bo = (lane & ~7) ^ 8, dl = (o_kbn - o_kb) >> 2, jo = (lane - rb) + dl;
                    src = (jo >= 0 && jo < 8) ? Bo + jo : -1, moved = true;
                } else if (act == 1) {
                    const int Bo = wide ? rb : (lane & ~7), Go = wide ? 16 : 8, dl = (kbn - kb) >> 2, jo = (lane - Bo) + dl;
                    src = (jo >= 0 && jo < Go) ? Bo + jo : -1, moved = true;
                } else if (widen) {
                    const int Bo = lane & ~7, dl = (kbn - kb) >> 2, jo = (lane - rb) + dl;
                    src = (jo >= 0 && jo < 8) ? Bo + jo : -1, moved = true;
                } else if (act == 3) {
                    const int dl = (kbn - kb) >> 2, jo = (lane - rb) + dl;
                    if (hi_half) src = -1, freed = true;
                    else src = (jo >= 0 && jo < 16) ? rb + jo : -1;
                    moved = true;
                }
            }
            if (__ballot(moved) != 0ull) {
                WFA_EVT(2, 1);
                const int sl = (src < 0 ? lane : src) << 2;
#pragma unroll
                for (int d = 0; d < R; d++)
#pragma unroll
                    for (int p = 0; p < PP; p++) {
                        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)M[d][p]);
                        M[d][p]          = src < 0 ? 0u : v;
                    }
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)I[p]);
                    const uint32_t u = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)D[p]);
                    I[p] = src < 0 ? 0u : v, D[p] = src < 0 ? 0u : u;
                }
                // lanes that join the neighbour's pair take its scalars
                if (__ballot(join) != 0ull) {
                    const int      o_n = other_half(n), o_m = other_half(m);
                    const uint32_t o_pidx = (uint32_t)other_half((int)pidx), o_si = (uint32_t)other_half((int)si);
                    const uint32_t o_cells = (uint32_t)other_half((int)cells), o_sbuf = (uint32_t)other_half((int)sbuf);
                    const int      o_fl = other_half((slow ? 1 : 0) | (first_eq ? 2 : 0));
                    int o_rlo[R], o_rhi[R];
#pragma unroll
                    for (int d = 0; d < R; d++) o_rlo[d] = other_half(rlo[d]), o_rhi[d] = other_half(rhi[d]);
                    if (join) {
                        n = o_n, m = o_m, pidx = o_pidx, si = o_si, cells = o_cells, sbuf = o_sbuf;
                        slow = (o_fl & 1) != 0, first_eq = (o_fl & 2) != 0;
#pragma unroll
                        for (int d = 0; d < R; d++) rlo[d] = o_rlo[d], rhi[d] = o_rhi[d];
                        st = 1;
                    }
                }
                if (join) kb = o_kbn;
                else if (moved && !freed) kb = kbn;
                if (join || widen) wide = true;
                if (act == 3) wide = false;
                if (freed) {
                    st = 0;
#pragma unroll
                    for (int d = 0; d < R; d++) rlo[d] = BK_BIG, rhi[d] = -BK_BIG;
                }
            }
            // ------------------------------------------------------------ free halves: parked pairs first, then the staged pair
            m_free = __ballot(st == 0);
            while (m_free != 0ull && (park_used != 0u || (pf & PF_STAGED) != 0u)) {
                const uint32_t oct  = (uint32_t)__builtin_ctzll(m_free) >> 3;  // wave-uniform
                const bool     mine = ((uint32_t)lane >> 3) == oct;
                if (park_used != 0u) {
                    const uint32_t rec = (uint32_t)__builtin_ctz(park_used);
                    park_used &= park_used - 1u;
                    WFA_EVT(4, 1);
                    const uint32_t *const pr = park0 + rec * DUO_PARK_WORDS;
                    if (mine) {
                        // (record row r -> M[(ph + r) mod R]: the phase of the resuming wave, not of the one that parked it)
                        const uint4 *const w = reinterpret_cast<const uint4 *>(pr + 12 * l7);
                        const uint4 v0 = w[0], v1 = R > 2 ? w[1] : make_uint4(0u, 0u, 0u, 0u), v2 = w[2];
                        const uint32_t pw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        const uint4 *const s4 = reinterpret_cast<const uint4 *>(pr + 96);
                        const uint4 a0 = s4[0], a1 = s4[1], a2 = s4[2], a3 = s4[3];
                        const uint32_t plo[4] = {a2.x, a2.y, a2.z, a2.w}, phi[4] = {a3.x, a3.y, a3.z, a3.w};
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            M[(ph + r) % R][0] = pw[2 * r] & 0xFFFFu, M[(ph + r) % R][1] = pw[2 * r] >> 16;
                            M[(ph + r) % R][2] = pw[2 * r + 1] & 0xFFFFu, M[(ph + r) % R][3] = pw[2 * r + 1] >> 16;
                            rlo[(ph + r) % R] = (int)plo[r], rhi[(ph + r) % R] = (int)phi[r];
                        }
                        I[0] = v2.x & 0xFFFFu, I[1] = v2.x >> 16, I[2] = v2.y & 0xFFFFu, I[3] = v2.y >> 16;
                        D[0] = v2.z & 0xFFFFu, D[1] = v2.z >> 16, D[2] = v2.w & 0xFFFFu, D[3] = v2.w >> 16;
                        pidx = a0.x, si = a0.y, cells = a0.z, sbuf = a0.w;
                        n = (int)a1.x, m = (int)a1.y, kb = (int)a1.z, slow = (a1.w & 1u) != 0u, first_eq = (a1.w & 2u) != 0u;
                        st = 1, wide = false;
                    }
                    m_free &= ~(0xFFull << (8 * oct));
                    continue;
                }
                // the next staged pair: header {n, m, status, -} + packed words, in the lowest buffer of st_mask
                pf_buf   = (uint32_t)__builtin_ctz(st_mask);
                st_mask &= st_mask - 1u;
                const uint32_t this_idx = pf_idx;
                pf_idx += 1u;
                const uint32_t *const hb = lds + pf_buf * PW;
                // (readfirstlane: an LDS load counts as divergent, and a branch on it would turn this wave-uniform loop and
                // every scalar it updates -- the prefetch flags, the buffer masks -- into per-lane values)
                const uint32_t nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb[0]), mt = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb[1]);
                const uint32_t status = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb[2]);
                if (st_mask == 0u) pf &= ~PF_STAGED;
                if (status != ST_PENDING) {  // empty / too long / does not fit / a byte outside ACGT: no alignment here
                    if (lane == 0) {
                        P.pair_meta[this_idx] = make_uint4(status, 0u, 0u, 0u);
                        if (status >= ST_REDO_BYTES) push_redo(P, pair_of(this_idx), status);
                    }
                    buf_free |= 1u << pf_buf;
                    continue;  // (the half stays free; the next staged pair will take it)
                }
                WFA_EVT(5, 1);
                if (mine) {
                    pidx = this_idx, sbuf = pf_buf;
                    n = (int)nq, m = (int)mt;
                    const int Ak = m - n;
                    si = 0, cells = 0, slow = false;
                    kb = -16 + PP * imax2(-3, imin2(3, Ak / (2 * PP)));  // k = 0 (the seed) inside, biased towards Ak
                    first_eq = ((hb[4] ^ hb[4 + SW]) & 3u) == 0u;  // q[0] == t[0] (wfa.go:155)
                    clear_rings();
                    st = 1, wide = false;
                }
                m_free &= ~(0xFFull << (8 * oct));
            }
            set_derived();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        }
        WFA_STAMP(1);  // restructuring
        return m_free == ~0ull && park_used == 0u && (pf & (PF_DRY | PF_LD | PF_STAGED)) == PF_DRY;
    };

    // ================================================================ one score step of every running pair
    const auto step = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_c)::value;
        const bool run = (st == 1);
        const unsigned long long wm = __ballot(wide);  // lanes of the 16-lane pairs (DuoRed's fourth stage)
        WFA_EVT(0, 1), WFA_EVT(6, __builtin_popcountll(__ballot(run)) / 8);

        uint32_t(&Mo)[PP] = M[(ph + R - DOE) % R];  // M[s-o-e]
        uint32_t(&Mx)[PP] = M[(ph + R - DX) % R];   // M[s-x]
        uint32_t(&Mn)[PP] = M[ph];                  // the slot of the row being computed (the oldest row of the ring)

        // ------------------------------------------------------------ WF_NEXT (wfa.go:549-700)
        uint32_t nM[PP], nI[PP], nD[PP], wd[PP], cc[PP];
        const uint32_t a_edge = dn1(Mo[PP - 1]), b_edge = dn1(I[PP - 1]);
        const uint32_t c_edge = up1(Mo[0]), d_edge = up1(D[0]);
        const bool     slow_any = __ballot(run && slow) != 0ull;
        if (__builtin_expect(!slow_any, 1)) {
#pragma unroll
            for (int p = 0; p < PP; p++) {
                const uint32_t a = p ? Mo[p - 1] : a_edge, b = p ? I[p - 1] : b_edge;
                const uint32_t c = p < PP - 1 ? Mo[p + 1] : c_edge, d = p < PP - 1 ? D[p + 1] : d_edge;
                const uint32_t x = Mx[p];
                const uint32_t mi = umax2(a, b), Isk = mi + (mi != 0u ? 1u : 0u);  // wfa.go:579-609
                const uint32_t Dsk = umax2(c, d);                                   // wfa.go:614-645
                const uint32_t x1  = x + (x != 0u ? 1u : 0u);
                const uint32_t Msk = umax3(Isk, Dsk, x1);                           // wfa.go:655
                nM[p] = Msk, nI[p] = Isk, nD[p] = Dsk;
                wd[p] = blk_word_asm(Msk, a, b, c, d, x1, umax2(Isk, Dsk), Isk, Dsk);  // wfa.go:590-600,626-636,657-693
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
                const bool fromX = x != 0u && Msk == x1;  // wfa.go:657-693
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
        if (__builtin_expect(__ballot(run && si <= seed_si) != 0ull, 0)) {  // (ONE compare on the step's path; `want` below is the exact test)
            const bool want = run && ((si == 0u && first_eq) || (si == seed_si && !first_eq));
#pragma unroll
            for (int p = 0; p < PP; p++)
                if (want && k0 + p == 0 && nM[p] == 0u)
                    nM[p] = 1u, wd[p] = first_eq ? BLK_SEED_MATCH : BLK_SEED_MISMATCH, cc[p] = CENSUS ? 1u : 0u;
        }
        // ------------------------------------------------------------ store the row's words (CompactView fmt 7)
        // 16-bit words -- a pre-extension offset below 4 096 and the four decisions: this kernel's reads are under 2 048
        // bases -- in tiles of 8 scores x 64 diagonals (1 KB), inside a tile [diagonal / 4][score & 7][diagonal & 3]: a
        // lane's four diagonals are one 8-byte store, and a 64-byte piece of a line holds 8 scores of 4 diagonals.  Half
        // the bytes of fmt 3 to write here and to fetch in the backtrace's walk (round 3; WRITE_SIZE was 31 GB per
        // 1e6 pairs with 32-bit words).
        const bool no_room = run && (int)si >= rows_cap;
        {
            uint32_t anyc = 0u;
#pragma unroll
            for (int p = 0; p < PP; p++) anyc |= nM[p];
            // (fmt 10: score index si -> word (si & ~7) * 4 + (si & 7) * 2 = 2 (si + (si & ~7)) behind the lane's base)
            const uint32_t row_off = DUO_ARENA_FMT == 10u ? 2u * (si + (si & ~7u)) : DUO_ARENA_FMT == 9u ? 0u : (((uint32_t)k0 & 60u) << 2);
            if (run && !no_room && anyc != 0u)
                *reinterpret_cast<uint2 *>(rowp + row_off) = make_uint2(wd[0] | (wd[1] << 16), wd[2] | (wd[3] << 16));
        }
        WFA_STAMP(2);  // next + store

        // ------------------------------------------------------------ WF_EXTEND (wfa.go:381-458), first 16 bases
        uint32_t cmask = 0u;
#pragma unroll
        for (int p = 0; p < PP; p++) {
            const int      h    = (int)nM[p];
            uint32_t       rem;  // max(lim - h, 0): one saturating subtraction
            asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(rem) : "v"(lim[p]), "v"(h));
            const uint32_t room = h ? rem : 0u;
            const int      v    = h - (k0 + p);
            const uint32_t xr   = SeqView<0>::win16(lq, v) ^ SeqView<0>::win16(lt, h);
            const uint32_t rn   = umin2(ffbl_raw(xr) >> 1, room);
            nM[p] += umin2(rn, 16u);
            if (rn > 16u) cmask |= 1u << p;
        }
        while (__ballot(cmask != 0u) != 0ull) {
            const int psel = (int)ffbl_raw(cmask);
            int       h = 0, lm = 0;
#pragma unroll
            for (int p = 0; p < PP; p++)
                if (psel == p) h = (int)nM[p], lm = lim[p];
            const int kd = k0 + psel;
            bool      go = cmask != 0u;
            do {
                const int      rem = lm - h;
                const uint32_t xr  = SeqView<0>::win16(lq, h - kd) ^ SeqView<0>::win16(lt, h);
                const uint32_t cnt = umin2(ffbl_raw(xr) >> 1, (uint32_t)imin2(imax2(rem, 0), 16));
                h += go ? (int)cnt : 0;
                go = go && xr == 0u && rem > 16;
            } while (__ballot(go) != 0ull);
#pragma unroll
            for (int p = 0; p < PP; p++)
                if (psel == p) nM[p] = (uint32_t)h;
            cmask &= cmask - 1u;
        }
        WFA_STAMP(3);  // extend

        // ------------------------------------------------------------ ends reached? termination (wfa.go:235-239)
        bool nz[PP], hit[PP], hitl = false;
#pragma unroll
        for (int p = 0; p < PP; p++) nz[p] = nM[p] != 0u, hit[p] = nM[p] >= (uint32_t)lim[p], hitl |= hit[p];
        bool       term    = false;
        const bool hit_any = __ballot(hitl) != 0ull;
        bool       ghit    = false;
        if (hit_any) {
            bool tl = false;
#pragma unroll
            for (int p = 0; p < PP; p++) tl |= (k0 + p == m - n && nz[p] && (int)nM[p] >= m);
            const int r = DuoRed::or1((hitl ? 1 : 0) | (tl ? 2 : 0), wm);
            ghit = (r & 1) != 0;
            slow |= ghit;
            term = run && (r & 2) != 0;
        }

        // ------------------------------------------------------------ band of the row + wf-adaptive (wfa.go:461-540)
        int      ilo = 0, ihi = -1;  // band to keep (window-relative)
        bool     anyM = false;
        uint32_t csum = 0u;
        if (__builtin_expect(!hit_any, 1)) {
            // (an absent cell's distance is ABSENT, above every threshold: the tests below need no "exists and")
            constexpr int ABSENT = BK_BIG + 1;
            int glo = BK_BIG, ghi = -BK_BIG, dd[PP];
#pragma unroll
            for (int p = PP - 1; p >= 0; p--) glo = nz[p] ? PP * j + p : glo;
#pragma unroll
            for (int p = 0; p < PP; p++) {
                ghi   = nz[p] ? PP * j + p : ghi;
                dd[p] = nz[p] ? lmx[p] - (int)nM[p] : ABSENT;
            }
            static_assert(PP == 4, "minimum of four distances");
            int mind = imin2(imin2(dd[0], dd[1]), imin2(dd[2], dd[3]));
            DuoRed::min_max_min(glo, ghi, mind, wm);
            anyM = ghi >= 0;
            const bool want = run && adaptive && anyM && (ghi - glo + 1) >= minwf;
            const int  thr  = want ? mind + mdd : BK_BIG;
            int        first_ok = BK_BIG, last_ok = -BK_BIG;
#pragma unroll
            for (int p = PP - 1; p >= 0; p--) first_ok = dd[p] <= thr ? PP * j + p : first_ok;
#pragma unroll
            for (int p = 0; p < PP; p++) last_ok = dd[p] <= thr ? PP * j + p : last_ok;
            DuoRed::min_max(first_ok, last_ok, wm);
            ilo = first_ok, ihi = last_ok;
#pragma unroll
            for (int p = 0; p < PP; p++) {  // Delete of wfa.go:526-535: the words never exist
                const int  ix   = PP * j + p;
                const bool keep = ix >= ilo && ix <= ihi;
                nM[p] = keep ? nM[p] : 0u, nI[p] = keep ? nI[p] : 0u, nD[p] = keep ? nD[p] : 0u;
                csum += keep ? cc[p] : 0u;
            }
        } else {
            int glo = BK_BIG, ghi = -BK_BIG;
#pragma unroll
            for (int p = PP - 1; p >= 0; p--) glo = nz[p] ? PP * j + p : glo;
#pragma unroll
            for (int p = 0; p < PP; p++) ghi = nz[p] ? PP * j + p : ghi;
            DuoRed::min_max(glo, ghi, wm);
            anyM = ghi >= 0;
            ilo = glo, ihi = ghi;
            csum = 0u;
#pragma unroll
            for (int p = 0; p < PP; p++) csum += cc[p];
            const bool want_reduce = run && !term && adaptive && anyM && (ghi - glo + 1) >= minwf;
            if (__ballot(want_reduce) != 0ull) {
                int  dd[PP], mind = BK_BIG, maxd = -BK_BIG;
                bool vd[PP];
#pragma unroll
                for (int p = 0; p < PP; p++) {
                    vd[p] = nz[p] && !hit[p];
                    dd[p] = lmx[p] - (int)nM[p];
                    mind  = vd[p] ? imin2(mind, dd[p]) : mind;
                    maxd  = vd[p] ? imax2(maxd, dd[p]) : maxd;
                }
                DuoRed::min_max(mind, maxd, wm);
                const int  thr   = mind + mdd;
                const bool found = want_reduce && mind != BK_BIG && maxd > thr;  // some distance fails (wfa.go:507)
                if (__ballot(found) != 0ull) {
                    int first_ok = BK_BIG, last_ok = -BK_BIG;
#pragma unroll
                    for (int p = PP - 1; p >= 0; p--) first_ok = (vd[p] && dd[p] <= thr) ? PP * j + p : first_ok;
#pragma unroll
                    for (int p = 0; p < PP; p++) last_ok = (vd[p] && dd[p] <= thr) ? PP * j + p : last_ok;
                    DuoRed::min_max(first_ok, last_ok, wm);
                    // wfa.go:509-511 (see wfa_blk.hpp: per pair, not per wave)
                    int leadp = -1;
#pragma unroll
                    for (int p = 0; p < PP; p++) leadp = (vd[p] && PP * j + p < first_ok) ? PP * j + p : leadp;
                    leadp = DuoRed::max1(leadp, wm);
                    const int newlo = ghit ? (leadp >= 0 ? leadp + 1 : glo) : first_ok;
                    if (found) ilo = newlo, ihi = last_ok;  // wfa.go:517-524
                    csum = 0u;
#pragma unroll
                    for (int p = 0; p < PP; p++) {
                        const int  ix   = PP * j + p;
                        const bool keep = ix >= ilo && ix <= ihi;
                        nM[p] = keep ? nM[p] : 0u, nI[p] = keep ? nI[p] : 0u, nD[p] = keep ? nD[p] : 0u;
                        csum += keep ? cc[p] : 0u;
                    }
                }
            }
        }

        WFA_STAMP(4);  // ranges + wf-adaptive
        // ------------------------------------------------------------ the row's census and position
        const bool keepl = anyM && ihi >= ilo && !no_room;
        if constexpr (CENSUS) {  // (instrumentation instance only: the pair's total, the same in all its lanes)
            int cs = (int)csum, dummy = 0;
            DuoRed::max_add(dummy, cs, wm);
            cells += keepl ? (uint32_t)cs : 0u;
        }
        if constexpr (DUO_ARENA_FMT != 10u) rowp += 2;
        if constexpr (DUO_ARENA_FMT == 7u) rowp += (((uint32_t)(uintptr_t)rowp & 0x38u) == 0u) ? 240 : 0;  // past the tile's 8th score: next tile

        // ------------------------------------------------------------ the new row enters the rings
#pragma unroll
        for (int p = 0; p < PP; p++) Mn[p] = nM[p], I[p] = nI[p], D[p] = nD[p];
        rlo[ph] = keepl ? kb + ilo : BK_BIG;
        rhi[ph] = keepl ? kb + ihi : -BK_BIG;

        // ------------------------------------------------------------ finish / next score
        const bool fin = run && (term || no_room);
        if (__ballot(fin) != 0ull) {
            const int ctot = (CENSUS && P.census) ? (int)cells : 0;
            int hf   = 0;  // extended offset of the end cell M[s][Ak]: where the backtrace starts
#pragma unroll
            for (int p = 0; p < PP; p++)
                if (k0 + p == m - n) hf = (int)Mn[p];
            hf = DuoRed::max1(hf, wm);
            if (fin && j == 0) {
                if (no_room) {
                    P.pair_meta[pidx] = make_uint4(ST_REDO_ARENA, 0u, 0u, 0u);
                    push_redo(P, pair_of(pidx), ST_REDO_ARENA);
                } else {
                    P.pair_meta[pidx] = make_uint4(ST_OK, si * P.g, (uint32_t)hf, (uint32_t)ctot);
                }
            }
            buf_free |= or_halves(fin ? (1u << sbuf) : 0u);
            if (fin) {
                st = 0, wide = false;
                clear_rings();
            }
        }
        if (run && !fin) si += 1u;
        WFA_STAMP(5);  // ring + finish
    };

    for (;;) {
        if (refill(std::integral_constant<int, 0>{})) break;
        step(std::integral_constant<int, 0>{});
        if constexpr (R > 1) {
            if (refill(std::integral_constant<int, 1 % R>{})) break;
            step(std::integral_constant<int, 1 % R>{});
        }
        if constexpr (R > 2) {
            if (refill(std::integral_constant<int, 2 % R>{})) break;
            step(std::integral_constant<int, 2 % R>{});
        }
        if constexpr (R > 3) {
            if (refill(std::integral_constant<int, 3 % R>{})) break;
            step(std::integral_constant<int, 3 % R>{});
        }
    }
#ifdef WFA_STAMPS
    if (lane == 0 && P.debug_info) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(P.debug_info);
        for (int i = 0; i < 8; i++) atomicAdd(acc + i, stamp_acc[i]);
        for (int i = 0; i < 8; i++) atomicAdd(acc + 8 + i, evt[i]);
    }
#endif
}

}  // namespace wfa
