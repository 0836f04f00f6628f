// wfa_teamc.hpp -- kernel E2 (round 5): the team kernel for WIDE wavefronts with the blocked kernels' cell.
//
// wfa_team_kernel (rounds 1-4, wfa_team.hpp) keeps the reference's storage as it is: three words per diagonal and score
// (M, I, D: offset<<3 | tag, wfa_wavefront.go:93) in the arena, every cell of a step sourced from the arena rows of earlier
// scores.  A 100 kbp semi-global pair whose band does not collapse stores 7.5e9 words (30 GB) that way -- which is what
// bounded the number of teams in flight to eight --, every cell costs five loads, three stores and ~230 vector instructions,
// and the end-cell search (wfa.go:270-375) re-reads every M row when the alignment is over.  Here:
//   * ONE backtrace word per diagonal and score goes to the arena -- blk_word(): the pre-extension offset backTrace
//     recomputes (wfa.go:766-817) and the four decisions of next() (wfa_device.hpp) -- a third of the footprint and of the
//     stores; the walk is back_trace_compact(), one load per step instead of five;
//   * the rows the next steps source -- the last max(x, o+e)/g rows of M and e/g rows of I and D, as BARE extended offsets
//     (0 = absent, exactly Get's semantics: what wf-adaptive deletes is zeroed) -- never go to the arena.  In STRIPE mode
//     (rows that fit T x 4 096 diagonals) workgroup b keeps them for its contiguous stripe [KB + 4 096 b, + 4 096) of a fixed
//     diagonal axis in LDS, with one halo cell at either end; a cell is five LDS reads, the exact rules of next() in the
//     blocked kernels' form (lean_next(): ~45 instructions), WF_EXTEND, one coalesced 4-byte store;
//   * the two edge cells of a stripe travel through the team's exchange rows in global memory (`xbuf`: (R+1) + 2 (E+1) rows of
//     n + m words, indexed by diagonal).  Rows wider than the stripes (the first scores of a semi-global pair: n + m - 1
//     seeds) are stepped in XBUF mode -- cells dealt round-robin, sources from the exchange rows --, rows of at most 64
//     diagonals in WAVE mode (one wave, LDS ring, as before); every change of mode, and every move of the axis, goes through
//     the exchange rows (dump, barrier, load);
//   * the reductions of a step travel WITH its barrier: every workgroup stores its eight values with the step's sequence
//     number into its slot, and one wave per workgroup polls all T slots until they carry that number and reduces them --
//     one store and one round trip instead of seven atomics, an arrive, a spin and seven loads;
//   * the ends of the band wf-adaptive keeps (wfa.go:496-524) are found by the cells' owners from their registers and
//     combined by a second exchange (the arena no longer holds extended offsets for everybody to scan);
//   * the semi-global end cell (wfa.go:270-375) is found while the rows are in flight: every cell that ends the reference's
//     scan on its side of the final diagonal -- it leaves the matrix, or lies on the last row / column -- is combined per
//     score with an atomic minimum on (distance from that diagonal, hit or stop) in the score's directory entry; when the
//     alignment is over the lowest score whose nearest such cell is a hit is read off the directory.
// Same results as wfa_team_kernel, which stays for the penalties this kernel does not take (e = 0, rings too deep for LDS)
// and for wfahip_debug_wavefronts (Plot, the word-for-word tests of the three-word storage).
#pragma once
#include "wfa_device.hpp"
#include "wfa_team.hpp"

namespace wfa {

#ifndef WFA_TC_THREADS
#define WFA_TC_THREADS 1024
#endif
constexpr int      TC_THREADS  = WFA_TC_THREADS;
constexpr int      TC_STRIPE   = 4096;             // diagonals of a workgroup's stripe
constexpr int      TC_U        = TC_STRIPE / TC_THREADS;
constexpr int      TC_ROWW     = TC_STRIPE + 2;    // LDS words of a ring row: the stripe and a halo cell at either end
constexpr int      TC_SLOT_U64 = 16;               // a workgroup's exchange slot: (sequence number, value) words -- eight values to reduce, six edge cells
constexpr int      TC_RED      = 80;               // ints of the workgroup's scratch in LDS (red[])
constexpr int      TC_MAX_T    = 64;               // workgroups per team (one lane of the polling wave each)
// per team in global memory: the control words of wfa_team_kernel (TEAM_CTL_WORDS: [0] barrier count [1] abort [2] work index
// [3] XCC mask [4] command [5] score [6..7] arena top [8..9] - [10] end flags [11] on one XCD [12..13] stored cells
// [112..114] page hand-over [115] KB [116] mode) followed by two sets of exchange slots
constexpr int      TC_TRACE_OFF = TEAM_CTL_WORDS + 2 * TC_MAX_T * TC_SLOT_U64 * 2;  // then a progress word per workgroup: (score << 8) | phase
constexpr int      TC_CTL_WORDS = TC_TRACE_OFF + 8 * TC_MAX_T;                       // ... the sequence number of its latest exchange, and the latest
                                                                                     // word of a few kinds (slot s at TC_MAX_T * (2 + s)): mode changes, wave mode, the end of the pair
enum : uint32_t { TC_XBUF = 0, TC_STRIPE_T = 1, TC_STRIPE_S = 2, TC_WAVE = 3 };

struct TcArgs {
    uint32_t *team_ctl;
    uint32_t *xbuf;        // exchange rows, xbuf_words per team
    uint64_t  xbuf_words;
    uint32_t  xw;          // words of one exchange row (>= n + m of the longest pair)
    uint32_t  T, n_teams, tpx;  // workgroups per team; teams; teams per XCD (0: team = blockIdx / T)
    uint32_t  solo_max, wave_rows, strict, slack, fast;
    uint32_t  pipe;        // 1: the team's stripe steps overlap a row's exchanges with the next row's cells (the pipelined steps)
    uint32_t  scout;       // 1: teams of one workgroup that hand a pair on (ST_REDO_WIDE) as soon as a row past the seeds is wider than a stripe
    uint32_t *dbg;         // debug (one pair): [0] directory entries [1] final score [2] start score [3] start diagonal
};

// next() for one diagonal on BARE offsets (0 = absent), by its exact rules -- the rejections (wfa.go:581-588,616-623,651-654),
// mismatch-wins ties (wfa.go:657-693) and backTrace's recomputation of the pre-extension offset from the unrejected sources
// (wfa.go:766-817) -- in the blocked kernels' form: wd = blk_word() (wfa_device.hpp), 0 when the M cell does not exist.
struct LCell {
    uint32_t M, I, D, wd;
};
WFA_DEV LCell lean_next(uint32_t a0, uint32_t b0, uint32_t c0, uint32_t d0, uint32_t x0, int k, int n, int m) {
    const uint32_t a = (int)a0 > m ? 0u : a0, b = (int)b0 > m ? 0u : b0;
    const uint32_t c = (int)c0 - k > n ? 0u : c0, d = (int)d0 - k > n ? 0u : d0;
    const uint32_t x = ((int)x0 > m || (int)x0 - k > n) ? 0u : x0;
    const uint32_t mi = umax2(a, b), Isk = mi + (mi != 0u ? 1u : 0u);
    const uint32_t Dsk = umax2(c, d);
    const uint32_t x1  = x + (x != 0u ? 1u : 0u);
    const uint32_t Msk = umax2(umax2(Isk, Dsk), x1);
    const bool     fromX = x != 0u && Msk == x1;
    const bool     fromI = !fromX && Msk == Isk;
    const uint32_t mu = umax2(a0, b0), Iu = mu + (mu != 0u ? 1u : 0u), Du = umax2(c0, d0), Xu = x0 + (x0 != 0u ? 1u : 0u);
    const bool     iext = a < b, dext = c < d;
    const uint32_t o0   = (fromI && iext) ? Iu : ((!fromX && !fromI && dext) ? Du : umax2(umax2(Iu, Du), Xu));
    LCell r;
    r.M = Msk, r.I = Isk, r.D = Dsk;
    r.wd = Msk != 0u ? blk_word(o0, iext, dext, fromX, fromI) : 0u;
    return r;
}
// The same where no source of any of the wave's cells lies past a sequence end (a ballot: all but the waves at the matrix's
// borders): nothing is rejected, the unrejected sources are the sources, and the offset backTrace would recompute is the M offset
// itself (from I: Iu = Isk = Msk; from D: Du = Dsk = Msk; else the maximum of the three, Msk).
WFA_DEV bool lean_rejects(uint32_t a0, uint32_t b0, uint32_t c0, uint32_t d0, uint32_t x0, int k, int n, int m) {
    const uint32_t hmax = umax2(umax2(a0, b0), x0), vsrc = umax2(umax2(c0, d0), x0);
    return (int)hmax > m || (int)vsrc - k > n;
}
WFA_DEV LCell lean_next_norej(uint32_t a0, uint32_t b0, uint32_t c0, uint32_t d0, uint32_t x0) {
    const uint32_t mi = umax2(a0, b0), Isk = mi + (mi != 0u ? 1u : 0u), Dsk = umax2(c0, d0), x1 = x0 + (x0 != 0u ? 1u : 0u);
    const uint32_t Msk = umax2(umax2(Isk, Dsk), x1);
    const bool     fromX = x0 != 0u && Msk == x1, fromI = !fromX && Msk == Isk;
    LCell r;
    r.M = Msk, r.I = Isk, r.D = Dsk;
    r.wd = Msk != 0u ? blk_word(Msk, a0 < b0, c0 < d0, fromX, fromI) : 0u;
    return r;
}
WFA_DEV LCell lean_next_fast(uint32_t a0, uint32_t b0, uint32_t c0, uint32_t d0, uint32_t x0, int k, int n, int m) {
    if (__ballot(lean_rejects(a0, b0, c0, d0, x0, k, n, m)) != 0ull) return lean_next(a0, b0, c0, d0, x0, k, n, m);
    return lean_next_norej(a0, b0, c0, d0, x0);
}
// the seed of initComponents on diagonal k that belongs to score s, as a bare offset (0: none); *match: its class
template <int MODE>
WFA_DEV uint32_t lean_seed(const SeqView<MODE> &sv, int k, uint32_t s, uint32_t x, bool glob, bool &match) {
    const uint32_t w = seed_word<MODE>(sv, k, s, x, glob);
    match            = (w & TAG_MASK) == TAG_MATCH;
    return w >> TAG_BITS;
}
// WF_EXTEND on a bare offset (wfa.go:394-455)
template <int MODE>
WFA_DEV uint32_t lean_extend(const SeqView<MODE> &sv, uint32_t h, int k) {
    const int v = (int)h - k;
    if (h == 0u || v <= 0 || v >= sv.n || (int)h >= sv.m) return h;
    return h + (uint32_t)sv.lcp(v, (int)h);
}
// remaining distance of wf-adaptive on a bare offset (wfa.go:474-494): -1 when absent or at / past a sequence end
WFA_DEV int lean_dist(uint32_t h, int k, int n, int m) {
    const int v = (int)h - k;
    if (h == 0u || v < 0 || v >= n || (int)h >= m) return -1;
    return imax2(m - (int)h, n - v);
}
// backtraceStartPosistion's view of a cell (wfa.go:301-361): 0 = the scan passes it, 1 = it ends the scan (the cell leaves the
// matrix), 2 = hit through v == n (h = n + k), 3 = hit through h == m
WFA_DEV uint32_t lean_endclass(uint32_t h, int k, int n, int m) {
    if (h == 0u) return 0u;
    const int v = (int)h - k;
    if (v <= 0 || v > n || (int)h > m) return 1u;
    if (v == n && (int)h >= n) return 2u;
    if ((int)h == m && v >= m) return 3u;
    return 0u;
}
// key of a scan-ending cell on its side of the final diagonal: (distance << 2) | 0 / 2 / 3 for a hit (class - 1 ... 1 = stop sorts last)
WFA_DEV uint32_t lean_endkey(uint32_t cls, uint32_t dist) { return (dist << 2) | (cls == 1u ? 3u : (cls == 2u ? 0u : 1u)); }

// The compact rows of the arena seen by a wave that walks the backtrace together: directory entries of a 64-score window in
// LDS (as ArenaViewWave), one word per cell.
struct DirCompactViewWave {
    const uint32_t *A;
    uint64_t        cap;
    uint32_t        g, n_ent;
    DirEnt         *win;
    mutable uint32_t win_lo, win_hi;
    WFA_DEV void init(const uint32_t *A_, uint64_t cap_, uint32_t g_, uint32_t n_ent_, DirEnt *lds_win) {
        A = A_, cap = cap_, g = g_, n_ent = n_ent_, win = lds_win, win_lo = 1u, win_hi = 0u;
    }
    WFA_DEV void refill(uint32_t top, int k) const {
        const uint32_t hi = top < n_ent ? top : n_ent - 1u;
        const uint32_t lo = hi >= 63u ? hi - 63u : 0u;
        const uint32_t j  = lo + (uint32_t)(threadIdx.x & 63);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (j <= hi) {
            const DirEnt d = load_dir(A + cap - (uint64_t)DIR_WORDS * (j + 1));
            win[j & 63u]   = d;
            if (d.w > 0) {  // the cells the walk can reach while the window lasts: their lines in one round trip
                const int k0 = k - 66 > d.lo ? k - 66 : d.lo, k1 = k + 66 < d.lo + d.w - 1 ? k + 66 : d.lo + d.w - 1;
                uint32_t  acc = 0;
                const uint32_t *row = A + d.base - d.lo;
                for (int kk = k0; kk <= k1; kk += 32) acc ^= row[kk];
                if (k1 >= k0) acc ^= row[k1];
                asm volatile("" ::"v"(acc));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        win_lo = lo, win_hi = hi;
    }
    WFA_DEV uint32_t word(uint32_t idx, int k) const {
        if (idx >= n_ent) return 0u;
        if (WFA_RARE(idx < win_lo || idx > win_hi)) refill(idx, k);
        const DirEnt e = win[idx & 63u];
        const bool   ok = e.w > 0 && k >= e.lo && k < e.lo + e.w;
        const uint32_t v = A[ok ? e.base + (uint32_t)(k - e.lo) : 0ull];
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)(ok ? v : 0u));
    }
    WFA_DEV uint32_t tag(int comp, uint32_t idx, int k, uint32_t &off0) const { return blk_tag(word(idx, k), comp, off0); }
};

template <int MODE>
__global__ __launch_bounds__(TC_THREADS) void wfa_teamc_kernel(const KParams P, const TcArgs X) {
    constexpr int G = TC_THREADS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *const lq   = lds;
    uint32_t *const lt   = lds + P.lds_seq_words;
    int *const      red  = reinterpret_cast<int *>(lds + 2 * (MODE == 0 ? P.lds_seq_words : 0));  // TC_RED ints
    DirEnt *const   ring = reinterpret_cast<DirEnt *>(red + TC_RED);                               // TEAM_RING entries
    uint32_t *const wring = reinterpret_cast<uint32_t *>(ring + TEAM_RING);                       // wave mode: [row][component][diagonal & 63]
    uint32_t *const wsc   = wring + X.wave_rows * 3u * 64u;                                       // stripe mode: wave 0's backtrace words of the row (64 x TC_U)
    uint32_t *const lrows = wsc + 64 * TC_U;                                                      // stripe mode: ring rows of TC_ROWW words

    const int      tid = threadIdx.x, lane = tid & 63;
    const uint32_t T   = X.T;
    // teams of one XCD's workgroups (tpx teams per XCD: workgroups are dealt round-robin over the eight XCDs, so those with
    // the same blockIdx % 8 share an L2), or plainly consecutive workgroups
    uint32_t team, b;
    if (X.tpx != 0u) {
        const uint32_t xcd = blockIdx.x % 8u, ix = blockIdx.x / 8u;
        team = xcd * X.tpx + ix / T, b = ix % T;
        if (ix / T >= X.tpx) return;
    } else {
        team = blockIdx.x / T, b = blockIdx.x % T;
    }
    if (team >= X.n_teams) return;
    uint32_t *const ctl = X.team_ctl + (uint64_t)team * TC_CTL_WORDS;
    unsigned long long *const slots = reinterpret_cast<unsigned long long *>(ctl + TEAM_CTL_WORDS);  // [set][workgroup][TC_SLOT_U64]
    uint32_t *const xb  = X.xbuf + (uint64_t)team * X.xbuf_words;
    // where every workgroup is: written by its thread 0 as it goes (one fire-and-forget store per phase), printed by the host when a
    // barrier has timed out -- a hang then names the score and the phase of each workgroup instead of nothing
    uint32_t *const trace = ctl + TC_TRACE_OFF;
#define TC_ABORT_RET                                                                                                    \
    do {                                                                                                                \
        if (tid == 0) __hip_atomic_store(trace + b, 0xAB000000u | (uint32_t)__LINE__, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        return;                                                                                                         \
    } while (0)
    // (-DWFA_TC_TRACE: a store in front of a barrier is waited for by it -- the trace costs a few hundred nanoseconds per step -- so the
    // shipped build only records where a workgroup LEFT after a barrier that timed out)
#ifdef WFA_TC_TRACE
#define TC_TRACEN(slot, sc, ph)                                                   \
    do {                                                                          \
        if (tid == 0) __hip_atomic_store(trace + TC_MAX_T * (2 + (slot)) + b, ((uint32_t)(sc) << 8) | (uint32_t)(ph), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    } while (0)
#define TC_TRACE(sc, ph)                                                          \
    do {                                                                          \
        if (tid == 0) __hip_atomic_store(trace + b, ((uint32_t)(sc) << 8) | (uint32_t)(ph), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    } while (0)
#else
#define TC_TRACEN(slot, sc, ph) \
    do {                        \
    } while (0)
#define TC_TRACE(sc, ph) \
    do {                 \
    } while (0)
#endif
    const bool      paged = P.page_ctl != nullptr;
    uint32_t *const A     = paged ? P.arena : P.arena + (uint64_t)team * P.arena_words;
    const uint64_t  cap   = paged ? P.arena_words - (uint64_t)team * P.dir_region_words : P.arena_words;
    const uint32_t  dir_entries = paged ? (uint32_t)(P.dir_region_words / DIR_WORDS) : 0u;
    const uint64_t  page_words  = 1ull << P.page_words_log2;
    uint32_t *const my_pages    = paged ? P.page_ctl + 4u + P.n_pages + team * (uint32_t)TEAM_MAX_PAGES : nullptr;
    const uint32_t x = P.x, oe = P.oe, e = P.e, g = P.g;
    const uint32_t dx = x / g, doe = oe / g, de = e / g;
    const uint32_t RM = dx > doe ? dx : doe, RE = de;  // rows of the M ring / of the I and D rings (LDS; the exchange rows hold one more each)
    const bool     strict = (X.strict & 1u) != 0u, xl_ok = (X.strict & 4u) != 0u;

    uint32_t bar_target = 0, xseq = 0;
    bool     aborted = false, xl = false;
    // Workgroup barrier for LDS data only: __syncthreads() also waits for every global store the wave has issued (s_waitcnt
    // vmcnt(0) in front of s_barrier), and a step has nine barriers -- the row's backtrace words, which nobody reads before the
    // pair's last (fenced) barrier, would be waited for at the next of them.  Where global data IS handed over (the stripe's edge
    // cells, rows in XBUF mode) the storing waves wait for their stores themselves, in front of the exchange.
    const auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#ifdef WFA_TEAM_STAMPS  // diagnostic build (scripts/team_stamps.sh): time per phase of workgroup 0's wave 0, summed over the team's pairs
    unsigned long long tacc[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memrealtime();
#define TC_STAMP(i)                                                     \
    do {                                                                \
        const unsigned long long _t = __builtin_amdgcn_s_memrealtime(); \
        tacc[i] += _t - tprev;                                          \
        tprev = _t;                                                     \
    } while (0)
#define TC_COUNT(i) (tacc[i]++)
#else
#define TC_STAMP(i) asm volatile("; TC_STAMP " #i)  // (a comment in the assembly listing: where a phase ends)
#define TC_COUNT(i) \
    do {            \
    } while (0)
#endif
    const auto ald = [](const uint32_t *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const auto ast = [](uint32_t *p_, uint32_t v_) { __hip_atomic_store(p_, v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // counted barrier with fences (mode switches, the start and the end of a pair): wfa_team_kernel's
    auto team_barrier = [&]() __attribute__((always_inline)) {
        __syncthreads();
        if (tid == 0) {
            bar_target += T;
            __threadfence();
            bool bad = false;
            if ((int32_t)(atomicAdd(&ctl[0], 1u) + 1u - bar_target) < 0) {
                uint32_t spins = 0;
                while ((int32_t)(ald(&ctl[0]) - bar_target) < 0) {
                    if ((++spins & 1023u) == 0u && (spins > TEAM_SPIN_LIMIT || ald(&ctl[1]) != 0u)) {
                        atomicExch(&ctl[1], 1u);
                        bad = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __threadfence();
            red[46] = bad ? 1 : 0;
        }
        __syncthreads();
        aborted = red[46] != 0;
    };
    // EXCHANGE: barrier + reduction in one.  Every workgroup hands in eight ints (red[16 .. 23], written by its thread 0 before
    // the call); when the call returns red[16 .. 23] hold the MINIMUM of each over the team (a maximum travels as its complement, a flag
    // as 0 / -1).  Thread 0 stores the eight values tagged with this exchange's sequence number into the workgroup's slot (eight
    // 64-bit relaxed atomic stores: no word can be seen half-written); wave 0 polls the T slots, a workgroup per lane, until
    // every word carries the number, and reduces.  Two sets of slots alternate: a workgroup can only overwrite a set two exchanges
    // later, which it reaches only after every workgroup has read this one.  What the workgroup stored before the call -- the
    // edge cells of its stripe in the exchange rows, rows in XBUF mode -- is ordered before the slot by the release (memory-side
    // protocol), or is in the XCD's L2 once the stores have been acknowledged, which the barrier in front waits for (teams on
    // one XCD; see wfa_team.hpp).
    // edges: the payload also carries the six words of the stripe's first and last cell (red[24 .. 26] and red[28 .. 30], written by their owners
    // before the call: M, I, D of the first, then of the last), and every workgroup gets its neighbours' -- the halo cells of its
    // stripe -- back in red[40 .. 45] (M, I, D below the stripe, then above it; zero at the team's ends): the k +- 1 sources that
    // cross a stripe's edge (wfa.go:579-650) travel with the barrier, no load, no store and no wait of their own.
    // exchange_core: the exchange without the barrier in front (the caller has ordered red[16 .. 30] before wave 0's
    // stores itself); nf = how many of the eight values travel.
    // A slot is 128 bytes, sixteen (sequence number, value) words: [0 .. 7] the values, [8 .. 10] the stripe's first cell (M, I, D),
    // [12 .. 14] its last.  Lane f of wave 0 stores word f -- one store instruction, one line.  Polling, lane l loads 16-byte piece
    // l % 4 (values 2 (l % 4) and the next) of workgroup l / 4's slot, and of workgroup 16 + l / 4's: four lanes make a 64-byte
    // request, a poll of 32 workgroups is 32 requests (a load per lane and value was 256, and the compiler waits behind every atomic
    // load of its own -- fourteen round trips a poll; here ALL the loads of a poll are in flight before the one wait).  Lanes 0 .. 3
    // also load the two pieces of each neighbour's edge cell.  The minimum of a value over the workgroups: two DPP steps inside each
    // row of sixteen lanes, then one LDS atomic per row into red[16 + value] (reset by this wave after it has read its own values).
    auto exchange_core = [&](int nf, bool edges, bool wgb = true) __attribute__((always_inline)) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        xseq += 1u;
#ifdef WFA_TC_TRACE
        if (tid == 0) __hip_atomic_store(trace + TC_MAX_T + b, xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        if (T != 1u && tid < 64) {  // (a team of one: the values are the reduction)
            unsigned long long *const set = slots + (size_t)(xseq & 1u) * TC_MAX_T * TC_SLOT_U64;
            {
                const bool sends = lane < 16 && (lane < 8 ? lane < nf : edges);
                // (red[16 .. 31] in the slot's order: the values, the first cell at [24 .. 26], the last at [28 .. 30]; [27] and [31] are padding)
                const uint32_t mine_v = sends ? (uint32_t)red[16 + lane] : 0u;
                if (lane < 8) red[16 + lane] = INT32_MAX;  // (the minima are collected here)
                if (sends) {
                    if (strict && !xl) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    // (a team on one XCD: plain stores -- the words stay in that XCD's L2, where the others' polling loads find them;
                    // a write-through store would drop them from it and send every poll to the memory side)
                    const unsigned long long wv = ((unsigned long long)xseq << 32) | mine_v;
                    if (xl) set[(size_t)b * TC_SLOT_U64 + lane] = wv;
                    else __hip_atomic_store(set + (size_t)b * TC_SLOT_U64 + lane, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (nf == 8) TC_STAMP(24);  // exchange 1 in detail: the slot stored
            const uint32_t pc = (uint32_t)lane & 3u;
            // lanes 0, 1: the last cell of workgroup b - 1 (words 12 .. 15); lanes 2, 3: the first cell of workgroup b + 1 (words 8 .. 11)
            const int  nb     = lane < 2 ? (int)b - 1 : (int)b + 1;
            const bool has_nb = edges && lane < 4 && nb >= 0 && nb < (int)T;
            u32x4      we     = {0u, 0u, 0u, 0u};
            uint32_t   spins  = 0;
            bool       bad    = false;
            int        va = INT32_MAX, vb = INT32_MAX;
            // (32 workgroups a round: a team of more polls the second half of the slots in a second round; a set has room for TC_MAX_T slots,
            // so a load beyond the team's last slot reads memory nobody writes, and is ignored)
            for (uint32_t base = 0; base < T && !bad; base += 32u) {
                const uint32_t sl = base + ((uint32_t)lane >> 2);
                const unsigned long long *const p0 = set + (size_t)sl * TC_SLOT_U64 + 2u * pc;
                const unsigned long long *const pe = (has_nb && base == 0u) ? set + (size_t)nb * TC_SLOT_U64 + (lane < 2 ? 12 + 2 * lane : 8 + 2 * (lane - 2)) : p0;
                const unsigned long long emask = __ballot(has_nb && base == 0u);
                u32x4 w0, w1;
                for (;;) {
                    unsigned long long sv_;
                    asm volatile(
                        "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:2048 sc1\n\t"
                        "s_mov_b64 %3, exec\n\ts_and_b64 exec, exec, %6\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_mov_b64 exec, %3\n\ts_waitcnt vmcnt(0)"
                        : "=&v"(w0), "=&v"(w1), "+v"(we), "=&s"(sv_)
                        : "v"(p0), "v"(pe), "s"(emask)
                        : "memory", "scc");
                    // the piece's two values: 2 pc and 2 pc + 1
                    const bool c0 = 2u * pc < (uint32_t)nf, c1 = 2u * pc + 1u < (uint32_t)nf;
                    bool ok = (!c0 || w0.y == xseq || sl >= T) && (!c1 || w0.w == xseq || sl >= T) && (!c0 || w1.y == xseq || sl + 16u >= T) && (!c1 || w1.w == xseq || sl + 16u >= T);
                    if (has_nb && base == 0u) ok = ok && we.y == xseq && we.w == xseq;
                    if (nf == 8) TC_COUNT(27);
                    if (__ballot(!ok) == 0ull) break;
                    if ((++spins & 255u) == 0u && (spins > (TEAM_SPIN_LIMIT >> 2) || ald(&ctl[1]) != 0u)) {
                        atomicExch(&ctl[1], 1u);
                        bad = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (sl < T) va = imin2(va, (int)w0.x), vb = imin2(vb, (int)w0.z);
                if (sl + 16u < T) va = imin2(va, (int)w1.x), vb = imin2(vb, (int)w1.z);
            }
            if (nf == 8) TC_STAMP(25);  // ... every slot read
            // the four workgroups of a row that hold the same piece: lanes pc, pc + 4, pc + 8, pc + 12 -> lanes 12 .. 15 of the row
            asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\ts_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1" : "+v"(va));
            asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\ts_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1" : "+v"(vb));
            if ((lane & 12) == 12) {
                if (2u * pc < (uint32_t)nf) atomicMin(&red[16 + 2 * (int)pc], va);
                if (2u * pc + 1u < (uint32_t)nf) atomicMin(&red[17 + 2 * (int)pc], vb);
            }
            if (edges && lane < 4) {
                // the halo cells of this workgroup's stripe: M, I, D below it (red[40 .. 42]) and above it (red[43 .. 45]); zero at the team's ends
                const int ex = has_nb ? (int)we.x : 0, ez = has_nb ? (int)we.z : 0;
                if (lane == 0) red[40] = ex, red[41] = ez;
                if (lane == 1) red[42] = ex;
                if (lane == 2) red[43] = ex, red[44] = ez;
                if (lane == 3) red[45] = ex;
            }
            if (lane == 0) red[46] = bad ? 1 : 0;
            if (nf == 8) TC_STAMP(26);  // ... reduced
            if (!wgb) {  // (the pipelined steps: wave 0 exchanges on its own, the others are computing cells)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                aborted = bad;
            }
        }
        if (wgb) {
            lds_barrier();
            aborted = red[46] != 0;
        }
    };
    auto exchange = [&](bool drain, bool edges) __attribute__((always_inline)) {
        if (drain) __syncthreads();  // (every wave's global stores acknowledged: rows in the exchange rows)
        else lds_barrier();          // (nothing global is handed over)
        exchange_core(8, edges);
    };

    // ---- where the team's workgroups sit (teams per XCD): one bit per XCC id seen
    if (X.tpx != 0u) {
        if (tid == 0) atomicOr(&ctl[3], 1u << (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 15u));
        team_barrier();
        if (aborted) TC_ABORT_RET;
        const uint32_t seen = ald(&ctl[3]);
        xl = xl_ok && (seen & (seen - 1u)) == 0u;
        if (b == 0 && tid == 0) ctl[11] = xl ? 1u : 0u;
    }
    // ---- pages (wfa_team_kernel's pool: a stack of free page ids behind a spin lock)
    uint32_t n_pg = 0;
    auto page_lock = [&]() -> bool {
        uint32_t spins = 0;
        for (;;) {
            uint32_t expect = 0u;
            if (__hip_atomic_compare_exchange_strong(&P.page_ctl[0], &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
            if (++spins > TEAM_SPIN_LIMIT) return false;
            __builtin_amdgcn_s_sleep(8);
        }
    };
    auto page_unlock = [&]() { __hip_atomic_store(&P.page_ctl[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); };
    auto page_alloc = [&]() -> uint32_t {
        if (n_pg >= (uint32_t)TEAM_MAX_PAGES) return 0xFFFFFFFFu;
        uint32_t pg = 0xFFFFFFFFu, spins = 0;
        bool     waiting = false;
        for (;;) {
            if (!page_lock()) break;
            const uint32_t nf = ald(&P.page_ctl[1]);
            if (nf != 0u) pg = ald(&P.page_ctl[4u + nf - 1u]), ast(&P.page_ctl[1], nf - 1u);
            page_unlock();
            if (pg != 0xFFFFFFFFu) break;
            if (!waiting) waiting = true, atomicAdd(&P.page_ctl[2], 1u);
            const uint32_t holders = ald(&P.page_ctl[3]), waiters = ald(&P.page_ctl[2]);
            const uint32_t others_holding = holders - (n_pg != 0u ? 1u : 0u), others_waiting = waiters - 1u;
            if (others_waiting >= others_holding || ++spins > (1u << 20)) break;
            __builtin_amdgcn_s_sleep(64);
        }
        if (waiting) atomicSub(&P.page_ctl[2], 1u);
        if (pg != 0xFFFFFFFFu) {
            if (n_pg == 0u) atomicAdd(&P.page_ctl[3], 1u);
            my_pages[n_pg++] = pg;
        }
        return pg;
    };
    auto page_free_all = [&]() {
        if (n_pg == 0u) return;
        if (page_lock()) {
            uint32_t nf = ald(&P.page_ctl[1]);
            for (uint32_t i = 0; i < n_pg; i++) ast(&P.page_ctl[4u + nf++], my_pages[i]);
            ast(&P.page_ctl[1], nf);
            page_unlock();
        }
        atomicSub(&P.page_ctl[3], 1u);
        n_pg = 0u;
    };

    for (;;) {
        // ---- the team's next pair
        if (b == 0 && tid == 0) {
            const uint32_t w0 = atomicAdd(P.queue_head, 1u);
            ast(&ctl[2], w0), ast(&ctl[12], 0u), ast(&ctl[13], 0u), ast(&ctl[4], (uint32_t)TEAM_CMD_NONE);
        }
        team_barrier();
        if (aborted) TC_ABORT_RET;
        const uint32_t wi = ald(&ctl[2]);
        if (wi >= P.n_work) return;
        const uint32_t pair = P.work ? P.work[wi] : wi;
        uint32_t *const rec = P.rec + (uint64_t)pair * REC_WORDS;
        const bool      lead_wg = (b == 0);

        const uint32_t nq = P.q_len[pair], mt = P.t_len[pair];
        if (nq == 0 || mt == 0 || nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu) {  // wfa.go:204-209
            if (lead_wg && tid < REC_WORDS) rec[tid] = (tid == REC_STATUS) ? ((nq == 0 || mt == 0) ? ST_EMPTY : ST_TOO_LONG) : 0u;
            continue;
        }
        const int n = (int)nq, m = (int)mt, Ak = m - n;
        if ((uint64_t)nq + mt > (uint64_t)X.xw || nq >= (1u << 27) || mt >= (1u << 27)) {  // (cannot happen: the host sizes the exchange rows by the longest pair)
            if (lead_wg && tid == 0) rec[REC_STATUS] = ST_REDO_ARENA, push_redo(P, pair, ST_REDO_ARENA);
            continue;
        }

        SeqView<MODE> sv;
        sv.n = n, sv.m = m;
        if constexpr (MODE == 0) {
            const uint32_t need = ((imax2(n, m) + 15) >> 4) + 1;
            if (need > P.lds_seq_words) {
                if (lead_wg && tid == 0) rec[REC_STATUS] = ST_REDO_LDS, push_redo(P, pair, ST_REDO_LDS);
                continue;
            }
            if (tid == 0) red[9] = 0;
            __syncthreads();
            bool bad = stage_pack<G>(P.blob, P.q_off[pair], nq, lq, tid);
            bad |= stage_pack<G>(P.blob, P.t_off[pair], mt, lt, tid);
            if (__ballot(bad) != 0ull && lane == 0) red[9] = 1;
            __syncthreads();
            if (red[9]) {
                if (lead_wg && tid == 0) rec[REC_STATUS] = ST_REDO_BYTES, push_redo(P, pair, ST_REDO_BYTES);
                continue;
            }
            sv.q = lq, sv.t = lt;
        } else {
            sv.q = P.blob + P.q_off[pair];
            sv.t = P.blob + P.t_off[pair];
        }

        // ---- score loop
        const bool glob    = P.global_alignment != 0;
        const int  seed_lo = glob ? 0 : -(n - 1), seed_hi = glob ? 0 : m - 1;
        const int  xoff    = n - 1;  // exchange rows are indexed by k + n - 1
        uint64_t   top = 0, page_end = 0;
        uint32_t   n_ent = 0, s_final = 0, h_final = 0;
        bool       overflow = false, done = false, too_wide = false;
        uint64_t   my_cells = 0;
        uint32_t   mode = TC_XBUF;  // every workgroup active: XBUF / STRIPE_T; workgroup 0 alone: STRIPE_S / WAVE
        int        KB   = 0;        // stripe modes: first diagonal of workgroup 0's stripe
        int        SWd  = TC_STRIPE;  // ... and the diagonals of a stripe: the row's width dealt evenly over the team's workgroups (a multiple of 64, at
                                      // most TC_STRIPE) -- every workgroup then has the same number of cells per thread, and the step takes what its
                                      // busiest workgroup takes
        bool       rings_in_lds = false;  // the LDS rings hold the rows of the scores before s (else they are in the exchange rows)
        auto dir_ptr = [&](uint32_t idx) { return A + cap - (uint64_t)DIR_WORDS * (idx + 1); };
        const DirEnt none = {0ull, 0, 0, 0u, {0u, 0u, 0u}};
        auto put_ent = [&](uint32_t idx, uint64_t base, int lo_, int w_) __attribute__((always_inline)) {
            if (tid == 0) {
                DirEnt d;
                d.base = base, d.lo = lo_, d.w = w_, d.stride = 0u, d.pad[0] = d.pad[1] = d.pad[2] = 0u;
                ring[idx % TEAM_RING] = d;
                if (lead_wg) store_dir(dir_ptr(idx), base, lo_, w_, 0u);
            }
        };
        // exchange rows: M row of score index i at slot i mod (RM + 1), I / D rows at i mod (RE + 1) behind them
        auto xrow = [&](int comp, uint32_t idx) {
            return xb + (size_t)(comp == 0 ? idx % (RM + 1u) : (RM + 1u) + (uint32_t)(comp - 1) * (RE + 1u) + idx % (RE + 1u)) * X.xw;
        };
        auto lrow = [&](int comp, uint32_t idx) {
            return lrows + (size_t)(comp == 0 ? idx % RM : RM + (uint32_t)(comp - 1) * RE + idx % RE) * TC_ROWW;
        };
        auto xld = [&](const uint32_t *p_) { return (mode == TC_XBUF || mode == TC_STRIPE_T) ? ald(p_) : *p_; };
        auto xst = [&](uint32_t *p_, uint32_t v_) {
            if ((mode == TC_XBUF || mode == TC_STRIPE_T) && !xl) ast(p_, v_);
            else *p_ = v_;
        };
        auto cst = [&](uint32_t *p_, uint32_t v_) { *p_ = v_; };  // compact words: nobody reads them before the pair's last (fenced) barrier
        // the rings of this workgroup's stripe -> exchange rows (every cell of the stripe: what lies outside a row's band is zero)
        auto dump_rings = [&](uint32_t si) {
            const int KBw = KB + (mode == TC_STRIPE_T ? (int)b * SWd : 0);
            for (uint32_t r = 1; r <= RM && r <= si; r++) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (c != 0 && r > RE) break;
                    const uint32_t *const src = lrow(c, si - r);
                    uint32_t *const       dst = xrow(c, si - r);
#pragma unroll
                    for (int u = 0; u < TC_U; u++) {
                        const int j = tid + u * G, kx = KBw + j + xoff;
                        if (j < SWd && kx >= 0 && kx < (int)X.xw) xst(dst + kx, src[j + 1]);
                    }
                }
            }
        };
        // exchange rows -> the rings of this workgroup's stripe (cells outside a row's kept band: zero; halo cells included)
        auto load_rings = [&](uint32_t si) {
            const int KBw = KB + (mode == TC_STRIPE_T ? (int)b * SWd : 0);
            for (uint32_t r = 1; r <= RM && r <= si; r++) {
                const DirEnt d = ring[(si - r) % TEAM_RING];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (c != 0 && r > RE) break;
                    const uint32_t *const src = xrow(c, si - r);
                    uint32_t *const       dst = lrow(c, si - r);
                    for (int j = tid - 1; j <= SWd; j += G) {
                        const int k = KBw + j;
                        dst[j + 1]  = (d.w > 0 && k >= d.lo && k < d.lo + d.w) ? xld(src + k + xoff) : 0u;
                    }
                }
            }
        };

        // the mode a row of W diagonals at score s_ asks for (every active workgroup computes the same W)
        auto want_mode = [&](int64_t W, uint32_t s_) __attribute__((always_inline)) -> uint32_t {
            const int64_t capT = (int64_t)T * TC_STRIPE - 2, capS = (int64_t)TC_STRIPE - 2;
            uint32_t want = W > capT ? (uint32_t)TC_XBUF
                            : (W > (int64_t)X.solo_max || W > capS) && T > 1u ? (uint32_t)TC_STRIPE_T
                            : (W <= 64 && X.wave_rows != 0u && s_ != 0u) ? (uint32_t)TC_WAVE
                                                                        : (W > capS ? (uint32_t)TC_XBUF : (uint32_t)TC_STRIPE_S);
            if (T == 1u && want == TC_STRIPE_T) want = W > capS ? (uint32_t)TC_XBUF : (uint32_t)TC_STRIPE_S;
            return want;
        };
        // a stripe mode whose axis no longer holds the row [lo, hi] is left and entered again; so is a team whose stripes have
        // become 128 diagonals wider than the row needs (the step takes what the workgroup with the most cells per thread takes;
        // SWd - 128 is a multiple of 64: "the width a team would pick now is at least 128 smaller" without a division per step)
        auto moved_now = [&](int lo, int hi, int64_t W) __attribute__((always_inline)) -> bool {
            if (mode != TC_STRIPE_T && mode != TC_STRIPE_S) return false;
            const int64_t capd = mode == TC_STRIPE_T ? (int64_t)T * SWd : (int64_t)SWd;
            const int64_t slk  = (int64_t)X.slack < 256 ? (int64_t)X.slack : 256;
            return lo < KB || (int64_t)hi >= (int64_t)KB + capd || (mode == TC_STRIPE_T && SWd > 128 && W + 2 * slk <= (int64_t)T * (SWd - 128));
        };

        uint32_t s = 0;
        int      fast_budget = -1;  // steps the plain stripe loop may take before the pipelined one is tried again (-1: no pipelined steps in this launch / mode)
        for (;; s += g) {
            // ---- PIPELINED STEPS (team stripes, wf-adaptive, de = 1): the exchanges of row S overlap the cells of row N = S + 1.
            // A step is a chain -- cells, exchange 1, band ends, exchange 2, deletions -- and only the cells near the band's ends and at the
            // stripe's two edges depend on what the exchanges of the row before decide (98 % of the steps cut fewer than PZ cells).  So
            // wave 0 ("sync wave") owns no interior cell: it runs row S's exchanges, band ends (from the row's two end zones in LDS), deletions,
            // directory entry and halo, then computes row N's LATE cells (the end zones and the stripe's first and last cell); waves 1 .. 15
            // compute row N's INTERIOR cells meanwhile.  One workgroup barrier joins them, the row is committed (statistics into the scratch
            // set, cells into the rings), a second barrier, next row.  Anything unusual finishes row S and leaves for the steps below with
            // row N dropped (nothing of it has been committed): a cut deeper than the zones, a cell at a sequence end below the first
            // passing one, a row that does not fit the stripes / the page, termination.
            if (X.pipe != 0u && mode == TC_STRIPE_T && T > 1u && rings_in_lds && P.adaptive && de == 1u && dx >= 2u && s > x && s >= oe && X.fast != 0u) {
                constexpr int PZ  = 48;        // depth of an end zone
                constexpr int PCW = G - 64;    // threads that own interior cells
                constexpr int PU  = 4;         // cells per thread of the cell waves: stripes of up to PU x PCW = 3 840 diagonals (wider ones: the steps below)
                const auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                const bool syncw = tid < 64;
                const int  ct0 = tid - 64, cbase0 = (int)rfl((uint32_t)(tid - lane)) - 64;  // cell-thread index; its wave's first
                const int  KBw0 = (int)rfl((uint32_t)(KB + (int)b * SWd)), SWf0 = (int)rfl((uint32_t)SWd);
                s = rfl(s);
                uint32_t si = rfl(s / g), pM = si % RM;
                uint64_t ftop = (uint64_t)rfl((uint32_t)top) | ((uint64_t)rfl((uint32_t)(top >> 32)) << 32);
                const int2 *const ring_lw = reinterpret_cast<const int2 *>(ring);
                int *acc = red + 48, *accn = red;
                const auto acc_reset = [](int i) { return (i == 2 || i == 10) ? 0 : ((i == 1 || i == 5 || i == 7 || i == 12 || i == 14) ? INT32_MIN : INT32_MAX); };
                int *const pf = red + 64;  // flags of the sync wave for everybody: [0] term [1] complex [2] deep [3] nlo [4] nhi [5] h_final [6] W of row S [7] abort
                if (tid < 16) acc[tid] = acc_reset(tid), accn[tid] = acc_reset(tid);
                lds_barrier();
                bool     first = true;
                int      loS = 0, hiS = -1, loSs = 0, iloS = 0, ihiS = -1;  // row S: its true range, its storage origin, its interior as it was dealt
                uint64_t baseS = 0;
                bool     leave = false;
                for (;;) {
                    // (the loop's invariants pass through an empty asm every iteration: what is derived from them is a handful of additions, and
                    // hoisted out of the loop it would sit in registers the kernel does not have -- reloaded from scratch memory inside the loop)
                    int ct = ct0, cbase = cbase0, KBw = KBw0, SWf = SWf0;
                    asm volatile("" : "+v"(ct), "+s"(cbase), "+s"(KBw), "+s"(SWf));
                    // ---- row N's storage range: from the sources' ranges, row S's uncut (its kept band is not known yet)
                    const int2 vx = ring_lw[((si - dx) % TEAM_RING) * 4u + 1u], vo = ring_lw[((si - doe) % TEAM_RING) * 4u + 1u];
                    const int  xlo = (int)rfl((uint32_t)vx.x), xw_ = (int)rfl((uint32_t)vx.y), olo = (int)rfl((uint32_t)vo.x), ow_ = (int)rfl((uint32_t)vo.y);
                    int elo, ew_;
                    if (first) {
                        const int2 ve = ring_lw[((si - 1u) % TEAM_RING) * 4u + 1u];
                        elo = (int)rfl((uint32_t)ve.x), ew_ = (int)rfl((uint32_t)ve.y);
                    } else {
                        elo = loS, ew_ = hiS - loS + 1;
                    }
                    int lo = INT32_MAX, hi = INT32_MIN;
                    if (xw_ > 0) lo = imin2(lo, xlo - 1), hi = imax2(hi, xlo + xw_);
                    if (ow_ > 0) lo = imin2(lo, olo - 1), hi = imax2(hi, olo + ow_);
                    if (ew_ > 0) lo = imin2(lo, elo - 1), hi = imax2(hi, elo + ew_);
                    lo = imax2(lo, -(n - 1)), hi = imin2(hi, m - 1);
                    const int W = hi >= lo ? hi - lo + 1 : 0;
                    // the interior: cells whose three sources in row S lie deeper than PZ inside its range; the two end zones before / behind it
                    const int ilo = elo + PZ + 2, ihi = elo + ew_ - 1 - PZ - 2;
                    bool last = W <= 0 || xw_ <= 0 || ow_ <= 0 || ew_ <= 0 || ihi < ilo || SWf > PU * PCW || ilo - lo > 64 || hi - ihi > 64 || want_mode((int64_t)W, s) != mode ||
                                moved_now(lo, hi, (int64_t)W) ||
                                (!paged ? ftop + (uint64_t)W + (uint64_t)DIR_WORDS * (si + 2) > cap : (si + 2u > dir_entries || (uint64_t)W > page_words || ftop + (uint64_t)W > page_end));
                    if (first && last) {
                        TC_COUNT(28);
                        fast_budget = 16;  // (not a row for these steps: a stretch of the plain ones)
                        break;
                    }
                    TC_STAMP(0);
                    const uint64_t baseN = ftop;
                    uint32_t *const rowC = A + baseN;
                    const uint32_t sO = pM >= doe ? pM - doe : pM + RM - doe, sX = pM >= dx ? pM - dx : pM + RM - dx, sS = pM >= 1u ? pM - 1u : RM - 1u;
                    const uint32_t *const lO = lrows + (size_t)sO * TC_ROWW, *const lXr = lrows + (size_t)sX * TC_ROWW;
                    uint32_t *const rS = lrows + (size_t)sS * TC_ROWW;                                          // row S's M cells (where the sync wave deletes)
                    uint32_t *const nM = lrows + (size_t)pM * TC_ROWW, *const nI = lrows + (size_t)RM * TC_ROWW, *const nD = lrows + (size_t)(RM + 1u) * TC_ROWW;
                    // (de = 1: one I and one D row; row S's until row N's replace them at the commit)

                    int      mlo = INT32_MAX, mhi = INT32_MIN, fvm = INT32_MAX, lvm = INT32_MIN, hminw = INT32_MAX;  // (wave-uniform)
                    int      mind = INT32_MAX, maxd = INT32_MIN;
                    bool     termw = false, endw = false;
                    uint32_t kM[PU], kI[PU], kD[PU], cnt = 0, ekd = 0xFFFFFFFFu, eku = 0xFFFFFFFFu;
#pragma unroll
                    for (int u = 0; u < PU; u++) kM[u] = kI[u] = kD[u] = 0u;
                    // one cell: sources from the rings, next, extend, the backtrace word, the statistics
                    auto cell = [&](int u, int j, int k, bool on) __attribute__((always_inline)) {
                        LCell c = {0u, 0u, 0u, 0u};
                        bool  valid = false;
                        if (on) {
                            c   = lean_next_fast(lO[j], nI[j], lO[j + 2], nD[j + 2], lXr[j + 1], k, n, m);
                            c.M = lean_extend<MODE>(sv, c.M, k);
                            cst(rowC + (k - lo), c.wd);
                            cnt += (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                            const int v = (int)c.M - k;
                            valid = c.M != 0u && (uint32_t)v < (uint32_t)n && (int)c.M < m;
                            const int d = imax2(m - (int)c.M, n - v);
                            if (valid) mind = imin2(mind, d), maxd = imax2(maxd, d);
                            if (c.M != 0u && k == Ak && (int)c.M >= m) termw = true, h_final = c.M;
                        }
                        kM[u] = c.M, kI[u] = c.I, kD[u] = c.D;
                        return valid;
                    };

                    if (syncw) {
                        // ================= the sync wave: row S's exchanges, band ends, deletions, entry; then row N's late cells
                        int  nloS = loS, nhiS = hiS;
                        bool termS = false, complexS = false, deepS = false;
                        if (!first) {
                            const uint32_t siS = si - 1u;
                            if (lane == 0) {
                                red[16] = acc[0], red[17] = ~acc[1], red[18] = acc[2] ? -1 : 0, red[19] = acc[3], red[20] = ~acc[12], red[21] = acc[13], red[22] = ~acc[14],
                                red[23] = acc[2] ? ~acc[10] : INT32_MAX;
                                if (lead_wg && !glob) {
                                    ast(dir_ptr(siS) + 5, 0xFFFFFFFFu), ast(dir_ptr(siS) + 6, 0xFFFFFFFFu);
                                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                }
                            }
                            exchange_core(8, true, false);
                            const int  amlo = (int)rfl((uint32_t)red[16]), amhi = ~(int)rfl((uint32_t)red[17]), amind = (int)rfl((uint32_t)red[19]), amaxd = ~(int)rfl((uint32_t)red[20]);
                            termS = rfl((uint32_t)red[18]) != 0u;
                            if (termS) h_final = ~rfl((uint32_t)red[23]);
                            nloS = amlo, nhiS = amhi;
                            if (lane == 0) pf[12] = amlo, pf[13] = amhi, pf[14] = amind;
                            if (!aborted && !termS && amhi >= amlo && (amhi - amlo + 1) >= (int)P.min_wf_len && amind != INT32_MAX && amaxd - amind > (int)P.max_dist_diff) {
                                const int thr = amind + (int)P.max_dist_diff;
                                // (a passing cell has a distance to go: the scans start at the row's first / last such cell, not at the ends of its M
                                // range -- a band whose end has reached a sequence end keeps a run of cells there that neither pass nor fail)
                                const int zl = (int)rfl((uint32_t)red[21]), zh = ~(int)rfl((uint32_t)red[22]);
                                int f_ok = INT32_MAX, l_ok = INT32_MIN;
                                {
                                    const int  k = zl + lane, j = k - KBw;
                                    const bool in = j >= 0 && j < SWf && k <= zh;
                                    const int  d = in ? lean_dist(rS[j + 1], k, n, m) : -1;
                                    const unsigned long long bb = __ballot(d >= 0 && d <= thr);
                                    if (bb != 0ull) f_ok = zl + (int)__builtin_ctzll(bb);
                                }
                                {
                                    const int  k = zh - 63 + lane, j = k - KBw;
                                    const bool in = j >= 0 && j < SWf && k >= zl;
                                    const int  d = in ? lean_dist(rS[j + 1], k, n, m) : -1;
                                    const unsigned long long bb = __ballot(d >= 0 && d <= thr);
                                    if (bb != 0ull) l_ok = zh - (int)__builtin_clzll(bb);
                                }
                                if (lane == 0) red[16] = f_ok, red[17] = ~l_ok, red[18] = acc[11];
                                exchange_core(3, false, false);
                                const int first_ok = (int)rfl((uint32_t)red[16]), last_ok = ~(int)rfl((uint32_t)red[17]), hitmin = (int)rfl((uint32_t)red[18]);
                                if (first_ok != INT32_MAX && last_ok != INT32_MIN && hitmin >= first_ok) {
                                    nloS = first_ok, nhiS = last_ok;
                                } else if (first_ok != INT32_MAX && last_ok != INT32_MIN && !aborted) {
                                    // a cell at a sequence end below the first passing one (wfa.go:503-516): _lo is one past the last valid entry
                                    // before the first non-failing one -- every cell below first_ok lies in the low zone, so one more exchange of the
                                    // sync waves settles it (pairs whose band's low end has reached a sequence end take this on every row)
                                    int lead = INT32_MIN;
                                    {
                                        const int  k = zl + lane, j = k - KBw;  // (a valid cell below first_ok lies in [zl, first_ok): inside the scanned zone)
                                        const bool in = j >= 0 && j < SWf && k <= zh && k < first_ok;
                                        const unsigned long long bb = __ballot(in && lean_dist(rS[j + 1], k, n, m) >= 0);
                                        if (bb != 0ull) lead = zl + 63 - (int)__builtin_clzll(bb);
                                    }
                                    if (lane == 0) red[16] = ~lead;
                                    exchange_core(1, false, false);
                                    lead = ~(int)rfl((uint32_t)red[16]);
                                    nloS = (lead != INT32_MIN) ? lead + 1 : amlo;
                                    nhiS = last_ok;
                                } else {
                                    complexS = true;  // (no passing cell within 64 diagonals of an end: the full pass over the row, by everybody, below)
                                }
                            }
                            if (!complexS && !aborted) {
                                deepS = !termS && (nloS > loS + PZ + 1 || nhiS < hiS - PZ - 1);
                                // Delete of wfa.go:526-535 in the rings and in the census: within 64 diagonals of the M range's ends (else: deep)
                                if (!deepS && !termS) {
#pragma unroll
                                    for (int z = 0; z < 2; z++) {
                                        const int  k = z == 0 ? amlo + lane : amhi - 63 + lane, j = k - KBw;
                                        const bool cut = j >= 0 && j < SWf && k >= amlo && k <= amhi && (k < nloS || k > nhiS) && (z == 0 || k > amlo + 63);
                                        if (cut) {
                                            my_cells -= (rS[j + 1] != 0u) + (nI[j + 1] != 0u) + (nD[j + 1] != 0u);
                                            rS[j + 1] = 0u, nI[j + 1] = 0u, nD[j + 1] = 0u;
                                        }
                                    }
                                }
                                // the end-cell keys of row S (semi-global): what the cell waves collected in the interior + the kept cells of the zones
                                if (!glob && !deepS) {
                                    // (the scratch words start at INT32_MAX: "none")
                                    uint32_t kd = acc[8] == INT32_MAX ? 0xFFFFFFFFu : (uint32_t)acc[8], ku = acc[9] == INT32_MAX ? 0xFFFFFFFFu : (uint32_t)acc[9];
                                    // row S's late cells as they were dealt: below its interior, above it, the stripe's first and last cell
#pragma unroll
                                    for (int z = 0; z < 3; z++) {
                                        int  k, j;
                                        bool in;
                                        if (z == 0) k = loSs + lane, j = k - KBw, in = j > 0 && j < SWf - 1 && k < iloS;
                                        else if (z == 1) k = ihiS + 1 + lane, j = k - KBw, in = j > 0 && j < SWf - 1 && k <= hiS;
                                        else j = lane == 0 ? 0 : SWf - 1, k = KBw + j, in = lane < 2 && (lane == 0 || SWf > 1);
                                        in = in && k >= loS && k <= hiS && k >= nloS && k <= nhiS;
                                        const uint32_t cls = in ? lean_endclass(rS[j + 1], k, n, m) : 0u;
                                        if (cls != 0u) {
                                            if (k <= Ak) kd = umin2(kd, lean_endkey(cls, (uint32_t)(Ak - k)));
                                            else ku = umin2(ku, lean_endkey(cls, (uint32_t)(k - Ak - 1)));
                                        }
                                    }
                                    if (__ballot(kd != 0xFFFFFFFFu || ku != 0xFFFFFFFFu) != 0ull) {
                                        kd = (uint32_t)wave_min((int)(kd ^ 0x80000000u)) ^ 0x80000000u;
                                        ku = (uint32_t)wave_min((int)(ku ^ 0x80000000u)) ^ 0x80000000u;
                                        if (lane == 0) {
                                            if (kd != 0xFFFFFFFFu) atomicMin(dir_ptr(siS) + 5, kd);
                                            if (ku != 0xFFFFFFFFu) atomicMin(dir_ptr(siS) + 6, ku);
                                        }
                                    }
                                }
                                if (!deepS) {
                                    if (nhiS >= nloS) put_ent(siS, baseS + (uint64_t)(nloS - loSs), nloS, nhiS - nloS + 1);
                                    else put_ent(siS, 0ull, 0, 0);
                                    if (lane < 6) {  // the halo cells of row S's ring slots: the neighbours' edge cells where the row keeps them
                                        const int halo_k = (lane & 1) ? KBw + SWf : KBw - 1;
                                        uint32_t *const row = (lane >> 1) == 0 ? rS : ((lane >> 1) == 1 ? nI : nD);
                                        row[(lane & 1) ? SWf + 1 : 0] = (nhiS >= nloS && halo_k >= nloS && halo_k <= nhiS) ? (uint32_t)red[40 + 3 * (lane & 1) + (lane >> 1)] : 0u;
                                    }
                                }
                            }
                        }
                        // ---- row N's late cells: its true range is known now
                        int loN = lo, hiN = hi;
                        if (!first) {
                            loN = INT32_MAX, hiN = INT32_MIN;
                            const int sw = nhiS >= nloS ? nhiS - nloS + 1 : 0;
                            if (xw_ > 0) loN = imin2(loN, xlo - 1), hiN = imax2(hiN, xlo + xw_);
                            if (ow_ > 0) loN = imin2(loN, olo - 1), hiN = imax2(hiN, olo + ow_);
                            if (sw > 0) loN = imin2(loN, nloS - 1), hiN = imax2(hiN, nloS + sw);
                            loN = imax2(loN, -(n - 1)), hiN = imin2(hiN, m - 1);
                        }
                        if (!last && !termS && !complexS && !deepS && !aborted) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            // pass 0: the zone below the interior; pass 1: the zone above it; pass 2: the stripe's first / last cell if they are interior cells' gap
#pragma unroll
                            for (int u = 0; u < 3; u++) {
                                int  k, j;
                                bool on;
                                if (u == 0) k = lo + lane, j = k - KBw, on = j > 0 && j < SWf - 1 && k < ilo && k >= loN && k <= hiN;
                                else if (u == 1) k = ihi + 1 + lane, j = k - KBw, on = j > 0 && j < SWf - 1 && k > ihi && k >= loN && k <= hiN;
                                else j = lane == 0 ? 0 : SWf - 1, k = KBw + j, on = lane < 2 && (lane == 0 || SWf > 1) && k >= loN && k <= hiN;
                                const bool valid = cell(u, on ? j : 0, k, on);
                                const unsigned long long bM = __ballot(kM[u] != 0u), bV = __ballot(valid), bZ = __ballot(valid && (int)kM[u] == k);
                                if (u < 2) {
                                    const int kb = u == 0 ? lo : ihi + 1;
                                    if (bM != 0ull) mlo = imin2(mlo, kb + (int)__builtin_ctzll(bM)), mhi = imax2(mhi, kb + 63 - (int)__builtin_clzll(bM));
                                    if (bV != 0ull) fvm = imin2(fvm, kb + (int)__builtin_ctzll(bV)), lvm = imax2(lvm, kb + 63 - (int)__builtin_clzll(bV));
                                    if ((bM & ~bV) != 0ull) hminw = imin2(hminw, kb + (int)__builtin_ctzll(bM & ~bV));
                                } else {  // (two unrelated diagonals)
#pragma unroll
                                    for (int q = 0; q < 2; q++) {
                                        const int kq = KBw + (q == 0 ? 0 : SWf - 1);
                                        if ((bM >> q) & 1ull) mlo = imin2(mlo, kq), mhi = imax2(mhi, kq);
                                        if ((bV >> q) & 1ull) fvm = imin2(fvm, kq), lvm = imax2(lvm, kq);
                                        if (((bM & ~bV) >> q) & 1ull) hminw = imin2(hminw, kq);
                                    }
                                }
                                endw = endw || bM != bV || bZ != 0ull;
                            }
                        }
                        if (lane == 0) {
                            pf[0] = termS ? 1 : 0, pf[1] = complexS ? 1 : 0, pf[2] = deepS ? 1 : 0, pf[3] = nloS, pf[4] = nhiS, pf[5] = (int)h_final, pf[7] = aborted ? 1 : 0;
                            pf[10] = loN, pf[11] = hiN;
                        }
                    } else if (!last) {
                        // ================= the cell waves: row N's interior
#pragma unroll
                        for (int u = 0; u < PU; u++) {
                            const int  j = ct + u * PCW, k = KBw + j;
                            const bool on = j > 0 && j < SWf - 1 && k >= ilo && k <= ihi;
                            const bool valid = cell(u, j, k, on);
                            const unsigned long long bM = __ballot(kM[u] != 0u), bV = __ballot(valid), bZ = __ballot(valid && (int)kM[u] == k);
                            const int kb = KBw + cbase + u * PCW;
                            if (bM != 0ull) {
                                if (mlo == INT32_MAX) mlo = kb + (int)__builtin_ctzll(bM);
                                mhi = kb + 63 - (int)__builtin_clzll(bM);
                            }
                            if (bV != 0ull) {
                                if (fvm == INT32_MAX) fvm = kb + (int)__builtin_ctzll(bV);
                                lvm = kb + 63 - (int)__builtin_clzll(bV);
                            }
                            endw = endw || bM != bV || bZ != 0ull;
                            if ((bM & ~bV) != 0ull && hminw == INT32_MAX) hminw = kb + (int)__builtin_ctzll(bM & ~bV);
                        }
                        // the end-cell keys of the interior cells (they are kept unless the row is dropped)
                        if (!glob && endw) {
#pragma unroll
                            for (int u = 0; u < PU; u++) {
                                const int      k = KBw + ct + u * PCW;
                                const uint32_t cls = lean_endclass(kM[u], k, n, m);  // (0 for a cell that does not exist)
                                if (cls != 0u) {
                                    if (k <= Ak) ekd = umin2(ekd, lean_endkey(cls, (uint32_t)(Ak - k)));
                                    else eku = umin2(eku, lean_endkey(cls, (uint32_t)(k - Ak - 1)));
                                }
                            }
                        }
                    }
                    TC_STAMP(29);
                    lds_barrier();  // (B1) row S is final, row N's cells are computed
                    TC_STAMP(30);
                    const bool termS = pf[0] != 0, complexS = pf[1] != 0, deepS = pf[2] != 0;
                    int        nloS = (int)rfl((uint32_t)pf[3]), nhiS = (int)rfl((uint32_t)pf[4]);
                    if (pf[7] != 0) {
                        aborted = true;
                        TC_ABORT_RET;
                    }
                    if (!first) n_ent = si;  // (row S = si - 1 has its entry, or gets it below)
                    if (termS) {
                        h_final = (uint32_t)pf[5];
                        done    = true;
                        s_final = s - g;
                        leave   = true;
                        break;
                    }
                    if (complexS) {
                        TC_COUNT(23);
                        // ---- the full pass over row S (wfa.go:497-524), everybody, out of the rings; then the row's deletions, keys, entry and halo
                        const uint32_t siS = si - 1u;
                        const int thr = (int)rfl((uint32_t)pf[14]) + (int)P.max_dist_diff;
                        const int amlo = (int)rfl((uint32_t)pf[12]);
                        int f_ok = INT32_MAX, l_ok = INT32_MIN;
                        if (!syncw) {
#pragma unroll
                            for (int u = 0; u < PU; u++) {
                                const int  j = ct + u * PCW, k = KBw + j;
                                const int  d = (j < SWf && k >= loS && k <= hiS) ? lean_dist(rS[j + 1], k, n, m) : -1;
                                const unsigned long long bOk = __ballot(d >= 0 && d <= thr);
                                const int kb = KBw + cbase + u * PCW;
                                if (bOk != 0ull) {
                                    if (f_ok == INT32_MAX) f_ok = kb + (int)__builtin_ctzll(bOk);
                                    l_ok = kb + 63 - (int)__builtin_clzll(bOk);
                                }
                            }
                            if (lane == 0 && f_ok != INT32_MAX) atomicMin(&acc[4], f_ok), atomicMax(&acc[5], l_ok);
                        }
                        lds_barrier();
                        if (tid == 0) red[16] = acc[4], red[17] = ~acc[5], red[18] = acc[11];
                        exchange_core(3, false);
                        if (aborted) TC_ABORT_RET;
                        const int first_ok = (int)rfl((uint32_t)red[16]), last_ok = ~(int)rfl((uint32_t)red[17]), hitmin = (int)rfl((uint32_t)red[18]);
                        if (hitmin >= first_ok) {
                            nloS = first_ok, nhiS = last_ok;
                        } else {
                            int lead = INT32_MIN;
                            if (!syncw) {
#pragma unroll
                                for (int u = 0; u < PU; u++) {
                                    const int j = ct + u * PCW, k = KBw + j;
                                    if (j < SWf && k >= loS && k <= hiS && k < first_ok && lean_dist(rS[j + 1], k, n, m) >= 0) lead = imax2(lead, k);
                                }
                            }
                            lead = wave_max(lead);
                            if (lane == 0) atomicMax(&acc[7], lead);
                            lds_barrier();
                            if (tid == 0) red[16] = ~acc[7];
                            exchange_core(1, false);
                            if (aborted) TC_ABORT_RET;
                            lead = ~(int)rfl((uint32_t)red[16]);
                            nloS = (lead != INT32_MIN) ? lead + 1 : amlo;
                            nhiS = last_ok;
                        }
                        uint32_t kd = 0xFFFFFFFFu, ku = 0xFFFFFFFFu;
                        if (!syncw) {
#pragma unroll
                            for (int u = 0; u < PU; u++) {
                                const int j = ct + u * PCW, k = KBw + j;
                                if (j < SWf && k >= loS && k <= hiS) {
                                    if (k < nloS || k > nhiS) {
                                        my_cells -= (rS[j + 1] != 0u) + (nI[j + 1] != 0u) + (nD[j + 1] != 0u);
                                        rS[j + 1] = 0u, nI[j + 1] = 0u, nD[j + 1] = 0u;
                                    } else if (!glob) {
                                        const uint32_t cls = lean_endclass(rS[j + 1], k, n, m);
                                        if (cls != 0u) {
                                            if (k <= Ak) kd = umin2(kd, lean_endkey(cls, (uint32_t)(Ak - k)));
                                            else ku = umin2(ku, lean_endkey(cls, (uint32_t)(k - Ak - 1)));
                                        }
                                    }
                                }
                            }
                        }
                        if (!glob && __ballot(kd != 0xFFFFFFFFu || ku != 0xFFFFFFFFu) != 0ull) {
                            kd = (uint32_t)wave_min((int)(kd ^ 0x80000000u)) ^ 0x80000000u;
                            ku = (uint32_t)wave_min((int)(ku ^ 0x80000000u)) ^ 0x80000000u;
                            if (lane == 0) {
                                if (kd != 0xFFFFFFFFu) atomicMin(dir_ptr(siS) + 5, kd);
                                if (ku != 0xFFFFFFFFu) atomicMin(dir_ptr(siS) + 6, ku);
                            }
                        }
                        if (nhiS >= nloS) put_ent(siS, baseS + (uint64_t)(nloS - loSs), nloS, nhiS - nloS + 1);
                        else put_ent(siS, 0ull, 0, 0);
                        if (tid < 6) {
                            const int halo_k = (tid & 1) ? KBw + SWf : KBw - 1;
                            uint32_t *const row = (tid >> 1) == 0 ? rS : ((tid >> 1) == 1 ? nI : nD);
                            row[(tid & 1) ? SWf + 1 : 0] = (nhiS >= nloS && halo_k >= nloS && halo_k <= nhiS) ? (uint32_t)red[40 + 3 * (tid & 1) + (tid >> 1)] : 0u;
                        }
                        lds_barrier();
                        leave = true;
                        break;
                    }
                    if (deepS) TC_COUNT(22);
                    if (deepS) {  // row S's deletions reach the interior's sources: the sync wave has not applied them -- everybody does, then row N is dropped
                        const uint32_t siS = si - 1u;
                        uint32_t kd = 0xFFFFFFFFu, ku = 0xFFFFFFFFu;
                        if (!syncw) {
#pragma unroll
                            for (int u = 0; u < PU; u++) {
                                const int j = ct + u * PCW, k = KBw + j;
                                if (j < SWf && k >= loS && k <= hiS) {
                                    if (k < nloS || k > nhiS) {
                                        my_cells -= (rS[j + 1] != 0u) + (nI[j + 1] != 0u) + (nD[j + 1] != 0u);
                                        rS[j + 1] = 0u, nI[j + 1] = 0u, nD[j + 1] = 0u;
                                    } else if (!glob) {
                                        const uint32_t cls = lean_endclass(rS[j + 1], k, n, m);
                                        if (cls != 0u) {
                                            if (k <= Ak) kd = umin2(kd, lean_endkey(cls, (uint32_t)(Ak - k)));
                                            else ku = umin2(ku, lean_endkey(cls, (uint32_t)(k - Ak - 1)));
                                        }
                                    }
                                }
                            }
                        }
                        if (!glob && __ballot(kd != 0xFFFFFFFFu || ku != 0xFFFFFFFFu) != 0ull) {
                            kd = (uint32_t)wave_min((int)(kd ^ 0x80000000u)) ^ 0x80000000u;
                            ku = (uint32_t)wave_min((int)(ku ^ 0x80000000u)) ^ 0x80000000u;
                            if (lane == 0) {
                                if (kd != 0xFFFFFFFFu) atomicMin(dir_ptr(siS) + 5, kd);
                                if (ku != 0xFFFFFFFFu) atomicMin(dir_ptr(siS) + 6, ku);
                            }
                        }
                        if (nhiS >= nloS) put_ent(siS, baseS + (uint64_t)(nloS - loSs), nloS, nhiS - nloS + 1);
                        else put_ent(siS, 0ull, 0, 0);
                        if (tid < 6) {
                            const int halo_k = (tid & 1) ? KBw + SWf : KBw - 1;
                            uint32_t *const row = (tid >> 1) == 0 ? rS : ((tid >> 1) == 1 ? nI : nD);
                            row[(tid & 1) ? SWf + 1 : 0] = (nhiS >= nloS && halo_k >= nloS && halo_k <= nhiS) ? (uint32_t)red[40 + 3 * (tid & 1) + (tid >> 1)] : 0u;
                        }
                        lds_barrier();
                        leave = true;
                        break;
                    }
                    if (last) {
                        TC_COUNT(21);
                        leave = true;
                        break;
                    }
                    // ---- commit row N: the arena, the census, the statistics into the scratch set, the cells into the rings
                    const int loN = (int)rfl((uint32_t)pf[10]), hiN = (int)rfl((uint32_t)pf[11]);
                    ftop += (uint64_t)W, top = ftop;
                    my_cells += cnt;
                    if (fvm != INT32_MAX) mind = wave_min(mind), maxd = wave_max(maxd);
                    if (__ballot(termw) != 0ull) {
                        if (lane == 0) accn[2] = 1;
                        if (termw) accn[10] = (int)h_final;
                    }
                    if (!glob && __ballot(ekd != 0xFFFFFFFFu || eku != 0xFFFFFFFFu) != 0ull) {
                        ekd = (uint32_t)wave_min((int)(ekd ^ 0x80000000u)) ^ 0x80000000u;
                        eku = (uint32_t)wave_min((int)(eku ^ 0x80000000u)) ^ 0x80000000u;
                        if (lane == 0) atomicMin(reinterpret_cast<unsigned int *>(&accn[8]), ekd), atomicMin(reinterpret_cast<unsigned int *>(&accn[9]), eku);
                    }
                    if (lane == 0) {
                        if (mlo != INT32_MAX) atomicMin(&accn[0], mlo), atomicMax(&accn[1], mhi);
                        if (fvm != INT32_MAX) atomicMin(&accn[13], fvm), atomicMax(&accn[14], lvm), atomicMin(&accn[3], mind), atomicMax(&accn[12], maxd);
                        if (hminw != INT32_MAX) atomicMin(&accn[11], hminw);
                    }
                    // (row N's I / D take row S's slots: every cell of row N has read them.)  The cell waves write every cell of the stripe that is
                    // not a late cell -- zero where row N has none: the slot held row N - RM --, the sync wave the late ones
                    if (syncw) {
#pragma unroll
                        for (int u = 0; u < 3; u++) {
                            int  j;
                            bool mine;
                            if (u == 0) j = lo + lane - KBw, mine = j > 0 && j < SWf - 1 && lo + lane < ilo;
                            else if (u == 1) j = ihi + 1 + lane - KBw, mine = j > 0 && j < SWf - 1 && ihi + 1 + lane <= hi;
                            else j = lane == 0 ? 0 : SWf - 1, mine = lane < 2 && (lane == 0 || SWf > 1);
                            if (mine) nM[j + 1] = kM[u], nI[j + 1] = kI[u], nD[j + 1] = kD[u];
                        }
                        // the stripe's first / last cell: the neighbours' halo, handed over with the next exchange
                        if (lane == 0) red[24] = (int)kM[2], red[25] = (int)kI[2], red[26] = (int)kD[2];
                        if (lane == (SWf > 1 ? 1 : 0)) red[28] = (int)kM[2], red[29] = (int)kI[2], red[30] = (int)kD[2];
                    } else {
#pragma unroll
                        for (int u = 0; u < PU; u++) {
                            const int j = ct + u * PCW, k = KBw + j;
                            if (j > 0 && j < SWf - 1 && !(k >= lo && k < ilo) && !(k > ihi && k <= hi)) nM[j + 1] = kM[u], nI[j + 1] = kI[u], nD[j + 1] = kD[u];
                        }
                    }
                    lds_barrier();  // (A) row N is in the rings, its statistics are complete
                    if (tid >= 64 && tid < 80) acc[tid - 64] = acc_reset(tid - 64);  // row S's set: idle until the row after next
                    {
                        int *const t_ = acc;
                        acc = accn, accn = t_;
                    }
                    loS = loN, hiS = hiN, loSs = lo, baseS = baseN, iloS = ilo, ihiS = ihi;
                    TC_STAMP(31);
                    TC_COUNT(20);
                    first = false;
                    s += g, si += 1u;
                    pM = pM + 1u == RM ? 0u : pM + 1u;
                }
                top = ftop;
                if (done) break;
                lds_barrier();
                if (leave) fast_budget = 2;  // (the row that made the pipelined steps leave, and one more; then they are tried again)
            }
            // ---- FAST STEPS: a stripe mode in its steady state -- every source score has its ring slot, no seeds, the row fits the
            // stripes as they are dealt, the page and the directory have room.  The same step as the general one below, on fewer
            // instructions and five workgroup barriers instead of nine: the ring slots and the score index are counters, not divisions;
            // the row's ranges come from ballots (a wave's cells are consecutive diagonals), not wave reductions; the workgroup's
            // partial results are collected in one of two scratch sets that alternate (the idle one is reset on the side); thread 0
            // builds the exchange's payload right behind the barrier that completes the set.  Anything else leaves the loop: the general
            // head below decides about score s.
            if ((mode == TC_STRIPE_T || mode == TC_STRIPE_S) && rings_in_lds && s > x && s >= oe && X.fast != 0u) {
                const auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                const bool team_now = mode == TC_STRIPE_T;
                bool       back_to_pipe = false;
                const int  KBw = (int)rfl((uint32_t)(KB + (team_now ? (int)b * SWd : 0))), SWf = (int)rfl((uint32_t)SWd), wbase = (int)rfl((uint32_t)(tid - lane));
                s = rfl(s);
                uint32_t si = rfl(s / g), pM = si % RM, pE = si % RE;
                uint64_t ftop = (uint64_t)rfl((uint32_t)top) | ((uint64_t)rfl((uint32_t)(top >> 32)) << 32);  // (the arena's fill in scalar registers)
                const int2 *const ring_lw = reinterpret_cast<const int2 *>(ring);
                int *acc = red + 48, *accn = red;  // (the general step uses red[0 .. 14] itself and resets it in its head)
                // reset values of a scratch set: minima start at INT32_MAX, maxima (1, 5, 7, 12, 14) at INT32_MIN, flags (2, 10) at zero
                const auto acc_reset = [](int i) { return (i == 2 || i == 10) ? 0 : ((i == 1 || i == 5 || i == 7 || i == 12 || i == 14) ? INT32_MIN : INT32_MAX); };
                if (tid < 16) acc[tid] = acc_reset(tid);
                lds_barrier();
                for (;;) {
                    // ---- head: the three source rows' ranges -> this row's
                    const int2 vx = ring_lw[((si - dx) % TEAM_RING) * 4u + 1u], vo = ring_lw[((si - doe) % TEAM_RING) * 4u + 1u], ve = ring_lw[((si - de) % TEAM_RING) * 4u + 1u];
                    const int  xlo = (int)rfl((uint32_t)vx.x), xw_ = (int)rfl((uint32_t)vx.y), olo = (int)rfl((uint32_t)vo.x), ow_ = (int)rfl((uint32_t)vo.y);
                    const int  elo = (int)rfl((uint32_t)ve.x), ew_ = (int)rfl((uint32_t)ve.y);
                    int lo = INT32_MAX, hi = INT32_MIN;
                    if (xw_ > 0) lo = imin2(lo, xlo - 1), hi = imax2(hi, xlo + xw_);
                    if (ow_ > 0) lo = imin2(lo, olo - 1), hi = imax2(hi, olo + ow_);
                    if (ew_ > 0) lo = imin2(lo, elo - 1), hi = imax2(hi, elo + ew_);
                    lo = imax2(lo, -(n - 1)), hi = imin2(hi, m - 1);  // wfa.go:562-563
                    if (hi < lo || xw_ <= 0 || ow_ <= 0 || ew_ <= 0) break;  // (a source score without a row has not cleared its ring slot)
                    const int W = hi - lo + 1;
                    if (want_mode((int64_t)W, s) != mode || moved_now(lo, hi, (int64_t)W)) break;
                    if (!paged ? ftop + (uint64_t)W + (uint64_t)DIR_WORDS * (si + 2) > cap
                               : (si + 2u > dir_entries || (uint64_t)W > page_words || ftop + (uint64_t)W > page_end))
                        break;
                    const uint64_t base = ftop;
                    uint32_t *const rowC = A + base;
                    // ring slots: row si - r of the M ring sits r slots behind pM (mod RM); the I / D rows of si - de share the slot the
                    // new row takes (de = RE rows back)
                    const uint32_t sO = pM >= doe ? pM - doe : pM + RM - doe, sX = pM >= dx ? pM - dx : pM + RM - dx;
                    const uint32_t *const lO = lrows + (size_t)sO * TC_ROWW, *const lXr = lrows + (size_t)sX * TC_ROWW;
                    uint32_t *const nM = lrows + (size_t)pM * TC_ROWW, *const nI = lrows + (size_t)(RM + pE) * TC_ROWW, *const nD = lrows + (size_t)(RM + RE + pE) * TC_ROWW;
                    if (lead_wg && !glob && tid == 0) ast(dir_ptr(si) + 5, 0xFFFFFFFFu), ast(dir_ptr(si) + 6, 0xFFFFFFFFu);
                    // the stripe's cells of this row: j in [jlo, jlo + jspan] (one unsigned comparison per cell)
                    const int      jhi_ = imin2(hi - KBw, SWf - 1), jlo = jhi_ >= imax2(lo - KBw, 0) ? imax2(lo - KBw, 0) : (1 << 30);
                    const uint32_t jspan = jhi_ >= jlo ? (uint32_t)(jhi_ - jlo) : 0u;
                    TC_STAMP(0);

                    // ---- P1: next + extend, the row's backtrace words; ranges from ballots
                    int      mlo = INT32_MAX, mhi = INT32_MIN, fvm = INT32_MAX, lvm = INT32_MIN;  // (wave-uniform)
                    int      mind = INT32_MAX, maxd = INT32_MIN;                                  // (per lane)
                    bool     termw = false, endw = false;                                         // (endw: wave-uniform)
                    int      hminw = INT32_MAX;  // (wave-uniform) the wave's first present cell at / past a sequence end
                    uint32_t kM[TC_U], kI[TC_U], kD[TC_U], cnt = 0;
                    // The thread's cells side by side, in straight-line code the compiler can interleave (a wave's step is a chain of LDS
                    // round trips and dependent instructions: four chains in flight instead of one after the other).  The sources are loaded
                    // unconditionally -- every index lies inside the ring rows -- and zeroed for a cell outside the row: next() of no
                    // sources is no cell, which neither extends nor counts.
                    LCell c[TC_U];
                    {
                        uint32_t sa[TC_U], sb[TC_U], sc_[TC_U], sd[TC_U], sx[TC_U];
                        bool     rej = false;
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int  j = tid + u * G;
                            const bool on = (uint32_t)(j - jlo) <= jspan;
                            sa[u] = on ? lO[j] : 0u, sb[u] = on ? nI[j] : 0u, sc_[u] = on ? lO[j + 2] : 0u, sd[u] = on ? nD[j + 2] : 0u, sx[u] = on ? lXr[j + 1] : 0u;
                            rej = rej || lean_rejects(sa[u], sb[u], sc_[u], sd[u], sx[u], KBw + j, n, m);
                        }
                        if (__ballot(rej) == 0ull) {
#pragma unroll
                            for (int u = 0; u < TC_U; u++) c[u] = lean_next_norej(sa[u], sb[u], sc_[u], sd[u], sx[u]);
                        } else {
#pragma unroll
                            for (int u = 0; u < TC_U; u++) c[u] = lean_next(sa[u], sb[u], sc_[u], sd[u], sx[u], KBw + tid + u * G, n, m);
                        }
                    }
                    // WF_EXTEND (wfa.go:394-455): the first sixteen bases of every cell side by side, the rare longer run in lean_extend's loop
                    if constexpr (MODE == 0) {
                        bool mo[TC_U], more = false;
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int      k = KBw + tid + u * G;
                            const uint32_t h = c[u].M;
                            const int      v = (int)h - k;
                            const bool     can = h != 0u && v > 0 && v < n && (int)h < m;
                            const int      vv = can ? v : 0, hh = can ? (int)h : 0;
                            const uint32_t xw = SeqView<0>::win16(sv.q, vv) ^ SeqView<0>::win16(sv.t, hh);
                            const int      rem = imin2(n - vv, m - hh), l = xw != 0u ? (int)(__builtin_ctz(xw) >> 1) : 16;
                            c[u].M = h + (can ? (uint32_t)imin2(l, rem) : 0u);
                            mo[u]  = can && xw == 0u && rem > 16;
                            more   = more || mo[u];
                        }
                        if (__ballot(more) != 0ull) {
#pragma unroll
                            for (int u = 0; u < TC_U; u++)
                                if (mo[u]) c[u].M = lean_extend<MODE>(sv, c[u].M, KBw + tid + u * G);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < TC_U; u++) c[u].M = lean_extend<MODE>(sv, c[u].M, KBw + tid + u * G);
                    }
#pragma unroll
                    for (int u = 0; u < TC_U; u++) {
                        const int  j = tid + u * G, k = KBw + j;
                        // (wave 0 polls the exchanges: loads that would queue behind its stores -- its words wait in LDS for the end of the step)
                        if (tid < 64) wsc[tid + 64 * u] = c[u].wd;
                        else if ((uint32_t)(j - jlo) <= jspan) cst(rowC + (k - lo), c[u].wd);
                        cnt += (c[u].M != 0u) + (c[u].I != 0u) + (c[u].D != 0u);
                        const int  v = (int)c[u].M - k;
                        const bool valid = c[u].M != 0u && (uint32_t)v < (uint32_t)n && (int)c[u].M < m;  // (lean_dist's cell: a distance to go)
                        const int  d = imax2(m - (int)c[u].M, n - v);
                        if (valid) mind = imin2(mind, d), maxd = imax2(maxd, d);
                        if (c[u].M != 0u && k == Ak && (int)c[u].M >= m) termw = true, h_final = c[u].M;  // wfa.go:235-239
                        kM[u] = c[u].M, kI[u] = c[u].I, kD[u] = c[u].D;
                        const unsigned long long bM = __ballot(c[u].M != 0u), bV = __ballot(valid), bZ = __ballot(valid && (int)c[u].M == k);
                        const int kb = KBw + wbase + u * G;
                        if (bM != 0ull) {
                            if (mlo == INT32_MAX) mlo = kb + (int)__builtin_ctzll(bM);
                            mhi = kb + 63 - (int)__builtin_clzll(bM);
                        }
                        if (bV != 0ull) {
                            if (fvm == INT32_MAX) fvm = kb + (int)__builtin_ctzll(bV);
                            lvm = kb + 63 - (int)__builtin_clzll(bV);
                        }
                        // (a present cell that is at / past a sequence end, or in row 0: the only ones lean_endclass can name)
                        endw = endw || bM != bV || bZ != 0ull;
                        if ((bM & ~bV) != 0ull && hminw == INT32_MAX) hminw = kb + (int)__builtin_ctzll(bM & ~bV);
                    }
                    my_cells += cnt;
                    if (team_now) {  // the stripe's first / last cell: its neighbours' halo, handed over with the exchange
                        if (tid == 0) red[24] = (int)kM[0], red[25] = (int)kI[0], red[26] = (int)kD[0];
                        if (tid == ((SWf - 1) & (G - 1))) {
                            const int uL = (SWf - 1) / G;
#pragma unroll
                            for (int u = 0; u < TC_U; u++)
                                if (u == uL) red[28] = (int)kM[u], red[29] = (int)kI[u], red[30] = (int)kD[u];
                        }
                    }
                    TC_STAMP(1);
                    if (fvm != INT32_MAX) mind = wave_min(mind), maxd = wave_max(maxd);
                    const int wmind = fvm != INT32_MAX ? mind : INT32_MAX;  // (wave-uniform: the wave's smallest distance to go)
                    if (__ballot(termw) != 0ull) {
                        if (lane == 0) acc[2] = 1;
                        if (termw) acc[10] = (int)h_final;  // (one cell of the team sits on the final diagonal)
                    }
                    if (lane == 0) {
                        if (mlo != INT32_MAX) atomicMin(&acc[0], mlo), atomicMax(&acc[1], mhi);
                        if (fvm != INT32_MAX) atomicMin(&acc[13], fvm), atomicMax(&acc[14], lvm), atomicMin(&acc[3], mind), atomicMax(&acc[12], maxd);
                    }
                    TC_STAMP(2);
                    lds_barrier();  // (A) the workgroup's partial results are complete; everybody has read the rows whose slots the new row takes
                    TC_STAMP(3);
                    if (tid == 0) {
                        // (a maximum travels as its complement ~v: order-reversing like the negation, and "none" -- INT32_MIN -- has one)
                        red[16] = acc[0], red[17] = ~acc[1], red[18] = acc[2] ? -1 : 0, red[19] = acc[3], red[20] = ~acc[12], red[21] = acc[13], red[22] = ~acc[14],
                        red[23] = acc[2] ? ~acc[10] : INT32_MAX;
                        // (the directory keys of this score were reset in the head: acknowledged before the slot says the workgroup is here)
                        if (lead_wg && !glob) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    if (tid >= 64 && tid < 80) accn[tid - 64] = acc_reset(tid - 64);  // the other set, for the next step
#pragma unroll
                    for (int u = 0; u < TC_U; u++)
                        if (tid + u * G < SWf) nM[tid + u * G + 1] = kM[u], nI[tid + u * G + 1] = kI[u], nD[tid + u * G + 1] = kD[u];
                    if (!team_now && tid < 6) {  // (one stripe: nothing lies beyond it)
                        uint32_t *const row = (tid >> 1) == 0 ? nM : ((tid >> 1) == 1 ? nI : nD);
                        row[(tid & 1) ? SWf + 1 : 0] = 0u;
                    }
                    TC_STAMP(4);
                    // ---- exchange 1: the row's ranges, termination, the distances (and the stripes' edge cells)
                    if (team_now) {
                        exchange_core(8, true);
                        if (aborted) TC_ABORT_RET;
                    } else {
                        lds_barrier();
                    }
                    uint32_t halo = 0u;
                    int      halo_k = 0;
                    if (team_now && tid < 6) {  // the neighbours' edge cells of this row: the halo of its ring slots
                        halo_k = (tid & 1) ? KBw + SWf : KBw - 1;
                        halo   = (uint32_t)red[40 + 3 * (tid & 1) + (tid >> 1)];
                    }
                    if (tid < 64) {  // wave 0's share of the row's backtrace words: on their way while the band ends are worked out (the next poll waits for them)
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int j = tid + u * G, k = KBw + j;
                            if ((uint32_t)(j - jlo) <= jspan) cst(rowC + (k - lo), wsc[tid + 64 * u]);
                        }
                    }
                    TC_STAMP(5);
                    const int  amlo = (int)rfl((uint32_t)red[16]), amhi = ~(int)rfl((uint32_t)red[17]), amind = (int)rfl((uint32_t)red[19]), amaxd = ~(int)rfl((uint32_t)red[20]);
                    const bool term = rfl((uint32_t)red[18]) != 0u;
                    if (term) h_final = ~rfl((uint32_t)red[23]);
                    ftop += (uint64_t)W, top = ftop;
                    n_ent = si + 1;

                    // ---- reduce (wfa.go:461-540): the band wf-adaptive keeps, from the cells' owners
                    int nlo = amlo, nhi = amhi;
                    if (!term && P.adaptive && amhi >= amlo && (amhi - amlo + 1) >= (int)P.min_wf_len && amind != INT32_MAX && amaxd - amind > (int)P.max_dist_diff) {
                        const int thr = amind + (int)P.max_dist_diff;
                        int f_ok = INT32_MAX, l_ok = INT32_MIN;  // (wave-uniform)
                        const int hmin = hminw;                  // a present cell at / past a sequence end (from P1's ballots)
                        if (wmind <= thr) {  // (only a wave that holds a cell within the threshold has candidates: a handful of the team's waves)
#pragma unroll
                            for (int u = 0; u < TC_U; u++) {
                                const int  k = KBw + tid + u * G;
                                const int  d = lean_dist(kM[u], k, n, m);  // (-1 for a cell that is absent: no cell outside the row)
                                const unsigned long long bOk = __ballot(d >= 0 && d <= thr);
                                const int kb = KBw + wbase + u * G;
                                if (bOk != 0ull) {
                                    if (f_ok == INT32_MAX) f_ok = kb + (int)__builtin_ctzll(bOk);
                                    l_ok = kb + 63 - (int)__builtin_clzll(bOk);
                                }
                            }
                        }
                        if (lane == 0) {
                            if (f_ok != INT32_MAX) atomicMin(&acc[4], f_ok), atomicMax(&acc[5], l_ok);
                            if (hmin != INT32_MAX) atomicMin(&acc[11], hmin);
                        }
                        lds_barrier();  // (D)
                        if (tid == 0) red[16] = acc[4], red[17] = ~acc[5], red[18] = acc[11];
                        TC_STAMP(6);
                        if (team_now) {
                            exchange_core(3, false);
                            if (aborted) TC_ABORT_RET;
                        } else {
                            lds_barrier();
                        }
                        TC_STAMP(7);
                        const int first_ok = (int)rfl((uint32_t)red[16]), last_ok = ~(int)rfl((uint32_t)red[17]), hitmin = (int)rfl((uint32_t)red[18]);
                        if (hitmin >= first_ok) {
                            // wfa.go:509-511 with no present-but-unusable cell below first_ok: the entries between the last leading failure
                            // and first_ok are holes, and dropping or keeping a hole is the same row
                            nlo = first_ok, nhi = last_ok;
                        } else {
                            // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                            int lead = INT32_MIN;
#pragma unroll
                            for (int u = 0; u < TC_U; u++) {
                                const int  j = tid + u * G, k = KBw + j;
                                const bool on = (uint32_t)(j - jlo) <= jspan;
                                if (on && k < first_ok && lean_dist(kM[u], k, n, m) >= 0) lead = imax2(lead, k);
                            }
                            lead = wave_max(lead);
                            if (lane == 0) atomicMax(&acc[7], lead);
                            lds_barrier();
                            if (tid == 0) red[16] = ~acc[7];
                            if (team_now) {
                                exchange_core(1, false);
                                if (aborted) TC_ABORT_RET;
                            } else {
                                lds_barrier();
                            }
                            lead = ~(int)rfl((uint32_t)red[16]);
                            nlo  = (lead != INT32_MIN) ? lead + 1 : amlo;
                            nhi  = last_ok;  // wfa.go:517-524
                        }
                        // Delete of wfa.go:526-535: the cells outside [nlo, nhi] stop existing -- in the rings (zero), in the census
                        if (KBw + wbase < nlo || KBw + wbase + (TC_U - 1) * G + 63 > nhi) {
#pragma unroll
                            for (int u = 0; u < TC_U; u++) {
                                const int  j = tid + u * G, k = KBw + j;
                                const bool on = (uint32_t)(j - jlo) <= jspan;
                                if (on && (k < nlo || k > nhi)) {
                                    my_cells -= (kM[u] != 0u) + (nI[j + 1] != 0u) + (nD[j + 1] != 0u);
                                    nM[j + 1] = 0u, nI[j + 1] = 0u, nD[j + 1] = 0u;
                                    kM[u] = 0u;
                                }
                            }
                        }
                    }
                    // ---- the cells that end the reference's end-cell scan (semi-global, wfa.go:301-361): nearest to the final diagonal on
                    // either side, among the cells the row keeps
                    if (!glob && endw) {
                        uint32_t kd = 0xFFFFFFFFu, ku = 0xFFFFFFFFu;
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int  j = tid + u * G, k = KBw + j;
                            const bool on = (uint32_t)(j - jlo) <= jspan && k >= nlo && k <= nhi;
                            const uint32_t cls = on ? lean_endclass(kM[u], k, n, m) : 0u;
                            if (cls != 0u) {
                                if (k <= Ak) kd = umin2(kd, lean_endkey(cls, (uint32_t)(Ak - k)));
                                else ku = umin2(ku, lean_endkey(cls, (uint32_t)(k - Ak - 1)));
                            }
                        }
                        if (__ballot(kd != 0xFFFFFFFFu || ku != 0xFFFFFFFFu) != 0ull) {
                            kd = (uint32_t)wave_min((int)(kd ^ 0x80000000u)) ^ 0x80000000u;
                            ku = (uint32_t)wave_min((int)(ku ^ 0x80000000u)) ^ 0x80000000u;
                            if (lane == 0) {
                                if (kd != 0xFFFFFFFFu) atomicMin(dir_ptr(si) + 5, kd);
                                if (ku != 0xFFFFFFFFu) atomicMin(dir_ptr(si) + 6, ku);
                            }
                        }
                    }
                    if (nhi >= nlo) put_ent(si, base + (uint64_t)(nlo - lo), nlo, nhi - nlo + 1);
                    else put_ent(si, 0ull, 0, 0);
                    // the halo cells of this row's ring slots: the neighbours' edge cells where the row keeps them
                    if (team_now && tid < 6) {
                        uint32_t *const row = (tid >> 1) == 0 ? nM : ((tid >> 1) == 1 ? nI : nD);
                        row[(tid & 1) ? SWf + 1 : 0] = (nhi >= nlo && halo_k >= nlo && halo_k <= nhi) ? halo : 0u;
                    }
                    if (term) {
                        done    = true;
                        s_final = s;
                        break;
                    }
                    lds_barrier();  // (G) the directory entry, the halo cells and the deletions, before the next head and its cells
                    TC_STAMP(8);
                    TC_COUNT(team_now ? 16 : 18);
                    s += g, si += 1u;
                    pM = pM + 1u == RM ? 0u : pM + 1u, pE = pE + 1u == RE ? 0u : pE + 1u;
                    int *const t_ = acc;
                    acc = accn, accn = t_;
                    if (fast_budget > 0 && --fast_budget == 0) {  // back to the pipelined steps
                        back_to_pipe = true;
                        break;
                    }
                }
                if (done) break;
                lds_barrier();  // (the general head resets red[0 .. 14]: nobody is still reading a scratch set)
                if (back_to_pipe) {
                    fast_budget = -1;
                    s -= g;  // (the loop head again, for score s)
                    continue;
                }
            }
            const uint32_t si = s / g;
            auto uni = [](DirEnt d) {
                auto r = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                DirEnt u;
                u.base   = (uint64_t)r((uint32_t)d.base) | ((uint64_t)r((uint32_t)(d.base >> 32)) << 32);
                u.lo     = (int)r((uint32_t)d.lo), u.w = (int)r((uint32_t)d.w), u.stride = 0u;
                u.pad[0] = u.pad[1] = u.pad[2] = 0u;
                return u;
            };
            const DirEnt eX = uni((s >= x) ? ring[(si - dx) % TEAM_RING] : none);
            const DirEnt eO = uni((s >= oe) ? ring[(si - doe) % TEAM_RING] : none);
            const DirEnt eE = uni((s >= e) ? ring[(si - de) % TEAM_RING] : none);
            const bool   seeded = (s == 0u) || (s == x);
            int lo = INT32_MAX, hi = INT32_MIN;
            if (eX.w > 0) lo = imin2(lo, eX.lo - 1), hi = imax2(hi, eX.lo + eX.w);
            if (eO.w > 0) lo = imin2(lo, eO.lo - 1), hi = imax2(hi, eO.lo + eO.w);
            if (eE.w > 0) lo = imin2(lo, eE.lo - 1), hi = imax2(hi, eE.lo + eE.w);
            lo = imax2(lo, -(n - 1));  // wfa.go:562-563
            hi = imin2(hi, m - 1);
            if (s == 0u) lo = INT32_MAX, hi = INT32_MIN;
            if (seeded) lo = imin2(lo, seed_lo), hi = imax2(hi, seed_hi);
            const int64_t W = (hi >= lo) ? ((int64_t)hi - lo + 1) : 0;
            TC_STAMP(11);  // head: ring entries, ranges
            TC_TRACE(s, 1u | (mode << 4));

            // ---- room for the row (one word per diagonal) and its directory entry
            if (!paged) {
                if (top + (uint64_t)W + (uint64_t)DIR_WORDS * (si + 2) > cap) {
                    TC_TRACE(s, 11u);
                    overflow = true;
                    break;
                }
            } else {
                if (si + 2u > dir_entries || (uint64_t)W > page_words) {
                    overflow = true;
                    break;
                }
                if (top + (uint64_t)W > page_end) {
                    uint32_t pg;
                    if (mode == TC_XBUF || mode == TC_STRIPE_T) {
                        if (lead_wg && tid == 0) ast(&ctl[112], page_alloc());
                        team_barrier();
                        if (aborted) TC_ABORT_RET;
                        pg = ald(&ctl[112]);
                    } else {
                        if (tid == 0) red[8] = (int)page_alloc();
                        __syncthreads();
                        pg = (uint32_t)red[8];
                        __syncthreads();
                    }
                    if (pg == 0xFFFFFFFFu) {
                        overflow = true;
                        break;
                    }
                    top = (uint64_t)pg << P.page_words_log2, page_end = top + page_words;
                }
            }
            TC_STAMP(12);  // head: room
            lds_barrier();  // everybody has read the ring entries before the slot of this score is rewritten
            TC_STAMP(13);  // head: first barrier

            // ---- scout pass: a row past the seeds that one workgroup's stripe cannot hold is a team's work
            if (X.scout != 0u && T == 1u && s > x && W > (int64_t)TC_STRIPE - 2) {
                too_wide = true;
                break;
            }
            // ---- the mode of this step (every active workgroup computes the same W)
            const uint32_t want = want_mode(W, s);
            const bool was_team = mode == TC_XBUF || mode == TC_STRIPE_T, will_team = want == TC_XBUF || want == TC_STRIPE_T;
            const int64_t slk   = (int64_t)X.slack < 256 ? (int64_t)X.slack : 256;
            const bool moved = want == mode && moved_now(lo, hi, W);
            if (want != mode || moved) {
                TC_TRACE(s, 2u | (want << 4));
                TC_TRACEN(0, s, mode | (want << 4));
                if (rings_in_lds) {  // the rings go to the exchange rows, where every mode can pick them up
                    dump_rings(si);
                    rings_in_lds = false;
                }
                if (was_team && T > 1u) {
                    team_barrier();  // (fenced: the dumps, and in XBUF mode the rows, are visible to whoever loads them next)
                    if (aborted) TC_ABORT_RET;
                } else {
                    __threadfence();
                    __syncthreads();
                }
                if (was_team && !will_team && T > 1u && !lead_wg) {
                    // ---- parked: workgroup 0 goes on alone until the row is wide again or the pair is over
                    bool resumed = false;
                    for (;;) {
                        team_barrier();
                        if (aborted) TC_ABORT_RET;
                        const uint32_t cmd = ald(&ctl[4]);
                        TC_TRACE(s, 6u | (cmd << 4));
                        TC_TRACEN(1, s, cmd);
                        if (cmd == TEAM_CMD_DONE) {
                            s_final = ald(&ctl[5]);
                            const uint32_t fl = ald(&ctl[10]);
                            done = (fl & 1u) != 0u, overflow = (fl & 2u) != 0u;
                            break;
                        }
                        if (cmd == TEAM_CMD_RESUME) {
                            s   = ald(&ctl[5]);
                            top = (uint64_t)ald(&ctl[6]) | ((uint64_t)ald(&ctl[7]) << 32);
                            page_end = (uint64_t)ald(&ctl[113]) | ((uint64_t)ald(&ctl[114]) << 32);
                            const uint32_t si2 = s / g;
                            if (tid < TEAM_RING && (uint32_t)tid < si2) {
                                const uint32_t idx = si2 - 1u - (uint32_t)tid;
                                ring[idx % TEAM_RING] = load_dir(dir_ptr(idx));
                            }
                            n_ent = si2;
                            mode  = TC_XBUF;  // (the loop head of score s decides the real one; the rings are in the exchange rows)
                            resumed = true;
                            __syncthreads();
                            break;
                        }
                    }
                    if (!resumed) break;
                    s -= g;
                    continue;
                }
                if (!was_team && will_team && T > 1u) {
                    // ---- workgroup 0 wakes the others: where the pair is
                    if (tid == 0) {
                        ast(&ctl[5], s), ast(&ctl[6], (uint32_t)top), ast(&ctl[7], (uint32_t)(top >> 32));
                        ast(&ctl[113], (uint32_t)page_end), ast(&ctl[114], (uint32_t)(page_end >> 32));
                        ast(&ctl[4], (uint32_t)TEAM_CMD_RESUME);
                    }
                    team_barrier();
                    if (aborted) TC_ABORT_RET;
                    // (like the workgroups it has woken: the loop head of score s again, as a team, the rings in the exchange rows --
                    // every workgroup then takes the same path into the mode the row asks for)
                    mode = TC_XBUF;
                    s -= g;
                    continue;
                }
                mode = want;
                if (mode == TC_STRIPE_T || mode == TC_STRIPE_S) {
                    uint32_t sw_t = (((uint32_t)(W + 2 * slk) + T - 1u) / T + 63u) & ~63u;
                    sw_t          = sw_t > (uint32_t)TC_STRIPE ? (uint32_t)TC_STRIPE : sw_t;
                    SWd = mode == TC_STRIPE_T ? (int)sw_t : TC_STRIPE;
                    const int64_t cd = mode == TC_STRIPE_T ? (int64_t)T * SWd : (int64_t)SWd, slack = (cd - W) / 2;
                    KB = lo - (int)(slack < (int64_t)X.slack ? slack : (int64_t)X.slack);
                    load_rings(si);
                    rings_in_lds = true;
                    __syncthreads();
                    TC_COUNT(19);
                }
            }

            // ---- WAVE mode (workgroup 0): wave 0 steps alone while the rows stay within 64 diagonals
            if (mode == TC_WAVE) {
                if (tid < 64) {
                    const uint32_t rmask = X.wave_rows - 1u;
                    auto wrow = [&](uint32_t idx, int comp) { return wring + (((idx & rmask) * 3u + (uint32_t)comp) << 6); };
                    auto rfl  = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                    // the rows the next steps can source: from the exchange rows into the LDS ring
                    for (uint32_t r = 1; r <= X.wave_rows && r <= si && r <= RM; r++) {
                        const DirEnt d = ring[(si - r) % TEAM_RING];
                        if (d.w > 0 && d.w <= 64 && lane < d.w) {
                            const uint32_t sl = (uint32_t)(d.lo + lane) & 63u;
#pragma unroll
                            for (int c = 0; c < 3; c++)
                                if (c == 0 || r <= RE) wrow(si - r, c)[sl] = xrow(c, si - r)[d.lo + lane + xoff];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int2 *const ring_lw = reinterpret_cast<const int2 *>(ring);
                    uint32_t wflags = 0, su = rfl(s), sj = rfl(si);
                    uint64_t utop = (uint64_t)rfl((uint32_t)top) | ((uint64_t)rfl((uint32_t)(top >> 32)) << 32);
                    uint64_t wcells = 0;
                    for (;; su += g, sj++) {
                        int xlo = 0, xw_ = 0, olo = 0, ow_ = 0, elo = 0, ew = 0;
                        if (su >= x) {
                            const int2 v = ring_lw[((sj - dx) % TEAM_RING) * 4u + 1u];
                            xlo = (int)rfl((uint32_t)v.x), xw_ = (int)rfl((uint32_t)v.y);
                        }
                        if (su >= oe) {
                            const int2 v = ring_lw[((sj - doe) % TEAM_RING) * 4u + 1u];
                            olo = (int)rfl((uint32_t)v.x), ow_ = (int)rfl((uint32_t)v.y);
                        }
                        if (su >= e) {
                            const int2 v = ring_lw[((sj - de) % TEAM_RING) * 4u + 1u];
                            elo = (int)rfl((uint32_t)v.x), ew = (int)rfl((uint32_t)v.y);
                        }
                        const bool wseed = (su == 0u) || (su == x);
                        int wlo = INT32_MAX, whi = INT32_MIN;
                        if (xw_ > 0) wlo = imin2(wlo, xlo - 1), whi = imax2(whi, xlo + xw_);
                        if (ow_ > 0) wlo = imin2(wlo, olo - 1), whi = imax2(whi, olo + ow_);
                        if (ew > 0) wlo = imin2(wlo, elo - 1), whi = imax2(whi, elo + ew);
                        wlo = imax2(wlo, -(n - 1)), whi = imin2(whi, m - 1);
                        if (su == 0u) wlo = INT32_MAX, whi = INT32_MIN;
                        if (wseed) wlo = imin2(wlo, seed_lo), whi = imax2(whi, seed_hi);
                        const int64_t WW = (whi >= wlo) ? ((int64_t)whi - wlo + 1) : 0;
                        if (WFA_RARE(dir_entries == 0u ? utop + (uint64_t)WW + (uint64_t)DIR_WORDS * (sj + 2) > cap
                                                       : (utop + (uint64_t)WW > page_end || sj + 2u > dir_entries))) {
                            wflags = WAVE_OVERFLOW;
                            break;
                        }
                        if (WFA_RARE(WW > 64)) {
                            wflags = WAVE_WIDE;
                            break;
                        }
                        auto wput = [&](uint64_t base_, int lo_, int w_, uint32_t kd, uint32_t ku) {
                            if (lane == 0) {
                                DirEnt d;
                                d.base = base_, d.lo = lo_, d.w = w_, d.stride = 0u, d.pad[0] = kd, d.pad[1] = ku, d.pad[2] = 0u;
                                ring[sj % TEAM_RING] = d;
                                uint32_t *const dp = dir_ptr(sj);
                                store_dir(dp, base_, lo_, w_, 0u);
                                dp[5] = kd, dp[6] = ku;
                            }
                        };
                        if (WFA_RARE(WW == 0)) {
                            wput(0ull, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            continue;
                        }
                        const uint64_t wbase = utop;
                        const int      k     = wlo + lane;
                        const bool     on    = lane < (int)WW;
                        const uint32_t sl    = (uint32_t)k & 63u;
                        auto wsrc = [&](int dlo, int dw, uint32_t idx, int comp, int kk) -> uint32_t {
                            return (kk >= dlo && kk < dlo + dw) ? wrow(idx, comp)[(uint32_t)kk & 63u] : 0u;
                        };
                        LCell c = {0u, 0u, 0u, 0u};
                        if (on) {
                            if (su != 0u)
                                c = lean_next(wsrc(olo, ow_, sj - doe, 0, k - 1), wsrc(elo, ew, sj - de, 1, k - 1), wsrc(olo, ow_, sj - doe, 0, k + 1),
                                              wsrc(elo, ew, sj - de, 2, k + 1), wsrc(xlo, xw_, sj - dx, 0, k), k, n, m);
                            if (wseed && c.M == 0u) {
                                bool mt_ = false;
                                c.M = lean_seed<MODE>(sv, k, su, x, glob, mt_);
                                c.wd = c.M != 0u ? (mt_ ? BLK_SEED_MATCH : BLK_SEED_MISMATCH) : 0u;
                            }
                            c.M = lean_extend<MODE>(sv, c.M, k);
                            A[wbase + lane] = c.wd;
                            wrow(sj, 0)[sl] = c.M, wrow(sj, 1)[sl] = c.I, wrow(sj, 2)[sl] = c.D;
                            // (write-through to the exchange rows: a wider row takes over from there)
                            xrow(0, sj)[k + xoff] = c.M, xrow(1, sj)[k + xoff] = c.I, xrow(2, sj)[k + xoff] = c.D;
                            wcells += (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                        }
                        utop += (uint64_t)WW;
                        const unsigned long long bM = __ballot(c.M != 0u);
                        if (WFA_RARE(bM == 0ull)) {
                            wput(0ull, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu);
                        } else {
                            const int  wmlo = wlo + (int)__builtin_ctzll(bM), wmhi = wlo + 63 - (int)__builtin_clzll(bM);
                            const int  dd   = lean_dist(c.M, k, n, m);
                            const bool hitf = c.M != 0u && k == Ak && (int)c.M >= m;
                            int wnlo = wmlo, wnhi = wmhi;
                            const bool fin = __ballot(hitf) != 0ull;
                            const unsigned long long bV = __ballot(dd >= 0);
                            if (!fin && P.adaptive && (wmhi - wmlo + 1) >= (int)P.min_wf_len && bV != 0ull) {
                                const int wmind = wave_min(dd >= 0 ? dd : INT32_MAX);
                                const unsigned long long bFail = __ballot(dd >= 0 && dd - wmind > (int)P.max_dist_diff);
                                const unsigned long long bOk   = bV & ~bFail;
                                if (bFail != 0ull) {
                                    const int first_ok = wlo + (int)__builtin_ctzll(bOk), last_ok = wlo + 63 - (int)__builtin_clzll(bOk);
                                    const unsigned long long bEnd = bM & ~bV;
                                    const int hitmin = bEnd != 0ull ? wlo + (int)__builtin_ctzll(bEnd) : INT32_MAX;
                                    if (hitmin >= first_ok) {
                                        wnlo = first_ok, wnhi = last_ok;
                                    } else {
                                        const unsigned long long below = bV & ((1ull << (first_ok - wlo)) - 1ull);
                                        wnlo = below != 0ull ? wlo + 63 - (int)__builtin_clzll(below) + 1 : wmlo;
                                        wnhi = last_ok;
                                    }
                                    if (on && (k < wnlo || k > wnhi)) wcells -= (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                                }
                            }
                            // the cells that end the reference's end-cell scan on either side of the final diagonal (semi-global)
                            uint32_t kd = 0xFFFFFFFFu, ku = 0xFFFFFFFFu;
                            if (!glob) {
                                const bool     keep = on && k >= wnlo && k <= wnhi;
                                const uint32_t cls  = keep ? lean_endclass(c.M, k, n, m) : 0u;
                                const uint32_t key  = cls != 0u ? lean_endkey(cls, (uint32_t)(k <= Ak ? Ak - k : k - Ak - 1)) : 0xFFFFFFFFu;
                                kd = (uint32_t)wave_min((int)((k <= Ak ? key : 0xFFFFFFFFu) ^ 0x80000000u)) ^ 0x80000000u;
                                ku = (uint32_t)wave_min((int)((k > Ak ? key : 0xFFFFFFFFu) ^ 0x80000000u)) ^ 0x80000000u;
                            }
                            if (wnhi >= wnlo) wput(wbase + (uint64_t)(wnlo - wlo), wnlo, wnhi - wnlo + 1, kd, ku);
                            else wput(0ull, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu);
                            if (WFA_RARE(fin)) {
                                const int hfl = hitf ? (int)c.M : 0;
                                h_final = (uint32_t)wave_max(hfl);
                                wflags  = WAVE_DONE;
                                break;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (tid == 0) {
                        unsigned int *const ur = reinterpret_cast<unsigned int *>(red);
                        ur[0] = su, ur[1] = (uint32_t)utop, ur[2] = (uint32_t)(utop >> 32), ur[3] = (wflags == WAVE_DONE) ? sj + 1u : sj, ur[4] = wflags,
                        ur[5] = h_final;
                    }
                    my_cells += wcells;
                }
                __syncthreads();
                {
                    const unsigned int *const ur = reinterpret_cast<const unsigned int *>(red);
                    s = ur[0], top = (uint64_t)ur[1] | ((uint64_t)ur[2] << 32), n_ent = ur[3];
                    const uint32_t wf = ur[4];
                    TC_TRACE(s, 7u | (wf << 4));
                    TC_TRACEN(2, s, wf);
                    if (wf & WAVE_DONE) done = true, s_final = s, h_final = ur[5];
                    if ((wf & WAVE_OVERFLOW) && !(paged && s / g + 2u <= dir_entries)) overflow = true;
                }
                __syncthreads();
                TC_STAMP(9);  // wave mode
                if (done || overflow) break;
                s -= g;  // the row at s is wider than 64 (or its page is full): redo the loop head for it
                continue;
            }

            if (W == 0) {
                put_ent(si, 0ull, 0, 0);
                if (lead_wg && tid == 0) dir_ptr(si)[5] = 0xFFFFFFFFu, dir_ptr(si)[6] = 0xFFFFFFFFu;
                n_ent = si + 1;
                __syncthreads();
                continue;
            }
            const uint64_t base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)top) |
                                  ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(top >> 32)) << 32);
            uint32_t *const rowC = A + base;  // the row's backtrace words, diagonal lo first
            const bool team_now = mode == TC_XBUF || mode == TC_STRIPE_T;
            if (tid == 0) {
                red[0] = INT32_MAX, red[1] = INT32_MIN, red[2] = 0, red[3] = INT32_MAX, red[4] = INT32_MAX, red[5] = INT32_MIN, red[6] = INT32_MAX,
                red[7] = INT32_MIN, red[11] = INT32_MAX, red[12] = INT32_MIN, red[13] = INT32_MAX, red[14] = INT32_MIN, red[10] = 0;
                // the end-cell keys of this score's directory entry start as "none" (atomic minima below; semi-global only)
                if (lead_wg && !glob) ast(dir_ptr(si) + 5, 0xFFFFFFFFu), ast(dir_ptr(si) + 6, 0xFFFFFFFFu);
            }
            const int KBw = KB + (mode == TC_STRIPE_T ? (int)b * SWd : 0);
            TC_STAMP(14);  // head: mode, scratch
            lds_barrier();

            TC_TRACE(s, 3u | (mode << 4));
            TC_STAMP(0);  // loop head: ranges, room, mode, halo cells
            // ---- P1: next + seeds + extend, the row's backtrace words, partial reductions
            int mlo = INT32_MAX, mhi = INT32_MIN, term = 0, mind = INT32_MAX, maxd = INT32_MIN, fvm = INT32_MAX, lvm = INT32_MIN;
            // the thread's cells of this row: the extended M offsets stay in registers to the end of the step (band ends, end-cell keys);
            // I and D only until they have entered the rings; the diagonals are recomputed where they are needed (cell_k) -- the kernel
            // lives on 128 registers, and every value that stays alive across the exchanges is one the compiler may have to spill
            uint32_t kM[TC_U], kI[TC_U], kD[TC_U];
#pragma unroll
            for (int u = 0; u < TC_U; u++) kM[u] = kI[u] = kD[u] = 0u;
            const bool stripe_mode = mode == TC_STRIPE_T || mode == TC_STRIPE_S;
            const bool hO = eO.w > 0, hX = eX.w > 0, hE = eE.w > 0;
            const uint32_t *const lO = lrow(0, s >= oe ? si - doe : 0u), *const lXr = lrow(0, s >= x ? si - dx : 0u);
            const uint32_t *const lE1 = lrow(1, s >= e ? si - de : 0u), *const lE2 = lrow(2, s >= e ? si - de : 0u);
            const uint32_t *const xO = xrow(0, s >= oe ? si - doe : 0u), *const xX = xrow(0, s >= x ? si - dx : 0u);
            const uint32_t *const xE1 = xrow(1, s >= e ? si - de : 0u), *const xE2 = xrow(2, s >= e ? si - de : 0u);
            auto xsrc = [&](const DirEnt &d, const uint32_t *row, int k) -> uint32_t {
                return (d.w > 0 && k >= d.lo && k < d.lo + d.w) ? ald(row + k + xoff) : 0u;
            };
            // XBUF mode: cells dealt round-robin over the team (or the workgroup), as many passes as the row needs
            const int64_t xstep = (int64_t)(team_now ? T : 1u) * G, x0i = team_now ? (int64_t)b * G + tid : tid;
            const int     npass = stripe_mode ? 1 : (int)((W + TC_U * xstep - 1) / (TC_U * xstep));
            // diagonal of the thread's u-th cell of the first pass; INT32_MIN: it has none
            const auto cell_k = [&](int u) -> int {
                if (stripe_mode) {
                    const int j = tid + u * G, k = KBw + j;
                    return (j < SWd && k >= lo && k <= hi) ? k : INT32_MIN;
                }
                const int64_t i = x0i + (int64_t)u * xstep;
                return i < W ? lo + (int)i : INT32_MIN;
            };
            for (int pass = 0; pass < npass; pass++) {
                // (one cell at a time: the five sources of a cell are five registers, not twenty -- the kernel has 128 to live on, and a
                // value spilled to scratch memory is a round trip that also waits for every store in front of it)
#pragma unroll
                for (int u = 0; u < TC_U; u++) {
                    int      k;
                    bool     on;
                    uint32_t sa = 0u, sb = 0u, sc_ = 0u, sd = 0u, sx = 0u;
                    if (stripe_mode) {
                        const int j = tid + u * G;
                        k = KBw + j, on = j < SWd && k >= lo && k <= hi;
                        // (the rings hold zero wherever a row has no cell: no range checks -- but a score without a row has no slot)
                        if (on) sa = hO ? lO[j] : 0u, sb = hE ? lE1[j] : 0u, sc_ = hO ? lO[j + 2] : 0u, sd = hE ? lE2[j + 2] : 0u, sx = hX ? lXr[j + 1] : 0u;
                    } else {
                        const int64_t i = x0i + ((int64_t)pass * TC_U + u) * xstep;
                        on = i < W, k = lo + (int)(on ? i : 0);
                        if (on && s != 0u) {
                            sa = xsrc(eO, xO, k - 1), sb = xsrc(eE, xE1, k - 1), sc_ = xsrc(eO, xO, k + 1), sd = xsrc(eE, xE2, k + 1);
                            sx = xsrc(eX, xX, k);
                        }
                    }
                    if (!on) continue;
                    LCell c = {0u, 0u, 0u, 0u};
                    if (s != 0u) c = lean_next(sa, sb, sc_, sd, sx, k, n, m);
                    if (seeded && c.M == 0u) {  // Set = last write wins (R2)
                        bool mt_ = false;
                        c.M  = lean_seed<MODE>(sv, k, s, x, glob, mt_);
                        c.wd = c.M != 0u ? (mt_ ? BLK_SEED_MATCH : BLK_SEED_MISMATCH) : 0u;
                    }
                    c.M = lean_extend<MODE>(sv, c.M, k);
                    // (wave 0 polls the exchanges: loads that would queue behind its stores -- its words wait in LDS for the end of the step)
                    if (stripe_mode && tid < 64) wsc[tid + 64 * u] = c.wd;
                    else cst(rowC + (k - lo), c.wd);
                    my_cells += (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                    if (!stripe_mode) {  // the row itself lives in the exchange rows
                        uint32_t *const nx = xrow(0, si), *const ni = xrow(1, si), *const nd = xrow(2, si);
                        xst(nx + k + xoff, c.M), xst(ni + k + xoff, c.I), xst(nd + k + xoff, c.D);
                    }
                    if (pass == 0) kM[u] = c.M, kI[u] = c.I, kD[u] = c.D;
                    if (c.M != 0u) {
                        mlo = imin2(mlo, k), mhi = imax2(mhi, k);
                        if (k == Ak && (int)c.M >= m) term = 1, h_final = c.M;  // wfa.go:235-239
                        const int d = lean_dist(c.M, k, n, m);
                        if (d >= 0) mind = imin2(mind, d), maxd = imax2(maxd, d), fvm = imin2(fvm, k), lvm = imax2(lvm, k);
                    }
                }
            }
            TC_STAMP(1);  // cells
            mlo = wave_min(mlo), mhi = wave_max(mhi), mind = wave_min(mind), maxd = wave_max(maxd);
            fvm = wave_min(fvm), lvm = wave_max(lvm);
            {
                const unsigned long long bt = __ballot(term);
                if (bt != 0ull && lane == 0) red[2] = 1;
                if (term) red[10] = (int)h_final;  // (one cell of the team sits on the final diagonal)
            }
            if (lane == 0) {
                atomicMin(&red[0], mlo), atomicMax(&red[1], mhi), atomicMin(&red[3], mind), atomicMax(&red[12], maxd);
                atomicMin(&red[13], fvm), atomicMax(&red[14], lvm);
            }
            TC_STAMP(2);  // wave reductions
            lds_barrier();
            TC_STAMP(3);  // the other waves of the workgroup
            if (stripe_mode) {
                // the new rows enter the rings (the slots of the oldest rows: everybody has read them), and the stripe's two edge
                // cells go to the exchange rows for the neighbours
                uint32_t *const nM = lrow(0, si), *const nI = lrow(1, si), *const nD = lrow(2, si);
#pragma unroll
                for (int u = 0; u < TC_U; u++)
                    if (tid + u * G < SWd) nM[tid + u * G + 1] = kM[u], nI[tid + u * G + 1] = kI[u], nD[tid + u * G + 1] = kD[u];
                if (mode == TC_STRIPE_T) {
#pragma unroll
                    for (int u = 0; u < TC_U; u++) {
                        const int j = tid + u * G;
                        // the stripe's first / last cell: its neighbours' halo, handed over with the exchange
                        if (j == 0) red[24] = (int)kM[u], red[25] = (int)kI[u], red[26] = (int)kD[u];
                        if (j == SWd - 1) red[28] = (int)kM[u], red[29] = (int)kI[u], red[30] = (int)kD[u];
                    }
                } else if (tid < 6) {  // (one stripe: nothing lies beyond it)
                    lrow(tid >> 1, si)[(tid & 1) ? SWd + 1 : 0] = 0u;
                }
            }
            // ---- exchange 1: the row's ranges, termination, the distances
            if (tid == 0) {
                // (a maximum travels as its complement ~v: order-reversing like the negation, and "none" -- INT32_MIN -- has one)
                red[16] = red[0], red[17] = ~red[1], red[18] = red[2] ? -1 : 0, red[19] = red[3], red[20] = ~red[12], red[21] = red[13], red[22] = ~red[14],
                red[23] = red[2] ? ~red[10] : INT32_MAX;
            }
            TC_TRACE(s, 4u | (mode << 4));
            TC_STAMP(4);  // rows into the rings, edge cells out
            if (team_now) {
                exchange(!stripe_mode, mode == TC_STRIPE_T);
                if (aborted) TC_ABORT_RET;
            } else {
                lds_barrier();
            }
            uint32_t halo = 0u;
            int      halo_k = 0;
            if (mode == TC_STRIPE_T && tid < 6) {  // the neighbours' edge cells of this row: the halo of its ring slots
                halo_k = (tid & 1) ? KBw + SWd : KBw - 1;
                halo   = (uint32_t)red[40 + 3 * (tid & 1) + (tid >> 1)];
            }
            TC_STAMP(5);  // exchange 1
            TC_TRACE(s, 5u | (mode << 4));
            mlo = red[16], mhi = ~red[17], term = red[18] != 0, mind = red[19], maxd = ~red[20], fvm = red[21], lvm = ~red[22];
            if (term) h_final = (uint32_t)(~red[23]);
            top += (uint64_t)W;
            n_ent = si + 1;

            // ---- reduce (wfa.go:461-540): the band wf-adaptive keeps, from the cells' owners
            int nlo = mlo, nhi = mhi;
            if (!term && P.adaptive && mhi >= mlo && (mhi - mlo + 1) >= (int)P.min_wf_len && mind != INT32_MAX && maxd - mind > (int)P.max_dist_diff) {
                const int thr = mind + (int)P.max_dist_diff;
                int f_ok = INT32_MAX, l_ok = INT32_MIN, hmin = INT32_MAX;
                auto p2 = [&](uint32_t hM, int k) {
                    const int d = lean_dist(hM, k, n, m);
                    if (d >= 0) {
                        if (d <= thr) f_ok = imin2(f_ok, k), l_ok = imax2(l_ok, k);
                    } else if (hM != 0u) {
                        hmin = imin2(hmin, k);  // a present cell at / past a sequence end
                    }
                };
#pragma unroll
                for (int u = 0; u < TC_U; u++)
                    if (cell_k(u) != INT32_MIN) p2(kM[u], cell_k(u));
                if (!stripe_mode)
                    for (int pass = 1; pass < npass; pass++)
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int64_t i = x0i + ((int64_t)pass * TC_U + u) * xstep;
                            if (i < W) p2(xld(xrow(0, si) + lo + (int)i + xoff), lo + (int)i);
                        }
                f_ok = wave_min(f_ok), l_ok = wave_max(l_ok), hmin = wave_min(hmin);
                if (lane == 0) atomicMin(&red[4], f_ok), atomicMax(&red[5], l_ok), atomicMin(&red[11], hmin);
                lds_barrier();
                if (tid == 0) {
                    red[16] = red[4], red[17] = ~red[5], red[18] = red[11];
#pragma unroll
                    for (int f = 19; f < 24; f++) red[f] = 0;
                }
                TC_STAMP(6);  // band ends: the owners' candidates
                if (team_now) {
                    exchange(!stripe_mode, false);
                    if (aborted) TC_ABORT_RET;
                } else {
                    lds_barrier();
                }
                TC_STAMP(7);  // exchange 2
                const int first_ok = red[16], last_ok = ~red[17], hitmin = red[18];
                if (hitmin >= first_ok) {
                    // wfa.go:509-511 with no present-but-unusable cell below first_ok: the entries between the last leading failure
                    // and first_ok are holes, and dropping or keeping a hole is the same row
                    nlo = first_ok, nhi = last_ok;
                } else {
                    // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                    int lead = INT32_MIN;
                    auto p3 = [&](uint32_t hM, int k) {
                        if (k < first_ok && lean_dist(hM, k, n, m) >= 0) lead = imax2(lead, k);
                    };
#pragma unroll
                    for (int u = 0; u < TC_U; u++)
                        if (cell_k(u) != INT32_MIN) p3(kM[u], cell_k(u));
                    if (!stripe_mode)
                        for (int pass = 1; pass < npass; pass++)
#pragma unroll
                            for (int u = 0; u < TC_U; u++) {
                                const int64_t i = x0i + ((int64_t)pass * TC_U + u) * xstep;
                                if (i < W) p3(xld(xrow(0, si) + lo + (int)i + xoff), lo + (int)i);
                            }
                    lead = wave_max(lead);
                    if (lane == 0) atomicMax(&red[7], lead);
                    lds_barrier();
                    if (tid == 0) {
                        red[16] = ~red[7];
#pragma unroll
                        for (int f = 17; f < 24; f++) red[f] = 0;
                    }
                    if (team_now) {
                        exchange(!stripe_mode, false);
                        if (aborted) TC_ABORT_RET;
                    } else {
                        lds_barrier();
                    }
                    lead = ~red[16];
                    nlo  = (lead != INT32_MIN) ? lead + 1 : mlo;
                    nhi  = last_ok;  // wfa.go:517-524
                }
                // Delete of wfa.go:526-535: the cells outside [nlo, nhi] stop existing -- in the rings (zero), in the census
                if (stripe_mode) {
                    uint32_t *const nM = lrow(0, si), *const nI = lrow(1, si), *const nD = lrow(2, si);
#pragma unroll
                    for (int u = 0; u < TC_U; u++) {
                        const int k = cell_k(u);
                        if (k != INT32_MIN && (k < nlo || k > nhi)) {
                            my_cells -= (kM[u] != 0u) + (nI[tid + u * G + 1] != 0u) + (nD[tid + u * G + 1] != 0u);
                            nM[tid + u * G + 1] = 0u, nI[tid + u * G + 1] = 0u, nD[tid + u * G + 1] = 0u;
                            kM[u] = 0u;
                        }
                    }
                } else {
                    for (int pass = 0; pass < npass; pass++)
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int64_t i = x0i + ((int64_t)pass * TC_U + u) * xstep;
                            const int     k = lo + (int)i;
                            if (i < W && (k < nlo || k > nhi)) {
                                my_cells -= (xld(xrow(0, si) + k + xoff) != 0u) + (xld(xrow(1, si) + k + xoff) != 0u) + (xld(xrow(2, si) + k + xoff) != 0u);
                                if (pass == 0) kM[u] = 0u;
                            }
                        }
                }
            }
            // ---- the cells that end the reference's end-cell scan (semi-global, wfa.go:301-361): nearest to the final diagonal on
            // either side, among the cells the row keeps
            if (!glob) {
                uint32_t kd = 0xFFFFFFFFu, ku = 0xFFFFFFFFu;
                auto p4 = [&](uint32_t hM, int k) {
                    if (k < nlo || k > nhi) return;
                    const uint32_t cls = lean_endclass(hM, k, n, m);
                    if (cls == 0u) return;
                    if (k <= Ak) kd = umin2(kd, lean_endkey(cls, (uint32_t)(Ak - k)));
                    else ku = umin2(ku, lean_endkey(cls, (uint32_t)(k - Ak - 1)));
                };
#pragma unroll
                for (int u = 0; u < TC_U; u++)
                    if (cell_k(u) != INT32_MIN) p4(kM[u], cell_k(u));
                if (!stripe_mode)
                    for (int pass = 1; pass < npass; pass++)
#pragma unroll
                        for (int u = 0; u < TC_U; u++) {
                            const int64_t i = x0i + ((int64_t)pass * TC_U + u) * xstep;
                            if (i < W) p4(xld(xrow(0, si) + lo + (int)i + xoff), lo + (int)i);
                        }
                if (__ballot(kd != 0xFFFFFFFFu || ku != 0xFFFFFFFFu) != 0ull) {
                    kd = (uint32_t)wave_min((int)(kd ^ 0x80000000u)) ^ 0x80000000u;
                    ku = (uint32_t)wave_min((int)(ku ^ 0x80000000u)) ^ 0x80000000u;
                    if (lane == 0) {
                        // (the entry's key words were set to "none" before exchange 1 of this step; the pair's last barrier is fenced)
                        if (kd != 0xFFFFFFFFu) atomicMin(dir_ptr(si) + 5, kd);
                        if (ku != 0xFFFFFFFFu) atomicMin(dir_ptr(si) + 6, ku);
                    }
                }
            }
            if (nhi >= nlo) put_ent(si, base + (uint64_t)(nlo - lo), nlo, nhi - nlo + 1);
            else put_ent(si, 0ull, 0, 0);
            // the halo cells of this row's ring slots: the neighbours' edge cells where the row keeps them
            if (mode == TC_STRIPE_T && tid < 6) lrow(tid >> 1, si)[(tid & 1) ? SWd + 1 : 0] = (nhi >= nlo && halo_k >= nlo && halo_k <= nhi) ? halo : 0u;
            if (stripe_mode && tid < 64) {  // (wave 0's share of the row's backtrace words)
#pragma unroll
                for (int u = 0; u < TC_U; u++) {
                    const int k = cell_k(u);
                    if (k != INT32_MIN) cst(rowC + (k - lo), wsc[tid + 64 * u]);
                }
            }
            if (term) {
                done    = true;
                s_final = s;
                break;
            }
            lds_barrier();
            TC_STAMP(8);  // deletions, end-cell keys, directory entry
            TC_COUNT(mode == TC_STRIPE_T ? 16 : mode == TC_XBUF ? 17 : 18);
        }
#ifdef WFA_TEAM_STAMPS
        if (lead_wg && tid == 0)
            for (int i = 0; i < 32; i++) atomicAdd(i < 24 ? reinterpret_cast<unsigned long long *>(ctl + 64) + i : reinterpret_cast<unsigned long long *>(trace + 64) + (i - 24), tacc[i]), tacc[i] = 0;
        // (and a workgroup in the middle of the team, for the skew between workgroups: in the trace words, unused in this build)
        if (!lead_wg && tid == 0)
            for (int i = 0; i < 32; i++) {
                if (b == T / 2u && i < 24) atomicAdd(reinterpret_cast<unsigned long long *>(trace) + i, tacc[i]);
                tacc[i] = 0;
            }
        tprev = __builtin_amdgcn_s_memrealtime();
#endif

        // ---- a pair that ends with workgroup 0 alone: wake the parked workgroups
        const bool alone = mode == TC_STRIPE_S || mode == TC_WAVE;
        TC_TRACEN(3, s, mode | (done ? 16u : 0u) | (overflow ? 32u : 0u) | (alone ? 64u : 0u));
        if (alone && lead_wg && T > 1u) {
            if (tid == 0) {
                ast(&ctl[5], s_final), ast(&ctl[10], (done ? 1u : 0u) | (overflow ? 2u : 0u));
                ast(&ctl[4], (uint32_t)TEAM_CMD_DONE);
            }
            team_barrier();
            if (aborted) TC_ABORT_RET;
        }
        // ---- stored cells across the team
        if (tid == 0) red[38] = 0, red[39] = 0;
        __syncthreads();
        {
            unsigned long long wsum = my_cells;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red[38]), wsum);
        }
        __syncthreads();
        if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 12), *reinterpret_cast<unsigned long long *>(&red[38]));
        TC_TRACE(s_final, 9u);
        team_barrier();  // every row, the directory with its end-cell keys and the cell count are visible to workgroup 0
        if (aborted) TC_ABORT_RET;
        TC_TRACE(s_final, 10u);
        if (!lead_wg) continue;  // what follows is workgroup 0's (the others wait at the next pair's barrier)

        if (overflow || !done) {
            if (tid == 0) {
                const uint32_t why = (too_wide && !overflow) ? (uint32_t)ST_REDO_WIDE : (uint32_t)ST_REDO_ARENA;
                rec[REC_STATUS] = why;
                push_redo(P, pair, why);
                if (paged) page_free_all();
            }
            continue;
        }
        // ---- the semi-global end cell (backtraceStartPosistion, wfa.go:270-375): the hit at the LOWEST score that has one; a score has
        // one when the nearest scan-ending cell above the final diagonal is a hit, else when the nearest one at / below it is
        // (the upward scan overrides the downward one at equal score, wfa.go:319-361)
        uint32_t minS = s_final, h_start = h_final;
        int      lastK = Ak;
        if (!glob) {
            unsigned long long best = ~0ull;
            for (uint32_t idx = (uint32_t)tid; idx <= s_final / g; idx += G) {
                const uint32_t *const dp = dir_ptr(idx);
                if ((int)dp[3] <= 0) continue;  // !M.HasScore(_s)
                const uint32_t kd = dp[5], ku = dp[6];
                uint32_t key = 0xFFFFFFFFu;
                int      kk  = 0;
                if (ku != 0xFFFFFFFFu && (ku & 3u) != 3u) key = ku, kk = Ak + 1 + (int)(ku >> 2);
                else if (kd != 0xFFFFFFFFu && (kd & 3u) != 3u) key = kd, kk = Ak - (int)(kd >> 2);
                if (key != 0xFFFFFFFFu) {
                    const unsigned long long v = ((unsigned long long)idx << 32) | (uint32_t)(kk + 0x40000000) << 1 | (key & 1u);
                    best = v < best ? v : best;
                }
            }
            // (minimum over the workgroup: score index first)
            unsigned long long wb = best;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long ot = __shfl_xor(wb, o, 64);
                wb = ot < wb ? ot : wb;
            }
            unsigned long long *const lb = reinterpret_cast<unsigned long long *>(&red[28]);
            if (tid == 0) *lb = ~0ull;
            __syncthreads();
            if (lane == 0 && wb != ~0ull) atomicMin(lb, wb);
            __syncthreads();
            const unsigned long long fb = *lb;
            if (fb != ~0ull) {
                minS  = (uint32_t)(fb >> 32) * g;
                lastK = (int)(((uint32_t)fb >> 1) & 0x7FFFFFFFu) - 0x40000000;
                h_start = (fb & 1ull) ? (uint32_t)m : (uint32_t)(n + lastK);  // hit through h == m / through v == n
            }
            __syncthreads();
        }
        if (X.dbg && tid == 0) X.dbg[0] = n_ent, X.dbg[1] = s_final, X.dbg[2] = minS, X.dbg[3] = (uint32_t)lastK;

        // ---- backtrace: wave 0 walks together over the compact rows; process() by its 64 lanes
        uint64_t scratch_end = 0;
        bool     no_scratch  = false;
        if (paged) {
            const uint64_t need = 2ull * ((uint64_t)n + (uint64_t)m + 8ull);
            if (tid == 0) red[8] = (page_end - top < need) ? (int)page_alloc() : -2;
            __syncthreads();
            const int r8 = red[8];
            if (r8 == -1) no_scratch = true;
            else if (r8 >= 0) top = (uint64_t)(uint32_t)r8 << P.page_words_log2, page_end = top + page_words;
            scratch_end = page_end;
            __syncthreads();
        }
        if (tid < 64) {
            bool ok = !no_scratch;
            if (ok) {
                DirCompactViewWave cv;
                cv.init(A, cap, g, n_ent, ring);
                const uint64_t scratch0 = (top + 1ull) & ~1ull;
                const uint64_t dir_lo   = scratch_end != 0ull ? scratch_end : cap - (uint64_t)DIR_WORDS * (uint64_t)n_ent;
                const uint64_t room     = dir_lo > scratch0 ? (dir_lo - scratch0) / 2ull : 0ull;
                OpsWriter ow;
                ow.init(reinterpret_cast<uint64_t *>(A + scratch0), (uint32_t)(room > 0xFFFFFFFFull ? 0xFFFFFFFFull : room));
                TraceOut to;
                back_trace_compact(cv, n, m, minS, lastK, h_start, x, P.o, e, ow, to, !glob);
                ok = !ow.overflow;
                if (ok) {
                    const uint32_t L = ow.n;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    uint32_t off_lo = 0, off_hi = 0;
                    if (lane == 0) {
                        const uint64_t o = atomicAdd(P.ops_cursor, (unsigned long long)L);
                        off_lo = (uint32_t)o, off_hi = (uint32_t)(o >> 32);
                    }
                    const uint64_t off = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)off_lo) |
                                         ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)off_hi) << 32);
                    int firstM = INT32_MAX, lastM = INT32_MIN;
                    for (uint32_t i = (uint32_t)lane; i < L; i += 64u) {
                        const uint64_t op = ow.buf[L - 1 - i];
                        if ((uint32_t)(op >> 32) == 'M') firstM = imin2(firstM, (int)i), lastM = imax2(lastM, (int)i);
                        if (off + i < P.ops_cap) P.ops[off + i] = op;
                    }
                    firstM = wave_min(firstM), lastM = wave_max(lastM);
                    const uint32_t begin = firstM != INT32_MAX ? (uint32_t)firstM : 0u, end = firstM != INT32_MAX ? (uint32_t)lastM : 0u;
                    unsigned int *const acc = reinterpret_cast<unsigned int *>(red);
                    if (lane == 0) acc[0] = acc[1] = acc[2] = acc[3] = 0u;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    uint32_t alen = 0, matches = 0, gaps = 0, regions = 0;
                    for (uint32_t i = begin + (uint32_t)lane; i <= end && i < L; i += 64u) {
                        const uint64_t op  = ow.buf[L - 1 - i];
                        const uint32_t cnt = (uint32_t)op, o = (uint32_t)(op >> 32);
                        alen += cnt;
                        if (o == 'M') matches += cnt;
                        else if (o == 'I' || o == 'D') gaps += cnt, regions++;
                    }
                    atomicAdd(&acc[0], alen), atomicAdd(&acc[1], matches), atomicAdd(&acc[2], gaps), atomicAdd(&acc[3], regions);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        rec[REC_STATUS] = ST_OK, rec[REC_SCORE] = to.score;
                        rec[REC_TBEGIN] = (uint32_t)to.tbegin, rec[REC_TEND] = (uint32_t)to.tend;
                        rec[REC_QBEGIN] = (uint32_t)to.qbegin, rec[REC_QEND] = (uint32_t)to.qend;
                        rec[REC_ALIGN_LEN] = acc[0], rec[REC_MATCHES] = acc[1], rec[REC_GAPS] = acc[2], rec[REC_GAP_REGIONS] = acc[3];
                        rec[REC_OPS_LEN] = L, rec[REC_OPS_OFF_LO] = (uint32_t)off, rec[REC_OPS_OFF_HI] = (uint32_t)(off >> 32);
                        rec[REC_CELLS_LO] = ald(&ctl[12]), rec[REC_CELLS_HI] = ald(&ctl[13]), rec[REC_N_SCORES] = s_final;
                    }
                }
            }
            if (!ok && tid == 0) {
                rec[REC_STATUS] = ST_REDO_ARENA;
                push_redo(P, pair, ST_REDO_ARENA);
            }
#ifdef WFA_TEAM_STAMPS
            TC_STAMP(10);  // end-cell lookup + backtrace + record
            if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 64) + 10, tacc[10]), tacc[10] = 0;
#endif
            if (paged && tid == 0 && !X.dbg) page_free_all();
        }
    }
}

}  // namespace wfa
