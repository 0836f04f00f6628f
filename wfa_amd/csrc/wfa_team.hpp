// wfa_team.hpp -- kernel E: the general alignment kernel for WIDE wavefronts, several workgroups per pair.
//
// Semi-global alignment seeds a cell on every diagonal (wfa.go:163-183), and wf-adaptive only removes diagonals
// that fall 50 bases behind the best one (wfa.go:507), so a 100 kbp pair keeps 2e5-diagonal wavefronts alive for
// thousands of scores: 1e9 and more stored cells per pair.  One workgroup per pair (wfa_generic_kernel) then runs
// at one CU's memory bandwidth while 255 CUs idle.  Here a TEAM of T workgroups (one per CU, all resident) works
// on one pair at a time: every score step cuts the diagonal range into T*1024-wide stripes, workgroup b takes the
// b-th 1024 diagonals of every stripe (coalesced rows), and the reductions of the step (tight range of M, the
// termination test, the wf-adaptive minimum distance, first/last surviving diagonal) go through block reductions
// in LDS, one global atomic per workgroup and a team barrier.  Same per-cell code as the generic kernel
// (next_cell / seed_word / extend_word / reduce_dist), same arena layout (rows + 32-byte directory), same
// backtrace.
//
// Inter-workgroup visibility (per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores):
// while the team works together every arena access is an agent-scope relaxed atomic load / store (sc1: served by
// the memory side, not by the XCD's L2), so the per-step barriers need no cache write-back or invalidate: they are
// __syncthreads (drains every wave's stores) -> one lane: atomic arrive, bounded spin on the counter with
// agent-scope loads -> __syncthreads.  Rows written before a barrier are read by other workgroups only after it.
// A write-through store whose vmcnt has returned is NOT yet at the memory side: a load issued straight after the
// barrier can overtake the last arriver's stores (measured on MI355X: a few stale words per 1e5 steps, none with a
// 0.25 us delay; with the same batch run twice the stale word is the right one and the error hides itself -- the
// tests poison the arena, option `arena_poison`).  So every barrier carries an agent-scope release (L2 write-back +
// wait) before the arrive: +1.7 us per score step, 0.54 -> 0.59 s on the configs[4] sample.  `team_strict = 0` drops
// it (every cross-workgroup read then comes one further memory-side round trip after the barrier: fast, and wrong
// once in ~40 runs of that sample).
//
// Narrow rows do not pay for barriers: when a row is at most TEAM_SOLO_MAX diagonals wide the team switches to
// SOLO mode -- workgroup 0 runs the steps alone with plain (L2-cached) accesses and __syncthreads only, the others
// park in a barrier -- and back to team mode when the row grows again (workgroup 0 publishes score, arena top and
// directory; the others reload the last directory entries).  The mode switches and the end of a pair use a FENCED
// barrier (agent-scope release before the arrive, acquire after the spin).  After wf-adaptive has collapsed the
// band of a semi-global pair the rest of the alignment runs solo at the generic kernel's speed.
//
// Rows of at most 64 diagonals -- what is left of a semi-global pair once wf-adaptive has collapsed its band:
// ~3e4 of the 3.3e4 score steps of a 100 kbp pair -- run in WAVE mode: wave 0 of workgroup 0 alone, one diagonal
// per lane, the last `wave_rows` rows of M, I and D in an LDS ring (slot = diagonal & 63: every source of a row of
// <= 64 diagonals lies inside the row's own range), the ranges and the wf-adaptive ends from ballots (lanes are
// ordered by diagonal), no barrier of any kind; rows and directory entries still go to the arena for the
// backtrace.  0.6 us per step against 5 us in solo mode.
//
// A barrier that does not complete within the spin bound raises the team's abort flag and every workgroup leaves
// (the host reports an internal error instead of hanging the device).
#pragma once
#include "wfa_device.hpp"
#include "wfa_wave.hpp"

namespace wfa {

constexpr int TEAM_THREADS   = 1024;
constexpr int TEAM_RING      = WAVE_DIR_RING;  // directory entries every workgroup keeps in LDS (sources reach back < 64 scores)
constexpr int TEAM_CTL_WORDS = 128;  // per team, in global memory: [0] barrier count [1] abort [2] work index
                                     // [3] XCC ids of the team (mask) [4] command [5] score [6..7] arena top [8..9] end-cell key (u64) [10] end flags [11] team on one XCD [12..13] stored cells (u64)
                                     // [16 + 16*set ..] three reduction sets [64..] diagnostic stamps
constexpr uint32_t TEAM_SPIN_LIMIT = 1u << 24;
constexpr int      TEAM_MAX_PAGES  = 1024;  // pages one pair can hold (its list in page_ctl)
#ifndef WFA_TEAM_U
#define WFA_TEAM_U 2
#endif
constexpr int      TEAM_U          = WFA_TEAM_U;     // cells of a thread in flight together / kept in registers per row
constexpr uint32_t TEAM_SOLO_MAX   = 4096;  // default: rows up to this width are done by workgroup 0 alone
enum : uint32_t { TEAM_CMD_NONE = 0, TEAM_CMD_RESUME = 1, TEAM_CMD_DONE = 2 };  // ctl[4]; ctl[5] = score, ctl[6..7] = top

struct TeamRed {  // one reduction set (global memory, 16 words)
    int mlo, mhi, term, mind, first_ok, last_ok, anyfail, lead, hitmin, maxd, fvm, lvm, pad[4];
};

template <int MODE>
__global__ __launch_bounds__(TEAM_THREADS) void wfa_team_kernel(const KParams P, uint32_t *team_ctl, uint32_t T,
                                                                uint32_t solo_max, uint32_t wave_rows, uint32_t strict) {
    constexpr int G = TEAM_THREADS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *const lq   = lds;
    uint32_t *const lt   = lds + P.lds_seq_words;
    int *const      red  = reinterpret_cast<int *>(lds + 2 * (MODE == 0 ? P.lds_seq_words : 0));  // 16 ints
    DirEnt *const   ring = reinterpret_cast<DirEnt *>(red + 16);                                   // TEAM_RING entries
    // wave mode: the last wave_rows (a power of two, 0 = wave mode off) rows, [row][component][diagonal & 63]
    uint32_t *const wring = reinterpret_cast<uint32_t *>(ring + TEAM_RING);

    const int      tid = threadIdx.x, lane = tid & 63;
    // Teams of one XCD's CUs (round 4; `strict` bit 1, n_teams in its upper half): team = blockIdx % 8.  Workgroups are dealt
    // round-robin over the eight XCDs, so such a team normally sits on ONE XCD, whose L2 all its CUs share.  A team of 32 CUs
    // steps as fast as one of 50 (a wide step is a chain of round trips, not throughput: 8 x 100 kbp 610 -> 589 ms).  Bit 2
    // (option team_xcd = 2, the default) adds the LOCAL protocol: every workgroup reports its XCC id, and a team that finds all
    // of them equal (`xl`) stores its rows plain -- they stay in the XCD's L2, where the team's sc1 loads find them -- and
    // drops the release from its barriers; any other placement runs the memory-side protocol unchanged, so nothing depends on
    // where the dispatcher puts a workgroup.  Why it is sound: within an XCD the L2 is the point of coherence; a plain store is
    // in it when the storing wave's vmcnt has returned, every barrier drains that (__syncthreads) before its arrive, and the
    // readers' sc1 loads bypass their own L1.  (Round 2's stale words were write-THROUGH stores overtaken on their way to the
    // memory side; no store travels that way here.)  Measured with four teams: 589 -> 580 ms; with the eight teams the paged
    // arena allows: 548 -> 512 ms; 310 passes of the configs[4] sample over a poisoned pool, every one bit-identical to the
    // memory-side protocol's result.  Teams beyond the number of arena slots leave at once.
    const bool     xmap    = (strict & 2u) != 0u, xl_ok = (strict & 4u) != 0u;
    const uint32_t n_teams = strict >> 16;
    strict &= 1u;
    const uint32_t team = xmap ? blockIdx.x % 8u : blockIdx.x / T, b = xmap ? blockIdx.x / 8u : blockIdx.x % T;
    if (xmap && team >= n_teams) return;
    uint32_t *const ctl = team_ctl + (uint64_t)team * TEAM_CTL_WORDS;
    // Paged arena (P.page_ctl != nullptr): the rows of a pair live in pages taken from a pool all teams share -- a pair holds
    // what it needs (0.2 .. 43 GB at 100 kbp) instead of a slot sized for the worst one, so eight teams run where four slots
    // fitted.  `base` of a directory entry is a word index into the pool either way; the directory of team t is the
    // dir_region_words below arena_words - t * dir_region_words.
    const bool      paged = P.page_ctl != nullptr;
    uint32_t *const A     = paged ? P.arena : P.arena + (uint64_t)team * P.arena_words;
    const uint64_t  cap   = paged ? P.arena_words - (uint64_t)team * P.dir_region_words : P.arena_words;
    const uint32_t  dir_entries = paged ? (uint32_t)(P.dir_region_words / DIR_WORDS) : 0u;
    const uint64_t  page_words  = 1ull << P.page_words_log2;
    uint32_t *const my_pages    = paged ? P.page_ctl + 4u + P.n_pages + team * (uint32_t)TEAM_MAX_PAGES : nullptr;
    const uint32_t x = P.x, oe = P.oe, e = P.e, g = P.g;
    const int64_t  stripe = (int64_t)T * G;

#ifdef WFA_TEAM_STAMPS
    unsigned long long tacc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memrealtime();
#define TEAM_STAMP(i)                                                  \
    do {                                                               \
        const unsigned long long _t = __builtin_amdgcn_s_memrealtime(); \
        tacc[i] += _t - tprev;                                         \
        tprev = _t;                                                    \
    } while (0)
#else
#define TEAM_STAMP(i) \
    do {              \
    } while (0)
#endif
    uint32_t bar_target = 0;  // barriers are counted: the n-th one completes at n*T arrivals
    bool     aborted    = false;
    bool     xl         = false;  // this team's workgroups share one XCD (checked below): rows travel through its L2
    auto team_barrier = [&](bool fenced) {
        __syncthreads();
        if (tid == 0) {
            bar_target += T;
            if (fenced)
                __threadfence();
            else if (strict && !xl)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (team_strict, the default: see the header comment)
            bool bad = false;
            // the last workgroup to arrive sees the full count in the value its own atomic returns.  The count runs
            // over all the pairs of a launch: compared modulo 2^32.
            if ((int32_t)(atomicAdd(&ctl[0], 1u) + 1u - bar_target) < 0) {
                uint32_t spins = 0;
                while ((int32_t)(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - bar_target) < 0) {
                    if ((++spins & 1023u) == 0u &&
                        (spins > TEAM_SPIN_LIMIT || __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        atomicExch(&ctl[1], 1u);
                        bad = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            if (fenced) __threadfence();
            red[15] = bad ? 1 : 0;
        }
        __syncthreads();
        aborted = red[15] != 0;
    };
    const auto ald = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const auto ast = [](uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto tred = [&](uint32_t set) { return reinterpret_cast<TeamRed *>(ctl + 16 + 16 * set); };
    auto reset_set = [&](TeamRed *r) {  // memory-side stores: the other workgroups' atomics must see them
        uint32_t *const w = reinterpret_cast<uint32_t *>(r);
        const int v[12] = {INT32_MAX, INT32_MIN, 0, INT32_MAX, INT32_MAX, INT32_MIN, 0, INT32_MIN, INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN};
#pragma unroll
        for (int i = 0; i < 12; i++) __hip_atomic_store(w + i, (uint32_t)v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- where the team's workgroups sit: one bit per XCC id seen (ctl[3]: zero at launch)
    if (xmap) {
        if (tid == 0) atomicOr(&ctl[3], 1u << (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 15u));  // HW_REG_XCC_ID [3:0]
        team_barrier(true);
        if (aborted) return;
        const uint32_t seen = __hip_atomic_load(&ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xl = xl_ok && (seen & (seen - 1u)) == 0u;
        if (b == 0 && tid == 0) ctl[11] = xl ? 1u : 0u;  // (diagnostics: WFAHIP_DEBUG_TIMING prints it)
    }
    // ---- pages: a stack of free page ids behind a spin lock (one thread of a team at a time; a pair turns a page every
    // ~100 wide score steps).  Everything inside the lock is a memory-side access.
    uint32_t n_pg = 0;  // pages the current pair holds (kept by thread 0 of workgroup 0)
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
    // An empty pool is WAITED for as long as some other team that holds pages is still working (its pair will end and give
    // them back): page_ctl[3] counts the teams that hold pages, page_ctl[2] those that wait.  When every other holder waits
    // too, nobody can free anything: this team gives up (its pair is re-run by the next ladder level) and its pages let the
    // others go on.  Failing at once instead cost a 32-pair batch a second launch for the pairs that had run dry.
    auto page_alloc = [&]() -> uint32_t {  // 0xFFFFFFFF: none to be had
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
        }  // (a lock that cannot be had: the pages stay out of the pool for the rest of the launch -- slower, never wrong)
        atomicSub(&P.page_ctl[3], 1u);
        n_pg = 0u;
    };
    for (;;) {
        // ---- the team's next pair: workgroup 0 pulls it, the barrier publishes it
        if (b == 0 && tid == 0) {
            const uint32_t w0 = atomicAdd(P.queue_head, 1u);
            __hip_atomic_store(&ctl[2], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[12], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // [12..13]: stored cells (64-bit: a
            __hip_atomic_store(&ctl[13], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // 100 kbp semi-global pair stores > 2^32 / 1.3)
            __hip_atomic_store(&ctl[4], (uint32_t)TEAM_CMD_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[8], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[9], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            reset_set(tred(0)), reset_set(tred(1)), reset_set(tred(2));
        }
        team_barrier(true);
        if (aborted) return;
        const uint32_t wi = __hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

        SeqView<MODE> sv;
        sv.n = n, sv.m = m;
        if constexpr (MODE == 0) {
            const uint32_t need = ((imax2(n, m) + 15) >> 4) + 1;
            if (need > P.lds_seq_words) {
                if (lead_wg && tid == 0) {
                    rec[REC_STATUS] = ST_REDO_LDS;
                    push_redo(P, pair, ST_REDO_LDS);
                }
                continue;
            }
            if (tid == 0) red[9] = 0;
            __syncthreads();
            bool bad = stage_pack<G>(P.blob, P.q_off[pair], nq, lq, tid);
            bad |= stage_pack<G>(P.blob, P.t_off[pair], mt, lt, tid);
            if (__ballot(bad) != 0ull && lane == 0) red[9] = 1;
            __syncthreads();
            if (red[9]) {  // every workgroup staged the same bytes and takes the same decision
                if (lead_wg && tid == 0) {
                    rec[REC_STATUS] = ST_REDO_BYTES;
                    push_redo(P, pair, ST_REDO_BYTES);
                }
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
        uint64_t   top     = 0;  // next free arena word (identical in every active workgroup)
        uint64_t   page_end = 0; // paged arena: end of the page the rows are being written to (0: no page yet)
        uint32_t   n_ent   = 0;
        bool       overflow = false, done = false;
        uint32_t   s_final  = 0;
        uint64_t   my_cells = 0;
        bool       teamed   = true;  // team mode (all workgroups step together) / solo mode (workgroup 0 alone)
        auto dir_ptr = [&](uint32_t idx) { return A + cap - (uint64_t)DIR_WORDS * (idx + 1); };
        const DirEnt none = {0ull, 0, 0, 0u, {0u, 0u, 0u}};
        auto put_ent = [&](uint32_t idx, uint64_t base, int lo_, int w_, uint32_t stride) {
            if (tid == 0) {
                DirEnt d;
                d.base = base, d.lo = lo_, d.w = w_, d.stride = stride, d.pad[0] = d.pad[1] = d.pad[2] = 0u;
                ring[idx % TEAM_RING] = d;
                if (lead_wg) store_dir(dir_ptr(idx), base, lo_, w_, stride);
            }
        };
        // arena words: coherent (memory-side) in team mode, plain (L2-cached) in solo mode.  A team on one XCD (xl) stores
        // plain too: the line stays in the XCD's L2 (an sc1 store drops it), where the team's sc1 loads -- they bypass the
        // reader's L1 only -- find it; a store is in that L2 when the storing wave's vmcnt has returned, which every barrier
        // waits for (__syncthreads), so no write-back stands between the rows and the barrier's arrive
        auto ldw = [&](const uint32_t *p_) { return teamed ? ald(p_) : *p_; };
        auto stw = [&](uint32_t *p_, uint32_t v_) {
            if (teamed && !xl)
                ast(p_, v_);
            else
                *p_ = v_;
        };
        // one reduction of the step: block result in LDS slot `slot`, team result through the set's global word
        auto team_min = [&](int *gword, int slot) {
            if (teamed && tid == 0) atomicMin(gword, red[slot]);
        };
        auto team_max = [&](int *gword, int slot) {
            if (teamed && tid == 0) atomicMax(gword, red[slot]);
        };
        auto team_or = [&](int *gword, int slot) {
            if (teamed && tid == 0 && red[slot]) atomicOr(gword, 1);
        };
        auto team_get = [&](int *gword, int slot) {
            if (teamed && tid == 0) red[slot] = __hip_atomic_load(gword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };

        uint32_t s = 0;
        for (;; s += g) {
            const uint32_t si = s / g;
            TeamRed *const tr = tred(si % 3u);
            // the set of the next score was last used three scores ago; every workgroup is past that score
            if (lead_wg && tid == 0) reset_set(tred((si + 1u) % 3u));
            // sources: M[s-x], M[s-o-e], I[s-e] / D[s-e]  (wfa.go:557-560; missing when diff > s)
            // (the entries are the same in every lane: into scalar registers, so that the row addresses and range tests
            // of the five source loads of a cell are scalar operands instead of 64-bit vector arithmetic)
            auto uni = [](DirEnt d) {
                auto r = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                DirEnt u;
                u.base   = (uint64_t)r((uint32_t)d.base) | ((uint64_t)r((uint32_t)(d.base >> 32)) << 32);
                u.lo     = (int)r((uint32_t)d.lo), u.w = (int)r((uint32_t)d.w), u.stride = r(d.stride);
                u.pad[0] = u.pad[1] = u.pad[2] = 0u;
                return u;
            };
            const DirEnt eX = uni((s >= x) ? ring[(si - x / g) % TEAM_RING] : none);
            const DirEnt eO = uni((s >= oe) ? ring[(si - oe / g) % TEAM_RING] : none);
            const DirEnt eE = uni((s >= e) ? ring[(si - e / g) % TEAM_RING] : none);
            const bool   seeded = (s == 0u) || (s == x);

            int lo = INT32_MAX, hi = INT32_MIN;
            if (eX.w > 0) lo = imin2(lo, eX.lo - 1), hi = imax2(hi, eX.lo + eX.w);
            if (eO.w > 0) lo = imin2(lo, eO.lo - 1), hi = imax2(hi, eO.lo + eO.w);
            if (eE.w > 0) lo = imin2(lo, eE.lo - 1), hi = imax2(hi, eE.lo + eE.w);
            lo = imax2(lo, -(n - 1));  // wfa.go:562-563
            hi = imin2(hi, m - 1);
            if (s == 0u) lo = INT32_MAX, hi = INT32_MIN;  // the reference never calls next(0)
            if (seeded) lo = imin2(lo, seed_lo), hi = imax2(hi, seed_hi);

            const int64_t W = (hi >= lo) ? ((int64_t)hi - lo + 1) : 0;
            if (!paged) {
                if (top + 3ull * (uint64_t)W + (uint64_t)DIR_WORDS * (si + 2) > cap) {
                    overflow = true;
                    break;
                }
            } else {
                if (si + 2u > dir_entries || 3ull * (uint64_t)W > page_words) {
                    overflow = true;
                    break;
                }
                if (top + 3ull * (uint64_t)W > page_end) {
                    // the row does not fit the page: the next one.  Every active workgroup gets here with the same values; in team
                    // mode workgroup 0 takes the page and a fenced barrier hands its id round
                    uint32_t pg;
                    if (teamed) {
                        if (lead_wg && tid == 0) ast(&ctl[112], page_alloc());
                        team_barrier(true);
                        if (aborted) return;
                        pg = ald(&ctl[112]);
                    } else {
                        if (tid == 0) red[8] = (int)page_alloc();
                        __syncthreads();
                        pg = (uint32_t)red[8];
                        __syncthreads();
                    }
                    if (pg == 0xFFFFFFFFu) {  // the pool is empty: the pair is re-run when fewer teams share it
                        overflow = true;
                        break;
                    }
                    top = (uint64_t)pg << P.page_words_log2, page_end = top + page_words;
                }
            }
            __syncthreads();  // everybody has read the ring entries before the slot of this score is rewritten

            // ---- mode switches (every active workgroup computes the same W)
            if (teamed && W <= (int64_t)solo_max && T > 1) {
                // team -> solo: the rows so far become visible to workgroup 0's plain loads; the others park
                team_barrier(true);
                if (aborted) return;
                teamed = false;
                if (!lead_wg) {
                    bool resumed = false;
                    for (;;) {  // parked: workgroup 0 arrives here when the row is wide again or the pair is over
                        team_barrier(true);
                        if (aborted) return;
                        const uint32_t cmd = ald(&ctl[4]);
                        if (cmd == TEAM_CMD_DONE) {  // how the pair ended: everybody takes part in the end-cell search
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
                            if (tid < TEAM_RING && (uint32_t)tid < si2) {  // the directory entries the next scores can source
                                const uint32_t idx = si2 - 1u - (uint32_t)tid;
                                ring[idx % TEAM_RING] = load_dir(dir_ptr(idx));
                            }
                            n_ent  = si2;
                            teamed = true;
                            resumed = true;
                            __syncthreads();
                            break;
                        }
                    }
                    if (!resumed) break;  // pair over
                    s -= g;               // redo the loop head for score s in team mode
                    continue;
                }
            } else if (!teamed && W > (int64_t)solo_max) {
                // solo -> team (only workgroup 0 is here): publish where we are and wake the others
                if (tid == 0) {
                    ast(&ctl[5], s), ast(&ctl[6], (uint32_t)top), ast(&ctl[7], (uint32_t)(top >> 32));
                    ast(&ctl[113], (uint32_t)page_end), ast(&ctl[114], (uint32_t)(page_end >> 32));
                    ast(&ctl[4], (uint32_t)TEAM_CMD_RESUME);
                }
                team_barrier(true);
                if (aborted) return;
                teamed = true;
            }

            // ---- wave mode (workgroup 0, solo): wave 0 steps alone while the rows stay within 64 diagonals
            if (!teamed && wave_rows != 0u && W <= 64) {
                if (tid < 64) {
                    unsigned long long *wsteps = nullptr;
#ifdef WFA_TEAM_STAMPS
                    wsteps = &tacc[9];
#endif
                    // (locals: the loop's own variables stay out of the callee's references)
                    uint32_t ws = s, wn = n_ent, wfin = s_final;
                    uint64_t wtop = top, wcells = 0;
                    const uint32_t wflags =
                        wave_mode_steps<MODE>(P, sv, A, cap, ring, wring, wave_rows, n, m, glob, ws, wtop, wn, wfin, wcells, wsteps, page_end, dir_entries);
                    my_cells += wcells;
                    if (tid == 0) {
                        unsigned int *const ur = reinterpret_cast<unsigned int *>(red);
                        ur[0] = ws, ur[1] = (uint32_t)wtop, ur[2] = (uint32_t)(wtop >> 32), ur[3] = wn, ur[4] = wflags, ur[5] = wfin;
                    }
                }
                __syncthreads();
                {
                    const unsigned int *const ur = reinterpret_cast<const unsigned int *>(red);
                    s = ur[0], top = (uint64_t)ur[1] | ((uint64_t)ur[2] << 32), n_ent = ur[3];
                    const uint32_t wf = ur[4];
                    if (wf & WAVE_DONE) done = true, s_final = ur[5];
                    // (paged arena: "overflow" = the page is full -- the loop head below turns the page -- unless it is the directory)
                    if ((wf & WAVE_OVERFLOW) && !(paged && s / g + 2u <= dir_entries)) overflow = true;
                }
                __syncthreads();
                TEAM_STAMP(8);
                if (done || overflow) break;
                // the row at s is wider than 64: redo the loop head for it in solo (or team) mode.  The loop heads of
                // the scores stepped here did not run: the reduction set of score s is still the one of s - 3g
                // (nobody else is using the sets: the other workgroups are parked)
                if (tid == 0) reset_set(tred((s / g) % 3u));
                s -= g;
                continue;
            }

            if (W == 0) {
                put_ent(si, 0ull, 0, 0, 0u);
                n_ent = si + 1;
                __syncthreads();
                continue;
            }
            const uint64_t base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)top) |
                                  ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(top >> 32)) << 32);
            uint32_t *const rowM = A + base, *const rowI = rowM + W, *const rowD = rowI + W;
            if (tid == 0) {
                red[0] = INT32_MAX, red[1] = INT32_MIN, red[2] = 0, red[3] = INT32_MAX;
                red[4] = INT32_MAX, red[5] = INT32_MIN, red[6] = 0, red[7] = INT32_MIN, red[11] = INT32_MAX, red[12] = INT32_MIN, red[13] = INT32_MAX, red[14] = INT32_MIN;
            }
            __syncthreads();
            const int64_t i0 = teamed ? (int64_t)b * G + tid : tid, istep = teamed ? stripe : G;

            auto src = [&](const DirEnt &d, int comp, int k) -> uint32_t {
                return (d.w > 0 && k >= d.lo && k < d.lo + d.w)
                           ? ldw(A + d.base + (uint64_t)comp * d.stride + (uint32_t)(k - d.lo))
                           : 0u;
            };

            // ---- P1: next + seeds + extend, store rows, partial reductions
            int mlo = INT32_MAX, mhi = INT32_MIN, term = 0, mind = INT32_MAX, maxd = INT32_MIN;
            int fvm = INT32_MAX, lvm = INT32_MIN;  // first / last diagonal whose M cell lies inside both sequences (d >= 0)
            // TEAM_U cells of a thread are in flight together (their source loads overlap: in team mode every load
            // is a memory-side round trip), and the thread's first TEAM_U cells of the row stay in registers for
            // the wf-adaptive passes below (kM = M word, kF = number of non-zero I / D words).
            uint32_t kM[TEAM_U], kF[TEAM_U];
#pragma unroll
            for (int u = 0; u < TEAM_U; u++) kM[u] = 0u, kF[u] = 0u;
            for (int64_t i = i0; i < W; i += TEAM_U * istep) {
                uint32_t sa[TEAM_U], sb[TEAM_U], sc_[TEAM_U], sd[TEAM_U], sx[TEAM_U];
#pragma unroll
                for (int u = 0; u < TEAM_U; u++) {
                    const int64_t iu = i + u * istep;
                    const int     k  = lo + (int)iu;
                    const bool    on = iu < W && s != 0u;
                    sa[u]  = on ? src(eO, 0, k - 1) : 0u;
                    sb[u]  = on ? src(eE, 1, k - 1) : 0u;
                    sc_[u] = on ? src(eO, 0, k + 1) : 0u;
                    sd[u]  = on ? src(eE, 2, k + 1) : 0u;
                    sx[u]  = on ? src(eX, 0, k) : 0u;
                }
#pragma unroll
                for (int u = 0; u < TEAM_U; u++) {
                    const int64_t iu = i + u * istep;
                    if (iu >= W) continue;
                    const int k = lo + (int)iu;
                    Cell      c = {0u, 0u, 0u};
                    if (s != 0u) c = next_cell(sa[u], sb[u], sc_[u], sd[u], sx[u], k, n, m);
                    if (seeded && c.M == 0u) c.M = seed_word<MODE>(sv, k, s, x, glob);  // Set = last write wins (R2)
                    c.M = extend_word<MODE>(sv, c.M, k);
                    stw(rowM + iu, c.M), stw(rowI + iu, c.I), stw(rowD + iu, c.D);
                    my_cells += (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                    if (i == i0) kM[u] = c.M, kF[u] = (c.I != 0u) + (c.D != 0u);
                    if (c.M != 0u) {
                        mlo = imin2(mlo, k), mhi = imax2(mhi, k);
                        if (k == Ak && (int)(c.M >> TAG_BITS) >= m) term = 1;  // wfa.go:235-239
                        const int d = reduce_dist(c.M, k, n, m);
                        if (d >= 0) mind = imin2(mind, d), maxd = imax2(maxd, d), fvm = imin2(fvm, k), lvm = imax2(lvm, k);
                    }
                }
            }
#ifdef WFA_TEAM_STAMPS
            if (teamed) TEAM_STAMP(18);  // loads, next, extend, stores issued (this wave)
#endif
            // the thread's j-th cell of this row: from registers for j < TEAM_U, else from the arena
            const int64_t i_rest = i0 + TEAM_U * istep;
            mlo = wave_min(mlo), mhi = wave_max(mhi), mind = wave_min(mind), maxd = wave_max(maxd);
            fvm = wave_min(fvm), lvm = wave_max(lvm);
            term = __ballot(term) != 0ull;
            if (lane == 0) {
                atomicMin(&red[0], mlo), atomicMax(&red[1], mhi), atomicMin(&red[3], mind), atomicMax(&red[12], maxd);
                atomicMin(&red[13], fvm), atomicMax(&red[14], lvm);
                if (term) red[2] = 1;
            }
#ifdef WFA_TEAM_STAMPS
            if (teamed) TEAM_STAMP(19);  // wave reductions
#endif
            __syncthreads();
#ifdef WFA_TEAM_STAMPS
            if (teamed) TEAM_STAMP(20);  // the other waves of the workgroup
#endif
            if (teamed) {
                team_min(&tr->mlo, 0), team_max(&tr->mhi, 1), team_or(&tr->term, 2), team_min(&tr->mind, 3), team_max(&tr->maxd, 12);
                team_min(&tr->fvm, 13), team_max(&tr->lvm, 14);
                TEAM_STAMP(0);
                team_barrier(false);  // B1: the rows of this score are visible to the whole team
                TEAM_STAMP(1);
                if (aborted) return;
                team_get(&tr->mlo, 0), team_get(&tr->mhi, 1), team_get(&tr->term, 2), team_get(&tr->mind, 3), team_get(&tr->maxd, 12);
                team_get(&tr->fvm, 13), team_get(&tr->lvm, 14);
                __syncthreads();
            }
            mlo = red[0], mhi = red[1], term = red[2], mind = red[3], maxd = red[12], fvm = red[13], lvm = red[14];
            top += 3ull * (uint64_t)W;
            n_ent = si + 1;
            if (term) {
                put_ent(si, base, lo, (int)W, (uint32_t)W);
                done    = true;
                s_final = s;
                break;
            }

            // ---- reduce (wfa.go:461-540) when M exists at s and its Lo..Hi span is wide enough
            int nlo = mlo, nhi = mhi;  // surviving band: I and D only hold cells where M does
            if (P.adaptive && mhi >= mlo && (mhi - mlo + 1) >= (int)P.min_wf_len && mind != INT32_MAX) {
                const int maxdiff = (int)P.max_dist_diff;
                // Team mode: the ends of the surviving band without a second team barrier.  Some cell fails the
                // distance test exactly when the LARGEST distance of the row does (reduced with the minimum before
                // B1); if none fails the row keeps its range.  Otherwise only the two ends of the row move
                // (wfa.go:496-524), and everything the rule looks at -- the first / last cell that passes, a present
                // cell at a sequence end below the first one, the last valid cell before it -- lies between the end of
                // the row and that first / last passing cell: every workgroup scans the first and the last
                // TEAM_THREADS / 2 cells of the row itself (visible since B1) and arrives at the same band.  When a
                // passing cell is not within that window (the band collapses by more than 512 diagonals in one step)
                // the full passes below run, with their barriers.
                bool windowed = false;
                if (teamed && maxd - mind <= maxdiff) {
                    windowed = true;  // nothing fails
#ifdef WFA_TEAM_STAMPS
                    tacc[12]++;
#endif
                } else if (teamed) {
                    // (windows: from the first cell inside both sequences upwards, from the last one downwards -- the
                    // stretches of cells that sit at a sequence end, thousands of diagonals in a semi-global row, hold no
                    // passing cell and no valid one; a present cell below fvm is by definition one of them)
                    constexpr int HALF = TEAM_THREADS / 2;
                    const bool    low = tid < HALF;
                    const int     kw  = low ? fvm + tid : lvm - (tid - HALF);
                    const int64_t iw  = (int64_t)kw - lo;
                    const bool    in  = kw >= fvm && kw <= lvm;  // (fvm <= lvm: mind exists)
                    // (rows written by the other workgroups: visible since the release of B1, see the header comment)
                    const uint32_t mw = in ? ldw(rowM + iw) : 0u;
                    const int     dw  = reduce_dist(mw, kw, n, m);
                    const bool    okw = dw >= 0 && dw - mind <= maxdiff;
                    int f_ok = okw ? kw : INT32_MAX, l_ok = okw ? kw : INT32_MIN, hmin = (dw < 0 && mw != 0u) ? kw : INT32_MAX;
                    f_ok = wave_min(f_ok), l_ok = wave_max(l_ok), hmin = wave_min(hmin);
                    if (lane == 0) atomicMin(&red[4], f_ok), atomicMax(&red[5], l_ok), atomicMin(&red[11], hmin);
                    __syncthreads();
                    const int first_ok = red[4], last_ok = red[5];
                    const int hitmin   = mlo < fvm ? mlo : red[11];
                    const bool whole = (int64_t)lvm - fvm < (int64_t)TEAM_THREADS;  // the two windows cover every valid cell
                    if (first_ok != INT32_MAX && (whole || (first_ok < fvm + HALF && last_ok > lvm - HALF))) {
                        windowed = true;
#ifdef WFA_TEAM_STAMPS
                        tacc[(first_ok - fvm < 64 && lvm - last_ok < 64) ? 13 : 14]++;
                        tacc[16] += (unsigned long long)(first_ok - fvm), tacc[17] += (unsigned long long)(lvm - last_ok);
#endif
                        if (hitmin >= first_ok) {
                            nlo = first_ok, nhi = last_ok;
                        } else {
                            // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                            int lead = (in && kw < first_ok && dw >= 0) ? kw : INT32_MIN;
                            lead = wave_max(lead);
                            if (lane == 0) atomicMax(&red[7], lead);
                            __syncthreads();
                            lead = red[7];
                            nlo  = (lead != INT32_MIN) ? lead + 1 : mlo;
                            nhi  = last_ok;
                        }
#pragma unroll
                        for (int u = 0; u < TEAM_U; u++) {
                            const int k = lo + (int)(i0 + u * istep);
                            if (i0 + u * istep < W && (k < nlo || k > nhi)) my_cells -= (kM[u] != 0u) + kF[u];
                        }
                        for (int64_t i = i_rest; i < W; i += istep) {
                            const int k = lo + (int)i;
                            if (k < nlo || k > nhi)
                                my_cells -= (ldw(rowM + i) != 0u) + (ldw(rowI + i) != 0u) + (ldw(rowD + i) != 0u);
                        }
                    } else {
#ifdef WFA_TEAM_STAMPS
                        tacc[15]++;
#endif
                        __syncthreads();
                        if (tid == 0) red[4] = INT32_MAX, red[5] = INT32_MIN, red[11] = INT32_MAX;
                        __syncthreads();
                    }
                    TEAM_STAMP(2);
                }
                int       first_ok = INT32_MAX, last_ok = INT32_MIN, anyfail = 0, hitmin = INT32_MAX;
                if (!windowed) {
                auto p2cell = [&](uint32_t mw, int k) {
                    const int d = reduce_dist(mw, k, n, m);
                    if (d >= 0) {
                        if (d - mind > maxdiff)
                            anyfail = 1;
                        else
                            first_ok = imin2(first_ok, k), last_ok = imax2(last_ok, k);
                    } else if (mw != 0u) {
                        hitmin = imin2(hitmin, k);  // a present cell at / past a sequence end
                    }
                };
#pragma unroll
                for (int u = 0; u < TEAM_U; u++)
                    if (i0 + u * istep < W) p2cell(kM[u], lo + (int)(i0 + u * istep));
                for (int64_t i = i_rest; i < W; i += istep) p2cell(ldw(rowM + i), lo + (int)i);
                first_ok = wave_min(first_ok), last_ok = wave_max(last_ok), hitmin = wave_min(hitmin);
                anyfail  = __ballot(anyfail) != 0ull;
                if (lane == 0) {
                    atomicMin(&red[4], first_ok), atomicMax(&red[5], last_ok), atomicMin(&red[11], hitmin);
                    if (anyfail) red[6] = 1;
                }
                __syncthreads();
                if (teamed) {
                    team_min(&tr->first_ok, 4), team_max(&tr->last_ok, 5), team_or(&tr->anyfail, 6), team_min(&tr->hitmin, 11);
                    TEAM_STAMP(2);
                    team_barrier(false);  // B2
                    TEAM_STAMP(1);
                    if (aborted) return;
                    team_get(&tr->first_ok, 4), team_get(&tr->last_ok, 5), team_get(&tr->anyfail, 6), team_get(&tr->hitmin, 11);
                    __syncthreads();
                }
                first_ok = red[4], last_ok = red[5], anyfail = red[6], hitmin = red[11];
                if (anyfail && hitmin >= first_ok) {
                    // wfa.go:509-511: _lo = one past the last usable entry before the first non-failing one.  With no
                    // present-but-unusable cell (a cell at a sequence end) below first_ok, the entries between that
                    // one and first_ok are holes, and dropping or keeping a hole is the same row: _lo = first_ok --
                    // one pass over the row and one team barrier less.
                    nlo = first_ok, nhi = last_ok;
                    auto fixcell = [&](uint32_t mw, uint32_t idw, int k) {
                        if (k < nlo || k > nhi) my_cells -= (mw != 0u) + idw;
                    };
#pragma unroll
                    for (int u = 0; u < TEAM_U; u++)
                        if (i0 + u * istep < W) fixcell(kM[u], kF[u], lo + (int)(i0 + u * istep));
                    for (int64_t i = i_rest; i < W; i += istep)
                        fixcell(ldw(rowM + i), (ldw(rowI + i) != 0u) + (ldw(rowD + i) != 0u), lo + (int)i);
                } else if (anyfail) {
                    // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                    int lead = INT32_MIN;
#pragma unroll
                    for (int u = 0; u < TEAM_U; u++) {
                        const int k = lo + (int)(i0 + u * istep);
                        if (i0 + u * istep < W && k < first_ok && reduce_dist(kM[u], k, n, m) >= 0) lead = imax2(lead, k);
                    }
                    for (int64_t i = i_rest; i < W; i += istep) {
                        const int k = lo + (int)i;
                        if (k < first_ok && reduce_dist(ldw(rowM + i), k, n, m) >= 0) lead = imax2(lead, k);
                    }
                    lead = wave_max(lead);
                    if (lane == 0) atomicMax(&red[7], lead);
                    __syncthreads();
                    if (teamed) {
                        team_max(&tr->lead, 7);
                        TEAM_STAMP(3);
                        team_barrier(false);  // B3
                        TEAM_STAMP(1);
                        if (aborted) return;
                        team_get(&tr->lead, 7);
                        __syncthreads();
                    }
                    lead = red[7];
                    nlo  = (lead != INT32_MIN) ? lead + 1 : mlo;
                    nhi  = last_ok;  // wfa.go:517-524
                    // wfa.go:526-535 deletes k outside [_lo,_hi] in M, I and D: here the rows are simply narrowed
#pragma unroll
                    for (int u = 0; u < TEAM_U; u++) {
                        const int k = lo + (int)(i0 + u * istep);
                        if (i0 + u * istep < W && (k < nlo || k > nhi)) my_cells -= (kM[u] != 0u) + kF[u];
                    }
                    for (int64_t i = i_rest; i < W; i += istep) {
                        const int k = lo + (int)i;
                        if (k < nlo || k > nhi)
                            my_cells -= (ldw(rowM + i) != 0u) + (ldw(rowI + i) != 0u) + (ldw(rowD + i) != 0u);
                    }
                }
                }  // !windowed
            }
            if (nhi >= nlo)
                put_ent(si, base + (uint64_t)(nlo - lo), nlo, nhi - nlo + 1, (uint32_t)W);
            else
                put_ent(si, 0ull, 0, 0, 0u);
            __syncthreads();
            TEAM_STAMP(teamed ? 4 : 5);
#ifdef WFA_TEAM_STAMPS
            tacc[teamed ? 11 : 10]++;
#endif
        }
#ifdef WFA_TEAM_STAMPS
        if (lead_wg && tid == 0)
            for (int i = 0; i < 24; i++)
                if (i < 6 || i >= 8) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 64) + i, tacc[i]), tacc[i] = 0;
        tprev = __builtin_amdgcn_s_memrealtime();
#endif

        // ---- a pair that ends in solo mode: wake the parked workgroups
        if (!teamed && lead_wg) {
            if (tid == 0) {
                ast(&ctl[5], s_final), ast(&ctl[10], (done ? 1u : 0u) | (overflow ? 2u : 0u));
                ast(&ctl[4], (uint32_t)TEAM_CMD_DONE);
            }
            team_barrier(true);
            if (aborted) return;
        }
        // ---- count stored cells across the team
        // (64 bits all the way: workgroup 0 alone stores more than 2^32 words of a long pair in solo / wave mode)
        if (tid == 0) red[10] = 0, red[11] = 0;
        __syncthreads();
        {
            unsigned long long wsum = my_cells;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
            if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red[10]), wsum);
        }
        __syncthreads();
        if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 12), *reinterpret_cast<unsigned long long *>(&red[10]));
        team_barrier(true);  // every row, the directory and the cell count are visible to workgroup 0
        if (aborted) return;

        if (overflow || !done) {
            if (lead_wg && tid == 0) {
                rec[REC_STATUS] = ST_REDO_ARENA;
                push_redo(P, pair, ST_REDO_ARENA);
                if (paged) page_free_all();  // (nobody reads the rows past the barrier above)
            }
            continue;
        }
        // ---- semi-global end cell (backtraceStartPosistion, wfa.go:270-375), whole team.  Per score the reference
        // scans down from Ak and up from Ak+1, skipping absent cells, until the first cell that either leaves the
        // matrix (break) or lies on the last row/column (hit): the NEAREST break-or-hit cell on each side.  Scanning
        // the scores downwards, every hit replaces the previous one (its score is never above the best so far) and
        // the upward scan overrides the downward one at equal score, so the answer is the hit at the LOWEST score
        // that has one.  Workgroup b takes the scores s_final/g - b, - b - T, ...; one 64-bit atomic min combines.
        uint32_t minS  = s_final;
        int      lastK = Ak;
        if (!glob) {
            unsigned int *const ured = reinterpret_cast<unsigned int *>(red);
            unsigned long long  best = ((unsigned long long)s_final << 32) | 0xFFFFFFFFull;  // low word all ones: no hit
            for (int64_t idx = (int64_t)(s_final / g) - (int64_t)b; idx >= 0; idx -= (int64_t)T) {
                const DirEnt en = load_dir(dir_ptr((uint32_t)idx));
                if (en.w <= 0) continue;  // !M.HasScore(_s)
                if (tid == 0) ured[4] = 0xFFFFFFFFu, ured[5] = 0xFFFFFFFFu;
                __syncthreads();
                const uint32_t *row = A + en.base;
                unsigned int keyD = 0xFFFFFFFFu, keyU = 0xFFFFFFFFu;
                for (int64_t i = tid; i < en.w; i += G) {
                    const uint32_t raw = row[i];
                    if (raw == 0u) continue;
                    const int  k = en.lo + (int)i, h = (int)(raw >> TAG_BITS), v = h - k;
                    const bool stop = (v <= 0 || v > n || h > m);
                    const bool hit  = !stop && ((v == n && h >= n) || (h == m && v >= m));
                    if (!(stop || hit)) continue;
                    if (k <= Ak)
                        keyD = min(keyD, ((unsigned int)(Ak - k) << 1) | (hit ? 0u : 1u));
                    else
                        keyU = min(keyU, ((unsigned int)(k - Ak - 1) << 1) | (hit ? 0u : 1u));
                }
                keyD = (unsigned int)wave_min((int)(keyD ^ 0x80000000u)) ^ 0x80000000u;  // unsigned min via signed min
                keyU = (unsigned int)wave_min((int)(keyU ^ 0x80000000u)) ^ 0x80000000u;
                if (lane == 0) atomicMin(&ured[4], keyD), atomicMin(&ured[5], keyU);
                __syncthreads();
                keyD = ured[4], keyU = ured[5];
                const unsigned long long sc = (unsigned long long)((uint32_t)idx * g) << 32;
                if (keyU != 0xFFFFFFFFu && (keyU & 1u) == 0u)
                    best = min(best, sc | (uint32_t)(Ak + 1 + (int)(keyU >> 1) + 0x40000000));
                else if (keyD != 0xFFFFFFFFu && (keyD & 1u) == 0u)
                    best = min(best, sc | (uint32_t)(Ak - (int)(keyD >> 1) + 0x40000000));
                __syncthreads();
            }
            if (tid == 0) atomicMin(reinterpret_cast<unsigned long long *>(ctl + 8), best);
            team_barrier(false);
            if (aborted) return;
            if (lead_wg) {
                const uint32_t blo = ald(&ctl[8]), bhi = ald(&ctl[9]);
                if (blo != 0xFFFFFFFFu) minS = bhi, lastK = (int)blo - 0x40000000;
            }
        }
        TEAM_STAMP(6);  // wake-up, cell count, end-cell search
        if (!lead_wg) continue;  // backtrace: workgroup 0 (the others wait at the next pair's barrier)

        // ---- backtrace: wave 0 walks together (same steps, same values in every lane; the directory entries of the
        // scores around the walk sit in LDS -- the ring is free now -- and are loaded 64 at a time); result record: one lane
        __syncthreads();
        // (paged arena: the ops scratch of the walk is the rest of the last page, or a page of its own when that is too small)
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
            if (no_scratch || !wave_backtrace_record(P, A, cap, n_ent, top, ring, reinterpret_cast<unsigned int *>(red), n, m, minS, lastK, glob, rec, scratch_end)) {
                if (tid == 0) {
                    rec[REC_STATUS] = ST_REDO_ARENA;
                    push_redo(P, pair, ST_REDO_ARENA);
                }
            } else {
                if (tid == 0) {
                    rec[REC_CELLS_LO] = __hip_atomic_load(&ctl[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rec[REC_CELLS_HI] = __hip_atomic_load(&ctl[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rec[REC_N_SCORES] = s_final;
                    if (P.debug_info) P.debug_info[0] = n_ent, P.debug_info[1] = s_final;  // (wfahip_debug_wavefronts)
                }
            }
#ifdef WFA_TEAM_STAMPS
            TEAM_STAMP(7);  // backtrace + result record
            if (tid == 0)
                for (int i = 6; i < 8; i++) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 64) + i, tacc[i]), tacc[i] = 0;
#endif
            if (paged && tid == 0) page_free_all();  // the pair's pages go back to the pool (the other workgroups wait at the next pair's barrier)
        }
    }
}

}  // namespace wfa
