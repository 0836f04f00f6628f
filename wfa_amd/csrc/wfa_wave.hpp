// wfa_wave.hpp -- WAVE mode of the long-pair kernels (wfa_team_kernel, wfa_generic_kernel): the score steps of rows
// that are at most 64 diagonals wide, done by ONE wave.
//
// A diagonal per lane (lanes ordered by diagonal), the last `wave_rows` rows of M, I and D in an LDS ring (slot =
// diagonal & 63: every source of a row of <= 64 diagonals lies inside the row's own range, so the slots of a row
// never collide), the directory entries of the last 64 scores in an LDS ring too, the M range and the ends of the
// wf-adaptive band from ballots (first / last set lane = lowest / highest diagonal), the loop control in scalar
// registers, no barrier of any kind.  Rows and directory entries are still stored to the arena: the backtrace, and
// the workgroup-wide steps that take over when a row grows past 64 diagonals, read them from there.  Same cell code
// (next_cell / seed_word / extend_word / reduce_dist) and the same rules as the workgroup-wide step:
// next(s) -> extend(s) -> termination test -> reduce(s)  (wfa.go:228-251, 461-540).
#pragma once
#include "wfa_device.hpp"

namespace wfa {

constexpr int WAVE_DIR_RING = 64;  // directory entries in LDS (sources reach back < 64 scores)
enum : uint32_t { WAVE_DONE = 1u, WAVE_OVERFLOW = 2u, WAVE_WIDE = 4u };

// Runs score steps from score s (whose row is known to be at most 64 diagonals wide) until the alignment ends
// (WAVE_DONE: s = s_final = the final score, its entry written), the arena is full (WAVE_OVERFLOW) or the row at
// s is wider than 64 diagonals (WAVE_WIDE: s is the score to redo); every score below s has its directory entry,
// n_ent counts them.  Called by the 64 lanes of one wave, converged.  `ring` must hold the directory entries of the
// scores s - g .. s - 64 g (those that exist); the rows the next steps can source are copied from the arena here.
// row_end / dir_entries: where the rows must end and how many directory entries there is room for.  dir_entries = 0: rows
// and directory share the slot (the rows grow up from its start, the directory down from `cap`: row_end is ignored); else the
// rows live in a page of their own that ends at row_end (wfa_team_kernel's paged arena) and WAVE_OVERFLOW means "this page
// is full", unless the directory is.
template <int MODE>
WFA_DEV uint32_t wave_mode_steps(const KParams &P, const SeqView<MODE> &sv, uint32_t *const A, const uint64_t cap, DirEnt *const ring,
                                 uint32_t *const wring, const uint32_t wave_rows, const int n, const int m, const bool glob, uint32_t &s,
                                 uint64_t &top, uint32_t &n_ent, uint32_t &s_final, uint64_t &my_cells, unsigned long long *n_steps,
                                 const uint64_t row_end = 0, const uint32_t dir_entries = 0) {
    const uint32_t x = P.x, oe = P.oe, e = P.e, g = P.g;
    const int      lane = (int)(threadIdx.x & 63u), Ak = m - n;
    const int      seed_lo = glob ? 0 : -(n - 1), seed_hi = glob ? 0 : m - 1;
    const uint32_t si = s / g;
    auto put_ent = [&](uint32_t idx, uint64_t base, int lo_, int w_, uint32_t stride) {
        if (lane == 0) {
            DirEnt d;
            d.base = base, d.lo = lo_, d.w = w_, d.stride = stride, d.pad[0] = d.pad[1] = d.pad[2] = 0u;
            ring[idx % WAVE_DIR_RING] = d;
            store_dir(A + cap - (uint64_t)DIR_WORDS * (idx + 1), base, lo_, w_, stride);
        }
    };
    const uint32_t rmask = wave_rows - 1u;
    auto wrow = [&](uint32_t idx, int comp) { return wring + (((idx & rmask) * 3u + (uint32_t)comp) << 6); };
    // the rows the next steps can source: from the arena into the LDS ring (a row wider than 64
    // is never read here: a step that sources it is itself wider than 64 and leaves wave mode)
    for (uint32_t r = 1; r <= wave_rows && r <= si; r++) {
        const DirEnt d = ring[(si - r) % WAVE_DIR_RING];
        if (d.w > 0 && d.w <= 64 && lane < d.w) {
            const uint32_t sl = (uint32_t)(d.lo + lane) & 63u;
#pragma unroll
            for (int c = 0; c < 3; c++) wrow(si - r, c)[sl] = A[d.base + (uint64_t)c * d.stride + (uint32_t)lane];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Everything that is the same in every lane is kept in scalar registers (readfirstlane): the
    // score, the arena top, the ranges of the source rows -- the loop control is scalar code.
    auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const int2 *const ring_lw = reinterpret_cast<const int2 *>(ring);  // entry i: [4 i + 1] = {lo, w}
    uint32_t wflags = 0;
    uint32_t su = rfl(s), sj = rfl(s / g);
    uint64_t utop = (uint64_t)rfl((uint32_t)top) | ((uint64_t)rfl((uint32_t)(top >> 32)) << 32);
    const uint32_t dx = x / g, doe = oe / g, de = e / g;
    for (;; su += g, sj++) {
        int xlo = 0, xw = 0, olo = 0, ow_ = 0, elo = 0, ew = 0;
        if (su >= x) {
            const int2 v = ring_lw[((sj - dx) % WAVE_DIR_RING) * 4u + 1u];
            xlo = (int)rfl((uint32_t)v.x), xw = (int)rfl((uint32_t)v.y);
        }
        if (su >= oe) {
            const int2 v = ring_lw[((sj - doe) % WAVE_DIR_RING) * 4u + 1u];
            olo = (int)rfl((uint32_t)v.x), ow_ = (int)rfl((uint32_t)v.y);
        }
        if (su >= e) {
            const int2 v = ring_lw[((sj - de) % WAVE_DIR_RING) * 4u + 1u];
            elo = (int)rfl((uint32_t)v.x), ew = (int)rfl((uint32_t)v.y);
        }
        const bool wseed = (su == 0u) || (su == x);
        int wlo = INT32_MAX, whi = INT32_MIN;
        if (xw > 0) wlo = imin2(wlo, xlo - 1), whi = imax2(whi, xlo + xw);
        if (ow_ > 0) wlo = imin2(wlo, olo - 1), whi = imax2(whi, olo + ow_);
        if (ew > 0) wlo = imin2(wlo, elo - 1), whi = imax2(whi, elo + ew);
        wlo = imax2(wlo, -(n - 1));
        whi = imin2(whi, m - 1);
        if (su == 0u) wlo = INT32_MAX, whi = INT32_MIN;
        if (wseed) wlo = imin2(wlo, seed_lo), whi = imax2(whi, seed_hi);
        const int64_t WW = (whi >= wlo) ? ((int64_t)whi - wlo + 1) : 0;
        if (WFA_RARE(dir_entries == 0u ? utop + 3ull * (uint64_t)WW + (uint64_t)DIR_WORDS * (sj + 2) > cap
                                       : (utop + 3ull * (uint64_t)WW > row_end || sj + 2u > dir_entries))) {
            wflags = WAVE_OVERFLOW;
            break;
        }
        if (WFA_RARE(WW > 64)) {
            wflags = WAVE_WIDE;
            break;
        }
        if (WFA_RARE(WW == 0)) {
            put_ent(sj, 0ull, 0, 0, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        const uint64_t wbase = utop;
        const int      k     = wlo + lane;
        const bool     on    = lane < (int)WW;
        const uint32_t sl    = (uint32_t)k & 63u;
        auto wsrc = [&](int dlo, int dw, uint32_t idx, int comp, int kk) -> uint32_t {
            return (kk >= dlo && kk < dlo + dw) ? wrow(idx, comp)[(uint32_t)kk & 63u] : 0u;  // (dw <= 0: never)
        };
        Cell c = {0u, 0u, 0u};
        if (on) {
            if (su != 0u) {
                const uint32_t sa = wsrc(olo, ow_, sj - doe, 0, k - 1), sb = wsrc(elo, ew, sj - de, 1, k - 1);
                const uint32_t sc2 = wsrc(olo, ow_, sj - doe, 0, k + 1), sd = wsrc(elo, ew, sj - de, 2, k + 1);
                const uint32_t sx = wsrc(xlo, xw, sj - dx, 0, k);
                c = next_cell(sa, sb, sc2, sd, sx, k, n, m);
            }
            if (wseed && c.M == 0u) c.M = seed_word<MODE>(sv, k, su, x, glob);
            c.M = extend_word<MODE>(sv, c.M, k);
            uint32_t *const rowM = A + wbase + lane;
            rowM[0] = c.M, rowM[WW] = c.I, rowM[2 * WW] = c.D;
            wrow(sj, 0)[sl] = c.M, wrow(sj, 1)[sl] = c.I, wrow(sj, 2)[sl] = c.D;
            my_cells += (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
        }
        utop += 3ull * (uint64_t)WW;
        // lanes are ordered by diagonal: first / last lane of a ballot = lowest / highest diagonal
        const unsigned long long bM = __ballot(c.M != 0u);
        if (WFA_RARE(bM == 0ull)) {  // no M cell: nothing to reduce, the entry is empty (mlo > mhi)
            put_ent(sj, 0ull, 0, 0, 0u);
        } else {
            const int  wmlo = wlo + (int)__builtin_ctzll(bM), wmhi = wlo + 63 - (int)__builtin_clzll(bM);
            const int  dd   = reduce_dist(c.M, k, n, m);
            const bool hit  = c.M != 0u && k == Ak && (int)(c.M >> TAG_BITS) >= m;
            if (WFA_RARE(__ballot(hit) != 0ull)) {
                put_ent(sj, wbase, wlo, (int)WW, (uint32_t)WW);
                wflags = WAVE_DONE;
                break;
            }
            int wnlo = wmlo, wnhi = wmhi;
            const unsigned long long bV = __ballot(dd >= 0);
            if (P.adaptive && (wmhi - wmlo + 1) >= (int)P.min_wf_len && bV != 0ull) {
                const int wmind = wave_min(dd >= 0 ? dd : INT32_MAX);
                const int maxdiff = (int)P.max_dist_diff;
                const unsigned long long bFail = __ballot(dd >= 0 && dd - wmind > maxdiff);
                const unsigned long long bOk   = bV & ~bFail;
                if (bFail != 0ull) {
                    // (bOk is never empty: the cell at the minimum distance passes)
                    const int first_ok = wlo + (int)__builtin_ctzll(bOk), last_ok = wlo + 63 - (int)__builtin_clzll(bOk);
                    const unsigned long long bEnd = bM & ~bV;  // present cells at / past a sequence end
                    const int hitmin = bEnd != 0ull ? wlo + (int)__builtin_ctzll(bEnd) : INT32_MAX;
                    if (hitmin >= first_ok) {
                        wnlo = first_ok, wnhi = last_ok;
                    } else {
                        // _lo: one past the last valid entry before the first non-failing one
                        const unsigned long long below = bV & ((1ull << (first_ok - wlo)) - 1ull);
                        wnlo = below != 0ull ? wlo + 63 - (int)__builtin_clzll(below) + 1 : wmlo;
                        wnhi = last_ok;
                    }
                    if (on && (k < wnlo || k > wnhi)) my_cells -= (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                }
            }
            if (wnhi >= wnlo)
                put_ent(sj, wbase + (uint64_t)(wnlo - wlo), wnlo, wnhi - wnlo + 1, (uint32_t)WW);
            else
                put_ent(sj, 0ull, 0, 0, 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n_steps) ++*n_steps;
    }
    // su: the score the loop stopped at (done: the final score; overflow / wide: the score to redo);
    // every score below it has its directory entry
    s = su, top = utop, n_ent = (wflags == WAVE_DONE) ? sj + 1u : sj;
    if (wflags == WAVE_DONE) s_final = su;
    return wflags;
}

// Backtrace of a finished pair by one wave walking together (same steps, same values in every lane; the directory
// entries around the walk sit in the LDS window `win`, 64 entries, loaded 64 at a time), then process()
// (wfa_cigar.go:136-214) by the 64 lanes: the forward list is the scratch list reversed, copied to P.ops, and the
// statistics of the span first-M .. last-M.  Lane 0 writes the record except REC_CELLS_* and REC_N_SCORES.  Returns
// false when the ops scratch between the rows and the directory was too small (the caller re-queues the pair).
// `acc`: four LDS words.  Called by the 64 lanes of one wave, converged.
// scratch_end = 0: the ops scratch is what lies between the rows (top) and the directory; else [top, scratch_end).
WFA_DEV bool wave_backtrace_record(const KParams &P, uint32_t *const A, const uint64_t cap, const uint32_t n_ent, const uint64_t top,
                                   DirEnt *const win, unsigned int *const acc, const int n, const int m, const uint32_t minS,
                                   const int lastK, const bool glob, uint32_t *const rec, const uint64_t scratch_end = 0) {
    const uint32_t x = P.x, oe = P.oe, e = P.e, g = P.g;
    const int      lane = (int)(threadIdx.x & 63u);
    ArenaViewWave  av;
    av.init(A, cap, g, n_ent, win, (uint32_t)imax2((int)x, imax2((int)oe, (int)e)) / g);
    uint64_t  scratch0 = (top + 1ull) & ~1ull;
    uint64_t  dir_lo   = scratch_end != 0ull ? scratch_end : cap - (uint64_t)DIR_WORDS * (uint64_t)n_ent;
    uint64_t  room     = dir_lo > scratch0 ? (dir_lo - scratch0) / 2ull : 0ull;
    OpsWriter ow;
    ow.init(reinterpret_cast<uint64_t *>(A + scratch0), (uint32_t)(room > 0xFFFFFFFFull ? 0xFFFFFFFFull : room));
    TraceOut to;
    back_trace(av, n, m, minS, lastK, !glob, x, P.o, e, ow, to);
    if (ow.overflow || av.missed) return false;  // (missed: see ArenaViewWave::get_raw -- the pair fails instead of a wrong CIGAR)
    const uint32_t L = ow.n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the list was written by this wave)
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
    if (lane == 0) acc[0] = acc[1] = acc[2] = acc[3] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t alen = 0, matches = 0, gaps = 0, regions = 0;
    for (uint32_t i = begin + (uint32_t)lane; i <= end && i < L; i += 64u) {
        const uint64_t op  = ow.buf[L - 1 - i];
        const uint32_t cnt = (uint32_t)op, o = (uint32_t)(op >> 32);
        alen += cnt;
        if (o == 'M')
            matches += cnt;
        else if (o == 'I' || o == 'D')
            gaps += cnt, regions++;
    }
    atomicAdd(&acc[0], alen), atomicAdd(&acc[1], matches), atomicAdd(&acc[2], gaps), atomicAdd(&acc[3], regions);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        rec[REC_STATUS]      = ST_OK;
        rec[REC_SCORE]       = to.score;
        rec[REC_TBEGIN]      = (uint32_t)to.tbegin;
        rec[REC_TEND]        = (uint32_t)to.tend;
        rec[REC_QBEGIN]      = (uint32_t)to.qbegin;
        rec[REC_QEND]        = (uint32_t)to.qend;
        rec[REC_ALIGN_LEN]   = acc[0];
        rec[REC_MATCHES]     = acc[1];
        rec[REC_GAPS]        = acc[2];
        rec[REC_GAP_REGIONS] = acc[3];
        rec[REC_OPS_LEN]     = L;
        rec[REC_OPS_OFF_LO]  = (uint32_t)off;
        rec[REC_OPS_OFF_HI]  = (uint32_t)(off >> 32);
    }
    return true;
}

}  // namespace wfa
