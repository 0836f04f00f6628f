// wfa_generic.hpp -- kernel A: the general wavefront-alignment kernel.
//
// One workgroup of 64*WAVES threads aligns one pair at a time (persistent: pairs are pulled from a
// device queue).  Any penalties with mismatch > 0 and gap_open + gap_ext > 0, any sequence length that fits the
// slot's arena, global or semi-global, wf-adaptive on or off.  The M/I/D rows of finished scores
// live in the slot's HBM arena (they are what the backtrace needs) and the sources of WF_NEXT are
// read back from there (L2-resident: the slot wrote them a few scores ago).
//
// While the rows are at most 64 diagonals wide -- a long read under wf-adaptive, for all of its alignment -- the
// steps are done by wave 0 alone out of LDS rings (wfa_wave.hpp: no workgroup barrier, no read-back from the arena;
// 1.7 us per step against ~5); the workgroup-wide step below takes over when a row is wider.
//
// Per score s (a multiple of g = gcd(x, o+e, e); other scores cannot exist) the kernel runs the
// reference's sequence  next(s) -> extend(s) -> termination test -> reduce(s)  (wfa.go:228-251)
// fused per diagonal, so each cell is produced in registers and stored once.
#pragma once
#include "wfa_device.hpp"
#include "wfa_wave.hpp"

namespace wfa {

// LDS words used besides the two packed sequences
constexpr int GEN_LDS_EXTRA_WORDS = 16;

template <int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) void wfa_generic_kernel(const KParams P) {
    constexpr int G = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *const lq  = lds;
    uint32_t *const lt  = lds + P.lds_seq_words;
    int *const      red = reinterpret_cast<int *>(lds + 2 * (MODE == 0 ? P.lds_seq_words : 0));
    // red[0]=mlo red[1]=mhi red[2]=term red[3]=minDist red[4]=first_ok red[5]=last_ok red[6]=anyfail
    // red[7]=lead  red[8]=pair broadcast  red[9]=bad  red[10..11] = cells count (lo, hi)
    // P.wave_bt != 0: an LDS window of 64 directory entries (wave mode's ring, the backtrace's window);
    // P.wave_rows != 0: behind it the ring of the last rows of wave mode
    DirEnt *const   gring = reinterpret_cast<DirEnt *>(lds + ((2 * (MODE == 0 ? P.lds_seq_words : 0) + GEN_LDS_EXTRA_WORDS + 3u) & ~3u));
    uint32_t *const wring = reinterpret_cast<uint32_t *>(gring + WAVE_DIR_RING);

    const int      tid  = threadIdx.x;
    const int      lane = tid & 63;
    uint32_t *const A   = P.arena + (uint64_t)blockIdx.x * P.arena_words;
    const uint64_t cap  = P.arena_words;
    const uint32_t x = P.x, oe = P.oe, e = P.e, g = P.g;

    for (;;) {
        // ---- dequeue one pair (thread 0) and broadcast
        __syncthreads();
        if (tid == 0) red[8] = (int)atomicAdd(P.queue_head, 1u);
        __syncthreads();
        const uint32_t wi = (uint32_t)red[8];
        if (wi >= P.n_work) break;
        const uint32_t pair = P.work ? P.work[wi] : wi;
        uint32_t *const rec = P.rec + (uint64_t)pair * REC_WORDS;

        const uint32_t nq = P.q_len[pair], mt = P.t_len[pair];
        if (nq == 0 || mt == 0 || nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu) {  // wfa.go:204-209
            if (tid < REC_WORDS) rec[tid] = (tid == REC_STATUS) ? ((nq == 0 || mt == 0) ? ST_EMPTY : ST_TOO_LONG) : 0u;
            continue;
        }
        const int n = (int)nq, m = (int)mt, Ak = m - n;

        SeqView<MODE> sv;
        sv.n = n, sv.m = m;
        if constexpr (MODE == 0) {
            // ---- stage + 2-bit pack both sequences into LDS
            const uint32_t need = ((imax2(n, m) + 15) >> 4) + 1;
            if (need > P.lds_seq_words) {
                if (tid == 0) {
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
            if (red[9]) {  // non-ACGT byte: the byte-compare configuration must take this pair
                if (tid == 0) {
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
        const bool     glob    = P.global_alignment != 0;
        const int      seed_lo = glob ? 0 : -(n - 1), seed_hi = glob ? 0 : m - 1;
        uint64_t       top     = 0;  // next free arena word
        uint32_t       n_ent   = 0;  // directory entries written
        bool           overflow = false, done = false;
        uint32_t       s_final = 0;
        uint64_t       my_cells = 0;
        auto dir_ptr  = [&](uint32_t idx) { return A + cap - (uint64_t)DIR_WORDS * (idx + 1); };
        auto load_ent = [&](uint32_t idx) { return load_dir(dir_ptr(idx)); };
        const DirEnt none = {0ull, 0, 0, 0u, {0u, 0u, 0u}};

        for (uint32_t s = 0;; s += g) {
            const uint32_t si = s / g;
            // sources: M[s-x], M[s-o-e], I[s-e] / D[s-e]  (wfa.go:557-560; missing when diff > s)
            const DirEnt eX = (s >= x) ? load_ent(si - x / g) : none;
            const DirEnt eO = (s >= oe) ? load_ent(si - oe / g) : none;
            // GapExt == 0 (e0): I[s-e] and D[s-e] are the rows of THIS score.  The reference reads them while it writes
            // them, k ascending (wfa.go:572-580,614-615): I[s][k-1] is the cell of the previous iteration -- an
            // insertion chains along the row at no cost -- and D[s][k+1] does not exist yet, so a deletion never extends.
            const bool   e0 = e == 0u;
            const DirEnt eE = (!e0 && s >= e) ? load_ent(si - e / g) : none;
            const bool   seeded = (s == 0u) || (s == x);

            int lo = INT32_MAX, hi = INT32_MIN;
            if (eX.w > 0) lo = imin2(lo, eX.lo - 1), hi = imax2(hi, eX.lo + eX.w);
            if (eO.w > 0) lo = imin2(lo, eO.lo - 1), hi = imax2(hi, eO.lo + eO.w);
            if (eE.w > 0) lo = imin2(lo, eE.lo - 1), hi = imax2(hi, eE.lo + eE.w);
            if (e0) {
                // KRange of a missing wavefront is (0, 0) (wfa_component.go:91-101), and here I[s] and D[s] always are
                // when next(s) starts: diagonals -1 .. 1 belong to the range, and the chain of insertions may run
                // through them even when every source lies to one side
                const int xl = eX.w > 0 ? eX.lo : 0, xh = eX.w > 0 ? eX.lo + eX.w - 1 : 0;
                const int ol = eO.w > 0 ? eO.lo : 0, oh = eO.w > 0 ? eO.lo + eO.w - 1 : 0;
                lo = imin2(imin2(xl, ol), 0) - 1, hi = imax2(imax2(xh, oh), 0) + 1;
            }
            lo = imax2(lo, -(n - 1));  // wfa.go:562-563
            hi = imin2(hi, m - 1);
            if (s == 0u) lo = INT32_MAX, hi = INT32_MIN;  // the reference never calls next(0)
            // (GapExt == 0: the chain of insertions runs exactly over next()'s own range, not over the seeds' diagonals)
            const int nx_lo = lo, nx_hi = hi;
            if (seeded) lo = imin2(lo, seed_lo), hi = imax2(hi, seed_hi);

            // room for 3 rows + this directory entry (+ ops scratch is checked later)
            const int64_t W = (hi >= lo) ? ((int64_t)hi - lo + 1) : 0;
            if (top + 3ull * (uint64_t)W + (uint64_t)DIR_WORDS * (si + 2) > cap) {
                overflow = true;
                break;
            }
            // ---- wave mode: wave 0 steps alone while the rows stay within 64 diagonals (wfa_wave.hpp)
            if (!e0 && P.wave_rows != 0u && W <= 64) {
                if (tid < 64) {
                    if ((uint32_t)tid < si) gring[(si - 1u - (uint32_t)tid) % WAVE_DIR_RING] = load_ent(si - 1u - (uint32_t)tid);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    uint32_t ws = s, wn = n_ent, wfin = s_final;
                    uint64_t wtop = top, wcells = 0;
                    const uint32_t wflags =
                        wave_mode_steps<MODE>(P, sv, A, cap, gring, wring, P.wave_rows, n, m, glob, ws, wtop, wn, wfin, wcells, nullptr);
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
                    if (ur[4] & WAVE_DONE) done = true, s_final = ur[5];
                    if (ur[4] & WAVE_OVERFLOW) overflow = true;
                }
                __syncthreads();
                if (done || overflow) break;
                s -= g;  // the row at s is wider than 64: the workgroup-wide step
                continue;
            }
            if (W == 0) {
                if (tid == 0) store_dir(dir_ptr(si), 0ull, 0, 0, 0u);
                n_ent = si + 1;
                __syncthreads();  // the entry may be a source of the very next score
                continue;
            }
            const uint64_t base = top;
            uint32_t *const rowM = A + base, *const rowI = rowM + W, *const rowD = rowI + W;

            if (tid == 0) {
                red[0] = INT32_MAX, red[1] = INT32_MIN, red[2] = 0, red[3] = INT32_MAX;
                red[4] = INT32_MAX, red[5] = INT32_MIN, red[6] = 0, red[7] = INT32_MIN;
                store_dir(dir_ptr(si), base, lo, (int)W, (uint32_t)W);
            }
            __syncthreads();

            auto src = [&](const DirEnt &d, int comp, int k) -> uint32_t {
                return (d.w > 0 && k >= d.lo && k < d.lo + d.w)
                           ? A[d.base + (uint64_t)comp * d.stride + (uint32_t)(k - d.lo)]
                           : 0u;
            };

            // ---- GapExt == 0: the I row is a serial scan along k (one lane; this is not a fast path)
            if (e0 && s != 0u) {
                if (tid == 0) {
                    uint32_t prev = 0u;  // I[s][k-1]
                    for (int64_t i = 0; i < W; i++) {
                        const int k = lo + (int)i;
                        Cell      c = {0u, 0u, 0u};
                        if (k >= nx_lo && k <= nx_hi) c = next_cell(src(eO, 0, k - 1), prev, 0u, 0u, 0u, k, n, m);
                        rowI[i] = prev = c.I;
                    }
                }
                __syncthreads();
            }

            // ---- P1: next + seeds + extend, store rows, partial reductions
            int mlo = INT32_MAX, mhi = INT32_MIN, term = 0, mind = INT32_MAX;
            for (int64_t i = tid; i < W; i += G) {
                const int k = lo + (int)i;
                Cell      c = {0u, 0u, 0u};
                if (s != 0u && (!e0 || (k >= nx_lo && k <= nx_hi)))
                    c = next_cell(src(eO, 0, k - 1), e0 ? (i > 0 ? rowI[i - 1] : 0u) : src(eE, 1, k - 1), src(eO, 0, k + 1),
                                  src(eE, 2, k + 1), src(eX, 0, k), k, n, m);
                if (seeded && c.M == 0u) c.M = seed_word<MODE>(sv, k, s, x, glob);  // Set = last write wins (R2)
                c.M = extend_word<MODE>(sv, c.M, k);
                rowM[i] = c.M, rowI[i] = c.I, rowD[i] = c.D;
                my_cells += (c.M != 0u) + (c.I != 0u) + (c.D != 0u);
                if (c.M != 0u) {
                    mlo = imin2(mlo, k), mhi = imax2(mhi, k);
                    if (k == Ak && (int)(c.M >> TAG_BITS) >= m) term = 1;  // wfa.go:235-239
                    const int d = reduce_dist(c.M, k, n, m);
                    if (d >= 0) mind = imin2(mind, d);
                }
            }
            mlo = wave_min(mlo), mhi = wave_max(mhi), mind = wave_min(mind);
            term = __ballot(term) != 0ull;
            if (lane == 0) {
                atomicMin(&red[0], mlo), atomicMax(&red[1], mhi), atomicMin(&red[3], mind);
                if (term) red[2] = 1;
            }
            __syncthreads();
            mlo = red[0], mhi = red[1], term = red[2], mind = red[3];
            top += 3ull * (uint64_t)W;
            n_ent = si + 1;
            if (term) {
                done    = true;
                s_final = s;
                break;
            }

            // ---- reduce (wfa.go:461-540) when M exists at s and its Lo..Hi span is wide enough
            int nlo = mlo, nhi = mhi;  // surviving band: I and D only hold cells where M does
            if (P.adaptive && mhi >= mlo && (mhi - mlo + 1) >= (int)P.min_wf_len && mind != INT32_MAX) {
                const int maxdiff = (int)P.max_dist_diff;
                int       first_ok = INT32_MAX, last_ok = INT32_MIN, anyfail = 0;
                for (int64_t i = tid; i < W; i += G) {
                    const int k = lo + (int)i;
                    const int d = reduce_dist(rowM[i], k, n, m);
                    if (d >= 0) {
                        if (d - mind > maxdiff)
                            anyfail = 1;
                        else
                            first_ok = imin2(first_ok, k), last_ok = imax2(last_ok, k);
                    }
                }
                first_ok = wave_min(first_ok), last_ok = wave_max(last_ok);
                anyfail  = __ballot(anyfail) != 0ull;
                if (lane == 0) {
                    atomicMin(&red[4], first_ok), atomicMax(&red[5], last_ok);
                    if (anyfail) red[6] = 1;
                }
                __syncthreads();
                first_ok = red[4], last_ok = red[5], anyfail = red[6];
                if (anyfail) {
                    // _lo: one past the last valid entry before the first non-failing one (wfa.go:503-516)
                    int lead = INT32_MIN;
                    for (int64_t i = tid; i < W; i += G) {
                        const int k = lo + (int)i;
                        if (k < first_ok && reduce_dist(rowM[i], k, n, m) >= 0) lead = imax2(lead, k);
                    }
                    lead = wave_max(lead);
                    if (lane == 0) atomicMax(&red[7], lead);
                    __syncthreads();
                    lead = red[7];
                    nlo  = (lead != INT32_MIN) ? lead + 1 : mlo;
                    nhi  = last_ok;  // wfa.go:517-524
                    // wfa.go:526-535 deletes k outside [_lo,_hi] in M, I and D: here the rows are simply narrowed
                    for (int64_t i = tid; i < W; i += G) {
                        const int k = lo + (int)i;
                        if (k < nlo || k > nhi) my_cells -= (rowM[i] != 0u) + (rowI[i] != 0u) + (rowD[i] != 0u);
                    }
                }
            }
            if (nlo != lo || nhi != hi) {  // narrow the directory entry to the live band
                if (tid == 0) {
                    if (nhi >= nlo)
                        store_dir(dir_ptr(si), base + (uint64_t)(nlo - lo), nlo, nhi - nlo + 1, (uint32_t)W);
                    else
                        store_dir(dir_ptr(si), 0ull, 0, 0, 0u);
                }
            }
            __syncthreads();
        }

        // ---- count stored cells across the group
        if (tid == 0) red[10] = 0, red[11] = 0;
        __syncthreads();
        atomicAdd(reinterpret_cast<unsigned int *>(&red[10]), (unsigned int)(my_cells & 0xFFFFFFFFull));
        __syncthreads();

        if (overflow || !done) {
            if (tid == 0) {
                rec[REC_STATUS] = ST_REDO_ARENA;
                push_redo(P, pair, ST_REDO_ARENA);
            }
            continue;
        }

        // ---- semi-global end cell (backtraceStartPosistion, wfa.go:270-375), all threads: per score the
        // reference scans down from Ak and up from Ak+1, skipping absent cells, until the first cell that either
        // leaves the matrix (break) or lies on the last row/column (hit).  That is the NEAREST break-or-hit cell
        // on each side, found here with one min-reduction per side; a hit at a score <= the best so far wins,
        // the upward scan overriding the downward one at equal score.
        __syncthreads();
        uint32_t minS  = s_final;
        int      lastK = Ak;
        if (!glob) {
            ArenaView av0;
            av0.A = A, av0.cap = cap, av0.g = g, av0.n_ent = n_ent;
            unsigned int *const ured = reinterpret_cast<unsigned int *>(red);
            for (uint32_t idx = s_final / g + 1; idx-- > 0;) {
                const DirEnt e = av0.ent(idx);
                if (e.w <= 0) continue;  // !M.HasScore(_s)
                if (tid == 0) ured[4] = 0xFFFFFFFFu, ured[5] = 0xFFFFFFFFu;
                __syncthreads();
                const uint32_t *row = A + e.base;
                unsigned int keyD = 0xFFFFFFFFu, keyU = 0xFFFFFFFFu;
                for (int64_t i = tid; i < e.w; i += G) {
                    const uint32_t raw = row[i];
                    if (raw == 0u) continue;
                    const int  k = e.lo + (int)i, h = (int)(raw >> TAG_BITS), v = h - k;
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
                const uint32_t _s = idx * g;
                if (keyD != 0xFFFFFFFFu && (keyD & 1u) == 0u && _s <= minS) lastK = Ak - (int)(keyD >> 1), minS = _s;
                if (keyU != 0xFFFFFFFFu && (keyU & 1u) == 0u && _s <= minS) lastK = Ak + 1 + (int)(keyU >> 1), minS = _s;
                __syncthreads();
            }
        }

        // ---- backtrace + result record: wave 0 walks together over an LDS window of the directory (wfa_wave.hpp) ...
        if (P.wave_bt != 0u) {
            const uint32_t cells_lo = (uint32_t)red[10];  // (red[0..3] are the accumulators of the record's statistics)
            __syncthreads();
            if (tid < 64) {
                if (!wave_backtrace_record(P, A, cap, n_ent, top, gring, reinterpret_cast<unsigned int *>(red), n, m, minS, lastK, glob, rec)) {
                    if (tid == 0) {
                        rec[REC_STATUS] = ST_REDO_ARENA;
                        push_redo(P, pair, ST_REDO_ARENA);
                    }
                } else if (tid == 0) {
                    rec[REC_CELLS_LO] = cells_lo, rec[REC_CELLS_HI] = 0u, rec[REC_N_SCORES] = s_final;
                }
                if (tid == 0 && P.debug_info) P.debug_info[0] = n_ent, P.debug_info[1] = s_final;
            }
            continue;
        }
        // ... or, without that window (it did not fit beside the sequences), lane 0 of wave 0
        if (tid == 0) {
            ArenaView av;
            av.A = A, av.cap = cap, av.g = g, av.n_ent = n_ent;

            // ops scratch: free arena words between the rows and the directory
            uint64_t  scratch0 = (top + 1ull) & ~1ull;
            uint64_t  dir_lo   = cap - (uint64_t)DIR_WORDS * (uint64_t)n_ent;
            uint64_t  room     = dir_lo > scratch0 ? (dir_lo - scratch0) / 2ull : 0ull;
            OpsWriter ow;
            ow.init(reinterpret_cast<uint64_t *>(A + scratch0), (uint32_t)(room > 0xFFFFFFFFull ? 0xFFFFFFFFull : room));
            TraceOut to;
            back_trace(av, n, m, minS, lastK, !glob, x, P.o, e, ow, to);

            if (ow.overflow) {
                rec[REC_STATUS] = ST_REDO_ARENA;
                push_redo(P, pair, ST_REDO_ARENA);
            } else {
                // process() (wfa_cigar.go:136-214): the forward list is the scratch list reversed
                const uint32_t L   = ow.n;
                const uint64_t off = atomicAdd(P.ops_cursor, (unsigned long long)L);
                uint32_t begin = 0, end = 0;
                bool     seenM = false;
                for (uint32_t i = 0; i < L; i++) {
                    const uint64_t op = ow.buf[L - 1 - i];
                    if ((uint32_t)(op >> 32) == 'M') {
                        if (!seenM) begin = i, seenM = true;
                        end = i;
                    }
                    if (off + i < P.ops_cap) P.ops[off + i] = op;
                }
                uint32_t alen = 0, matches = 0, gaps = 0, regions = 0;
                for (uint32_t i = begin; i <= end && i < L; i++) {
                    const uint64_t op  = ow.buf[L - 1 - i];
                    const uint32_t cnt = (uint32_t)op, o = (uint32_t)(op >> 32);
                    alen += cnt;
                    if (o == 'M')
                        matches += cnt;
                    else if (o == 'I' || o == 'D')
                        gaps += cnt, regions++;
                }
                rec[REC_STATUS]      = ST_OK;
                rec[REC_SCORE]       = to.score;
                rec[REC_TBEGIN]      = (uint32_t)to.tbegin;
                rec[REC_TEND]        = (uint32_t)to.tend;
                rec[REC_QBEGIN]      = (uint32_t)to.qbegin;
                rec[REC_QEND]        = (uint32_t)to.qend;
                rec[REC_ALIGN_LEN]   = alen;
                rec[REC_MATCHES]     = matches;
                rec[REC_GAPS]        = gaps;
                rec[REC_GAP_REGIONS] = regions;
                rec[REC_OPS_LEN]     = L;
                rec[REC_OPS_OFF_LO]  = (uint32_t)off;
                rec[REC_OPS_OFF_HI]  = (uint32_t)(off >> 32);
                rec[REC_CELLS_LO]    = (uint32_t)red[10];
                rec[REC_CELLS_HI]    = 0u;
                rec[REC_N_SCORES]    = s_final;
            }
            if (P.debug_info) P.debug_info[0] = n_ent, P.debug_info[1] = s_final;
        }
    }
}

}  // namespace wfa
