// wfa_packed.hpp -- kernel B: the throughput path for short/medium global alignments, plus the
// lane-per-pair backtrace kernel that follows it.
//
// Measured on 1 kbp @5 % pairs (wf-adaptive 10/50/1) the live band of a wavefront is ~18 diagonals wide
// (p99 39, never above 51), so a whole wave64 per pair leaves most lanes idle.  Here a wave carries TWO
// pairs, one per 32-lane half ("subgroup"); lane j of a subgroup owns diagonals lo+j and lo+32+j (the
// second tile only runs when some subgroup's range is wider than 32), i.e. up to 64 diagonals per pair.
// A pair whose range would exceed 64 is handed to the generic kernel (ST_REDO_BAND).
//
// Data placement per subgroup:
//   LDS   2-bit packed query/target; a ring of the last max(x,o+e)/g+1 M rows and e/g+1 I/D rows
//         indexed by (k & 63), with each row's live range [lo, lo+w) kept beside it.  The five
//         sources of WF_NEXT (wfa.go:579,580,614,615,650) are ds_read_b32 at neighbouring indices.
//   HBM   every finished row is stored once (only the band that survives wf-adaptive) into the
//         pair's arena + a 16-byte directory entry; the backtrace kernel walks them afterwards.
//
// Both halves of a wave run the same instruction stream with their own score counter; a half that
// finishes its pair pulls the next one from the device queue while the other keeps stepping.
#pragma once
#include <type_traits>
#include "wfa_device.hpp"

namespace wfa {

// Diagnostic build only (-DWFA_STAMPS): per-phase s_memtime shares, summed per wave into debug_info
// (never part of the shipped library; see scripts/stamps.sh).
#ifdef WFA_STAMPS
#define WFA_STAMP(i)                                                                                  \
    do {                                                                                              \
        unsigned long long _t;                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        stamp_acc[i] += _t - stamp_prev;                                                              \
        stamp_prev = _t;                                                                              \
    } while (0)
#else
#define WFA_STAMP(i) \
    do {             \
    } while (0)
#endif

constexpr int PK_G     = 32;  // lanes per pair
constexpr int PK_TILES = 2;   // diagonals per lane
constexpr int PK_WCAP  = PK_G * PK_TILES;

// LDS words one subgroup needs
__host__ __device__ inline uint32_t packed_sub_lds_words(uint32_t seq_words, uint32_t dm, uint32_t di) {
    return 2u * seq_words + (dm + 2u * di) * PK_WCAP + 2u * (dm + di) + 2u;
}

WFA_DEV uint32_t half_of(unsigned long long ballot, int sub) {
    return sub ? (uint32_t)(ballot >> 32) : (uint32_t)ballot;
}
// Reductions over the 32 lanes of a half: four DPP steps inside each 16-lane row (quad_perm x2,
// row_half_mirror, row_mirror), then one ds_swizzle SWAP16 to combine the two rows of the half.
WFA_DEV int sub_min(int v) {
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_ds_swizzle(v, 0x401F));
    return v;
}
WFA_DEV uint32_t sub_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
    return v;
}

#ifndef WFA_NO_AUX_KERNELS  // (wfa_duo.hip includes this header for the device functions only)
__global__ __launch_bounds__(64, 7) void wfa_packed_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x, j = lane & 31, sub = lane >> 5;
    const int lead = sub << 5;

    const uint32_t SW = P.lds_seq_words, DM = P.dm, DI = P.di;
    uint32_t *const L     = lds + sub * P.sub_lds_words;
    uint32_t *const lq    = L;
    uint32_t *const lt    = L + SW;
    uint32_t *const ringM = L + 2 * SW;
    uint32_t *const ringI = ringM + DM * PK_WCAP;
    uint32_t *const ringD = ringI + DI * PK_WCAP;
    int *const      metaM = reinterpret_cast<int *>(ringD + DI * PK_WCAP);  // [DM][2] = {lo, w}
    int *const      metaE = metaM + 2 * DM;                                 // [DI][2]

    const uint32_t x = P.x, oe = P.oe, e = P.e, g = P.g;
    const uint64_t cap = P.arena_words;

    // per-subgroup state (identical in the 32 lanes of a half)
    int       st = 0;  // 0 = needs a pair, 1 = running, 2 = queue exhausted
    uint32_t  pidx = 0, pair = 0;
    int       n = 0, m = 0, Ak = 0;
    uint32_t  s = 0, si = 0, cm = 0, ce = 0;
    uint32_t  top = 0;
    uint32_t *A = nullptr;
    uint32_t  my_cells = 0;
    SeqView<0> sv;
    sv.q = lq, sv.t = lt, sv.n = 0, sv.m = 0;
#ifdef WFA_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif

    for (;;) {
        // ---------------------------------------------------------------- refill (divergent per half)
        if (st == 0) {
            uint32_t wi = 0;
            if (j == 0) wi = atomicAdd(P.queue_head, 1u);
            wi = __shfl(wi, lead, 64);
            if (wi >= P.chunk_n) {
                st = 2;
            } else {
                pidx = wi;
                pair = P.work ? P.work[wi] : P.chunk_first + wi;
                const uint32_t nq = P.q_len[pair], mt = P.t_len[pair];
                uint32_t status = ST_PENDING;
                if (nq == 0 || mt == 0)
                    status = ST_EMPTY;  // wfa.go:204-206
                else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
                    status = ST_TOO_LONG;  // wfa.go:207-209
                else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW)
                    status = ST_REDO_LDS;
                if (status == ST_PENDING) {
                    bool bad = stage_pack<PK_G>(P.blob, P.q_off[pair], nq, lq, j);
                    bad |= stage_pack<PK_G>(P.blob, P.t_off[pair], mt, lt, j);
                    if (half_of(__ballot(bad), sub) != 0u) status = ST_REDO_BYTES;
                }
                if (status != ST_PENDING) {
                    if (j == 0) {
                        P.pair_meta[pidx] = make_uint4(status, 0u, 0u, 0u);
                        if (status >= ST_REDO_BYTES) push_redo(P, pair, status);
                    }
                    // stay in state 0: the next loop iteration pulls another pair
                } else {
                    n = (int)nq, m = (int)mt, Ak = m - n;
                    sv.n = n, sv.m = m;
                    s = 0, si = 0, cm = 0, ce = 0, top = 0, my_cells = 0;
                    A  = P.arena + (uint64_t)pidx * cap;
                    st = 1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__ballot(st != 2) == 0ull) break;
        const bool run = (st == 1);
        WFA_STAMP(0);  // refill

        // ---------------------------------------------------------------- one score step
        // sources: M[s-x], M[s-o-e], I[s-e], D[s-e]  (wfa.go:557-560; missing when diff > s)
        const bool hasX = run && s >= x, hasO = run && s >= oe, hasE = run && s >= e;
        uint32_t   slotX = cm + DM - P.dx, slotO = cm + DM - P.doe, slotE = ce + DI - P.de;
        slotX -= (slotX >= DM) ? DM : 0u, slotO -= (slotO >= DM) ? DM : 0u, slotE -= (slotE >= DI) ? DI : 0u;
        int loX = 0, wX = 0, loO = 0, wO = 0, loE = 0, wE = 0;
        if (hasX) loX = metaM[2 * slotX], wX = metaM[2 * slotX + 1];
        if (hasO) loO = metaM[2 * slotO], wO = metaM[2 * slotO + 1];
        if (hasE) loE = metaE[2 * slotE], wE = metaE[2 * slotE + 1];

        int lo = INT32_MAX, hi = INT32_MIN;
        if (wX > 0) lo = imin2(lo, loX - 1), hi = imax2(hi, loX + wX);
        if (wO > 0) lo = imin2(lo, loO - 1), hi = imax2(hi, loO + wO);
        if (wE > 0) lo = imin2(lo, loE - 1), hi = imax2(hi, loE + wE);
        lo = imax2(lo, -(n - 1));  // wfa.go:562-563
        hi = imin2(hi, m - 1);
        const bool seeded = run && (s == 0u || s == x);  // global: M[0][0] or M[x][0] (wfa.go:155-160)
        if (seeded) lo = imin2(lo, 0), hi = imax2(hi, 0);
        if (!run) lo = 0, hi = -1;
        const int W = (hi >= lo) ? hi - lo + 1 : 0;
        const bool too_wide = W > PK_WCAP;
        const bool wide     = __ballot(W > PK_G && !too_wide) != 0ull;  // wave-uniform: second tile needed

        uint32_t cM[PK_TILES], cI[PK_TILES], cD[PK_TILES], cO[PK_TILES];
        uint32_t mbits[PK_TILES];
        bool     term = false;
        WFA_STAMP(1);  // ring meta + range
#pragma unroll
        for (int t = 0; t < PK_TILES; t++) {
            cM[t] = cI[t] = cD[t] = cO[t] = 0u;
            mbits[t]                      = 0u;
            if (t == 1 && !wide) continue;
            const int  k   = lo + PK_G * t + j;
            const bool act = run && !too_wide && k <= hi;
            uint32_t   mo_km1 = 0, ie_km1 = 0, mo_kp1 = 0, de_kp1 = 0, mx_k = 0;
            if (act) {
                if ((uint32_t)(k - 1 - loO) < (uint32_t)wO) mo_km1 = ringM[slotO * PK_WCAP + ((k - 1) & (PK_WCAP - 1))];
                if ((uint32_t)(k + 1 - loO) < (uint32_t)wO) mo_kp1 = ringM[slotO * PK_WCAP + ((k + 1) & (PK_WCAP - 1))];
                if ((uint32_t)(k - 1 - loE) < (uint32_t)wE) ie_km1 = ringI[slotE * PK_WCAP + ((k - 1) & (PK_WCAP - 1))];
                if ((uint32_t)(k + 1 - loE) < (uint32_t)wE) de_kp1 = ringD[slotE * PK_WCAP + ((k + 1) & (PK_WCAP - 1))];
                if ((uint32_t)(k - loX) < (uint32_t)wX) mx_k = ringM[slotX * PK_WCAP + (k & (PK_WCAP - 1))];
                Cell c = next_cell(mo_km1, ie_km1, mo_kp1, de_kp1, mx_k, k, n, m);
                if (seeded && k == 0 && c.M == 0u) c.M = seed_word<0>(sv, 0, s, x, true);
                c.M   = extend_word<0>(sv, c.M, k);
                cM[t] = c.M, cI[t] = c.I, cD[t] = c.D;
                cO[t] = c.rej ? off0_unrejected(mo_km1, ie_km1, mo_kp1, de_kp1, mx_k, c.M & TAG_MASK) : c.off0;
                if (k == Ak && (int)(c.M >> TAG_BITS) >= m) term = true;  // wfa.go:235-239
            }
            mbits[t] = half_of(__ballot(cM[t] != 0u), sub);
        }
        WFA_STAMP(2);  // sources + next + extend
        term = half_of(__ballot(term), sub) != 0u;
        const unsigned long long mmask = (unsigned long long)mbits[0] | ((unsigned long long)mbits[1] << 32);
        int nlo = 0, nhi = -1;  // band to keep; M.Lo/Hi of the reference = tight range of M cells
        if (mmask != 0ull) nlo = lo + __builtin_ctzll(mmask), nhi = lo + 63 - __builtin_clzll(mmask);

        // ---------------------------------------------------------------- wf-adaptive (wfa.go:461-540)
        const bool want_reduce = run && !term && P.adaptive && mmask != 0ull && (nhi - nlo + 1) >= (int)P.min_wf_len;
        if (__ballot(want_reduce) != 0ull) {
            int d[PK_TILES], dm = INT32_MAX;
#pragma unroll
            for (int t = 0; t < PK_TILES; t++) {
                d[t] = reduce_dist(cM[t], lo + PK_G * t + j, n, m);
                if (d[t] >= 0) dm = imin2(dm, d[t]);
            }
            const int mind = sub_min(dm);
            unsigned long long vmask = 0ull, okmask = 0ull;
#pragma unroll
            for (int t = 0; t < PK_TILES; t++) {
                const bool valid = d[t] >= 0;
                const bool okc   = valid && (d[t] - mind <= (int)P.max_dist_diff);
                vmask |= (unsigned long long)half_of(__ballot(valid), sub) << (32 * t);
                okmask |= (unsigned long long)half_of(__ballot(okc), sub) << (32 * t);
            }
            if (want_reduce && mind != INT32_MAX && (vmask & ~okmask) != 0ull) {  // some distance failed
                const int                first_ok = __builtin_ctzll(okmask);
                const unsigned long long leadm    = vmask & ((1ull << first_ok) - 1ull);
                if (leadm != 0ull) nlo = lo + 64 - __builtin_clzll(leadm);  // one past the last leading failure
                nhi = lo + 63 - __builtin_clzll(okmask);                    // last valid non-failing entry
            }
        }

        WFA_STAMP(3);  // masks + wf-adaptive
        // ---------------------------------------------------------------- store the surviving band
        const int  wn       = (nhi >= nlo) ? nhi - nlo + 1 : 0;
        const bool no_room  = run && ((uint64_t)top + (uint32_t)wn + 4ull * (si + 2u) > cap);
        const bool give_up  = run && (too_wide || no_room);
        const bool store_ok = run && !give_up;
        if (store_ok) {
            uint32_t *const rowM = A + top;
#pragma unroll
            for (int t = 0; t < PK_TILES; t++) {
                if (t == 1 && !wide) continue;
                const int k = lo + PK_G * t + j;
                if (k >= nlo && k <= nhi) {
                    rowM[k - nlo] = compact_word(cM[t], cI[t], cD[t], cO[t]);  // one compact backtrace word per diagonal
                    const uint32_t r = (uint32_t)k & (PK_WCAP - 1);
                    ringM[cm * PK_WCAP + r] = cM[t], ringI[ce * PK_WCAP + r] = cI[t], ringD[ce * PK_WCAP + r] = cD[t];
                    my_cells += (cM[t] != 0u) + (cI[t] != 0u) + (cD[t] != 0u);
                }
            }
            if (j == 0) {
                metaM[2 * cm] = nlo, metaM[2 * cm + 1] = wn;
                metaE[2 * ce] = nlo, metaE[2 * ce + 1] = wn;
                *reinterpret_cast<uint4 *>(A + cap - 4ull * (si + 1)) =
                    wn > 0 ? make_uint4(top, (uint32_t)nlo, (uint32_t)wn, 0u) : make_uint4(0u, 0u, 0u, 0u);
            }
            top += (uint32_t)wn;
        }
        WFA_STAMP(4);  // stores
        // ---------------------------------------------------------------- finish / advance
        if (give_up || (run && term)) {
            const uint32_t cells = sub_sum(my_cells);
            int            hf    = 0;  // extended offset of the end cell M[s][Ak]: where the backtrace starts
#pragma unroll
            for (int t = 0; t < PK_TILES; t++)
                if (lo + PK_G * t + j == Ak) hf = (int)(cM[t] >> TAG_BITS);
            hf = -sub_min(-hf);
            if (j == 0) {
                if (give_up) {
                    P.pair_meta[pidx] = make_uint4(too_wide ? ST_REDO_BAND : ST_REDO_ARENA, 0u, 0u, 0u);
                    push_redo(P, pair, too_wide ? ST_REDO_BAND : ST_REDO_ARENA);
                } else {
                    P.pair_meta[pidx] = make_uint4(ST_OK, s, (uint32_t)hf, cells);
                }
            }
            st = 0;
        } else if (run) {
            s += g, si += 1u;
            cm = (cm + 1u == DM) ? 0u : cm + 1u;
            ce = (ce + 1u == DI) ? 0u : ce + 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        WFA_STAMP(5);  // finish / advance
    }
#ifdef WFA_STAMPS
    if (lane == 0 && P.debug_info) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(P.debug_info);
        for (int i = 0; i < 8; i++) atomicAdd(acc + i, stamp_acc[i]);
    }
#endif
}

#endif  // WFA_NO_AUX_KERNELS

// Backtrace (wfa.go:703-983) + process() statistics + result record of ONE finished pair of the sub-wave pipeline
// (global alignment only): idx = the pair's index in the chunk (= its arena slot), h_end = extended offset of the
// end cell M[s_final][m - n].
// entries of the ops region of a pair of final score s: 2 * score / min(x, e) + 8, rounded up to the writer's store group
// (with the buffer aligned every region ends on a store boundary)
WFA_DEV uint32_t ops_bound(const KParams &P, uint32_t s_final) {
    // (semi-global: up to three flank runs more -- I / H in front, I or H behind: wfa.go:746-750,970-976)
    return (2u * (s_final / P.min_xe) + (P.global_alignment ? 8u : 12u) + OpsWriterRev::GROUP - 1u) & ~(OpsWriterRev::GROUP - 1u);
}

// off_given: where the pair's ops region starts (the caller has carved it), or OPS_OFF_OWN: carve it here
constexpr uint64_t OPS_OFF_OWN = ~0ull;
WFA_DEV void backtrace_one(const KParams &P, uint32_t idx, uint32_t s_final, uint32_t h_end, uint32_t cells, bool coherent,
                           uint64_t off_given = OPS_OFF_OWN) {
    const uint32_t  pair = P.work ? P.work[idx] : P.chunk_first + idx;
    uint32_t *const rec  = P.rec + (uint64_t)pair * REC_WORDS;
    const int       n = (int)P.q_len[pair], m = (int)P.t_len[pair];
    CompactView cv;
    cv.A = P.arena + (uint64_t)idx * P.arena_words, cv.cap = P.arena_words, cv.g = P.g, cv.n_ent = s_final / P.g + 1u;
    cv.fmt = P.compact_fmt, cv.coherent = coherent;

    // ops region: carved from the shared ops buffer
    const uint32_t bound = ops_bound(P, s_final);
    const uint64_t off   = off_given != OPS_OFF_OWN ? off_given : atomicAdd(P.ops_cursor, (unsigned long long)bound);
    OpsWriterRev   ow;
    const bool     fits = off + bound <= P.ops_cap;
    ow.init(P.ops + off, fits ? bound : 0u);
    TraceOut to;
    if (P.compact_fmt == 11u)  // wfa_wide_kernel: the walk starts at the cell its forward pass found (semi-global: wfa.go:270-375) -- offset | (diagonal + 32768) << 16
        back_trace_compact(cv, n, m, s_final, (int)(h_end >> 16) - 32768, h_end & 0xFFFFu, P.x, P.o, P.e, ow, to, P.global_alignment == 0u);
    else
        back_trace_compact(cv, n, m, s_final, m - n, h_end, P.x, P.o, P.e, ow, to);
    ow.finish();
    // (when the ops buffer is too small the host sees ops_cursor > ops_cap and re-runs with a bigger one)
    const uint64_t first = off + bound - ow.n;
    uint4 *r4            = reinterpret_cast<uint4 *>(rec);
    r4[0] = make_uint4(ST_OK, to.score, (uint32_t)to.tbegin, (uint32_t)to.tend);
    r4[1] = make_uint4((uint32_t)to.qbegin, (uint32_t)to.qend, ow.alen, ow.matches);
    r4[2] = make_uint4(ow.gaps, ow.regions, ow.n, (uint32_t)first);
    r4[3] = make_uint4((uint32_t)(first >> 32), cells, 0u, s_final);
}

// Lane-per-pair backtrace of a chunk the forward kernel has finished.  Final statuses (empty / too long) are recorded
// here.  A pair that was handed on (ST_REDO_*) gets its record from the pass that finishes it -- which may already
// be running beside this kernel, so this kernel must not touch that record.
// Streamed mode (P.done_q != nullptr): the finished pairs are the done_q entries the streaming kernel has not taken.
// The ops regions of a workgroup's pairs are carved with ONE atomic: every wave adding its own total to the one cursor
// serialized 1 563 waves of 1e5 short pairs at the L2 -- 20 of the kernel's 41 us went to waiting for that atomic's return.
constexpr int BT_THREADS = 512;
// One finished pair walked by the 64 lanes of a wave together (idx: its index in the chunk; region: CompactViewWave::WORDS words of LDS).
// The walk is the same in all 64 lanes: its inputs pass through v_readfirstlane here, and so does every word the view
// returns, so that the compiler keeps the walk's state in scalar registers and branches on SCC instead of masking EXEC.
// LDSV: `region` IS the pair's arena (CompactViewLds; wfa_blk_kernel<.., LDSA = true>).
template <bool LDSV = false>
WFA_DEV void backtrace_wave_one(const KParams &P, uint32_t idx_, uint32_t *region) {
    const auto     rfl  = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx  = rfl(idx_);
    uint4          meta;
    if constexpr (LDSV) {  // (the forward pass left the walk's start in front of the rows; a pair it handed on: pair_meta)
        meta = *reinterpret_cast<const uint4 *>(region - 4);
        if (rfl(meta.x) != ST_OK) meta = P.pair_meta[idx];
    } else {
        meta = P.pair_meta[idx];
    }
    meta = make_uint4(rfl(meta.x), rfl(meta.y), rfl(meta.z), rfl(meta.w));
    const uint32_t pair = rfl(P.work ? P.work[idx] : P.chunk_first + idx);
    uint4 *const   r4   = reinterpret_cast<uint4 *>(P.rec + (uint64_t)pair * REC_WORDS);
    if (meta.x != ST_OK) {
        // (a pair that was handed on gets its record from the pass that finishes it; wfahip_align_pair's LDS instance reports it:
        // the host decides what runs next)
        if ((LDSV || meta.x < ST_REDO_BYTES) && lane == 0u) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            r4[0] = make_uint4(meta.x, 0u, 0u, 0u), r4[1] = z, r4[2] = z, r4[3] = z;
        }
        return;
    }
    const uint32_t s_final = meta.y, h_end = meta.z, cells = meta.w;
    const int      n = LDSV ? (int)P.one_n : (int)rfl(P.q_len[pair]), m = LDSV ? (int)P.one_m : (int)rfl(P.t_len[pair]);
    const uint32_t bound = ops_bound(P, s_final);
    unsigned long long off = 0ull;
    if constexpr (!LDSV) {  // (the LDS instance's launch holds ONE pair: its ops region starts at 0, no cursor to move)
        if (lane == 0u) off = atomicAdd(P.ops_cursor, (unsigned long long)bound);
        off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
    }
    typename std::conditional<LDSV, CompactViewLds, CompactViewWave>::type cv;
    if constexpr (LDSV) {
        cv.A = region, cv.g = P.g, cv.n_ent = s_final / P.g + 1u;
    } else {
        cv.A = P.arena + (uint64_t)idx * P.arena_words, cv.cap = P.arena_words, cv.g = P.g, cv.n_ent = s_final / P.g + 1u;
        cv.fmt = P.compact_fmt, cv.reg = region;
    }
    OpsWriterRev ow;
    const bool   fits = off + bound <= P.ops_cap;
    ow.combine = false;
    ow.init(P.ops + off, fits ? bound : 0u);
    ow.active = lane == 0u;
    TraceOut to;
    back_trace_compact(cv, n, m, s_final, m - n, h_end, P.x, P.o, P.e, ow, to);
    ow.finish();
    if (lane == 0u) {
        const uint64_t first = off + bound - ow.n;
        r4[0] = make_uint4(ST_OK, to.score, (uint32_t)to.tbegin, (uint32_t)to.tend);
        r4[1] = make_uint4((uint32_t)to.qbegin, (uint32_t)to.qend, ow.alen, ow.matches);
        r4[2] = make_uint4(ow.gaps, ow.regions, ow.n, (uint32_t)first);
        r4[3] = make_uint4((uint32_t)(first >> 32), cells, 0u, s_final);
    }
}

#ifndef WFA_NO_AUX_KERNELS
__global__ __launch_bounds__(BT_THREADS) void wfa_backtrace_kernel(const KParams P) {
    __shared__ uint32_t           wsum[BT_THREADS / 64];
    __shared__ unsigned long long wbase;
    const uint32_t idx  = blockIdx.x * blockDim.x + threadIdx.x;
    const bool     have = idx < P.chunk_n;
    uint4          meta = make_uint4(ST_PENDING, 0u, 0u, 0u);
    if (have) meta = P.pair_meta[idx];
    if (have && meta.x != ST_OK && meta.x < ST_REDO_BYTES) {
        const uint32_t pair = P.work ? P.work[idx] : P.chunk_first + idx;
        const uint4    z    = make_uint4(0u, 0u, 0u, 0u);
        uint4 *r4           = reinterpret_cast<uint4 *>(P.rec + (uint64_t)pair * REC_WORDS);
        r4[0] = make_uint4(meta.x, 0u, 0u, 0u), r4[1] = z, r4[2] = z, r4[3] = z;
    }
    // what this thread walks: slot, final score, final offset, cells
    bool     walk = have && meta.x == ST_OK;
    uint32_t slot = idx, s_final = meta.y, h_end = meta.z, cells = meta.w;
    if (P.done_q) {
        const uint4 e = have ? P.done_q[idx] : make_uint4(0u, 0u, 0u, 0u);
        walk = e.x != 0u && e.x != DONE_TAKEN && (e.x & DONE_NOT_OK) == 0u;
        slot = e.x - 1u, s_final = e.y, h_end = e.z, cells = e.w;
    }
    const uint32_t bound = walk ? ops_bound(P, s_final) : 0u;
    // exclusive scan of `bound` over the workgroup
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t       incl = bound;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) wsum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (uint32_t w = 0; w < blockDim.x / 64u; w++) tot += wsum[w];
        wbase = tot ? atomicAdd(P.ops_cursor, tot) : 0ull;
    }
    __syncthreads();
    uint64_t off = wbase + (incl - bound);
    for (uint32_t w = 0; w < wv; w++) off += wsum[w];
    if (walk) backtrace_one(P, slot, s_final, h_end, cells, false, off);
}


// The same for LONG pairs (round 4: the sliding-window instances of wfa_blk_kernel) and for the one pair of
// wfahip_align_pair: a WAVE per pair.  All 64 lanes run the walk on the same values, the arena cells come from an LDS
// region the wave loads together (CompactViewWave), lane 0 stores the CIGAR ops and the record.  500 x 50 kbp pairs:
// ~4 500 ops each.
constexpr int BTW_WAVES = 4;
__global__ __launch_bounds__(64 * BTW_WAVES) void wfa_backtrace_wave_kernel(const KParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t region[BTW_WAVES][CompactViewWave::WORDS];
    const uint32_t wv  = threadIdx.x >> 6;
    const uint32_t idx = blockIdx.x * (uint32_t)BTW_WAVES + wv;
    if (idx >= P.chunk_n) return;
    backtrace_wave_one(P, idx, region[wv]);
}
#endif  // WFA_NO_AUX_KERNELS

// Streaming backtrace (called by waves of wfa_blk_kernel<.., STREAM = true>): the done_q entries are taken in
// completion order while the forward waves of the same launch are still running.  The walk is a chain of dependent
// DRAM reads: ~100 waves keep up with the whole forward kernel, while the same work costs ~1.8 ms of the whole GPU
// when it runs afterwards.  The first n_stream_wgs workgroups of the launch do nothing else; a forward wave joins in
// when the pair queue is exhausted, so the last pairs in flight are walked by the whole GPU.
// A wave takes 64 consecutive entries -- pairs that finished at about the same time -- waits until all of them are
// written, and walks them lane per pair with agent-scope loads (the producer's stores are write-through, its entry
// is stored after them).  Every pair of the chunk pushes exactly one entry, so the tickets run out when all pairs
// are done.  A wave never waits for long: when an entry does not show up within 20 ms it leaves, and
// wfa_backtrace_kernel, launched afterwards, sweeps up every entry not taken.
WFA_DEV void stream_backtrace(const KParams &P) {
    const int      lane = threadIdx.x;
    uint32_t *const ctl = P.done_ctl;
    for (;;) {
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(ctl + 1, 64u);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= P.chunk_n) break;
        const uint32_t idx   = base + (uint32_t)lane;
        const bool     valid = idx < P.chunk_n;
        uint32_t *const ew   = reinterpret_cast<uint32_t *>(P.done_q + idx);
        bool            ok   = !valid;
        const uint64_t  t0   = wall_clock64();
        bool            late = false;
        for (;;) {
            if (!ok) ok = __hip_atomic_load(ew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (__ballot(!ok) == 0ull) break;
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > (uint64_t)P.stream_wait) {  // (20 ms unless the host says otherwise)
                late = true;
                break;
            }
        }
        if (late) break;  // (the entries stay untaken: the sweep kernel does them)
        if (valid) {
            const uint32_t ex = __hip_atomic_load(ew + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t ey = __hip_atomic_load(ew + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t ez = __hip_atomic_load(ew + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t e3 = __hip_atomic_load(ew + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((ex & DONE_NOT_OK) == 0u) backtrace_one(P, ex - 1u, ey, ez, e3, true);
            ew[0] = DONE_TAKEN;
        }
    }
}

}  // namespace wfa
