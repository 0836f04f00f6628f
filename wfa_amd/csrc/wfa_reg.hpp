// wfa_reg.hpp -- kernel C: register-window forward kernel, FOUR pairs per wave64.
//
// Each 16-lane DPP row of a wave owns one pair; lane j of a row holds the four diagonals
// kb + 16*t + j (t = 0..3) of a 64-diagonal window whose base kb follows the band in steps of 16.
// The wavefront rows WF_NEXT needs -- M[s-x], M[s-o-e], I[s-e], D[s-e] (wfa.go:557-560) -- never leave
// the register file: the last max(x,o+e)/g M rows and e/g I/D rows are kept per lane, the k-1 / k+1
// neighbours (wfa.go:579,580,614,615) are fetched with DPP row shifts (row_shr:1 / row_shl:1, with
// row_ror carrying the lane across the 16-lane tile boundary), and the ring advances by register
// moves.  Cells outside a row's surviving band are kept at 0, so "exists" is simply word != 0.
// LDS holds only the 2-bit packed sequences.  Every finished row is stored once to the pair's HBM
// arena for the backtrace kernel (wfa_packed.hpp), exactly like the packed kernel does.
//
// The ring depths are compile-time (template on x/g, (o+e)/g, e/g); the default penalties 4/6/2 and
// every set with the same ratios use <2,4,1>.  Other penalty shapes take the LDS-ring packed kernel.
// A pair whose band does not fit the 64-diagonal window is handed on (ST_REDO_BAND).
#pragma once
#include "wfa_device.hpp"
#include "wfa_packed.hpp"

namespace wfa {

constexpr int RG_G = 16;  // lanes per pair = one DPP row
constexpr int RG_T = 4;   // diagonals per lane
constexpr int RG_W = RG_G * RG_T;

// R[k-1] for the lanes of tile `cur`: lanes 1..15 take their left neighbour, lane 0 takes lane 15 of the
// tile below (row_ror:1 rotates it into lane 0; row_shr:1 leaves lane 0 untouched: bound_ctrl off).
WFA_DEV uint32_t row_prev(uint32_t cur, uint32_t prev_tile) {
    const int carry = __builtin_amdgcn_update_dpp(0, (int)prev_tile, 0x121, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(carry, (int)cur, 0x111, 0xf, 0xf, false);
}
// R[k+1]: lanes 0..14 take their right neighbour, lane 15 takes lane 0 of the tile above (row_ror:15).
WFA_DEV uint32_t row_next(uint32_t cur, uint32_t next_tile) {
    const int carry = __builtin_amdgcn_update_dpp(0, (int)next_tile, 0x12F, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(carry, (int)cur, 0x101, 0xf, 0xf, false);
}
WFA_DEV int grp_min(int v) {  // min over the 16 lanes of a DPP row
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = imin2(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    return v;
}
WFA_DEV uint32_t grp_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false);
    return v;
}
WFA_DEV uint32_t grp_bits(unsigned long long ballot, int grp) { return (uint32_t)(ballot >> (16 * grp)) & 0xFFFFu; }
WFA_DEV int ctz_pair(uint32_t lo32, uint32_t hi32) { return lo32 ? __builtin_ctz(lo32) : 32 + __builtin_ctz(hi32); }
WFA_DEV int msb_pair(uint32_t lo32, uint32_t hi32) { return hi32 ? 63 - __builtin_clz(hi32) : 31 - __builtin_clz(lo32); }

constexpr int RG_EMPTY_LO = 0x3FFFFFFF, RG_EMPTY_HI = -0x3FFFFFFF;

template <int DX, int DOE, int DE>
__global__ __launch_bounds__(64, 5) void wfa_reg_kernel(const KParams P) {
    constexpr int RM = DX > DOE ? DX : DOE;  // M rows kept: scores s-g .. s-RM*g
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x, j = lane & 15, grp = lane >> 4;

    const uint32_t  SW = P.lds_seq_words;
    uint32_t *const lq = lds + grp * 2 * SW;
    uint32_t *const lt = lq + SW;

    const uint32_t x = P.x, g = P.g;
    const uint64_t cap = P.arena_words;

    // per-pair state (identical in the 16 lanes of a row)
    int        st = 0;  // 0 = needs a pair, 1 = running, 2 = queue exhausted
    uint32_t   pidx = 0, pair = 0;
    int        n = 0, m = 0, Ak = 0, kb = 0;
    uint32_t   s = 0, si = 0, top = 0, my_cells = 0;
    uint32_t  *A = nullptr;
    SeqView<0> sv;
    sv.q = lq, sv.t = lt, sv.n = 0, sv.m = 0;

    uint32_t Mh[RM][RG_T], Ih[DE][RG_T], Dh[DE][RG_T];  // [0] = previous score, [d] = d+1 scores back
    int      rlo[RM], rhi[RM], elo[DE], ehi[DE];         // live range of each kept row (absolute k)
#pragma unroll
    for (int d = 0; d < RM; d++) {
        rlo[d] = RG_EMPTY_LO, rhi[d] = RG_EMPTY_HI;
#pragma unroll
        for (int t = 0; t < RG_T; t++) Mh[d][t] = 0u;
    }
#pragma unroll
    for (int d = 0; d < DE; d++) {
        elo[d] = RG_EMPTY_LO, ehi[d] = RG_EMPTY_HI;
#pragma unroll
        for (int t = 0; t < RG_T; t++) Ih[d][t] = 0u, Dh[d][t] = 0u;
    }

#ifdef WFA_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    for (;;) {
        // ---------------------------------------------------------------- refill: rows that need a pair are
        // served one after the other, and ALL 64 lanes stage + 2-bit pack that row's sequences (coalesced
        // dword loads), so a refill costs one pass of the wave instead of four passes of 16 lanes.
        for (unsigned long long need = __ballot(st == 0); need != 0ull;) {
            const int r = __builtin_ctzll(need) >> 4;  // wave-uniform row index
            need &= ~(0xFFFFull << (16 * r));
            uint32_t wi = 0;
            if (lane == 16 * r) wi = atomicAdd(P.queue_head, 1u);
            wi = __shfl(wi, 16 * r, 64);
            const bool mine = (grp == r);
            if (wi >= P.chunk_n) {
                if (mine) st = 2;
                continue;
            }
            const uint32_t pr = P.work ? P.work[wi] : P.chunk_first + wi;
            const uint32_t nq = P.q_len[pr], mt = P.t_len[pr];
            uint32_t status = ST_PENDING;
            if (nq == 0 || mt == 0)
                status = ST_EMPTY;  // wfa.go:204-206
            else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
                status = ST_TOO_LONG;  // wfa.go:207-209
            else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW)
                status = ST_REDO_LDS;
            if (status == ST_PENDING) {
                uint32_t *const rq = lds + r * 2 * SW;
                bool bad = stage_pack<64>(P.blob, P.q_off[pr], nq, rq, lane);
                bad |= stage_pack<64>(P.blob, P.t_off[pr], mt, rq + SW, lane);
                if (__ballot(bad) != 0ull) status = ST_REDO_BYTES;
            }
            if (status != ST_PENDING) {
                if (lane == 16 * r) {
                    P.pair_meta[wi] = make_uint4(status, 0u, 0u, 0u);
                    if (status >= ST_REDO_BYTES) push_redo(P, pr, status);
                }
                continue;  // the row stays in state 0 and pulls another pair in the next round
            }
            if (mine) {
                pidx = wi, pair = pr;
                n = (int)nq, m = (int)mt, Ak = m - n;
                sv.n = n, sv.m = m;
                s = 0, si = 0, top = 0, my_cells = 0;
                kb = (Ak / 2) - RG_W / 2;            // window [kb, kb+64) around the main diagonals
                if (kb > -RG_G / 2) kb = -RG_G / 2;  // k = 0 (the seed) must be inside, in tile 0 if possible
                if (kb + RG_W <= 0) kb = -RG_W + RG_G / 2;
                A = P.arena + (uint64_t)pidx * cap;
#pragma unroll
                for (int d = 0; d < RM; d++) {
                    rlo[d] = RG_EMPTY_LO, rhi[d] = RG_EMPTY_HI;
#pragma unroll
                    for (int t = 0; t < RG_T; t++) Mh[d][t] = 0u;
                }
#pragma unroll
                for (int d = 0; d < DE; d++) {
                    elo[d] = RG_EMPTY_LO, ehi[d] = RG_EMPTY_HI;
#pragma unroll
                    for (int t = 0; t < RG_T; t++) Ih[d][t] = 0u, Dh[d][t] = 0u;
                }
                st = 1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__ballot(st != 2) == 0ull) break;
        const bool run = (st == 1);
        WFA_STAMP(0);  // refill

        // ---------------------------------------------------------------- range of this score (wfa.go:557-563)
        int lo = INT32_MAX, hi = INT32_MIN;
        if (rhi[DX - 1] >= rlo[DX - 1]) lo = imin2(lo, rlo[DX - 1] - 1), hi = imax2(hi, rhi[DX - 1] + 1);
        if (rhi[DOE - 1] >= rlo[DOE - 1]) lo = imin2(lo, rlo[DOE - 1] - 1), hi = imax2(hi, rhi[DOE - 1] + 1);
        if (ehi[DE - 1] >= elo[DE - 1]) lo = imin2(lo, elo[DE - 1] - 1), hi = imax2(hi, ehi[DE - 1] + 1);
        lo = imax2(lo, -(n - 1));
        hi = imin2(hi, m - 1);
        const bool seeded = run && (s == 0u || s == x);  // global: M[0][0] or M[x][0] (wfa.go:155-160)
        if (seeded) lo = imin2(lo, 0), hi = imax2(hi, 0);
        if (!run) lo = 0, hi = -1;
        const bool nonempty = hi >= lo;

        // ---------------------------------------------------------------- keep the band inside the window
        int all_lo = RG_EMPTY_LO, all_hi = RG_EMPTY_HI;  // every row still in the ring must stay inside too
#pragma unroll
        for (int d = 0; d < RM; d++) all_lo = imin2(all_lo, rlo[d]), all_hi = imax2(all_hi, rhi[d]);
#pragma unroll
        for (int d = 0; d < DE; d++) all_lo = imin2(all_lo, elo[d]), all_hi = imax2(all_hi, ehi[d]);
        if (nonempty) all_lo = imin2(all_lo, lo), all_hi = imax2(all_hi, hi);
        const bool need_dn = run && nonempty && all_lo < kb;                // window moves down 16 diagonals
        const bool need_up = run && nonempty && !need_dn && all_lo - 1 >= kb + RG_G;  // keep the band start in tile 0
        if (__ballot(need_dn || need_up) != 0ull) {
            const int sh = need_dn ? -1 : (need_up ? 1 : 0);
            auto shift4 = [&](uint32_t(&R)[RG_T]) {
                uint32_t o0 = R[0], o1 = R[1], o2 = R[2], o3 = R[3];
                R[0] = sh < 0 ? 0u : (sh > 0 ? o1 : o0);
                R[1] = sh < 0 ? o0 : (sh > 0 ? o2 : o1);
                R[2] = sh < 0 ? o1 : (sh > 0 ? o3 : o2);
                R[3] = sh < 0 ? o2 : (sh > 0 ? 0u : o3);
            };
#pragma unroll
            for (int d = 0; d < RM; d++) shift4(Mh[d]);
#pragma unroll
            for (int d = 0; d < DE; d++) shift4(Ih[d]), shift4(Dh[d]);
            kb += sh * RG_G;
        }
        const bool too_wide = run && nonempty && (all_lo < kb || all_hi > kb + RG_W - 1);

        WFA_STAMP(1);  // range + window
        // ---------------------------------------------------------------- next + seeds + extend, tile by tile
        uint32_t cM[RG_T], cI[RG_T], cD[RG_T], cO[RG_T], mb[RG_T];  // cO = backtrace off0 of the M cell
        int      ev[RG_T], eh[RG_T];  // position of the next window of a cell that is still matching (ev < 0: done)
        bool     act[RG_T];           // wave-uniform: some pair of this wave touches tile t
        uint32_t termbits = 0u;
#pragma unroll
        for (int t = 0; t < RG_T; t++) {
            cM[t] = cI[t] = cD[t] = cO[t] = 0u, mb[t] = 0u, ev[t] = -1, eh[t] = 0;
            const int  k   = kb + RG_G * t + j;
            const bool inr = run && !too_wide && k >= lo && k <= hi;
            act[t]         = __ballot(inr) != 0ull;
            if (!act[t]) continue;
            const uint32_t mo_km1 = row_prev(Mh[DOE - 1][t], t > 0 ? Mh[DOE - 1][t - 1] : 0u);
            const uint32_t ie_km1 = row_prev(Ih[DE - 1][t], t > 0 ? Ih[DE - 1][t - 1] : 0u);
            const uint32_t mo_kp1 = row_next(Mh[DOE - 1][t], t < RG_T - 1 ? Mh[DOE - 1][t + 1] : 0u);
            const uint32_t de_kp1 = row_next(Dh[DE - 1][t], t < RG_T - 1 ? Dh[DE - 1][t + 1] : 0u);
            const uint32_t mx_k   = Mh[DX - 1][t];
            Cell c = next_cell(mo_km1, ie_km1, mo_kp1, de_kp1, mx_k, k, n, m);
            if (!inr) c.M = c.I = c.D = 0u;
            if (seeded && k == 0 && c.M == 0u && inr) c.M = seed_word<0>(sv, 0, s, x, true);
            cO[t] = c.off0;
            if (__ballot(c.rej && inr) != 0ull)  // rare: a source was rejected near a sequence end
                cO[t] = c.rej ? off0_unrejected(mo_km1, ie_km1, mo_kp1, de_kp1, mx_k, c.M & TAG_MASK) : cO[t];
            // WF_EXTEND, first 16-base window (wfa.go:394-455): almost every off-path diagonal stops here
            const int h = (int)(c.M >> TAG_BITS), v = h - k;
            if (c.M != 0u && v > 0 && v < n && h < m) {
                const int      rem = imin2(n - v, m - h);
                const uint32_t xr  = SeqView<0>::win16(lq, v) ^ SeqView<0>::win16(lt, h);
                const int      cnt = imin2(xr ? (__builtin_ctz(xr) >> 1) : 16, rem);
                c.M += (uint32_t)cnt << TAG_BITS;
                if (xr == 0u && rem > 16) ev[t] = v + 16, eh[t] = h + 16;
            }
            cM[t] = c.M, cI[t] = c.I, cD[t] = c.D;
        }
        // the few cells (normally the one on the alignment path) whose first window matched completely
        for (;;) {
            bool more = false;
#pragma unroll
            for (int t = 0; t < RG_T; t++) more |= (ev[t] >= 0);
            if (__ballot(more) == 0ull) break;
#pragma unroll
            for (int t = 0; t < RG_T; t++) {
                if (!act[t]) continue;
                if (ev[t] >= 0) {
                    const int      rem = imin2(n - ev[t], m - eh[t]);
                    const uint32_t xr  = SeqView<0>::win16(lq, ev[t]) ^ SeqView<0>::win16(lt, eh[t]);
                    const int      cnt = imin2(xr ? (__builtin_ctz(xr) >> 1) : 16, rem);
                    cM[t] += (uint32_t)cnt << TAG_BITS;
                    if (xr == 0u && rem > 16)
                        ev[t] += 16, eh[t] += 16;
                    else
                        ev[t] = -1;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < RG_T; t++) {
            if (!act[t]) continue;
            const int k = kb + RG_G * t + j;
            mb[t]       = grp_bits(__ballot(cM[t] != 0u), grp);
            termbits |= grp_bits(__ballot(k == Ak && (int)(cM[t] >> TAG_BITS) >= m && cM[t] != 0u), grp);  // wfa.go:235-239
        }
        WFA_STAMP(2);  // sources + next + extend
        const bool     term = termbits != 0u;
        const uint32_t mlo32 = mb[0] | (mb[1] << 16), mhi32 = mb[2] | (mb[3] << 16);
        const bool     anyM = (mlo32 | mhi32) != 0u;
        int nlo = 0, nhi = -1;  // band to keep = M.Lo..M.Hi of the reference (tight range of M cells)
        if (anyM) nlo = kb + ctz_pair(mlo32, mhi32), nhi = kb + msb_pair(mlo32, mhi32);

        // ---------------------------------------------------------------- wf-adaptive (wfa.go:461-540)
        // The remaining distance of wfa.go:488 is max(m-h, n-v) = max(m, n+k) - h, and an entry is usable iff
        // k <= h < min(m, n+k) (wfa.go:483).  The first/last non-failing positions and the last failing one
        // before them are min/max reductions over the 16 lanes of the row (DPP), no masks needed.
        const bool want_reduce = run && !term && P.adaptive && anyM && (nhi - nlo + 1) >= (int)P.min_wf_len;
        if (__ballot(want_reduce) != 0ull) {
            int  dd[RG_T], dmin = INT32_MAX;
            bool vd[RG_T];
#pragma unroll
            for (int t = 0; t < RG_T; t++) {
                dd[t] = 0, vd[t] = false;
                if (!act[t]) continue;
                const int k = kb + RG_G * t + j, h = (int)(cM[t] >> TAG_BITS), nk = n + k;
                vd[t] = cM[t] != 0u && h >= k && h < imin2(m, nk);
                dd[t] = imax2(m, nk) - h;
                if (vd[t]) dmin = imin2(dmin, dd[t]);
            }
            const int mind = grp_min(dmin);
            const int thr  = mind == INT32_MAX ? INT32_MAX : mind + (int)P.max_dist_diff;
            int       okmin = RG_W, okmax = -1;
            bool      failed = false;
#pragma unroll
            for (int t = 0; t < RG_T; t++) {
                if (!act[t]) continue;
                const bool okc = vd[t] && dd[t] <= thr;
                if (okc) okmin = imin2(okmin, RG_G * t + j), okmax = RG_G * t + j;
                failed |= vd[t] && !okc;
            }
            const bool anyfail = grp_bits(__ballot(failed && mind != INT32_MAX), grp) != 0u;
            if (__ballot(want_reduce && anyfail) != 0ull) {
                const int first_ok = grp_min(okmin);
                const int last_ok  = -grp_min(-okmax);
                int       vmax     = -1;  // last usable entry before the first non-failing one (all of them failed)
#pragma unroll
                for (int t = 0; t < RG_T; t++)
                    if (act[t] && vd[t] && RG_G * t + j < first_ok) vmax = RG_G * t + j;
                const int leadp = -grp_min(-vmax);
                if (want_reduce && anyfail) {
                    if (leadp >= 0) nlo = kb + leadp + 1;  // wfa.go:509-511
                    nhi = kb + last_ok;                      // wfa.go:517-524
                }
            }
        }

        WFA_STAMP(3);  // masks + wf-adaptive
        // ---------------------------------------------------------------- store the surviving band
        const int  wn       = (nhi >= nlo) ? nhi - nlo + 1 : 0;
        const bool no_room  = run && ((uint64_t)top + (uint32_t)wn + 4ull * (si + 2u) > cap);
        const bool give_up  = run && (too_wide || no_room);
        const bool store_ok = run && !give_up;
        uint32_t *const row = A + top;  // one compact backtrace word per surviving diagonal
#pragma unroll
        for (int t = 0; t < RG_T; t++) {
            if (!act[t]) continue;  // nothing was computed in this tile: its registers are already 0
            const int  k    = kb + RG_G * t + j;
            const bool keep = store_ok && k >= nlo && k <= nhi;
            if (!keep) cM[t] = cI[t] = cD[t] = 0u;  // Delete of wfa.go:526-535: the words never exist
            if (__ballot(keep) == 0ull) continue;
            if (keep) {
                row[k - nlo] = compact_word(cM[t], cI[t], cD[t], cO[t]);
                my_cells += (cM[t] != 0u) + (cI[t] != 0u) + (cD[t] != 0u);
            }
        }
        if (store_ok && j == 0)
            *reinterpret_cast<uint4 *>(A + cap - 4ull * (si + 1)) =
                wn > 0 ? make_uint4(top, (uint32_t)nlo, (uint32_t)wn, 0u) : make_uint4(0u, 0u, 0u, 0u);
        if (store_ok) top += (uint32_t)wn;

        WFA_STAMP(4);  // stores
        // ---------------------------------------------------------------- advance the register ring
#pragma unroll
        for (int d = RM - 1; d > 0; d--) {
            rlo[d] = rlo[d - 1], rhi[d] = rhi[d - 1];
#pragma unroll
            for (int t = 0; t < RG_T; t++) Mh[d][t] = Mh[d - 1][t];
        }
#pragma unroll
        for (int d = DE - 1; d > 0; d--) {
            elo[d] = elo[d - 1], ehi[d] = ehi[d - 1];
#pragma unroll
            for (int t = 0; t < RG_T; t++) Ih[d][t] = Ih[d - 1][t], Dh[d][t] = Dh[d - 1][t];
        }
        rlo[0] = elo[0] = wn > 0 ? nlo : RG_EMPTY_LO;
        rhi[0] = ehi[0] = wn > 0 ? nhi : RG_EMPTY_HI;
#pragma unroll
        for (int t = 0; t < RG_T; t++) Mh[0][t] = cM[t], Ih[0][t] = cI[t], Dh[0][t] = cD[t];

        // ---------------------------------------------------------------- finish / next score
        if (give_up || (run && term)) {
            const uint32_t cells = grp_sum(my_cells);
            int            hf    = 0;  // extended offset of the end cell M[s][Ak]: where the backtrace starts
#pragma unroll
            for (int t = 0; t < RG_T; t++)
                if (kb + RG_G * t + j == Ak) hf = (int)(Mh[0][t] >> TAG_BITS);
            hf = -grp_min(-hf);
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
        }
        WFA_STAMP(5);  // ring advance + finish
    }
#ifdef WFA_STAMPS
    if (lane == 0 && P.debug_info) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(P.debug_info);
        for (int i = 0; i < 8; i++) atomicAdd(acc + i, stamp_acc[i]);
    }
#endif
}

}  // namespace wfa
