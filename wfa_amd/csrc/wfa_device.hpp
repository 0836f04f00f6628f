// wfa_device.hpp -- device functions shared by the gfx950 alignment kernels.
//
// Each function cites the reference code it reproduces (file:line into shenwei356/wfa v0.4.0).
// Nothing here is GEMM-shaped: it is integer max/compare work on u32 offsets, so no MFMA; the
// levers are wave64 lane<->diagonal mapping, LDS-resident 2-bit sequences and one coalesced store
// per finished wavefront row.
#pragma once
#include "wfa_common.hpp"

namespace wfa {

#define WFA_DEV __device__ __forceinline__

WFA_DEV uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
WFA_DEV uint32_t umin2(uint32_t a, uint32_t b) { return a < b ? a : b; }
WFA_DEV int      imin2(int a, int b) { return a < b ? a : b; }
WFA_DEV int      imax2(int a, int b) { return a > b ? a : b; }

// A pair this launch cannot finish: queue it (with the reason) for the host's next configuration.
WFA_DEV void push_redo(const KParams &P, uint32_t pair, uint32_t status) {
    const uint32_t i     = atomicAdd(P.redo_count, 1u);
    P.redo_list[2u * i]      = pair;
    P.redo_list[2u * i + 1u] = status;
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions (DPP/ds_swizzle via __shfl_xor; 64 lanes, not 32)
// Four DPP butterfly stages reduce each 16-lane row, row_bcast15 / row_bcast31 carry the row results up to lane 63
// (the classic GCN wave reduction: no LDS round trips, unlike __shfl_xor = ds_bpermute), v_readlane broadcasts.
#define WFA_WAVE_REDUCE(OP)                                                                                   \
    asm("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP          \
        " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP                          \
        " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP                              \
        " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP                                   \
        " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t" OP                                 \
        " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"                                         \
        : "+v"(v))
WFA_DEV int wave_min(int v) {
    WFA_WAVE_REDUCE("v_min_i32_dpp");
    return __builtin_amdgcn_readlane(v, 63);
}
WFA_DEV int wave_max(int v) {
    WFA_WAVE_REDUCE("v_max_i32_dpp");
    return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------------------------
// Sequence access.  MODE 0: 2-bit packed in LDS, 16 bases per u32 (base i at bits 2(i%16)),
// valid only when both sequences are pure uppercase ACGT.  MODE 1: raw bytes in global memory (any
// alphabet; the reference compares raw bytes, wfa.go:408-454).
template <int MODE>
struct SeqView;

template <>
struct SeqView<0> {
    const uint32_t *q, *t;
    int             n, m;
    WFA_DEV uint32_t qbase(int i) const { return (q[i >> 4] >> ((i & 15) * 2)) & 3u; }
    WFA_DEV uint32_t tbase(int i) const { return (t[i >> 4] >> ((i & 15) * 2)) & 3u; }
    // 16-base window starting at base p (needs one readable pad word after the last data word)
    static WFA_DEV uint32_t win16(const uint32_t *s, int p) {
        int w = p >> 4;
        return __funnelshift_r(s[w], s[w + 1], (uint32_t)(p & 15) * 2u);
    }
    // longest common prefix of q[v:], t[h:] (0-based v, h), clamped to the sequence ends.
    // Equals what the 8-byte block loop + byte tail of wfa.go:410-454 add up to.
    WFA_DEV int lcp(int v, int h) const {
        int rem = imin2(n - v, m - h);
        int tot = 0;
        while (tot < rem) {
            uint32_t x = win16(q, v + tot) ^ win16(t, h + tot);
            if (x) {
                tot += __builtin_ctz(x) >> 1;
                break;
            }
            tot += 16;
        }
        return imin2(tot, rem);
    }
};

template <>
struct SeqView<1> {
    const uint8_t *q, *t;
    int            n, m;
    WFA_DEV uint32_t qbase(int i) const { return q[i]; }
    WFA_DEV uint32_t tbase(int i) const { return t[i]; }
    WFA_DEV int      lcp(int v, int h) const {
        int rem = imin2(n - v, m - h);
        int tot = 0;
        while (tot < rem && q[v + tot] == t[h + tot]) tot++;
        return tot;
    }
};

// Pack `len` bytes at blob[off..) into 2-bit words dst[0..ceil(len/16)] (last index = zero pad word).
// Threads tid, tid+G, ... each produce one word from up to five aligned dword loads (coalesced across
// the group).  Returns true if this thread saw a byte outside {A,C,G,T}.
// word j (bases 16j .. 16j+15; 0 past the end) of a sequence, 2-bit packed; bad |= a byte outside {A,C,G,T}
WFA_DEV uint32_t stage_word(const uint8_t *blob, uint64_t off, uint32_t len, uint32_t j, bool &bad) {
    const uint32_t nw   = (len + 15u) >> 4;
    uint32_t       word = 0;
    if (j < nw) {
        const uintptr_t a  = (uintptr_t)(blob + off) + 16ull * j;
        const uint32_t *p  = (const uint32_t *)(a & ~(uintptr_t)3);
        const uint32_t  sh = (uint32_t)(a & 3) * 8u;
        const uint32_t  nb = (len - 16u * j) < 16u ? (len - 16u * j) : 16u;
        const uint32_t  nd = ((uint32_t)(a & 3) + nb + 3u) >> 2;  // dwords that hold valid bytes: 1..5
        uint32_t        d[5];
#pragma unroll
        for (int i = 0; i < 5; i++) d[i] = ((uint32_t)i < nd) ? p[i] : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t w = __funnelshift_r(d[i], d[i + 1], sh);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t idx = 4u * i + b;
                uint32_t c   = (w >> (8 * b)) & 0xFFu;
                bool     ok  = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
                if (idx < nb) {
                    bad |= !ok;
                    word |= ((c >> 1) & 3u) << (2u * idx);
                }
            }
        }
    }
    return word;
}

template <int G>
WFA_DEV bool stage_pack(const uint8_t *blob, uint64_t off, uint32_t len, uint32_t *dst, int tid) {
    const uint32_t nw  = (len + 15u) >> 4;
    bool           bad = false;
    for (uint32_t j = tid; j <= nw; j += G) dst[j] = stage_word(blob, off, len, j, bad);
    return bad;
}

// ---------------------------------------------------------------------------------------------
// One diagonal of WF_NEXT (wfa.go:572-699).  Inputs are the raw source words (0 = absent):
//   mo_km1 = M[s-o-e][k-1], ie_km1 = I[s-e][k-1], mo_kp1 = M[s-o-e][k+1], de_kp1 = D[s-e][k+1],
//   mx_k   = M[s-x][k].   n = len(q), m = len(t).
struct Cell {
    uint32_t M, I, D;  // raw words; 0 = nothing stored
    uint32_t off0;     // what backTrace recomputes for this M cell (wfa.go:766-817): the pre-extension offset
                       // from the sources WITHOUT the bounds rejections of next(); 0 = no source at all
    bool     rej;      // a source was rejected by next(): off0 must come from off0_unrejected()
};

WFA_DEV Cell next_cell(uint32_t mo_km1, uint32_t ie_km1, uint32_t mo_kp1, uint32_t de_kp1, uint32_t mx_k,
                       int k, int n, int m) {
    Cell c;
    // insertion (wfa.go:579-609): a source is rejected when its offset is > m (not >=)
    uint32_t v1 = mo_km1 >> TAG_BITS, v2 = ie_km1 >> TAG_BITS;
    bool     fM = mo_km1 != 0 && (int)v1 <= m;
    bool     fI = ie_km1 != 0 && (int)v2 <= m;
    v1          = fM ? v1 : 0u;
    v2          = fI ? v2 : 0u;
    const bool     uI  = fM | fI;
    const uint32_t Isk = uI ? umax2(v1, v2) + 1u : 0u;
    const uint32_t tI  = (fM && (!fI || v1 >= v2)) ? TAG_INS_OPEN : TAG_INS_EXT;
    c.I                = uI ? ((Isk << TAG_BITS) | tI) : 0u;

    // deletion (wfa.go:614-645): rejected when offset - k > n
    v1 = mo_kp1 >> TAG_BITS;
    v2 = de_kp1 >> TAG_BITS;
    fM = mo_kp1 != 0 && (int)v1 - k <= n;
    bool fD = de_kp1 != 0 && (int)v2 - k <= n;
    v1      = fM ? v1 : 0u;
    v2      = fD ? v2 : 0u;
    const bool     uD  = fM | fD;
    const uint32_t Dsk = uD ? umax2(v1, v2) : 0u;
    const uint32_t tD  = (fM && (!fD || v1 >= v2)) ? TAG_DEL_OPEN : TAG_DEL_EXT;
    c.D                = uD ? ((Dsk << TAG_BITS) | tD) : 0u;

    // mismatch (wfa.go:650-698): rejected when offset > m or offset - k > n; mismatch wins ties,
    // then insertion, then deletion; M carries I's / D's tag when it comes from them.
    v1 = mx_k >> TAG_BITS;
    fM = mx_k != 0 && !((int)v1 > m || (int)v1 - k > n);
    v1 = fM ? v1 : 0u;
    const uint32_t Msk = umax2(umax2(Isk, Dsk), v1 + 1u);
    uint32_t       tM;
    if (fM && Msk == v1 + 1u)
        tM = TAG_MISMATCH;
    else if (uI && (Msk == Isk || !uD))
        tM = tI;
    else
        tM = tD;
    c.M = (uI | uD | fM) ? ((Msk << TAG_BITS) | tM) : 0u;
    // backTrace's view of the same cell (wfa.go:766-817) uses plain Gets with no rejection.  Unless one of the
    // five sources was rejected above, that is simply Isk (InsExt tag), Dsk (DelExt tag) or Msk (any other tag);
    // c.rej flags the rare cells near a sequence end where the caller must use off0_unrejected() instead.
    c.off0 = c.M == 0u ? 0u : (tM == TAG_INS_EXT ? Isk : (tM == TAG_DEL_EXT ? Dsk : Msk));
    c.rej  = (mo_km1 != 0u && (int)(mo_km1 >> TAG_BITS) > m) | (ie_km1 != 0u && !fI) |
            (mo_kp1 != 0u && (int)(mo_kp1 >> TAG_BITS) - k > n) | (de_kp1 != 0u && !fD) | (mx_k != 0u && !fM);
    return c;
}

// The reference's recomputation without bounds rejection; tag = the tag the M cell finally carries.
WFA_DEV uint32_t off0_unrejected(uint32_t mo_km1, uint32_t ie_km1, uint32_t mo_kp1, uint32_t de_kp1, uint32_t mx_k,
                                 uint32_t tag) {
    const uint32_t Iu = (mo_km1 | ie_km1) ? umax2(mo_km1 >> TAG_BITS, ie_km1 >> TAG_BITS) + 1u : 0u;
    const uint32_t Du = (mo_kp1 | de_kp1) ? umax2(mo_kp1 >> TAG_BITS, de_kp1 >> TAG_BITS) : 0u;
    const uint32_t Xu = mx_k ? (mx_k >> TAG_BITS) + 1u : 0u;
    return tag == TAG_INS_EXT ? Iu : (tag == TAG_DEL_EXT ? Du : umax2(umax2(Iu, Du), Xu));
}

// Compact backtrace word (wfa_reg_kernel / wfa_packed_kernel, CompactView fmt 0): everything backTrace needs from a diagonal of one score.
//   bits 0-2  tag of the M cell (0 = no M cell)      bits 3-4  I cell: 0 none, 1 InsOpen, 2 InsExt
//   bits 5-6  D cell: 0 none, 1 DelOpen, 2 DelExt     bits 7-31 off0 of the M cell (25 bits)
// The walk never re-reads a cell's extended offset: it tracks h itself (wfa.go:851-853,886-909) and only
// fetches the next cell's tag (wfa.go:915-920), so the offsets themselves need not be stored.
WFA_DEV uint32_t compact_word(uint32_t M, uint32_t I, uint32_t D, uint32_t off0) {
    // InsOpen/InsExt are tags 1/2 and DelOpen/DelExt 3/4 (wfa_backtrace_types.go:27-31): I -> tag, D -> tag - 2
    const uint32_t td = D ? (D & TAG_MASK) - 2u : 0u;
    const uint32_t wd = (M & TAG_MASK) | ((I & 3u) << 3) | (td << 5) | (off0 << 7);
    return M ? wd : 0u;
}

// Word of the BLOCKED kernels' arenas (wfa_blk_kernel; CompactView fmt 1, 3, 4, 5).  The forward pass is bound by the
// number of vector instructions it issues, the walk by DRAM latency, so the word is what is cheapest to PRODUCE: the
// pre-extension offset with the four comparison results of next() shifted in under it (one add-with-carry each),
// not the reference's tag values -- the walk derives those:
//   bit 0  fromI: the M cell took the insertion's offset (wfa.go:664,672: Msk == Isk, after the mismatch's tie)
//   bit 1  fromX: the M cell took the mismatch's offset  (wfa.go:660,677,687: "mismatch is prefered")
//   bit 2  dext:  the D cell is a DeleteExt (v1 < v2, wfa.go:626-636), else a DeleteOpen
//   bit 3  iext:  the I cell is an InsertExt (v1 < v2, wfa.go:590-600), else an InsertOpen
//   bits 4-31  off0 of the M cell; 0 marks a seed of initComponents (wfa.go:155-160): bit 0 = Match, bit 1 = Mismatch
// Whether an I, D or M cell EXISTS is not recorded: the walk only ever steps to a cell that a stored decision
// names as a source, and a source existed (after its own row's wf-adaptive) when the decision was taken.
constexpr uint32_t BLK_SEED_MATCH = 1u, BLK_SEED_MISMATCH = 2u;
WFA_DEV uint32_t blk_word(uint32_t off0, bool iext, bool dext, bool fromX, bool fromI) {
    uint32_t w = off0;
    w = w + w + (iext ? 1u : 0u);
    w = w + w + (dext ? 1u : 0u);
    w = w + w + (fromX ? 1u : 0u);
    w = w + w + (fromI ? 1u : 0u);
    return w;
}
// tag of the cell of component comp (0 = M, 1 = I, 2 = D) a blocked-kernel word describes; 0 for a word never written as zero
WFA_DEV uint32_t blk_tag(uint32_t wd, int comp, uint32_t &off0) {
    off0 = wd >> 4;
    if (wd == 0u) return 0u;
    const uint32_t ti = (wd & 8u) ? TAG_INS_EXT : TAG_INS_OPEN, td = (wd & 4u) ? TAG_DEL_EXT : TAG_DEL_OPEN;
    if (comp == 1) return ti;
    if (comp == 2) return td;
    if (off0 == 0u) return (wd & 1u) ? TAG_MATCH : TAG_MISMATCH;  // a seed: the walk ends here (offset0 == 0, wfa.go:822-825)
    return (wd & 2u) ? (uint32_t)TAG_MISMATCH : ((wd & 1u) ? ti : td);
}

// Seeds of initComponents (wfa.go:143-184) that belong to score s, as a raw word for diagonal k
// (0 if none).  Global: only k = 0.  Semi-global: first row k = 1..m-1 (offset k+1), first column
// k = -1..-(n-1) (offset 1).  Class: score 0 / Match when the bases agree, else score x / Mismatch.
template <int MODE>
WFA_DEV uint32_t seed_word(const SeqView<MODE> &sv, int k, uint32_t s, uint32_t x, bool global_alignment) {
    if (k != 0 && global_alignment) return 0u;
    if (k > sv.m - 1 || k < -(sv.n - 1)) return 0u;
    bool     match = (k >= 0) ? (sv.qbase(0) == sv.tbase(k)) : (sv.qbase(-k) == sv.tbase(0));
    uint32_t h     = (k >= 0) ? (uint32_t)(k + 1) : 1u;
    uint32_t cls   = match ? 0u : x;
    if (cls != s) return 0u;
    return (h << TAG_BITS) | (match ? TAG_MATCH : TAG_MISMATCH);
}

// WF_EXTEND for one diagonal (wfa.go:394-455): only cells with 0 < v < n and h < m are extended.
template <int MODE>
WFA_DEV uint32_t extend_word(const SeqView<MODE> &sv, uint32_t raw, int k) {
    if (raw == 0u) return 0u;
    int h = (int)(raw >> TAG_BITS);
    int v = h - k;
    if (v <= 0 || v >= sv.n || h >= sv.m) return raw;
    return raw + ((uint32_t)sv.lcp(v, h) << TAG_BITS);
}

// remaining-distance of wf-adaptive (wfa.go:474-494): -1 when absent or past a sequence end
WFA_DEV int reduce_dist(uint32_t raw, int k, int n, int m) {
    if (raw == 0u) return -1;
    int h = (int)(raw >> TAG_BITS);
    int v = h - k;
    if (v < 0 || v >= n || h >= m) return -1;
    return imax2(m - h, n - v);
}

// ---------------------------------------------------------------------------------------------
// Arena access used by the end-cell search and the backtrace.  The directory grows downward from the
// end of the slot: entry i sits at arena + cap - DIR_WORDS*(i+1) words.
WFA_DEV DirEnt load_dir(const uint32_t *p) {
    const uint4 r = *reinterpret_cast<const uint4 *>(p);
    const uint  st = p[4];
    DirEnt      e;
    e.base   = (uint64_t)r.x | ((uint64_t)r.y << 32);
    e.lo     = (int)r.z;
    e.w      = (int)r.w;
    e.stride = st;
    return e;
}
WFA_DEV void store_dir(uint32_t *p, uint64_t base, int lo, int w, uint32_t stride) {
    *reinterpret_cast<uint4 *>(p) = make_uint4((uint32_t)base, (uint32_t)(base >> 32), (uint32_t)lo, (uint32_t)w);
    p[4]                          = stride;
}

struct ArenaView {
    const uint32_t *A;
    uint64_t        cap;    // words
    uint32_t        g;      // score granularity gcd(x, o+e, e): only multiples of g can exist
    uint32_t        n_ent;  // directory entries written (scores 0, g, .., (n_ent-1)*g)

    WFA_DEV DirEnt ent(uint32_t idx) const { return load_dir(A + cap - (uint64_t)DIR_WORDS * (idx + 1)); }
    WFA_DEV void   prepare(uint32_t, int) const {}  // (ArenaViewWave: loads the directory window of a backtrace step)
    // Component.GetRaw (wfa_component.go:150-155 + wfa_wavefront.go:163-169); s may have wrapped
    // below zero (uint32), which lands beyond the directory like the reference's len check.
    WFA_DEV uint32_t get_raw(int comp, uint32_t s, int k) const {
        if (s % g != 0u) return 0u;
        uint32_t idx = s / g;
        if (idx >= n_ent) return 0u;
        DirEnt e = ent(idx);
        if (e.w <= 0 || k < e.lo || k >= e.lo + e.w) return 0u;
        return A[e.base + (uint64_t)comp * e.stride + (uint32_t)(k - e.lo)];
    }
};

// The same view for a whole wave walking the backtrace together (every lane executes the same steps on the same
// values): the directory entries of a 64-score window live in LDS, loaded by the 64 lanes at once, so a step
// costs the round trip of its cells only.  back_trace() calls prepare() once per step: the window then holds the
// entries of the scores s - dmax*g .. s, all a step can look at, and the lookups themselves carry no refill code.
// On a refill every lane also touches the cells of ITS entry that the walk can reach while the window lasts (the
// walk changes diagonal by at most one per entry it descends): the misses of ~20 steps are taken together, in
// one round trip, instead of one after the other.
struct ArenaViewWave {
    const uint32_t *A;
    uint64_t        cap;
    uint32_t        g, n_ent;
    DirEnt         *win;      // LDS, 64 entries, slot = index & 63
    uint32_t        dmax;     // farthest source of a step, in entries (< 64)
    int             g_shift;  // log2(g) when g is a power of two (no division per lookup), else -1
    mutable uint32_t win_lo, win_hi;  // entries [win_lo, win_hi] are loaded; win_lo > win_hi: none
    mutable bool     missed;          // a lookup fell outside the window

    WFA_DEV void init(const uint32_t *A_, uint64_t cap_, uint32_t g_, uint32_t n_ent_, DirEnt *lds_win, uint32_t dmax_) {
        A = A_, cap = cap_, g = g_, n_ent = n_ent_, win = lds_win, dmax = dmax_, missed = false;
        g_shift = (g & (g - 1u)) == 0u ? (int)__builtin_ctz(g) : -1;
        win_lo = 1u, win_hi = 0u;
    }
    WFA_DEV bool split(uint32_t s, uint32_t &idx) const {  // s = idx * g ?
        if (g_shift >= 0) {
            idx = s >> g_shift;
            return (s & (g - 1u)) == 0u;
        }
        idx = s / g;
        return s % g == 0u;
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
            if (d.w > 0) {
                const int reach = 66;
                const int k0 = k - reach > d.lo ? k - reach : d.lo, k1 = k + reach < d.lo + d.w - 1 ? k + reach : d.lo + d.w - 1;
                uint32_t  acc = 0;
                for (int c = 0; c < 3; c++) {
                    const uint32_t *row = A + d.base + (uint64_t)c * d.stride - d.lo;  // indexed by diagonal
                    for (int kk = k0; kk <= k1; kk += 32) acc ^= row[kk];
                    if (k1 >= k0) acc ^= row[k1];
                }
                asm volatile("" ::"v"(acc));  // the loads are wanted for the lines they bring in
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        win_lo = lo, win_hi = hi;
    }
    WFA_DEV void prepare(uint32_t s, int k) const {  // wave-uniform
        uint32_t idx;
        (void)split(s, idx);
        if (n_ent == 0u) return;
        if (idx >= n_ent) idx = n_ent - 1u;
        const uint32_t need_lo = idx >= dmax ? idx - dmax : 0u;
        if (need_lo < win_lo || idx > win_hi) refill(idx, k);
    }
    // No early return and no branch around the cell load: the five lookups of a backtrace step then have their loads
    // in flight together (a lookup that finds nothing reads word 0 of the arena and drops it).
    WFA_DEV uint32_t get_raw(int comp, uint32_t s, int k) const {
        uint32_t idx;
        bool     ok = split(s, idx);
        ok          = ok && idx < n_ent;
        // prepare() has loaded every entry a step reads.  No second path that reads the entry from memory: where the
        // two meet the compiler has to wait for ALL outstanding loads, which puts the five cell loads of a step one
        // after the other (3.3 -> 1.x us per step).  A lookup outside the window (cannot happen) poisons the walk
        // instead: the caller fails the pair.
        missed = missed || (ok && (idx < win_lo || idx > win_hi));
        const DirEnt e = win[idx & 63u];
        ok = ok && e.w > 0 && k >= e.lo && k < e.lo + e.w;
        const uint64_t off = ok ? e.base + (uint64_t)comp * e.stride + (uint32_t)(k - e.lo) : 0ull;
        const uint32_t v   = A[off];
        return ok ? v : 0u;
    }
};

// AlignmentResult being built by one lane.  Ops are appended in backtrace order into scratch and
// merged with the previous one when the letter repeats -- merging adjacent equal ops commutes with
// the reversal done by process() (wfa_cigar.go:136-166).
struct OpsWriter {
    uint64_t *buf;
    uint32_t  cap;
    uint32_t  n;
    uint64_t  cur;  // pending op (0 = none)
    bool      overflow;
    WFA_DEV void init(uint64_t *b, uint32_t c) {
        buf = b, cap = c, n = 0, cur = 0, overflow = false;
    }
    WFA_DEV void add(uint32_t letter, uint32_t cnt) {  // AddN, wfa_cigar.go:118-124
        if (cur != 0 && (uint32_t)(cur >> 32) == letter) {
            cur += cnt;
            return;
        }
        flush();
        cur = ((uint64_t)letter << 32) | cnt;
    }
    WFA_DEV void flush() {
        if (cur != 0) {
            if (n < cap)
                buf[n] = cur;
            else
                overflow = true;
            n++;
            cur = 0;
        }
    }
};

// Same interface, for the lane-per-pair backtrace kernel: ops are written from the END of the pair's
// region towards its start, so the finished list is already in forward order (process()'s reversal,
// wfa_cigar.go:142-146, becomes a no-op) at buf[cap-n .. cap).  The statistics of process()
// (wfa_cigar.go:168-211: span first-M .. last-M) are accumulated while emitting: in emission order the
// span runs from the first emitted M to the last emitted M.
#ifndef WFA_OPS_GROUP
#define WFA_OPS_GROUP 8  // CIGAR ops per combined store of the backtrace: 1 (off), 2, 4 or 8 (16 / 32 / 64 bytes)
#endif
struct OpsWriterRev {
    uint64_t *buf;
    uint32_t  cap;
    uint32_t  n;
    uint64_t  cur;
    bool      overflow;
    bool      seenM;
    uint32_t  alen, matches, gaps, regions;      // committed (up to the latest M)
    uint32_t  p_len, p_gaps, p_regions;          // pending since the latest M
    uint64_t  last;                              // last flushed op (for the no-M case)
    // Write combining: the list is written backwards, entry cap-1-n at the n-th flush.  Single 8-byte stores reach HBM
    // as partial lines; with the end of the region on a 8*GROUP-byte boundary every GROUP-th flush lands on a boundary
    // and stores itself together with the GROUP-1 entries flushed before it (which sit right above it and skipped
    // their own store).  Separate backtrace kernel, 1e6 x 1 kbp: 1.78 ms (1) / 1.35 (2) / 1.25 (4) / 1.16 (8); WRITE_SIZE
    // 2.4 -> 0.8 GB, the algorithmic 0.74 GB of ops + 0.06 GB of records.
    static constexpr uint32_t GROUP = WFA_OPS_GROUP;
    uint64_t  hist[GROUP > 1 ? GROUP - 1 : 1];   // hist[i] = the op flushed i+1 flushes ago
    bool      grouped;
    bool      active = true;                     // false: count and keep statistics, store nothing (the lanes of a wave that walks
                                                 // one pair together all run the writer; one of them stores)
    bool      combine = true;                    // false: every op is stored on its own (a wave's walk: one lane stores, the history
                                                 // of the write combining is fourteen moves per op for nothing)
    WFA_DEV void init(uint64_t *b, uint32_t c) {
        buf = b, cap = c, n = 0, cur = 0, overflow = false, seenM = false;
        grouped = combine && GROUP > 1 && (reinterpret_cast<uintptr_t>(b + c) & (8u * GROUP - 1u)) == 0u;
#pragma unroll
        for (uint32_t i = 0; i + 1 < GROUP; i++) hist[i] = 0;
        alen = matches = gaps = regions = 0;
        p_len = p_gaps = p_regions = 0;
        last = 0;
    }
    WFA_DEV void add(uint32_t letter, uint32_t cnt) {
        if (cur != 0 && (uint32_t)(cur >> 32) == letter) {
            cur += cnt;
            return;
        }
        flush();
        cur = ((uint64_t)letter << 32) | cnt;
    }
    WFA_DEV void flush() {
        if (cur == 0) return;
        if (n < cap) {
            if (!active) {
            } else if (!grouped) {
                buf[cap - 1 - n] = cur;
            } else if ((n & (GROUP - 1u)) == GROUP - 1u) {
                ulonglong2 *d = reinterpret_cast<ulonglong2 *>(buf + (cap - 1 - n));
                d[0] = make_ulonglong2(cur, hist[0]);
#pragma unroll
                for (uint32_t i = 1; i < GROUP / 2; i++) d[i] = make_ulonglong2(hist[2 * i - 1], hist[2 * i]);
            }
        } else {
            overflow = true;
        }
        n++;
        // (selects, no branches: the lanes of a wave flush different letters; committed values are all zero before the first M)
        const uint32_t letter = (uint32_t)(cur >> 32), cnt = (uint32_t)cur;
        const bool     isM = letter == 'M', isG = letter == 'I' || letter == 'D';
        const bool     join = isM && seenM;  // what was pending lies between two Ms: it counts
        alen += (join ? p_len : 0u) + (isM ? cnt : 0u), matches += isM ? cnt : 0u;
        gaps += join ? p_gaps : 0u, regions += join ? p_regions : 0u;
        p_len     = isM ? 0u : p_len + cnt;
        p_gaps    = isM ? 0u : p_gaps + (isG ? cnt : 0u);
        p_regions = isM ? 0u : p_regions + (isG ? 1u : 0u);
        seenM     = seenM || isM;
        if (grouped) {
#pragma unroll
            for (uint32_t i = GROUP > 1 ? GROUP - 2 : 0; i > 0; i--) hist[i] = hist[i - 1];
            hist[0] = cur;
        }
        last    = cur;
        cur     = 0;
    }
    // process() with no M op at all: begin = end = 0 -> only the first op of the forward list counts
    WFA_DEV void finish() {
        if (grouped && n <= cap && active) {  // the newest entries that did not fill a store
            const uint32_t r = n & (GROUP - 1u);
#pragma unroll
            for (uint32_t i = 0; i + 1 < GROUP; i++)
                if (i < r) buf[cap - n + i] = hist[i];
        }
        if (!seenM && n > 0) {
            const uint32_t letter = (uint32_t)(last >> 32), cnt = (uint32_t)last;
            alen = cnt, matches = 0;
            gaps    = (letter == 'I' || letter == 'D') ? cnt : 0u;
            regions = (letter == 'I' || letter == 'D') ? 1u : 0u;
        }
    }
};

// Compact arena of the sub-wave pipeline: one word per diagonal per score, 16-byte directory entries
// {base, lo, w, -} growing down from the slot end.
//
// Fixed-pitch variant (fmt 1, written by the blocked register-window kernel): the row of score index i starts at
// word 64*i and diagonal k sits at slot k & 63 -- a row never spans more than 64 diagonals -- so a backtrace step
// is ONE load with no directory lookup in front of it.  Slots outside a row's surviving band are never written;
// the walk never looks at them: every tag names a source cell that existed (after its own row's wf-adaptive)
// when the cell was computed.  fmt 4 is the same with 256 words per score (the wave-per-pair kernel's window).
//
// Tiled variant (fmt 3, the default of the 64-diagonal blocked kernels): tiles of 8 scores x 64 diagonals (2 KB),
// inside a tile [diagonal / 4][score & 7][diagonal & 3].  A lane's four diagonals are still one 16-byte store,
// but a 128-byte line now holds 8 consecutive scores of 4 diagonals, so the next cell of the walk (2-4 scores
// earlier, the same or a neighbouring diagonal) is often in the line just fetched: the backtrace kernel -- bound
// by random DRAM accesses -- went from 2.15 to 1.78 ms per 1e6 1 kbp pairs.
struct CompactView {
    const uint32_t *A;
    uint64_t        cap;
    uint32_t        g, n_ent, fmt;
    bool            coherent = false;  // the producer kernel is still running: agent-scope loads (never a stale L2 line)
    WFA_DEV uint32_t ld(uint64_t i) const {
        return coherent ? __hip_atomic_load(A + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : A[i];
    }
    // word of score index idx = score / g (the walk steps in index units: every penalty is a multiple of g, so a score it
    // reaches is one too, and it pays no division per step; an index that has wrapped below zero is >= n_ent: absent)
    WFA_DEV uint32_t word(uint32_t idx, int k) const {
        if (idx >= n_ent) return 0u;
        if (fmt == 3u) return ld(512ull * (idx >> 3) + (((uint32_t)k & 60u) << 3) + ((idx & 7u) << 2) + ((uint32_t)k & 3u));
        if (fmt == 1u) return ld(64ull * idx + ((uint32_t)k & 63u));
        if (fmt == 4u) return ld(256ull * idx + ((uint32_t)k & 255u));
        if (fmt == 5u) return ld(32ull * idx + ((uint32_t)k & 31u));
        if (fmt == 6u) return ld(128ull * idx + ((uint32_t)k & 127u));
        if (fmt == 7u)  // fmt 3's tiles with 16-bit words (wfa_duo_kernel: reads under 2 048 bases)
            return reinterpret_cast<const uint16_t *>(A)[512ull * (idx >> 3) + (((uint32_t)k & 60u) << 3) + ((idx & 7u) << 2) + ((uint32_t)k & 3u)];
        if (fmt == 10u)  // pairs of groups, 16-bit words (wfa_duo_kernel, round 6): [diagonal / 8 & 7][index / 8][diagonal / 4 & 1][index & 7][diagonal & 3]
            return reinterpret_cast<const uint16_t *>(A)[(uint64_t)(((uint32_t)k & 56u) >> 3) * (uint32_t)(cap >> 5) * 8u + 64ull * (idx >> 3) + (((uint32_t)k & 4u) << 3) +
                                                         ((idx & 7u) << 2) + ((uint32_t)k & 3u)];
        if (fmt == 9u)  // group-major, 16-bit words (wfa_duo_kernel, round 6): [diagonal / 4 & 15][score index][diagonal & 3], cap / 32 score indices
            return reinterpret_cast<const uint16_t *>(A)[(uint64_t)(((uint32_t)k & 60u) >> 2) * (uint32_t)(cap >> 5) * 4u + 4ull * idx + ((uint32_t)k & 3u)];
        if (fmt == 8u)  // 32 diagonals per score, 16-bit words (wfa_lane_kernel: reads of at most 240 bases)
            return reinterpret_cast<const uint16_t *>(A)[32ull * idx + ((uint32_t)k & 31u)];
        const uint4 e = *reinterpret_cast<const uint4 *>(A + cap - 4ull * (idx + 1));
        const int   lo = (int)e.y, w = (int)e.z;
        if (w <= 0 || k < lo || k >= lo + w) return 0u;
        if (fmt == 11u) return reinterpret_cast<const uint16_t *>(A)[e.x + (uint32_t)(k - lo)];  // wfa_wide_kernel: rows of 16-bit blk_word()s
        return A[e.x + (uint32_t)(k - lo)];
    }
    // tag of the cell of component comp (0 = M, 1 = I, 2 = D) at (score index idx, k); 0 = absent
    WFA_DEV uint32_t tag(int comp, uint32_t idx, int k, uint32_t &off0) const {
        const uint32_t wd = word(idx, k);
        if (fmt != 0u) return blk_tag(wd, comp, off0);  // the blocked kernels' word; fmt 0: compact_word()
        off0              = wd >> 7;
        if (comp == 0) return wd & TAG_MASK;
        if (comp == 1) {
            const uint32_t t = (wd >> 3) & 3u;
            return t == 1u ? TAG_INS_OPEN : (t == 2u ? TAG_INS_EXT : 0u);
        }
        const uint32_t t = (wd >> 5) & 3u;
        return t == 1u ? TAG_DEL_OPEN : (t == 2u ? TAG_DEL_EXT : 0u);
    }
};

// The blocked kernels' arenas seen by a whole WAVE that walks one pair together (every lane runs the same walk on the same
// values): a region of ROWS score indices x 32 diagonals (4 KB at 32 rows; 64 rows measured the same) lives in LDS, loaded by the 64 lanes in one round
// of 16-byte loads -- whole 128-byte lines of the tiled layout -- whenever the walk asks for a cell outside it.  A step of the
// walk descends 1, 2 or 4 score indices and moves at most one diagonal, so a region lasts ~10 CIGAR ops: one DRAM round trip
// per ~10 ops instead of one per op (the lane-per-pair walk of a 50 kbp pair is a chain of ~5 000 dependent misses).
#ifndef WFA_BTW_ROWS
#define WFA_BTW_ROWS 32
#endif
struct CompactViewWave {
    const uint32_t *A;
    uint64_t        cap;
    uint32_t        g, n_ent, fmt;
    static constexpr int ROWS = WFA_BTW_ROWS;  // score indices the region holds (a multiple of 8)
    static constexpr int WORDS = ROWS * 32;
    uint32_t       *reg;          // LDS, WORDS words: [score index - r0][diagonal - d0]
    mutable int     r0 = -2 * ROWS, d0 = 0;  // the region holds score indices [r0, r0 + ROWS) x diagonals [d0, d0 + 32)
    WFA_DEV uint64_t widx(uint32_t idx, int k) const {  // word index of (score index, diagonal), CompactView's layouts
        if (fmt == 3u) return 512ull * (idx >> 3) + (((uint32_t)k & 60u) << 3) + ((idx & 7u) << 2) + ((uint32_t)k & 3u);
        if (fmt == 1u) return 64ull * idx + ((uint32_t)k & 63u);
        if (fmt == 4u) return 256ull * idx + ((uint32_t)k & 255u);
        return 128ull * idx + ((uint32_t)k & 127u);  // fmt 6
    }
    WFA_DEV void refill(uint32_t idx, int k) const {
        r0 = (int)(idx & ~7u) - (ROWS - 8);  // the cell's tile row on top: ROWS-7 .. ROWS rows to descend through
        d0 = (k - 16) & ~3;
        const int lane = threadIdx.x & 63;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the 16-byte loads of a lane in flight TOGETHER: no branch between them (a row outside the arena loads row 0 and
        // stores zeros), the layout decided once -- one round trip per refill
        constexpr int NL = WORDS / 256;
        uint4 v[NL];
        bool  ok[NL];
        if (fmt == 3u) {
#pragma unroll
            for (int r = 0; r < NL; r++) {
                const int q = lane + 64 * r, si = r0 + (q >> 3), kk = d0 + 4 * (q & 7);
                ok[r] = si >= 0 && (uint32_t)si < n_ent;
                const uint32_t su = ok[r] ? (uint32_t)si : 0u;
                v[r] = *reinterpret_cast<const uint4 *>(A + 512ull * (su >> 3) + (((uint32_t)kk & 60u) << 3) + ((su & 7u) << 2));
            }
        } else {
            const uint32_t Wd = fmt == 1u ? 64u : (fmt == 4u ? 256u : 128u);
#pragma unroll
            for (int r = 0; r < NL; r++) {
                const int q = lane + 64 * r, si = r0 + (q >> 3), kk = d0 + 4 * (q & 7);
                ok[r] = si >= 0 && (uint32_t)si < n_ent;
                const uint32_t su = ok[r] ? (uint32_t)si : 0u;
                v[r] = *reinterpret_cast<const uint4 *>(A + (uint64_t)Wd * su + ((uint32_t)kk & (Wd - 1u)));
            }
        }
#pragma unroll
        for (int r = 0; r < NL; r++) *reinterpret_cast<uint4 *>(reg + 4 * (lane + 64 * r)) = ok[r] ? v[r] : make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    WFA_DEV uint32_t word(uint32_t idx, int k) const {
        if (idx >= n_ent) return 0u;
        if (WFA_RARE((uint32_t)((int)idx - r0) >= (uint32_t)ROWS || (uint32_t)(k - d0) >= 32u)) refill(idx, k);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)reg[((int)idx - r0) * 32 + (k - d0)]);  // (uniform: see backtrace_wave_one)
    }
    WFA_DEV uint32_t tag(int comp, uint32_t idx, int k, uint32_t &off0) const { return blk_tag(word(idx, k), comp, off0); }
};

// The lone pair's rows where the forward pass wrote them: in LDS, tiled (fmt 3).  Uniform like CompactViewWave's words.
struct CompactViewLds {
    const uint32_t *A;  // LDS
    uint32_t        g, n_ent;
    WFA_DEV uint32_t word(uint32_t idx, int k) const {
        if (idx >= n_ent) return 0u;
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)A[512u * (idx >> 3) + (((uint32_t)k & 60u) << 3) + ((idx & 7u) << 2) + ((uint32_t)k & 3u)]);
    }
    WFA_DEV uint32_t tag(int comp, uint32_t idx, int k, uint32_t &off0) const { return blk_tag(word(idx, k), comp, off0); }
};

WFA_DEV uint32_t op_letter(uint32_t tag) {  // wfaOps = ".IIDDXMH" (wfa_backtrace_types.go:37)
    const uint32_t tbl0 = ('.') | ('I' << 8) | ('I' << 16) | ('D' << 24);
    const uint32_t tbl1 = ('D') | ('X' << 8) | ('M' << 16) | ('H' << 24);
    return ((tag < 4 ? tbl0 : tbl1) >> (8 * (tag & 3))) & 0xFFu;
}

struct TraceOut {
    uint32_t score;
    int      tbegin, tend, qbegin, qend;
};

// backtraceStartPosistion (wfa.go:270-375): semi-global end cell.  Serial, one lane.
WFA_DEV void backtrace_start(const ArenaView &av, int n, int m, uint32_t s, uint32_t &outS, int &outK) {
    uint32_t minS = s;
    const int Ak  = m - n;
    int      lastK = Ak;
    for (uint32_t idx = s / av.g + 1; idx-- > 0;) {
        const uint32_t _s = idx * av.g;
        DirEnt         e  = av.ent(idx);
        if (e.w <= 0) continue;  // !M.HasScore(_s)
        const int lo = e.lo, hi = e.lo + e.w - 1;
        const uint32_t *row = av.A + e.base;
        bool hit = false;
        int  k   = Ak;
        for (;;) {  // wfa.go:301-326
            if (k < lo) break;
            uint32_t raw = (k <= hi) ? row[k - lo] : 0u;
            if (raw == 0u) {
                k--;
                continue;
            }
            int h = (int)(raw >> TAG_BITS), v = h - k;
            if (v <= 0 || v > n || h > m) break;
            if ((v == n && h >= n) || (h == m && v >= m)) {
                hit = true;
                break;
            }
            k--;
        }
        if (hit && _s <= minS) {
            lastK = k;
            minS  = _s;
        }
        hit = false;
        k   = Ak + 1;
        for (;;) {  // wfa.go:336-361
            if (k > hi) break;
            uint32_t raw = (k >= lo) ? row[k - lo] : 0u;
            if (raw == 0u) {
                k++;
                continue;
            }
            int h = (int)(raw >> TAG_BITS), v = h - k;
            if (v <= 0 || v > n || h > m) break;
            if ((v == n && h >= n) || (h == m && v >= m)) {
                hit = true;
                break;
            }
            k++;
        }
        if (hit && _s <= minS) {
            lastK = k;
            minS  = _s;
        }
    }
    outS = minS;
    outK = lastK;
}

// backTrace (wfa.go:703-983), one lane.  Source lookups are plain Gets with NO bounds rejection,
// exactly as the reference recomputes the pre-extension offset.
//
// The cell read at the end of a step (wfa.go:915-920) is always one of the source cells the step has just read
// (M[s-x][k], M[s-o-e][k-+1], I/D[s-e][k-+1]), so it is taken from those registers: one memory round trip per
// step instead of two.  View: ArenaView, or ArenaViewWave when a whole wave walks together.
template <class View, class Writer>
WFA_DEV void back_trace(const View &av, int lenQ, int lenT, uint32_t s, int Ak, bool semiGlobal,
                        uint32_t px, uint32_t po, uint32_t pe, Writer &ow, TraceOut &out) {
    out.score  = s;
    out.tbegin = out.tend = out.qbegin = out.qend = 0;

    int      k = Ak, h, v, h0;
    uint32_t offset, wfaType;
    int      qBegin = 0, tBegin = 0;
    uint32_t v1, v2, Isk = 0, Dsk = 0, offset0 = 0;
    bool     fromMI, fromMD, fromItself = false, fromM, fromX;
    uint32_t sMismatch, sGapOpen, sGapExt;
    bool     previousFromM = true, firstMatch = true;
    int      nMatches;
    int      M0 = 0;  // component to read the next tag from: 0 = M, 1 = I, 2 = D
    uint32_t nxMis = 0, nxOpenI = 0, nxOpenD = 0, nxExt = 0;  // the source cells of this step, by the move that reaches them

    av.prepare(s, k);
    offset  = av.get_raw(0, s, k);  // wfa.go:738
    wfaType = offset & TAG_MASK;
    h       = (int)(offset >> TAG_BITS);
    v       = h - k;

    if (h < lenT)  // wfa.go:746-750
        ow.add('I', (uint32_t)lenT - (uint32_t)h);
    else if (v < lenQ)
        ow.add('H', (uint32_t)lenQ - (uint32_t)v);

    while (v > 0 && h > 0) {  // wfa.go:753
        av.prepare(s, k);
        sMismatch = s - px;
        sGapOpen  = s - po - pe;
        sGapExt   = s - pe;
        fromMI = false, fromMD = false;
        if (wfaType == TAG_INS_EXT) {  // wfa.go:767-777
            uint32_t r1 = av.get_raw(0, sGapOpen, k - 1), r2 = av.get_raw(1, sGapExt, k - 1);
            if (r1 != 0 || r2 != 0) {
                fromMI  = true;
                offset0 = umax2(r1 >> TAG_BITS, r2 >> TAG_BITS) + 1u;
            } else {
                offset0 = 0;
            }
            nxExt = r2;
            M0    = 1;
        } else if (wfaType == TAG_DEL_EXT) {  // wfa.go:778-788
            uint32_t r1 = av.get_raw(0, sGapOpen, k + 1), r2 = av.get_raw(2, sGapExt, k + 1);
            if (r1 != 0 || r2 != 0) {
                fromMD  = true;
                offset0 = umax2(r1 >> TAG_BITS, r2 >> TAG_BITS);
            } else {
                offset0 = 0;
            }
            nxExt = r2;
            M0    = 2;
        } else {  // wfa.go:789-817
            uint32_t r1 = av.get_raw(0, sGapOpen, k - 1), r2 = av.get_raw(1, sGapExt, k - 1);
            uint32_t r3 = av.get_raw(0, sGapOpen, k + 1), r4 = av.get_raw(2, sGapExt, k + 1);
            uint32_t r5 = av.get_raw(0, sMismatch, k);
            nxMis = r5, nxOpenI = r1, nxOpenD = r3;
            v1 = r1 >> TAG_BITS, v2 = r2 >> TAG_BITS;
            if (r1 != 0 || r2 != 0) {
                fromMI = true;
                Isk    = umax2(v1, v2) + 1u;
            } else {
                Isk = 0;
            }
            v1 = r3 >> TAG_BITS, v2 = r4 >> TAG_BITS;
            if (r3 != 0 || r4 != 0) {
                fromMD = true;
                Dsk    = umax2(v1, v2);
            } else {
                Dsk = 0;
            }
            fromX = r5 != 0;
            v1    = r5 >> TAG_BITS;
            if (fromMI || fromMD || fromX) {
                offset0    = umax2(umax2(Isk, Dsk), v1 + 1u);
                fromItself = false;
            } else {
                fromItself = true;
            }
            M0 = 0;
        }
        (void)fromM, (void)M0;
        if (fromItself) break;    // wfa.go:818-821
        if (offset0 == 0) break;  // wfa.go:822-825
        h0 = (int)offset0;

        if (previousFromM) {  // wfa.go:833-869
            nMatches = h - h0;
            if (nMatches > 0) {
                if (firstMatch) {
                    firstMatch = false;
                    out.tend   = h;
                    out.qend   = v;
                }
                ow.add('M', (uint32_t)nMatches);
            }
            h = h0;
            v = h - k;
            if (wfaType == TAG_MATCH) {
                tBegin = h, qBegin = v;
            } else if (nMatches > 0) {
                tBegin = h + 1, qBegin = v + 1;
            }
            if (h <= 0 || v <= 0) break;
        }

        ow.add(op_letter(wfaType), 1);  // wfa.go:872-873

        if (semiGlobal && (h == 1 || v == 1)) break;  // wfa.go:876-879

        previousFromM = true;  // wfa.go:885-909
        bool stop     = false;
        switch (wfaType) {
        case TAG_MISMATCH: s = sMismatch; h--; offset = nxMis; break;
        case TAG_INS_OPEN: s = sGapOpen; k--; h--; offset = nxOpenI; break;
        case TAG_INS_EXT: s = sGapExt; k--; h--; previousFromM = false; offset = nxExt; break;
        case TAG_DEL_OPEN: s = sGapOpen; k++; offset = nxOpenD; break;
        case TAG_DEL_EXT: s = sGapExt; k++; previousFromM = false; offset = nxExt; break;
        default: stop = true; break;
        }
        if (stop) break;
        v = h - k;
        // wfa.go:915-920: offset = component M0 at (s, k) -- the source cell read above
        if (offset == 0) break;
        wfaType = offset & TAG_MASK;
    }

    if (h > 0 && v > 0) {  // wfa.go:930-968
        nMatches = imin2(h, v) - 1;
        if (nMatches > 0) {
            if (firstMatch) {
                firstMatch = false;
                out.tend   = h;
                out.qend   = v;
            }
            ow.add('M', (uint32_t)nMatches);
            h -= nMatches;
            v -= nMatches;
            if (wfaType == TAG_MATCH) {
                tBegin = h, qBegin = v;
            } else {
                tBegin = h + 1, qBegin = v + 1;
            }
        } else if (wfaType == TAG_MATCH) {
            tBegin = h, qBegin = v;
            if (firstMatch) {
                firstMatch = false;
                out.tend   = h;
                out.qend   = v;
            }
        }
        ow.add(op_letter(wfaType), 1);
    }
    if (v > 1) ow.add('H', (uint32_t)(v - 1));  // wfa.go:970-972
    if (h > 1) ow.add('I', (uint32_t)(h - 1));  // wfa.go:974-976
    ow.flush();
    out.tbegin = tBegin;  // wfa.go:979
    out.qbegin = qBegin;
}

// backTrace (wfa.go:703-983) over the compact arena, global alignment.  h_start = extended offset of the end
// cell M[s][Ak].  Same control flow as back_trace(); the source recomputation is replaced by the stored off0
// (M cells) -- for a cell reached inside the I or D component the reference only tests its offset0 against 0,
// and it cannot be 0 there: an InsExt/InsOpen (DelExt/DelOpen) tag is only given when that source existed.
// semiGlobal (round 5, wfa_teamc_kernel): the walk starts at the end cell backtraceStartPosistion picked -- (s, Ak) name
// that cell, h_start its extended offset -- and stops at the first row / column (wfa.go:876-879).
template <class View, class Writer>
WFA_DEV void back_trace_compact(const View &cv, int lenQ, int lenT, uint32_t s, int Ak, uint32_t h_start,
                                uint32_t px, uint32_t po, uint32_t pe, Writer &ow, TraceOut &out, bool semiGlobal = false) {
    out.score  = s;
    out.tbegin = out.tend = out.qbegin = out.qend = 0;
    int      k = Ak, h = (int)h_start, v = h - k, h0;
    int      qBegin = 0, tBegin = 0, nMatches;
    bool     previousFromM = true, firstMatch = true;
    int      comp = 0;  // component the current cell lives in
    uint32_t off0 = 0;
    uint32_t       si = s / cv.g;  // score index; the penalties in index units:
    const uint32_t dxi = px / cv.g, doei = (po + pe) / cv.g, dei = pe / cv.g;
    uint32_t wfaType = cv.tag(0, si, k, off0);  // wfa.go:738-742

    if (h < lenT)  // wfa.go:746-750
        ow.add('I', (uint32_t)lenT - (uint32_t)h);
    else if (v < lenQ)
        ow.add('H', (uint32_t)lenQ - (uint32_t)v);

    while (v > 0 && h > 0) {
        int      M0;
        uint32_t offset0;
        if (wfaType == TAG_INS_EXT)
            M0 = 1, offset0 = comp == 0 ? off0 : 1u;
        else if (wfaType == TAG_DEL_EXT)
            M0 = 2, offset0 = comp == 0 ? off0 : 1u;
        else
            M0 = 0, offset0 = comp == 0 ? off0 : 1u;
        if (offset0 == 0u) break;  // fromItself / offset0 == 0 (wfa.go:818-825)
        h0 = (int)offset0;
        if (previousFromM) {  // only ever true for cells of the M component (wfa.go:833-869)
            nMatches = h - h0;
            if (nMatches > 0) {
                if (firstMatch) firstMatch = false, out.tend = h, out.qend = v;
                ow.add('M', (uint32_t)nMatches);
            }
            h = h0;
            v = h - k;
            if (wfaType == TAG_MATCH)
                tBegin = h, qBegin = v;
            else if (nMatches > 0)
                tBegin = h + 1, qBegin = v + 1;
            if (h <= 0 || v <= 0) break;
        }
        ow.add(op_letter(wfaType), 1);  // wfa.go:872-873
        if (semiGlobal && (h == 1 || v == 1)) break;  // wfa.go:876-879
        previousFromM = true;           // wfa.go:885-909
        bool stop     = false;
        switch (wfaType) {
        case TAG_MISMATCH: si -= dxi; h--; break;
        case TAG_INS_OPEN: si -= doei; k--; h--; break;
        case TAG_INS_EXT: si -= dei; k--; h--; previousFromM = false; break;
        case TAG_DEL_OPEN: si -= doei; k++; break;
        case TAG_DEL_EXT: si -= dei; k++; previousFromM = false; break;
        default: stop = true; break;
        }
        if (stop) break;
        v = h - k;
        uint32_t       noff0;
        const uint32_t nt = cv.tag(M0, si, k, noff0);  // wfa.go:915-920: a missing cell ends the walk, the old
        if (nt == 0u) break;                           // tag stays in wfaType for the tail below
        wfaType = nt, off0 = noff0, comp = M0;
    }
    if (h > 0 && v > 0) {  // wfa.go:930-968
        nMatches = imin2(h, v) - 1;
        if (nMatches > 0) {
            if (firstMatch) firstMatch = false, out.tend = h, out.qend = v;
            ow.add('M', (uint32_t)nMatches);
            h -= nMatches, v -= nMatches;
            if (wfaType == TAG_MATCH)
                tBegin = h, qBegin = v;
            else
                tBegin = h + 1, qBegin = v + 1;
        } else if (wfaType == TAG_MATCH) {
            tBegin = h, qBegin = v;
            if (firstMatch) firstMatch = false, out.tend = h, out.qend = v;
        }
        ow.add(op_letter(wfaType), 1);
    }
    if (v > 1) ow.add('H', (uint32_t)(v - 1));  // wfa.go:970-972
    if (h > 1) ow.add('I', (uint32_t)(h - 1));  // wfa.go:974-976
    ow.flush();
    out.tbegin = tBegin, out.qbegin = qBegin;
}

}  // namespace wfa
