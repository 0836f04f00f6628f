/*
 * wfa_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see wfa_oracle.h).
 *
 * Plain-C restatement of shenwei356/wfa v0.4.0's alignment path.  The storage
 * follows the reference's semantics exactly (zig-zag diagonal index, 0 = absent,
 * Lo/Hi maintained by Set/Delete the way the reference does) because several
 * results depend on them (see the R1..R4 notes at each function).
 *
 * Reference citations are file:line into the reference checkout.
 */
#include "wfa_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OFFSETS_BASE_SIZE    128  /* wfa_wavefront.go:31 */
#define WAVEFRONTS_BASE_SIZE 2048 /* wfa_component.go:30 */

static const char WFA_OPS[8] = {'.', 'I', 'I', 'D', 'D', 'X', 'M', 'H'}; /* wfa_backtrace_types.go:37 */

/* ------------------------------------------------------------------ WaveFront
 * wfa_wavefront.go:45-183 */
typedef struct {
    int       lo, hi;
    uint32_t *off;
    int       len; /* len(Offsets) */
    int       cap; /* cap(Offsets) */
} wf_t;

static inline int k2i(int k) { /* wfa_wavefront.go:77-82 */
    return k >= 0 ? (k << 1) : (((-k) << 1) - 1);
}

/* poolWaveFront (wfa_wavefront.go:62-74): recycled wavefronts keep their slice capacity.  Go's
   sync.Pool is effectively per-thread; the oracle keeps one free list per aligner. */
typedef struct wf_pool {
    wf_t **items;
    size_t n, cap;
} wf_pool;

static wf_t *wf_new(wf_pool *pool) { /* wfa_wavefront.go:52-60: Lo=MaxInt, Hi=MinInt, 128 zeroed slots */
    wf_t *w;
    if (pool->n > 0) {
        w = pool->items[--pool->n];
    } else {
        w      = (wf_t *)malloc(sizeof(wf_t));
        w->cap = OFFSETS_BASE_SIZE;
        w->off = (uint32_t *)malloc((size_t)w->cap * sizeof(uint32_t));
    }
    w->lo  = INT_MAX;
    w->hi  = INT_MIN;
    w->len = OFFSETS_BASE_SIZE; /* wf.Offsets = wf.Offsets[:OFFSETS_BASE_SIZE]; clear(wf.Offsets) */
    memset(w->off, 0, (size_t)OFFSETS_BASE_SIZE * sizeof(uint32_t));
    return w;
}

static void wf_recycle(wf_pool *pool, wf_t *w) { /* poolWaveFront.Put */
    if (pool->n == pool->cap) {
        pool->cap   = pool->cap ? pool->cap * 2 : 256;
        pool->items = (wf_t **)realloc(pool->items, pool->cap * sizeof(wf_t *));
    }
    pool->items[pool->n++] = w;
}

static void wf_free(wf_t *w) {
    if (w) {
        free(w->off);
        free(w);
    }
}

static inline void wf_grow(wf_t *w, int i) { /* wfa_wavefront.go:87-92: grow in 128-slot zeroed chunks */
    if (i >= w->len) {
        int chunks = (i - w->len + OFFSETS_BASE_SIZE) / OFFSETS_BASE_SIZE;
        int nlen   = w->len + chunks * OFFSETS_BASE_SIZE;
        if (nlen > w->cap) {
            w->cap = nlen > 2 * w->cap ? nlen : 2 * w->cap;
            w->off = (uint32_t *)realloc(w->off, (size_t)w->cap * sizeof(uint32_t));
        }
        memset(w->off + w->len, 0, (size_t)(nlen - w->len) * sizeof(uint32_t)); /* append(zeros...) */
        w->len = nlen;
    }
}

/* wfa_wavefront.go:85-104 Set: overwrite (last write wins), widen Lo/Hi.  Returns 1 if the
   slot was empty before (oracle-only accounting). */
static inline int wf_set(wf_t *w, int k, uint32_t offset, uint32_t tag) {
    int i = k2i(k);
    wf_grow(w, i);
    int fresh = (w->off[i] == 0);
    w->off[i] = (offset << WFAO_TYPE_BITS) | tag;
    if (k < w->lo) w->lo = k;
    if (k > w->hi) w->hi = k;
    return fresh;
}

/* wfa_wavefront.go:131-150 Increase */
static inline void wf_increase(wf_t *w, int k, uint32_t delta) {
    int i = k2i(k);
    wf_grow(w, i);
    w->off[i] += delta << WFAO_TYPE_BITS;
    if (k < w->lo) w->lo = k;
    if (k > w->hi) w->hi = k;
}

/* wfa_wavefront.go:153-159 Get: range check, then raw != 0 */
static inline int wf_get(const wf_t *w, int k, uint32_t *offset, uint32_t *tag) {
    if (k < w->lo || k > w->hi) {
        *offset = 0;
        *tag    = 0;
        return 0;
    }
    uint32_t raw = w->off[k2i(k)];
    *offset      = raw >> WFAO_TYPE_BITS;
    *tag         = raw & WFAO_TYPE_MASK;
    return raw > 0;
}

/* wfa_wavefront.go:163-169 GetRaw */
static inline int wf_get_raw(const wf_t *w, int k, uint32_t *raw) {
    if (k < w->lo || k > w->hi) {
        *raw = 0;
        return 0;
    }
    *raw = w->off[k2i(k)];
    return *raw > 0;
}

/* wfa_wavefront.go:171-183 Delete: zero the slot; shrink an edge only when k is exactly it
   (Hi tested first). */
static inline void wf_delete(wf_t *w, int k) {
    if (k < w->lo || k > w->hi) return;
    w->off[k2i(k)] = 0;
    if (k == w->hi)
        w->hi--;
    else if (k == w->lo)
        w->lo++;
}

/* ------------------------------------------------------------------ Component
 * wfa_component.go:37-187 */
typedef struct {
    wf_t   **wfs;
    uint32_t len;
    wf_pool *pool;
} comp_t;

static void comp_init(comp_t *c, wf_pool *pool) { /* wfa_component.go:46-55,73-78 */
    c->pool = pool;
    c->len  = WAVEFRONTS_BASE_SIZE;
    c->wfs = (wf_t **)calloc(c->len, sizeof(wf_t *));
}

static void comp_reset(comp_t *c) { /* wfa_component.go:57-64: scans every slot */
    for (uint32_t i = 0; i < c->len; i++) {
        if (c->wfs[i]) {
            wf_recycle(c->pool, c->wfs[i]);
            c->wfs[i] = NULL;
        }
    }
}

static void comp_destroy(comp_t *c) {
    comp_reset(c);
    free(c->wfs);
    c->wfs = NULL;
    c->len = 0;
}

static inline int comp_has_score(const comp_t *c, uint32_t s) { /* wfa_component.go:81-86 */
    return s < c->len && c->wfs[s] != NULL;
}

/* wfa_component.go:91-101 KRange: (0,0) when diff > s or the wavefront is missing (R4) */
static inline void comp_krange(const comp_t *c, uint32_t s, uint32_t diff, int *lo, int *hi) {
    *lo = 0;
    *hi = 0;
    if (diff > s) return;
    s -= diff;
    if (s >= c->len || c->wfs[s] == NULL) return;
    *lo = c->wfs[s]->lo;
    *hi = c->wfs[s]->hi;
}

/* wfa_component.go:104-115 Set.  The reference grows by ONE 2048 chunk and would panic with
   an index error if s were still out of range; the oracle reports that as an internal error
   through *panic. */
static inline int comp_set(comp_t *c, uint32_t s, int k, uint32_t offset, uint32_t tag, int *panic) {
    if (s >= c->len) {
        uint32_t nlen = c->len + WAVEFRONTS_BASE_SIZE;
        if (s >= nlen) {
            *panic = 1;
            while (s >= nlen) nlen += WAVEFRONTS_BASE_SIZE; /* keep the oracle itself memory safe */
        }
        c->wfs = (wf_t **)realloc(c->wfs, (size_t)nlen * sizeof(wf_t *));
        memset(c->wfs + c->len, 0, (size_t)(nlen - c->len) * sizeof(wf_t *));
        c->len = nlen;
    }
    if (c->wfs[s] == NULL) c->wfs[s] = wf_new(c->pool);
    return wf_set(c->wfs[s], k, offset, tag);
}

/* wfa_component.go:142-147 Get */
static inline int comp_get(const comp_t *c, uint32_t s, int k, uint32_t *offset, uint32_t *tag) {
    if (s >= c->len || c->wfs[s] == NULL) {
        *offset = 0;
        *tag    = 0;
        return 0;
    }
    return wf_get(c->wfs[s], k, offset, tag);
}

/* wfa_component.go:150-155 GetRaw */
static inline int comp_get_raw(const comp_t *c, uint32_t s, int k, uint32_t *raw) {
    if (s >= c->len || c->wfs[s] == NULL) {
        *raw = 0;
        return 0;
    }
    return wf_get_raw(c->wfs[s], k, raw);
}

/* wfa_component.go:158-167 GetAfterDiff: diff > s guards the uint32 underflow (R1) */
static inline int comp_get_after_diff(const comp_t *c, uint32_t s, uint32_t diff, int k,
                                      uint32_t *offset, uint32_t *tag) {
    if (diff > s) {
        *offset = 0;
        *tag    = 0;
        return 0;
    }
    return comp_get(c, s - diff, k, offset, tag);
}

/* wfa_component.go:182-187 Delete */
static inline void comp_delete(comp_t *c, uint32_t s, int k) {
    if (s >= c->len || c->wfs[s] == NULL) return;
    wf_delete(c->wfs[s], k);
}

/* ------------------------------------------------------------------ Aligner */
struct wfao_aligner {
    wfao_params p;
    comp_t      M, I, D;
    wf_pool     pool;
    wfao_hook   hook;
    void       *hook_ud;
    int        *dist; /* reduce() scratch (wfa.go:543-546 poolDist) */
    size_t      dist_cap;
    int         panic;
    uint64_t    cells[3];
    uint64_t    lcp_bases;
};

wfao_aligner *wfao_new(const wfao_params *p) {
    wfao_aligner *a = (wfao_aligner *)calloc(1, sizeof(*a));
    a->p            = *p;
    comp_init(&a->M, &a->pool);
    comp_init(&a->I, &a->pool);
    comp_init(&a->D, &a->pool);
    return a;
}

void wfao_free(wfao_aligner *a) {
    if (!a) return;
    comp_destroy(&a->M);
    comp_destroy(&a->I);
    comp_destroy(&a->D);
    for (size_t i = 0; i < a->pool.n; i++) wf_free(a->pool.items[i]);
    free(a->pool.items);
    free(a->dist);
    free(a);
}

void wfao_set_hook(wfao_aligner *a, wfao_hook h, void *ud) {
    a->hook    = h;
    a->hook_ud = ud;
}

static inline void call_hook(wfao_aligner *a, int phase, uint32_t s) {
    if (a->hook) a->hook(a->hook_ud, phase, s);
}

static inline uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return umax2(umax2(a, b), c); }
static inline int imin2(int a, int b) { return a < b ? a : b; }
static inline int imax2(int a, int b) { return a > b ? a : b; }

/* ---- wfa.go:143-184 initComponents */
static void init_components(wfao_aligner *a, const uint8_t *q, int n, const uint8_t *t, int m) {
    comp_reset(&a->M);
    comp_reset(&a->I);
    comp_reset(&a->D);

    uint32_t x = a->p.mismatch;
    /* first cell is always consumed as match or mismatch (wfa.go:155-160) */
    if (q[0] == t[0])
        a->cells[0] += comp_set(&a->M, 0, 0, 1, WFAO_MATCH, &a->panic);
    else
        a->cells[0] += comp_set(&a->M, x, 0, 1, WFAO_MISMATCH, &a->panic);

    if (!a->p.global_alignment) { /* wfa.go:163-183 */
        for (int k = 1; k < m; k++) { /* first row */
            if (q[0] == t[k])
                a->cells[0] += comp_set(&a->M, 0, k, (uint32_t)(k + 1), WFAO_MATCH, &a->panic);
            else
                a->cells[0] += comp_set(&a->M, x, k, (uint32_t)(k + 1), WFAO_MISMATCH, &a->panic);
        }
        for (int k = 1; k < n; k++) { /* first column */
            if (q[k] == t[0])
                a->cells[0] += comp_set(&a->M, 0, -k, 1, WFAO_MATCH, &a->panic);
            else
                a->cells[0] += comp_set(&a->M, x, -k, 1, WFAO_MISMATCH, &a->panic);
        }
    }
}

static inline uint64_t load_be64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v); /* binary.BigEndian.Uint64, wfa.go:377,415 (little-endian host) */
}

/* ---- wfa.go:381-458 extend: for every existing k of M[s] with 0 < v < n and h < m the offset
 * grows by the longest common prefix of q[v:], t[h:].  The 8-byte block loop and the byte tail
 * are restated as written (their exit conditions always sum to the full LCP). */
static void extend(wfao_aligner *a, const uint8_t *q, int lenQ, const uint8_t *t, int lenT,
                   uint32_t s, int *out_lo, int *out_hi) {
    wf_t *wf = a->M.wfs[s];
    int   lo = wf->lo, hi = wf->hi;
    for (int k = hi; k >= lo; k--) {
        uint32_t offset, tag;
        if (!wf_get(wf, k, &offset, &tag)) continue;
        int h = (int)offset;
        int v = h - k;
        if (v <= 0 || v >= lenQ || h >= lenT) continue; /* wfa.go:404 */

        int n = 0, N;
        if (v + 8 <= lenQ && h + 8 <= lenT) { /* wfa.go:411-435 */
            N = 0;
            for (;;) {
                uint64_t x = load_be64(q + v) ^ load_be64(t + h);
                n          = x ? (__builtin_clzll(x) >> 3) : 8;
                v += n;
                h += n;
                N += n;
                if (n < 8 || v + 8 >= lenQ || h + 8 >= lenT) break;
            }
            if (N == 0) continue;
            wf_increase(wf, k, (uint32_t)N);
            a->lcp_bases += (uint64_t)N;
            if (!(n == 8 && v < lenQ && h < lenT)) continue;
        }
        N = 0; /* wfa.go:439-454 */
        while (q[v] == t[h]) {
            v++;
            h++;
            N++;
            if (v == lenQ || h == lenT) break;
        }
        if (N == 0) continue;
        wf_increase(wf, k, (uint32_t)N);
        a->lcp_bases += (uint64_t)N;
    }
    *out_lo = lo; /* wfa.go:457: the Lo/Hi read before the loop (Increase never widens them here) */
    *out_hi = hi;
}

/* ---- wfa.go:461-540 reduce (wf-adaptive) */
static void reduce(wfao_aligner *a, int lenQ, int lenT, uint32_t s) {
    wf_t *wf = a->M.wfs[s];
    int   lo = wf->lo, hi = wf->hi;
    size_t need = (size_t)(hi - lo + 1);
    if (need > a->dist_cap) {
        a->dist_cap = need * 2 + 128;
        a->dist     = (int *)realloc(a->dist, a->dist_cap * sizeof(int));
    }
    int *ds      = a->dist;
    int  nds     = 0;
    int  minDist = INT_MAX;
    for (int k = lo; k <= hi; k++) { /* wfa.go:474-494 */
        uint32_t offset, tag;
        if (!wf_get(wf, k, &offset, &tag)) {
            ds[nds++] = -1;
            continue;
        }
        int h = (int)offset;
        int v = h - k;
        if (v < 0 || v >= lenQ || h >= lenT) {
            ds[nds++] = -1;
            continue;
        }
        int d     = imax2(lenT - h, lenQ - v);
        ds[nds++] = d;
        if (d < minDist) minDist = d;
    }

    int _lo = lo, _hi = hi;
    int maxDistDiff = (int)a->p.max_dist_diff;
    int updateLo = 1, found = 0;
    for (int i = 0; i < nds; i++) { /* wfa.go:503-516 */
        int d = ds[i];
        if (d < 0) continue;
        if (d - minDist > maxDistDiff) {
            found = 1;
            if (updateLo) _lo = lo + i + 1;
            ds[i] = -1;
        } else {
            updateLo = 0;
        }
    }
    if (found) { /* wfa.go:517-524 */
        for (int i = nds - 1; i >= 0; i--) {
            if (ds[i] >= 0) {
                _hi = lo + i;
                break;
            }
        }
    }
    for (int k = lo; k < _lo; k++) { /* wfa.go:526-530 */
        wf_delete(wf, k);
        comp_delete(&a->I, s, k);
        comp_delete(&a->D, s, k);
    }
    for (int k = _hi + 1; k <= hi; k++) { /* wfa.go:531-535 */
        wf_delete(wf, k);
        comp_delete(&a->I, s, k);
        comp_delete(&a->D, s, k);
    }
    wf->lo = _lo; /* wfa.go:537 */
    wf->hi = _hi;
}

/* ---- wfa.go:549-700 next */
static void next(wfao_aligner *a, int lenQ, int lenT, uint32_t s) {
    comp_t *M = &a->M, *I = &a->I, *D = &a->D;
    uint32_t x = a->p.mismatch, oe = a->p.gap_open + a->p.gap_ext, e = a->p.gap_ext;

    int loX, hiX, loO, hiO, loI, hiI, loD, hiD;
    comp_krange(M, s, x, &loX, &hiX);
    comp_krange(M, s, oe, &loO, &hiO);
    comp_krange(I, s, e, &loI, &hiI);
    comp_krange(D, s, e, &loD, &hiD);

    int hi = imin2(lenT - 1, imax2(imax2(hiX, hiO), imax2(hiI, hiD)) + 1); /* wfa.go:562 */
    int lo = imax2(-(lenQ - 1), imin2(imin2(loX, loO), imin2(loI, loD)) - 1); /* wfa.go:563 */

    for (int k = lo; k <= hi; k++) {
        uint32_t v1, v2, tg, Isk, Dsk, Msk;
        int      fromM, fromI, fromD;
        int      updatedI = 0, updatedD = 0;
        uint32_t typeI = 0, typeD = 0, typeM = 0;

        /* insertion: wfa.go:579-609 (reject when value > lenT, not >=) */
        fromM = comp_get_after_diff(M, s, oe, k - 1, &v1, &tg);
        fromI = comp_get_after_diff(I, s, e, k - 1, &v2, &tg);
        if (fromM && (int)v1 > lenT) {
            fromM = 0;
            v1    = 0;
        }
        if (fromI && (int)v2 > lenT) {
            fromI = 0;
            v2    = 0;
        }
        Isk = umax2(v1, v2) + 1;
        if (fromM || fromI) {
            if (fromM && fromI)
                typeI = (v1 >= v2) ? WFAO_INS_OPEN : WFAO_INS_EXT;
            else if (fromM)
                typeI = WFAO_INS_OPEN;
            else
                typeI = WFAO_INS_EXT;
            updatedI = 1;
            a->cells[1] += comp_set(I, s, k, Isk, typeI, &a->panic);
        } else {
            Isk = 0;
        }

        /* deletion: wfa.go:614-645 (reject when value - k > lenQ) */
        fromM = comp_get_after_diff(M, s, oe, k + 1, &v1, &tg);
        fromD = comp_get_after_diff(D, s, e, k + 1, &v2, &tg);
        if (fromM && (int)v1 - k > lenQ) {
            fromM = 0;
            v1    = 0;
        }
        if (fromD && (int)v2 - k > lenQ) {
            fromD = 0;
            v2    = 0;
        }
        Dsk = umax2(v1, v2);
        if (fromM || fromD) {
            if (fromM && fromD)
                typeD = (v1 >= v2) ? WFAO_DEL_OPEN : WFAO_DEL_EXT;
            else if (fromM)
                typeD = WFAO_DEL_OPEN;
            else
                typeD = WFAO_DEL_EXT;
            updatedD = 1;
            a->cells[2] += comp_set(D, s, k, Dsk, typeD, &a->panic);
        } else {
            Dsk = 0;
        }

        /* mismatch: wfa.go:650-698 */
        fromM = comp_get_after_diff(M, s, x, k, &v1, &tg);
        if (fromM && ((int)v1 > lenT || (int)v1 - k > lenQ)) {
            fromM = 0;
            v1    = 0;
        }
        Msk = umax3(Isk, Dsk, v1 + 1);
        if (updatedI || updatedD || fromM) {
            if (updatedI && updatedD && fromM) {
                if (Msk == v1 + 1)
                    typeM = WFAO_MISMATCH;
                else if (Msk == Isk)
                    typeM = typeI;
                else
                    typeM = typeD;
            } else if (updatedI) {
                if (updatedD) {
                    typeM = (Msk == Isk) ? typeI : typeD;
                } else if (fromM) {
                    typeM = (Msk == v1 + 1) ? WFAO_MISMATCH : typeI;
                } else {
                    typeM = typeI;
                }
            } else if (updatedD) {
                if (fromM)
                    typeM = (Msk == v1 + 1) ? WFAO_MISMATCH : typeD;
                else
                    typeM = typeD;
            } else {
                typeM = WFAO_MISMATCH;
            }
            a->cells[0] += comp_set(M, s, k, Msk, typeM, &a->panic);
        }
    }
}

/* ---- wfa.go:270-375 backtraceStartPosistion (semi-global) */
static void backtrace_start_position(wfao_aligner *a, int n, int m, uint32_t s,
                                     uint32_t *out_s, int *out_k) {
    comp_t  *M    = &a->M;
    uint32_t minS = s;
    int      Ak   = m - n;
    int      lastK = Ak;

    for (uint32_t _s = s;; _s--) {
        if (!comp_has_score(M, _s)) {
            if (_s == 0) break;
            continue;
        }
        int lo, hi;
        comp_krange(M, _s, 0, &lo, &hi);

        int hit = 0;
        int k   = Ak;
        for (;;) { /* downwards from Ak: wfa.go:301-326 */
            if (k < lo) break;
            uint32_t offset, tag;
            if (!comp_get_after_diff(M, _s, 0, k, &offset, &tag)) {
                k--;
                continue;
            }
            int h = (int)offset, v = h - k;
            if (v <= 0 || v > n || h > m) break;
            if ((v == n && h >= n) || (h == m && v >= m)) {
                hit = 1;
                break;
            }
            k--;
        }
        if (hit && _s <= minS) {
            lastK = k;
            minS  = _s;
        }

        hit = 0;
        k   = Ak + 1;
        for (;;) { /* upwards from Ak+1: wfa.go:336-361 */
            if (k > hi) break;
            uint32_t offset, tag;
            if (!comp_get_after_diff(M, _s, 0, k, &offset, &tag)) {
                k++;
                continue;
            }
            int h = (int)offset, v = h - k;
            if (v <= 0 || v > n || h > m) break;
            if ((v == n && h >= n) || (h == m && v >= m)) {
                hit = 1;
                break;
            }
            k++;
        }
        if (hit && _s <= minS) {
            lastK = k;
            minS  = _s;
        }
        if (_s == 0) break;
    }
    *out_s = minS;
    *out_k = lastK;
}

/* ---- AlignmentResult helpers: wfa_cigar.go:118-124 AddN */
static void res_add(wfao_result *r, char op, uint32_t n) {
    if (r->n_ops == r->cap_ops) {
        r->cap_ops = r->cap_ops ? r->cap_ops * 2 : 1024;
        r->ops     = (uint64_t *)realloc(r->ops, r->cap_ops * sizeof(uint64_t));
    }
    r->ops[r->n_ops++] = ((uint64_t)(uint8_t)op << 32) | (uint64_t)n;
}

/* ---- wfa_cigar.go:136-214 process: reverse, merge equal neighbours, stats over the span
 * first-M .. last-M (begin/end default to 0 when there is no M op). */
static int res_process(wfao_result *r) {
    if (r->n_ops == 0) return -1; /* the reference would panic on (*s)[0] */
    uint64_t *s = r->ops;
    size_t    L = r->n_ops;
    for (size_t i = 0, j = L - 1; i < j; i++, j--) {
        uint64_t tmp = s[i];
        s[i]         = s[j];
        s[j]         = tmp;
    }
    size_t   j     = 0;
    uint64_t opPre = s[0];
    for (size_t i = 1; i < L; i++) {
        uint64_t op = s[i];
        if ((op >> 32) == (opPre >> 32)) {
            opPre += op & 0xFFFFFFFFull;
            s[j] = opPre;
            continue;
        }
        j++;
        if (i != j) s[j] = s[i];
        opPre = op;
    }
    r->n_ops = L = j + 1;

    size_t begin = 0, end = 0;
    for (size_t i = 0; i < L; i++) {
        if ((s[i] >> 32) == (uint64_t)'M') {
            begin = i;
            break;
        }
    }
    for (size_t i = L; i-- > 0;) {
        if ((s[i] >> 32) == (uint64_t)'M') {
            end = i;
            break;
        }
    }
    uint32_t alen = 0, matches = 0, gaps = 0, regions = 0;
    for (size_t i = begin; i <= end; i++) {
        uint32_t cnt = (uint32_t)(s[i] & 0xFFFFFFFFull);
        alen += cnt;
        uint64_t o = s[i] >> 32;
        if (o == (uint64_t)'M') {
            matches += cnt;
        } else if (o == (uint64_t)'I' || o == (uint64_t)'D') {
            gaps += cnt;
            regions++;
        }
    }
    r->align_len   = alen;
    r->matches     = matches;
    r->gaps        = gaps;
    r->gap_regions = regions;
    return 0;
}

/* ---- wfa.go:703-983 backTrace.  Source lookups use plain Get (no bounds rejection), exactly
 * as written; fromItself / offset0 keep their value across iterations like the Go locals do. */
static int back_trace(wfao_aligner *a, int lenQ, int lenT, uint32_t s, int Ak, wfao_result *cg) {
    int      semiGlobal = !a->p.global_alignment;
    comp_t  *M = &a->M, *I = &a->I, *D = &a->D, *M0 = NULL;
    uint32_t px = a->p.mismatch, po = a->p.gap_open, pe = a->p.gap_ext;

    cg->score = s;

    int      k = Ak, h, v, h0;
    uint32_t offset = 0, wfaType, tg;
    int      qBegin = 0, tBegin = 0;
    uint32_t v1, v2, Isk = 0, Dsk = 0, offset0 = 0;
    int      fromMI, fromMD, fromItself = 0, fromI, fromD, fromM;
    uint32_t sMismatch, sGapOpen, sGapExt;
    int      previousFromM = 1, nMatches, firstMatch = 1;

    comp_get_raw(M, s, k, &offset); /* wfa.go:738 */
    wfaType = offset & WFAO_TYPE_MASK;
    h       = (int)(offset >> WFAO_TYPE_BITS);
    v       = h - k;

    if (h < lenT) /* wfa.go:746-750: target flank is I, query flank is H */
        res_add(cg, WFA_OPS[WFAO_INS_OPEN], (uint32_t)lenT - (uint32_t)h);
    else if (v < lenQ)
        res_add(cg, 'H', (uint32_t)lenQ - (uint32_t)v);

    while (v > 0 && h > 0) { /* wfa.go:753 */
        sMismatch = s - px;
        sGapOpen  = s - po - pe;
        sGapExt   = s - pe;

        fromMI = 0;
        fromMD = 0;
        if (wfaType == WFAO_INS_EXT) { /* wfa.go:767-777 */
            fromM = comp_get(M, sGapOpen, k - 1, &v1, &tg);
            fromI = comp_get(I, sGapExt, k - 1, &v2, &tg);
            if (fromM || fromI) {
                fromMI  = 1;
                offset0 = umax2(v1, v2) + 1;
            } else {
                offset0 = 0;
            }
            M0 = I;
        } else if (wfaType == WFAO_DEL_EXT) { /* wfa.go:778-788 */
            fromM = comp_get(M, sGapOpen, k + 1, &v1, &tg);
            fromD = comp_get(D, sGapExt, k + 1, &v2, &tg);
            if (fromM || fromD) {
                fromMD  = 1;
                offset0 = umax2(v1, v2);
            } else {
                offset0 = 0;
            }
            M0 = D;
        } else { /* wfa.go:789-817 */
            fromM = comp_get(M, sGapOpen, k - 1, &v1, &tg);
            fromI = comp_get(I, sGapExt, k - 1, &v2, &tg);
            if (fromM || fromI) {
                fromMI = 1;
                Isk    = umax2(v1, v2) + 1;
            } else {
                Isk = 0;
            }
            fromM = comp_get(M, sGapOpen, k + 1, &v1, &tg);
            fromD = comp_get(D, sGapExt, k + 1, &v2, &tg);
            if (fromM || fromD) {
                fromMD = 1;
                Dsk    = umax2(v1, v2);
            } else {
                Dsk = 0;
            }
            fromM = comp_get(M, sMismatch, k, &v1, &tg);
            if (fromMI || fromMD || fromM) {
                offset0    = umax3(Isk, Dsk, v1 + 1);
                fromItself = 0;
            } else {
                fromItself = 1;
            }
            M0 = M;
        }
        if (fromItself) break;   /* wfa.go:818-821 */
        if (offset0 == 0) break; /* wfa.go:822-825 */

        h0 = (int)offset0;

        if (previousFromM) { /* wfa.go:833-869 */
            nMatches = h - h0;
            if (nMatches > 0) {
                if (firstMatch) {
                    firstMatch = 0;
                    cg->tend   = h;
                    cg->qend   = v;
                }
                res_add(cg, WFA_OPS[WFAO_MATCH], (uint32_t)nMatches);
            }
            offset = offset0;
            h      = (int)offset;
            v      = h - k;
            if (wfaType == WFAO_MATCH) {
                tBegin = h;
                qBegin = v;
            } else if (nMatches > 0) {
                tBegin = h + 1;
                qBegin = v + 1;
            }
            if (h <= 0 || v <= 0) break;
        }

        res_add(cg, WFA_OPS[wfaType], 1); /* wfa.go:872-873 */

        if (semiGlobal && (h == 1 || v == 1)) break; /* wfa.go:876-879 */

        previousFromM = 1; /* wfa.go:885-909 */
        int stop      = 0;
        switch (wfaType) {
        case WFAO_MISMATCH:
            s = sMismatch;
            h--;
            break;
        case WFAO_INS_OPEN:
            s = sGapOpen;
            k--;
            h--;
            break;
        case WFAO_INS_EXT:
            s = sGapExt;
            k--;
            h--;
            previousFromM = 0;
            break;
        case WFAO_DEL_OPEN:
            s = sGapOpen;
            k++;
            break;
        case WFAO_DEL_EXT:
            s = sGapExt;
            k++;
            previousFromM = 0;
            break;
        default:
            stop = 1;
            break;
        }
        if (stop) break; /* break LOOP, wfa.go:906-908 */
        v = h - k;

        if (!comp_get_raw(M0, s, k, &offset)) break; /* wfa.go:915-919 */
        wfaType = offset & WFAO_TYPE_MASK;
    }

    if (h > 0 && v > 0) { /* wfa.go:930-968 */
        nMatches = imin2(h, v) - 1;
        if (nMatches > 0) {
            if (firstMatch) {
                firstMatch = 0;
                cg->tend   = h;
                cg->qend   = v;
            }
            res_add(cg, WFA_OPS[WFAO_MATCH], (uint32_t)nMatches);
            h -= nMatches;
            v -= nMatches;
            if (wfaType == WFAO_MATCH) {
                tBegin = h;
                qBegin = v;
            } else if (nMatches > 0) {
                tBegin = h + 1;
                qBegin = v + 1;
            }
        } else if (wfaType == WFAO_MATCH) {
            tBegin = h;
            qBegin = v;
            if (firstMatch) {
                firstMatch = 0;
                cg->tend   = h;
                cg->qend   = v;
            }
        }
        res_add(cg, WFA_OPS[wfaType], 1);
    }
    if (v > 1) res_add(cg, 'H', (uint32_t)(v - 1));                    /* wfa.go:970-972 */
    if (h > 1) res_add(cg, WFA_OPS[WFAO_INS_OPEN], (uint32_t)(h - 1)); /* wfa.go:974-976 */

    cg->tbegin = tBegin; /* wfa.go:979 */
    cg->qbegin = qBegin;
    return res_process(cg); /* wfa.go:981 */
}

/* ---- wfa.go:196-268 Align / AlignPointers */
int wfao_align(wfao_aligner *a, const uint8_t *q, size_t nq, const uint8_t *t, size_t mt,
               wfao_result *res) {
    if (nq == 0 || mt == 0) return WFAO_ERR_EMPTY;                              /* wfa.go:204-206 */
    if (nq > WFAO_MAX_SEQ_LEN || mt > WFAO_MAX_SEQ_LEN) return WFAO_ERR_TOO_LONG; /* wfa.go:207-209 */
    int n = (int)nq, m = (int)mt;

    /* NewAlignmentResult+reset (wfa_cigar.go:69-89).  TEnd/QEnd/TBegin/QBegin are NOT reset by
       the reference (stale pool values when the alignment has no match run); the oracle defines
       them as those of a fresh object: 0. */
    res->n_ops = 0;
    res->score = 0;
    res->tbegin = res->tend = res->qbegin = res->qend = 0;
    res->align_len = res->matches = res->gaps = res->gap_regions = 0;

    a->panic     = 0;
    a->cells[0] = a->cells[1] = a->cells[2] = 0;
    a->lcp_bases = 0;

    init_components(a, q, n, t, m);
    call_hook(a, WFAO_PH_INIT, 0);

    int      Ak      = m - n;
    uint32_t Aoffset = (uint32_t)m;
    uint32_t s       = 0;
    int      lo = 0, hi = 0;
    int      do_reduce = a->p.adaptive != 0;
    int      minWFLen  = do_reduce ? (int)a->p.min_wf_len : 0;

    for (;;) { /* wfa.go:228-251 */
        if (comp_has_score(&a->M, s)) {
            extend(a, q, n, t, m, s, &lo, &hi);
            call_hook(a, WFAO_PH_EXTEND, s);
            uint32_t offset, tag;
            comp_get_after_diff(&a->M, s, 0, Ak, &offset, &tag);
            if (offset >= Aoffset) break;
            if (do_reduce && hi - lo + 1 >= minWFLen) {
                reduce(a, n, m, s);
                call_hook(a, WFAO_PH_REDUCE, s);
            }
        }
        s++;
        next(a, n, m, s);
        call_hook(a, WFAO_PH_NEXT, s);
        if (a->panic) return WFAO_ERR_INTERNAL;
    }
    res->n_scores = s;

    uint32_t minS  = s;
    int      lastK = Ak;
    if (!a->p.global_alignment) backtrace_start_position(a, n, m, s, &minS, &lastK); /* wfa.go:258-261 */

    int rc = back_trace(a, n, m, minS, lastK, res);
    res->cells[0]  = a->cells[0];
    res->cells[1]  = a->cells[1];
    res->cells[2]  = a->cells[2];
    res->lcp_bases = a->lcp_bases;
    if (rc != 0 || a->panic) return WFAO_ERR_INTERNAL;
    return WFAO_OK;
}

void wfao_result_release(wfao_result *res) {
    free(res->ops);
    res->ops     = NULL;
    res->n_ops   = 0;
    res->cap_ops = 0;
}

/* ---- wfa_cigar.go:217-233 trimOps + :236-255 CIGAR */
size_t wfao_cigar(const wfao_result *res, int only_aligned, char *buf, size_t cap) {
    size_t b = 0, e = res->n_ops; /* [b, e) */
    if (only_aligned) {
        long start = -1, end = -1;
        for (size_t i = 0; i < res->n_ops; i++)
            if ((res->ops[i] >> 32) == (uint64_t)'M') {
                start = (long)i;
                break;
            }
        for (size_t i = res->n_ops; i-- > 0;)
            if ((res->ops[i] >> 32) == (uint64_t)'M') {
                end = (long)i;
                break;
            }
        if (start < 0) { /* the reference would panic slicing [-1:0] */
            if (cap) buf[0] = 0;
            return 0;
        }
        b = (size_t)start;
        e = (size_t)end + 1;
    }
    size_t need = 0;
    for (size_t i = b; i < e; i++) {
        char tmp[24];
        int  l = snprintf(tmp, sizeof tmp, "%u%c", (unsigned)(res->ops[i] & 0xFFFFFFFFull),
                          (char)(res->ops[i] >> 32));
        if (need + (size_t)l < cap) memcpy(buf + need, tmp, (size_t)l);
        need += (size_t)l;
    }
    if (cap) buf[need < cap ? need : cap - 1] = 0;
    return need;
}

/* ---- state inspection */
static const comp_t *pick(const wfao_aligner *a, int comp) {
    return comp == 0 ? &a->M : (comp == 1 ? &a->I : &a->D);
}

uint32_t wfao_num_scores(const wfao_aligner *a, int comp) { return pick(a, comp)->len; }

int wfao_get_wavefront(const wfao_aligner *a, int comp, uint32_t s, int32_t *lo, int32_t *hi,
                       uint32_t *raw, size_t cap) {
    const comp_t *c = pick(a, comp);
    if (s >= c->len || c->wfs[s] == NULL) return 0;
    const wf_t *w = c->wfs[s];
    *lo           = w->lo;
    *hi           = w->hi;
    if (raw && w->hi >= w->lo && cap >= (size_t)(w->hi - w->lo + 1)) {
        for (int k = w->lo; k <= w->hi; k++) {
            uint32_t r;
            wf_get_raw(w, k, &r);
            raw[k - w->lo] = r;
        }
    }
    return 1;
}

/* ---- batch driver (mirrors wfa-go/wfa-go.go:96-141: one aligner per worker, reused) */
typedef struct {
    const wfao_params *p;
    const uint8_t     *blob;
    const uint64_t    *q_off, *t_off;
    const uint32_t    *q_len, *t_len;
    uint64_t           begin, end;
    int32_t           *status;
    uint32_t          *score;
    int32_t           *tbegin, *tend, *qbegin, *qend;
    uint32_t          *align_len, *matches, *gaps, *gap_regions;
    uint64_t          *cells;
    int                want_ops;
    uint64_t          *ops; /* per-thread growing buffer */
    size_t             n_ops, cap_ops;
    uint64_t          *ops_off_local; /* offset inside this thread's buffer */
    uint32_t          *ops_len;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job    *j = (batch_job *)arg;
    wfao_aligner *a = wfao_new(j->p);
    wfao_result   r;
    memset(&r, 0, sizeof r);
    for (uint64_t i = j->begin; i < j->end; i++) {
        int st = wfao_align(a, j->blob + j->q_off[i], j->q_len[i], j->blob + j->t_off[i],
                            j->t_len[i], &r);
        if (j->status) j->status[i] = st;
        if (st != WFAO_OK) {
            r.n_ops = 0;
            r.score = 0;
            r.tbegin = r.tend = r.qbegin = r.qend = 0;
            r.align_len = r.matches = r.gaps = r.gap_regions = 0;
            r.cells[0] = r.cells[1] = r.cells[2] = 0;
        }
        if (j->score) j->score[i] = r.score;
        if (j->tbegin) j->tbegin[i] = r.tbegin;
        if (j->tend) j->tend[i] = r.tend;
        if (j->qbegin) j->qbegin[i] = r.qbegin;
        if (j->qend) j->qend[i] = r.qend;
        if (j->align_len) j->align_len[i] = r.align_len;
        if (j->matches) j->matches[i] = r.matches;
        if (j->gaps) j->gaps[i] = r.gaps;
        if (j->gap_regions) j->gap_regions[i] = r.gap_regions;
        if (j->cells) {
            j->cells[3 * i + 0] = r.cells[0];
            j->cells[3 * i + 1] = r.cells[1];
            j->cells[3 * i + 2] = r.cells[2];
        }
        if (j->want_ops) {
            if (j->n_ops + r.n_ops > j->cap_ops) {
                j->cap_ops = (j->n_ops + r.n_ops) * 2 + 1024;
                j->ops     = (uint64_t *)realloc(j->ops, j->cap_ops * sizeof(uint64_t));
            }
            memcpy(j->ops + j->n_ops, r.ops, r.n_ops * sizeof(uint64_t));
            j->ops_off_local[i] = j->n_ops;
            j->ops_len[i]       = (uint32_t)r.n_ops;
            j->n_ops += r.n_ops;
        }
    }
    wfao_result_release(&r);
    wfao_free(a);
    return NULL;
}

int wfao_align_batch(const wfao_params *p, const uint8_t *blob, const uint64_t *q_off,
                     const uint32_t *q_len, const uint64_t *t_off, const uint32_t *t_len,
                     uint64_t n_pairs, int n_threads, int32_t *status, uint32_t *score,
                     int32_t *tbegin, int32_t *tend, int32_t *qbegin, int32_t *qend,
                     uint32_t *align_len, uint32_t *matches, uint32_t *gaps, uint32_t *gap_regions,
                     uint64_t *cells, uint64_t **ops_out, uint64_t *ops_off, uint32_t *ops_len) {
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_pairs && n_pairs > 0) n_threads = (int)n_pairs;
    batch_job *jobs = (batch_job *)calloc((size_t)n_threads, sizeof(batch_job));
    pthread_t *th   = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    uint64_t   per  = n_threads ? (n_pairs + (uint64_t)n_threads - 1) / (uint64_t)n_threads : 0;
    for (int i = 0; i < n_threads; i++) {
        batch_job *j = &jobs[i];
        j->p = p, j->blob = blob, j->q_off = q_off, j->q_len = q_len, j->t_off = t_off, j->t_len = t_len;
        j->begin = (uint64_t)i * per;
        j->end   = j->begin + per > n_pairs ? n_pairs : j->begin + per;
        if (j->begin > n_pairs) j->begin = n_pairs;
        j->status = status, j->score = score, j->tbegin = tbegin, j->tend = tend, j->qbegin = qbegin;
        j->qend = qend, j->align_len = align_len, j->matches = matches, j->gaps = gaps;
        j->gap_regions = gap_regions, j->cells = cells;
        j->want_ops      = ops_out != NULL;
        j->ops_off_local = ops_off;
        j->ops_len       = ops_len;
        if (n_threads == 1)
            batch_worker(j);
        else
            pthread_create(&th[i], NULL, batch_worker, j);
    }
    if (n_threads > 1)
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    if (ops_out) {
        size_t total = 0;
        for (int i = 0; i < n_threads; i++) total += jobs[i].n_ops;
        uint64_t *all = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
        size_t    base = 0;
        for (int i = 0; i < n_threads; i++) {
            memcpy(all + base, jobs[i].ops, jobs[i].n_ops * sizeof(uint64_t));
            for (uint64_t k = jobs[i].begin; k < jobs[i].end; k++) ops_off[k] += base;
            base += jobs[i].n_ops;
            free(jobs[i].ops);
        }
        *ops_out = all;
    }
    free(jobs);
    free(th);
    return 0;
}
