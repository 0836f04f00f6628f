/*
 * wfa_oracle.h -- CPU oracle for the wavefront-alignment hot path of shenwei356/wfa v0.4.0.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (Align -> extend/next/reduce -> backtrace -> process).  It is the
 * checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under wfa_amd/ links or calls it.
 *
 * Parity status: the reference is Go and no Go toolchain exists in the build
 * container or on the GPU box, so the reference itself cannot be executed.  The
 * oracle is pinned against every known answer the reference publishes for this
 * path (README.md KA1..KA5 incl. the KA1 M-table, wfa_test.go:83,94) -- see
 * tests/test_oracle_golden.py.  Beyond those vectors parity is oracle-vs-HIP
 * self-consistency.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).
 */
#ifndef WFA_ORACLE_H
#define WFA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* backtrace tags: wfa_backtrace_types.go:24-35 */
enum {
    WFAO_INS_OPEN = 1,
    WFAO_INS_EXT  = 2,
    WFAO_DEL_OPEN = 3,
    WFAO_DEL_EXT  = 4,
    WFAO_MISMATCH = 5,
    WFAO_MATCH    = 6
};

#define WFAO_TYPE_BITS 3u
#define WFAO_TYPE_MASK 7u
/* wfa.go:190  MaxSeqLen = 1<<(32-3) - 1 */
#define WFAO_MAX_SEQ_LEN ((1u << (32 - WFAO_TYPE_BITS)) - 1u)

/* status codes of wfao_align (wfa.go:186-193,204-209) */
enum {
    WFAO_OK           = 0,
    WFAO_ERR_EMPTY    = 1, /* ErrEmptySeq   */
    WFAO_ERR_TOO_LONG = 2, /* ErrSeqTooLong */
    WFAO_ERR_INTERNAL = 9  /* a state in which the reference would panic */
};

/* wfa.go:32-36 Penalties, :46-50 AdaptiveReductionOption, :64-66 Options */
typedef struct {
    uint32_t mismatch, gap_open, gap_ext;
    int32_t  global_alignment; /* Options.GlobalAlignment */
    int32_t  adaptive;         /* 0: algn.ad == nil */
    uint32_t min_wf_len, max_dist_diff, cutoff_step;
} wfao_params;

/* wfa_cigar.go:30-48 AlignmentResult (ops are op<<32|n, wfa_cigar.go:118-124) */
typedef struct {
    uint64_t *ops;
    size_t    n_ops, cap_ops;
    uint32_t  score;
    int32_t   tbegin, tend, qbegin, qend;
    uint32_t  align_len, matches, gaps, gap_regions;
    /* oracle-only accounting (not in the reference): cells stored per component
       (a Set that turns a zero slot non-zero), used for the algorithmic-bytes model */
    uint64_t  cells[3]; /* M, I, D */
    uint64_t  lcp_bases; /* bases matched by extend */
    uint32_t  n_scores;  /* final loop score s (before semi-global end search) */
} wfao_result;

typedef struct wfao_aligner wfao_aligner;

/* phases reported to the step hook */
enum { WFAO_PH_INIT = 0, WFAO_PH_NEXT = 1, WFAO_PH_EXTEND = 2, WFAO_PH_REDUCE = 3 };
typedef void (*wfao_hook)(void *ud, int phase, uint32_t s);

wfao_aligner *wfao_new(const wfao_params *p);        /* wfa.go:120 New + :134 AdaptiveReduction */
void          wfao_free(wfao_aligner *a);            /* wfa.go:102 RecycleAligner */
void          wfao_set_hook(wfao_aligner *a, wfao_hook h, void *ud);

/* wfa.go:196-268 Align/AlignPointers.  res must be zero-initialised before first use;
   it may be reused across calls (ops buffer is recycled).  Returns a WFAO_* status. */
int  wfao_align(wfao_aligner *a, const uint8_t *q, size_t n, const uint8_t *t, size_t m,
                wfao_result *res);
void wfao_result_release(wfao_result *res);

/* wfa_cigar.go:236-255 CIGAR(onlyAlignedRegion); writes a NUL-terminated string, returns
   the length it needs (excluding NUL).  only_aligned uses trimOps (wfa_cigar.go:217-233). */
size_t wfao_cigar(const wfao_result *res, int only_aligned, char *buf, size_t cap);

/* state inspection for golden per-step dumps: comp 0=M 1=I 2=D.
   Returns 0 if no wavefront exists for (comp,s); else 1 and fills lo/hi (WaveFront.Lo/Hi).
   If raw != NULL and cap >= hi-lo+1 the raw words (offset<<3|tag, 0 = absent) for k=lo..hi
   are copied (wfa_wavefront.go:163-169 GetRaw). */
int  wfao_get_wavefront(const wfao_aligner *a, int comp, uint32_t s, int32_t *lo, int32_t *hi,
                        uint32_t *raw, size_t cap);
uint32_t wfao_num_scores(const wfao_aligner *a, int comp);

/* batch helper used by the cpu_baseline leg and by tests: aligns n_pairs pairs laid out like
   the C-ABI input (blob + offsets + lengths) with n_threads worker threads, one aligner per
   thread reused across its contiguous shard (mirrors wfa-go/wfa-go.go:96-141).
   out_* arrays may be NULL.  ops are appended per pair to a malloc'd array returned through
   ops_out/ops_off/ops_len when ops_out != NULL (caller frees *ops_out).  Returns 0. */
int wfao_align_batch(const wfao_params *p, const uint8_t *blob,
                     const uint64_t *q_off, const uint32_t *q_len,
                     const uint64_t *t_off, const uint32_t *t_len, uint64_t n_pairs,
                     int n_threads,
                     int32_t *status, uint32_t *score,
                     int32_t *tbegin, int32_t *tend, int32_t *qbegin, int32_t *qend,
                     uint32_t *align_len, uint32_t *matches, uint32_t *gaps, uint32_t *gap_regions,
                     uint64_t *cells /* [n_pairs*3] or NULL */,
                     uint64_t **ops_out, uint64_t *ops_off, uint32_t *ops_len);

#ifdef __cplusplus
}
#endif
#endif
