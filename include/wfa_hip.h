/*
 * wfa_hip.h -- C-ABI of libwfahip.so: the MI355X (gfx950) wavefront-alignment hot path behind
 * the shenwei356/wfa Aligner API.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, and is what a cgo binding of
 * the reference's Go package would call in place of its pure-Go path (INTEGRATION.md shows the
 * binding).  Citations are file:line into the reference checkout.
 *
 *   reference interface (Go)                                    replaced by
 *   ----------------------------------------------------------  ---------------------------------
 *   wfa.New(p *Penalties, opt *Options)            wfa.go:120    wfahip_create + wfahip_params
 *   (*Aligner).AdaptiveReduction(ad)               wfa.go:134    wfahip_params.adaptive/min_wf_len/...
 *   (*Aligner).Align / AlignPointers(q, t)         wfa.go:196,201  wfahip_align_batch (n_pairs = 1 or N)
 *   ErrEmptySeq / ErrSeqTooLong / MaxSeqLen        wfa.go:186-193  per-pair status WFAHIP_PAIR_EMPTY / _TOO_LONG
 *   AlignmentResult{Ops,Score,TBegin,...}          wfa_cigar.go:30-48  wfahip_results (struct of arrays)
 *   RecycleAlignmentResult                         wfa_cigar.go:92   wfahip_results_free
 *   RecycleAligner                                 wfa.go:102    wfahip_destroy
 *
 * Threading: one wfahip_ctx may be used by one thread at a time; different contexts may be used
 * concurrently (mirrors "one Aligner per goroutine", wfa.go:73-78).
 *
 * First call of a workload: a context allocates its device buffers on demand and keeps them -- wavefront arenas sized
 * for the batch (34 KB per 1 kbp pair: 32 GiB for a million pairs), staging buffers, the page-locked blocks of the
 * host entry.  hipMalloc of tens of GB and hipHostMalloc of hundreds of MB take SECONDS (2.5-4 s measured for the first
 * 1e6 x 1 kbp call of a context against 20-60 ms for every later one), so a service should align one batch of its
 * largest expected shape right after wfahip_create, before it takes traffic; later calls of the same or a smaller shape
 * allocate nothing.  A context also LEARNS per class of batches (length bucket, penalties, wf-adaptive): the arena level
 * long pairs need, the rows per pair and the window width that high-error batches need.  What is learned changes only
 * how fast a call is, never its results (options "learn", and the tests of it, say how).
 *
 * Results are owned by the library: every array of a wfahip_results is handed back through wfahip_results_free, never
 * free()d by the caller (blocks circulate through a cache and are page-locked while they do).
 */
#ifndef WFA_HIP_H
#define WFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFAHIP_VERSION 400 /* 0.4.0 */

/* whole-call return codes (0 = success, negative = failure) */
enum {
    WFAHIP_OK              = 0,
    WFAHIP_ERR_NO_DEVICE   = -1,
    WFAHIP_ERR_BAD_ARG     = -2,
    WFAHIP_ERR_OOM         = -3,
    WFAHIP_ERR_HIP         = -4,
    WFAHIP_ERR_UNSUPPORTED = -5, /* mismatch == 0 or gap_open + gap_ext == 0 (the reference's own backtrace does not
                                    terminate for mismatch == 0 when first bases differ, and next() would read the row it is writing);
                                    non-ACGT input to the packer */
    WFAHIP_ERR_INTERNAL    = -6
};

/* per-pair status (wfahip_results.status) */
enum {
    WFAHIP_PAIR_OK        = 0,
    WFAHIP_PAIR_EMPTY     = 1, /* ErrEmptySeq,   wfa.go:204-206 */
    WFAHIP_PAIR_TOO_LONG  = 2, /* ErrSeqTooLong, wfa.go:207-209 */
    WFAHIP_PAIR_NO_MEMORY = 4  /* wavefront arena could not be grown enough for this pair */
};

/* wfa.go:190 MaxSeqLen */
#define WFAHIP_MAX_SEQ_LEN ((1u << 29) - 1u)

/* Penalties (wfa.go:32-36) + Options (wfa.go:64-66) + AdaptiveReductionOption (wfa.go:46-50) */
typedef struct {
    uint32_t mismatch, gap_open, gap_ext;
    uint8_t  global_alignment; /* Options.GlobalAlignment */
    uint8_t  adaptive;         /* 0: AdaptiveReduction was never called (algn.ad == nil) */
    uint8_t  reserved[2];
    uint32_t min_wf_len, max_dist_diff, cutoff_step; /* cutoff_step is unused by the reference too (wfa.go:49) */
} wfahip_params;

/* AlignmentResult as a struct of arrays, one element per pair (wfa_cigar.go:30-48).
 * ops holds, for pair i, ops_len[i] entries starting at ops[ops_off[i]], each `op<<32 | n`
 * (wfa_cigar.go:118-124), already reversed and merged the way AlignmentResult.process() leaves
 * them (wfa_cigar.go:136-214).  All arrays are malloc'd by the library. */
typedef struct {
    uint64_t  n;
    int32_t  *status;
    uint32_t *score;
    int32_t  *tbegin, *tend, *qbegin, *qend;
    uint32_t *align_len, *matches, *gaps, *gap_regions;
    uint64_t *ops_off;
    uint32_t *ops_len;
    uint64_t *ops;
    uint64_t  n_ops;
} wfahip_results;

/* device-side result record: 16 x u32 per pair, one 64-byte line */
enum {
    WFAHIP_REC_STATUS = 0, WFAHIP_REC_SCORE, WFAHIP_REC_TBEGIN, WFAHIP_REC_TEND, WFAHIP_REC_QBEGIN,
    WFAHIP_REC_QEND, WFAHIP_REC_ALIGN_LEN, WFAHIP_REC_MATCHES, WFAHIP_REC_GAPS, WFAHIP_REC_GAP_REGIONS,
    WFAHIP_REC_OPS_LEN, WFAHIP_REC_OPS_OFF_LO, WFAHIP_REC_OPS_OFF_HI,
    WFAHIP_REC_CELLS_LO, WFAHIP_REC_CELLS_HI, /* non-zero wavefront words stored (M+I+D); 0 unless option "census" is on */
    WFAHIP_REC_N_SCORES,
    WFAHIP_REC_WORDS = 16
};

/* timing / accounting of the most recent wfahip_align_batch* call on a context */
typedef struct {
    double   kernel_ms;        /* sum of the alignment kernels' durations (hipEvent, on their stream) */
    double   total_ms;         /* whole device-side call: first launch -> last kernel done */
    uint32_t n_launches;       /* alignment kernel launches (1 + retries for bigger arenas / byte path) */
    uint32_t n_retried_pairs;  /* pairs that needed a second configuration */
    uint64_t cells_stored;     /* sum over pairs of non-zero wavefront words stored */
    uint64_t ops_written;      /* CIGAR ops written */
    uint64_t arena_bytes;      /* arena footprint allocated */
    double   main_kernel_ms;   /* total duration of the dominant kernel's launches (packed forward kernel when it
                                  ran, else the first generic launch) */
    uint32_t n_main_launches;  /* launches of that kernel (one per chunk) */
    uint32_t n_packed_pairs;   /* pairs finished by the sub-wave forward + backtrace kernels */
    uint32_t main_kernel_kind; /* 0 = wfa_generic_kernel, 1 = wfa_packed_kernel, 2 = wfa_reg_kernel, 3 = wfa_blk_kernel<16>, 4 = wfa_blk_kernel<8>,
                                  5 = wfa_blk_kernel<64>, 6 = wfa_blk_kernel<8, 8, false, 4> (short reads), 7 = wfa_team_kernel,
                                  8 = wfa_duo_kernel (8 or 16 lanes per pair), 9 = wfa_blk_kernel<32> (128 diagonals),
                                  10 = wfa_lane_kernel (a lane per pair, short reads), 11 / 12 / 13 = wfa_blk_kernel<16 / 32 / 64, .., LONG>
                                  (sliding sequence windows: reads of any length, 64 / 128 / 256 diagonals), 14 / 15 = wfa_blk_kernel<64, 1, false, 1 / 2, .., LONG>
                                  (a wave per pair, one / two diagonals per lane: batches too small to fill the GPU),
                                  16 = wfa_blk_kernel<64, 1, false, 1, false, false> (wfahip_align_pair: one launch, forward pass and backtrace),
                                  18 = wfa_wide_kernel (round 6: semi-global reads up to 2 047 bases, a workgroup per pair with the rows in 16-bit LDS rings),
                                  17 = wfa_teamc_kernel (wide wavefronts: a team of workgroups per pair, one backtrace word per diagonal) */
    uint32_t ladder_start_level; /* arena level the long-pair ladder of this call started on (0 unless a learned hint applied) */
} wfahip_timing;

typedef struct wfahip_ctx wfahip_ctx;

int         wfahip_version(void);
const char *wfahip_strerror(int code);
/* detail of the most recent WFAHIP_ERR_HIP / WFAHIP_ERR_OOM / WFAHIP_ERR_INTERNAL on a context (the failing runtime call and its
 * message; "" if none): for logs -- the code is what a binding acts on.  Valid until the next call on the context. */
const char *wfahip_last_error(const wfahip_ctx *ctx);
int         wfahip_device_count(void);

/* One context per GPU (one process per GPU in multi-GPU jobs).  device_id < 0 = current device. */
int  wfahip_create(int device_id, wfahip_ctx **out);
void wfahip_destroy(wfahip_ctx *ctx);

/* The cgo entry: host inputs, host outputs, synchronous.  seq_blob holds all sequences; pair i is
 * query seq_blob[q_off[i] .. +q_len[i]) vs target seq_blob[t_off[i] .. +t_len[i]).  Inputs are
 * borrowed for the duration of the call only.  out is filled with malloc'd arrays; release with
 * wfahip_results_free, which keeps the large blocks for the next call (RecycleAlignmentResult,
 * wfa_cigar.go:92: fresh pages cost more than the download).  Replaces Aligner.Align (wfa.go:196).
 * Batches of >= 200 000 pairs whose pairs lie in order in the blob are aligned in four slices while the rest of
 * the blob is still uploading and the results of earlier slices are already downloading.  Environment (diagnostics): WFAHIP_NO_UPLOAD_OVERLAP=1 switches that off,
 * WFAHIP_DL_THREADS=n sets the number of copy-out threads of the result download, WFAHIP_DEBUG_TIMING=1 prints
 * the phases of a call to stderr. */
int  wfahip_align_batch(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *seq_blob,
                        uint64_t blob_bytes, const uint64_t *q_off, const uint32_t *q_len,
                        const uint64_t *t_off, const uint32_t *t_len, uint64_t n_pairs,
                        wfahip_results *out);
void wfahip_results_free(wfahip_results *r);

/* Device-resident variant: every pointer is a device address on the context's GPU (caller-owned).
 * d_rec receives n_pairs records of WFAHIP_REC_WORDS u32; d_ops receives the CIGAR ops
 * (capacity ops_cap entries; record fields OPS_OFF/OPS_LEN index into it).  max_len = an upper
 * bound of every q_len/t_len (0 = let the library compute it).  stream = hipStream_t to launch on
 * (NULL = the context's own stream).  Synchronous: returns when results are complete.
 * If ops_cap is too small the call returns WFAHIP_ERR_OOM and *ops_needed (if non-NULL) holds
 * the required capacity. */
int  wfahip_align_batch_device(wfahip_ctx *ctx, const wfahip_params *p, const void *d_seq_blob,
                               uint64_t blob_bytes, const void *d_q_off, const void *d_q_len,
                               const void *d_t_off, const void *d_t_len, uint64_t n_pairs,
                               uint32_t max_len, void *d_rec, void *d_ops, uint64_t ops_cap,
                               uint64_t *ops_needed, void *stream);

/* Pre-packed input: the sequences arrive 2-bit packed, 16 bases per uint32 (base i of a
 * sequence in bits 2(i%16).. of word i/16, code = (ascii >> 1) & 3: A 0, C 1, T 2, G 3), every sequence starting at a
 * word boundary and followed by one pad word; pair i is packed[q_woff[i] ..] (q_len[i] bases) vs packed[t_woff[i] ..].
 * A quarter of the bytes cross PCIe, and they stay words on the device: a pass that fetches pre-packed pairs -- the first
 * pass of a large batch of 240+ base reads, the long-read instances -- has its fixed-stride slots {n, m, status, -, q words,
 * t words} copied from the uploaded words (wfa_prepack_words_kernel), and a pass that reads bytes (the retry rungs, the
 * lane-per-pair kernel, the long-pair kernels) has exactly its pairs expanded first (wfa_unpack_pairs_kernel: a thousandth
 * of a 1e6 x 1 kbp batch).  Rounds 2-3 expanded everything to bytes and packed it again (option "unpack_all" = 1).  Valid
 * only for pure uppercase ACGT input -- the reference compares raw bytes (wfa.go:408-454), so anything else must use
 * wfahip_align_batch.  wfahip_pack_pairs is the host-side packer (n_threads host threads; returns
 * WFAHIP_ERR_UNSUPPORTED if a byte outside ACGT is found); packed must hold the sum over all sequences of
 * wfahip_packed_words(len) words.  Results are identical to wfahip_align_batch on the unpacked bytes. */
uint64_t wfahip_packed_words(uint32_t len);
int  wfahip_pack_pairs(const uint8_t *seq_blob, const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off,
                       const uint32_t *t_len, uint64_t n_pairs, int n_threads, uint32_t *packed, uint64_t *q_woff,
                       uint64_t *t_woff, uint64_t *n_words);
int  wfahip_align_batch_packed(wfahip_ctx *ctx, const wfahip_params *p, const uint32_t *packed, uint64_t n_words,
                               const uint64_t *q_woff, const uint32_t *q_len, const uint64_t *t_woff,
                               const uint32_t *t_len, uint64_t n_pairs, wfahip_results *out);

/* One pair at a time behind the batch: a caller that loops over pairs like the reference's CLI
 * (wfa-go/wfa-go.go:166-178: one Align per ">"/"<" record) submits each pair -- the bytes are copied, the call returns
 * at once with the pair's ticket (0, 1, 2, ... since the last collect) -- and collects all results with ONE batch
 * alignment: out->...[ticket] is the result of that submission.  Same thread rule as every other call on a context. */
int      wfahip_submit(wfahip_ctx *ctx, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m, uint64_t *ticket);

/* Aligner.Align (wfa.go:196) for ONE pair, without the batch plumbing: no offset arrays in, no malloc'd result arrays
 * out.  rec receives the pair's record (WFAHIP_REC_WORDS u32: status, score, region, statistics, op count; OPS_OFF = 0),
 * ops the CIGAR ops (op<<32 | n, forward order, merged; capacity ops_cap entries), *n_ops their number.  If ops_cap is
 * too small the call returns WFAHIP_ERR_OOM with *n_ops = the capacity needed (at most n + m + 2).  Per-pair failures
 * are statuses in rec[WFAHIP_REC_STATUS] (EMPTY / TOO_LONG), like the batch entry.  A pair whose shape allows it (global,
 * penalties of a shape the register-ring kernels are built for -- x : o+e : e = 2:4:1 (4/6/2), 1:3:1, 1:2:1, 2:3:1, 2:2:1 or
 * 3:3:1 --, both sequences within ~30 kbp) takes ONE kernel launch and no copy -- a wave with a lane per
 * diagonal reads the sequences from, and writes record and CIGAR to, a page-locked block mapped into the GPU's address
 * space, and walks its own backtrace -- 0.144 ms for a 1 kbp pair as the round-4 driver measured it (round 3: 0.29;
 * wfahip_align_batch with n_pairs = 1: 0.45); every other pair, and one whose band leaves 64 diagonals, goes through that
 * entry.  Same results either way.
 * A caller that CAN batch should: a batch aligns ~55 million pairs a second, this call seven thousand. */
int      wfahip_align_pair(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m,
                           uint32_t *rec, uint64_t *ops, uint64_t ops_cap, uint64_t *n_ops);
uint64_t wfahip_pending(const wfahip_ctx *ctx);
int      wfahip_collect(wfahip_ctx *ctx, const wfahip_params *p, wfahip_results *out);

/* A set of contexts, one per GPU, behind one call (wfahip_create(device_ids, n_devices)).
 * wfahip_align_batch_multi cuts the batch into contiguous shards of pairs balanced by sequence bytes, aligns every
 * shard on its own GPU from its own host thread (alignments share no state: one Aligner per goroutine in the
 * reference, wfa.go:73-78) and merges the results in pair order.  device_ids == NULL: devices 0 .. n_devices-1
 * (n_devices <= 0: every device).  The same id may appear more than once (several contexts on one GPU). */
typedef struct wfahip_multi wfahip_multi;
int         wfahip_create_multi(const int *device_ids, int n_devices, wfahip_multi **out);
void        wfahip_destroy_multi(wfahip_multi *m);
int         wfahip_multi_size(const wfahip_multi *m);
wfahip_ctx *wfahip_multi_ctx(wfahip_multi *m, int i); /* for wfahip_set_option / wfahip_last_timing; owned by m */
int         wfahip_align_batch_multi(wfahip_multi *m, const wfahip_params *p, const uint8_t *seq_blob,
                                     uint64_t blob_bytes, const uint64_t *q_off, const uint32_t *q_len,
                                     const uint64_t *t_off, const uint32_t *t_len, uint64_t n_pairs,
                                     wfahip_results *out);

int  wfahip_last_timing(const wfahip_ctx *ctx, wfahip_timing *out);

/* Options (optional; results never depend on them).
 *
 * PUBLIC keys -- always accepted:
 *   "census"  0|1          count the wavefront words every pair stores (WFAHIP_REC_CELLS, timing.cells_stored)     default 0
 *   "learn"  0|1           long pairs start on the arena level the previous call of the same kind needed            default 1
 *   "mem_limit"  bytes     device memory the context may plan with (0 = all of it)
 *   "arena_budget_pct"     share of that memory the long-pair arenas may take                                       default 80
 *   "autopack"  0|1        the host entry 2-bit packs ACGT-only batches before the upload                           default 1
 *   "pair_fast", "pair_lds"   wfahip_align_pair's path (described at the end of the list below)
 *
 * DEBUG keys -- every other key below: a routing experiment, a test aid or the knob of one kernel family.  They are refused
 * with WFAHIP_ERR_UNSUPPORTED unless WFAHIP_DEBUG=1 is set in the environment of the process (tests/conftest.py, bench.py
 * --opt and the scripts under scripts/ set it).  Some change what is SAFE, not only what is fast ("team_strict" 0,
 * "fail_pass"); none belongs in a deployment.
 *   "blk"  16 | 8 | 0      blocked register-window forward kernel, lanes per pair (0 = off)       default 16
 *   "reg", "packed"  0|1   allow the strided register-window / LDS-ring forward kernels           default 1
 *   "packed_arena_bytes"   per-pair arena of the sub-wave pipeline (0 = automatic)
 *   "chunk_pairs", "packed_waves_per_cu", "overlap"   chunking of the sub-wave pipeline
 *   "tail_overlap"  0|1    retry passes run beside the first pass's backtrace kernel             default 1
 *   "pilot"  0|1           wf-adaptive off, >= 65 536 pairs of >= 200 bases: 4 096 pairs go first and decide whether
 *                          the rest starts on the 64-diagonal kernel, on the 256-diagonal one, or on the
 *                          generic kernel                                                          default 1
 *   "blk_batch"  0..8      short reads: a group of the blocked kernel stages several pairs per refill
 *                          (1 = as many as spreads the chunk evenly, at most 8; 2..8 = that many)  default 1
 *   "bt_stream"  n         n waves of the first pass's launch backtrace finished pairs while the other
 *                          waves are still aligning (0 = backtrace kernel after the forward kernel)  default 96
 *   "bt_stream_min"        ... for chunks of at least this many pairs                              default 393216
 *   "bt_stream_single" 0|1 switches the streamed backtrace on (any number of chunks); off by default since round 2:
 *                          a backtrace kernel per chunk is faster now (3e6 x 1 kbp pairs: 65.2 vs 70.5 ms)        default 0
 *   "bt_stream_wait_us"    a streaming wave that waits longer than this for a finished pair leaves the
 *                          rest to the backtrace kernel that follows the launch                     default 20000
 *   "blk_narrow"  0|1      reads under 200 bases start with eight pairs per wave (8 lanes, 32 diagonals each);
 *                          what outgrows that retries on the 16-lane instance                         default 1
 *   "blk_wide"  0|1        pairs whose band leaves the 64-diagonal window retry on the same kernel with a
 *                          wave per pair (256 diagonals) before the generic kernel takes them       default 1
 *   "wide"  0|1|3          semi-global batches of reads up to "wide_max_len" bases (default 2 047, the kernel's limit) and at least
 *                          "wide_min_pairs" pairs (default 64) start on wfa_wide_kernel (round 6: a workgroup per pair, the
 *                          rows in 16-bit LDS rings of any width; two launches per chunk under wf-adaptive -- wide rows, then
 *                          the narrow tail from a checkpoint); 3: one launch per chunk; 0: on the generic ladder     default 1
 *   "wide_waves"  0|1|4    waves per pair in its first launch: 0 = by the rings' size (four above 12 KB)          default 0
 *   "arena_bytes_per_slot", "slots", "threads_per_pair"   generic kernel (one workgroup per pair)
 *   "prepack"  0|1         the sequences of a chunk are 2-bit packed by a kernel of their own before the 16-lane forward
 *                          kernel, whose refill then is one round of loads (forward pass -2 %, packing kernel +4 %)   default 0
 *   "narrow_long"  0|1     reads of any length start on the 8-lanes-per-pair instance (experiment: the refills of the
 *                          hand-over cost more than the narrow steps save)                                            default 0
 *   "census"  0|1          count the wavefront words every pair stores (WFAHIP_REC_CELLS, timing.cells_stored: the roofline
 *                          accounting of bench.py); instrumentation, 4 % of the forward pass on 1 kbp pairs.  The kernels for
 *                          long / semi-global pairs always count                                                default 0
 *   "learn"  0|1            long pairs (team kernel): start on the arena level by which 90 % of the long pairs of the
 *                          previous call of the same kind had finished, instead of climbing from the smallest      default 1
 *   "team_min_len"         pairs at least this long use the team kernel (0 = never)              default 8192
 *   "team_wgs", "team_solo_max"   workgroups per team (0 = automatic), widest row done by one workgroup
 *   "team_wave"                   1 (default): rows of at most 64 diagonals are stepped by one wave out of an LDS ring
 *                                 (team kernel and the one-workgroup-per-pair kernel: wfa_wave.hpp)
 *   "team_strict"                 1 (default): every team barrier carries an agent-scope release (L2 write-back + wait).
 *                                 0: no release -- 10 % faster on 100 kbp pairs and NOT safe: a row word can be read before
 *                                 its write-through has landed (measured; wfa_team.hpp)
 *   "arena_poison"                tests: fill the arena with a pattern before every forward launch (no kernel may read a word
 *                                 it did not write in this launch)
 *   "team_paged"  0|1             1 (default): the teams of the long-pair kernel share ONE pool of arena pages -- a pair holds what it
 *                                 needs, up to eight teams run at once -- instead of a slot each sized for the worst pair
 *   "team_xcd"  0|1|2             teams of one XCD's 32 CUs (1); 2 (default): ... and a team that finds all its workgroups on one
 *                                 XCD keeps its rows in that XCD's L2 (checked at run time; any other placement: the memory-side protocol)
 *   "long"  0|1                   1 (default): global pairs longer than "long_min_len" (4 000) bases, penalties shaped 4/6/2, take the
 *                                 sub-wave kernels with sliding 2-bit sequence windows in LDS (any read length); 0: as in round 3
 *   "long_first"  0|11..15        which of those instances a batch starts on: 0 = by batch size (up to six pairs per SIMD, 6 144: a
 *                                 wave per pair with two diagonals per lane; else four pairs per wave)
 *   "long_window_words"           packed words of each sequence a pair keeps in LDS (default 240 = 3 840 bases; 64..4096)
 *   "long_wave_bt"  0|1|2         backtrace of those pairs by a wave per pair (1: for chunks of at most "long_wave_bt_pairs" pairs --
 *                                 default twenty per CU, 5 120; 2: always; 0: never)
 *   "long_mid_lone"  0|1          1 (default): long pairs handed on for their band, when they are at most six per SIMD, go to the
 *                                 wave-per-pair instance with two diagonals per lane (128 diagonals); 0: two pairs per wave
 *   "pair_fast"  0|1|2|3          wfahip_align_pair: 1 (default) one launch of the lone-pair instance (the wave walks its own
 *                                 backtrace); 3 / 2: round 3's one- / two-launch paths; 0: the batch entry
 *   "pair_lds"  0|1               1 (default): that instance keeps the pair's arena rows in LDS (160 KB: scores up to ~1 240 at
 *                                 penalties 4/6/2; a pair that needs more is re-run with the rows in global memory, and the next 64
 *                                 calls start there); 0: rows in global memory */
int  wfahip_set_option(wfahip_ctx *ctx, const char *key, int64_t value);

/* Debug / parity aid: align ONE pair and return every stored wavefront row.  rows[] receives
 * n_rows descriptors {score, lo, width, offset into words}; words[] holds, per row, the M, I and D
 * raw words (offset<<3|tag, 0 = absent; wfa_wavefront.go:93) for k = lo .. lo+width-1,
 * 3*width words per row.  Caller frees both with wfahip_free. */
typedef struct { uint32_t score; int32_t lo; uint32_t width; uint64_t word_off; } wfahip_row;
int  wfahip_debug_wavefronts(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n,
                             const uint8_t *t, uint32_t m, wfahip_row **rows, uint64_t *n_rows,
                             uint32_t **words, uint64_t *n_words, wfahip_results *res);
void wfahip_free(void *p);

/* Debug / parity aid: the compact backtrace arena of pair `pair` (an index of the most recent batch, inside its first
 * chunk) exactly as the first-pass sub-wave forward kernel left it in HBM, + the pair's meta words {status, final
 * score, end offset, cells}.  One word per diagonal and score: bit 0 the M cell took the insertion's offset, bit 1 it
 * took the mismatch's (which wins ties, wfa.go:657-693; neither: the deletion's), bit 2 the D cell is a DeleteExt
 * (else DeleteOpen), bit 3 the I cell is an InsertExt (else InsertOpen), bits 4-31 the pre-extension offset backTrace
 * recomputes (wfa.go:766-817); a seed of initComponents has offset 0 and bit 0 = Match / bit 1 = Mismatch.  Whether a
 * cell exists is not recorded (the walk only visits cells a stored decision names).  *fmt = layout: 1 = 64 words per
 * score index (score / gcd), diagonal k at slot k & 63; 3 = tiles of 8 score indices x 64 diagonals,
 * [(k & 63) / 4][index & 7][k & 3]; 4 = 256 words per index, slot k & 255; 5 = 32 words per index, slot k & 31; 6 = 128 words
 * per index, slot k & 127; 8 = 32 HALFWORDS per index (wfa_lane_kernel); 10 = halfwords, groups of four diagonals two by two
 * (wfa_duo_kernel, round 6): [(k & 63) / 8][index / 8][(k / 4) & 1][index & 7][k & 3] with n_words / 32 indices per pair of groups
 * (7: fmt 3 with halfwords, its tiled predecessor; 9: one group after the other, [(k & 63) / 4][index][k & 3]).
 * Slots the kernel never wrote hold stale bytes.  Caller frees *words with wfahip_free. */
/* Round 5: the same for a pair on wfa_teamc_kernel (wide wavefronts: option team_wgs = the team's size must be set): every row as
 * the kernel leaves it in the arena -- ONE backtrace word per diagonal (the pre-extension offset backTrace recomputes,
 * wfa.go:766-817, shifted left by four over the decisions of next(): bit 0 the M cell took the insertion's offset, bit 1 the
 * mismatch's, bit 2 the D cell is a DeleteExt, bit 3 the I cell is an InsertExt; offset 0 = a seed of initComponents, bit 0
 * Match / bit 1 Mismatch), rows[i].width words at rows[i].word_off, restricted to the band wf-adaptive kept. */
int  wfahip_debug_team_compact(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m,
                               wfahip_row **rows, uint64_t *n_rows, uint32_t **words, uint64_t *n_words, wfahip_results *res);
int  wfahip_debug_compact_arena(wfahip_ctx *ctx, uint64_t pair, uint32_t **words, uint64_t *n_words, uint32_t *fmt,
                                uint32_t *meta4);

/* Synthetic input generator (host): the seeded dataset spec of DESIGN.md (mirrors what
 * WFA's generate_dataset, used by README.md:298-306, produces: random ACGT pattern of length
 * `length`, text = pattern with round(length*error_rate) random edits).  Pair i uses
 * splitmix64(seed ^ (first_index+i)*0x9E3779B97F4A7C15).  query = pattern, target = text
 * (wfa-go/wfa-go.go:166-178).  blob must hold n_pairs*stride bytes with
 * stride = wfahip_gen_stride(length, error_rate); offsets/lengths arrays hold n_pairs entries. */
uint64_t wfahip_gen_stride(uint32_t length, double error_rate);
int      wfahip_generate_pairs(uint64_t seed, uint64_t first_index, uint64_t n_pairs, uint32_t length,
                               double error_rate, int n_threads, uint8_t *blob, uint64_t *q_off,
                               uint32_t *q_len, uint64_t *t_off, uint32_t *t_len);

/* The same dataset generated on the device: byte for byte what wfahip_generate_pairs writes
 * into the sequences (padding bytes between sequences are not defined), with nothing crossing PCIe.  d_blob must hold
 * n_pairs * wfahip_gen_stride(length, error_rate) + 16 bytes, the other arrays n_pairs entries; all device addresses
 * on the context's GPU.  stream = hipStream_t (NULL = the context's own); returns when the data is there.
 * WFAHIP_ERR_UNSUPPORTED when a pair's text does not fit the LDS it is edited in (length + edits > 160 KB). */
/* Diagnostics for the bench (round 5): the shader clock UNDER LOAD.  Launches a short kernel that keeps every SIMD issuing
 * dependent vector instructions (~0.3 ms) and compares s_memtime (shader cycles) with s_memrealtime (the constant 100 MHz
 * wall clock) in every wave: *mhz = mean shader clock while the GPU was busy with it, *mhz_min / *mhz_max over the waves
 * (either may be NULL).  A kernel-time figure can then be read in cycles as well as in nanoseconds, and two runs on
 * two boxes compared by more than their wall clocks. */
int      wfahip_debug_clock(wfahip_ctx *ctx, double *mhz, double *mhz_min, double *mhz_max);
int      wfahip_generate_pairs_device(wfahip_ctx *ctx, uint64_t seed, uint64_t first_index, uint64_t n_pairs, uint32_t length,
                                      double error_rate, void *d_blob, void *d_q_off, void *d_q_len, void *d_t_off,
                                      void *d_t_len, void *stream);

#ifdef __cplusplus
}
#endif
#endif
