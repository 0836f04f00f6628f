// wfa_common.hpp -- shared definitions of the gfx950 wavefront-alignment kernels.
//
// Semantics follow shenwei356/wfa v0.4.0 (citations are file:line into the reference checkout):
//   * a wavefront word is  offset<<3 | tag , 0 = absent      (wfa_wavefront.go:93,153-159)
//   * tags 1..6 = InsOpen, InsExt, DelOpen, DelExt, Mismatch, Match   (wfa_backtrace_types.go:27-35)
//   * offsets count target bases consumed (1-based h); diagonal k = h - v
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Kernels that are not templates are defined in headers that more than one translation unit includes (wfa_host.hip and
// wfa_duo.hip): the unit that does not launch them defines WFA_NO_AUX_KERNELS and gets the device functions only.

// Wave-uniform branches that are almost never / almost always taken: the hint moves the cold block out of the fall-through
// path.  A wave that is alone on its SIMD pays a fetch restart (~25 cycles) for every TAKEN branch, and the compiler's default
// layout makes the common case jump over every rare block: with the hints a lone-wave score step is straight-line code
// (round 4: 500 x 50 kbp 11.0 -> 9.3 ms, a single Align 166 -> 149 us).
#define WFA_RARE(x) __builtin_expect(!!(x), 0)
#define WFA_OFTEN(x) __builtin_expect(!!(x), 1)

namespace wfa {

enum : uint32_t {
    TAG_INS_OPEN = 1, TAG_INS_EXT = 2, TAG_DEL_OPEN = 3, TAG_DEL_EXT = 4, TAG_MISMATCH = 5, TAG_MATCH = 6
};
constexpr uint32_t TAG_BITS = 3, TAG_MASK = 7;

// internal per-pair status values written to rec[STATUS] while a batch is in flight
enum : uint32_t {
    ST_OK = 0, ST_EMPTY = 1, ST_TOO_LONG = 2, ST_NO_MEMORY = 4,
    ST_PENDING = 0xFFFFFFFFu,  // not processed yet
    DONE_NOT_OK = 0x80000000u, // done_q entry .x = (index in chunk + 1) | DONE_NOT_OK for pairs without a backtrace
    DONE_TAKEN  = 0xFFFFFFFFu, // done_q entry .x after the streaming backtrace kernel has processed it
    ST_REDO_BYTES = 100,       // non-ACGT byte found: needs the byte-compare path
    ST_REDO_ARENA = 101,       // wavefront arena too small: needs a bigger slot
    ST_REDO_LDS   = 102,       // sequences do not fit this launch's LDS budget
    ST_REDO_BAND  = 103,       // register-window kernel: diagonal band left the tile range
    ST_REDO_WIDE  = 104        // scout pass of the team kernel (one workgroup per pair): the band stays wider than a workgroup's stripe -- a team's work
};

constexpr int REC_WORDS = 16;
enum { REC_STATUS = 0, REC_SCORE, REC_TBEGIN, REC_TEND, REC_QBEGIN, REC_QEND, REC_ALIGN_LEN, REC_MATCHES,
       REC_GAPS, REC_GAP_REGIONS, REC_OPS_LEN, REC_OPS_OFF_LO, REC_OPS_OFF_HI, REC_CELLS_LO, REC_CELLS_HI,
       REC_N_SCORES };

// One directory entry per score index (score / g): the M, I and D rows of that score share the
// diagonal range [lo, lo+w) and sit at arena[base], arena[base+stride], arena[base+2*stride] (32-byte entry).
// w == 0 means no wavefront exists at that score in any component (Component.HasScore false,
// wfa_component.go:81-86).  After wf-adaptive pruning the entry is narrowed to the surviving band
// (base moves right, stride keeps the row pitch), so later scores only visit live diagonals.
struct alignas(16) DirEnt {
    uint64_t base;  // word index inside the slot (64-bit: 100 kbp semi-global pairs need > 16 Gi words)
    int32_t  lo;
    int32_t  w;
    uint32_t stride;
    uint32_t pad[3];
};
constexpr int DIR_WORDS = 8;  // 32-byte entries

struct KParams {
    // input (device pointers)
    const uint8_t  *blob;
    uint64_t        blob_bytes;
    const uint64_t *q_off;
    const uint32_t *q_len;
    const uint64_t *t_off;
    const uint32_t *t_len;
    const uint32_t *work;  // pair ids to process (nullptr = identity 0..n_work-1)
    uint32_t        n_work;
    // scheduling
    uint32_t *queue_head;
    // per-slot scratch
    uint32_t *arena;
    uint64_t  arena_words;  // per slot
    // penalties / options
    uint32_t x, o, e, oe, g;
    uint32_t global_alignment, adaptive, min_wf_len, max_dist_diff;
    // output
    uint32_t           *rec;  // [n_pairs][REC_WORDS]
    uint64_t           *ops;
    uint64_t            ops_cap;
    unsigned long long *ops_cursor;
    uint32_t           *redo_list;  // {pair, status} of pairs needing another configuration
    uint32_t           *redo_count;
    // LDS budget of this launch: words available for EACH packed sequence (incl. 1 pad word)
    uint32_t lds_seq_words;
    // debug: keep slot 0's arena intact and publish its final directory size
    uint32_t *debug_info;  // [0] = number of directory entries, [1] = final score

    // ---- packed (sub-wave) forward kernel + deferred backtrace kernel
    uint32_t  chunk_first, chunk_n;  // pairs [chunk_first, chunk_first + chunk_n); arena slot = index in chunk
    uint4    *pair_meta;             // per pair of the chunk: {status, final score, directory entries, cells}
    uint32_t  dx, doe, de;           // x/g, (o+e)/g, e/g
    uint32_t  dm, di;                // ring depths: max(dx,doe)+1 (M), de+1 (I and D)
    uint32_t  sub_lds_words;         // LDS words owned by one 32-lane subgroup
    uint32_t  min_xe;                // min(x, e): bounds the number of CIGAR ops by 2*score/min_xe + 8
    // streamed backtrace (wfa_blk_kernel<.., STREAM = true> + wfa_backtrace_stream_kernel): finished pairs are pushed to
    // done_q in completion order; done_ctl = {entries pushed, entries ticketed}
    uint4    *done_q;
    uint32_t *done_ctl;
    uint32_t  n_stream_wgs;  // the first workgroups of the launch only backtrace
    uint32_t  stream_wait;   // longest wait for a done_q entry, in ticks of the 100 MHz wall clock
    uint32_t  blk_batch_n;           // wfa_blk_kernel<.., BATCH > 1>: queue entries a group takes per refill (1 .. BATCH)
    // pre-packed sequences of the chunk (wfa_prepack_kernel): slot i = {q_len, t_len, status, 0, q words [lds_seq_words],
    // t words [lds_seq_words]} of pair chunk_first + i; nullptr: the forward kernel packs the bytes itself in its refill
    const uint32_t *prepack;
    uint32_t        prepack_words;   // words per slot
    uint32_t  wave_rows;             // wfa_generic_kernel: rows of its wave mode's LDS ring (a power of two), 0 = wave mode off
    uint32_t  wave_bt;               // wfa_generic_kernel: 1 = the LDS directory window exists (backtrace walked by a wave)
    uint32_t  census;                // sub-wave forward kernels: report the number of stored wavefront words (REC_CELLS), else 0
    // wfa_team_kernel, paged arena (round 4): the teams share ONE pool of pages instead of owning a slot each -- a 100 kbp
    // semi-global pair takes anything from 0.2 to 43 GB, and slots sized for the worst pair left room for four teams.
    // page_ctl: [0] lock [1] free pages [2] teams waiting for a page [3] teams holding pages [4 ..] stack of free page ids, then
    // per team a list of the pages its pair holds;
    // nullptr = one slot of arena_words per team.  The directories live at the end of the pool, dir_region_words per team.
    uint32_t *page_ctl;
    uint32_t  page_words_log2, n_pages;
    uint64_t  dir_region_words;
    uint32_t  fuse_bt;               // wfa_blk_kernel<64, 1, false, 1> (one pair, wfahip_align_pair): 1 = the wave walks its pair's backtrace itself when
                                     // the pair queue is empty (one launch for the whole Align)
    uint32_t  lds_arena_off;         // wfa_blk_kernel<.., LDSA = true>: word offset of the pair's arena rows inside the workgroup's LDS
    uint32_t  one_n, one_m;          // ... and the lengths of its one pair (query at blob offset 0, target at (one_n + 15) & ~15): the kernel arguments carry
                                     // them, so that neither the refill nor the walk fetches them from the host's mapped block (a PCIe round trip each)
    uint32_t *wide_ckpt;             // wfa_wide_kernel: WIDE_CKPT_WORDS words per pair of the chunk -- what its first launch hands its second (wfa_wide.hpp)
    uint32_t  wide_ckpt_on;          // ... 1: the first launch hands pairs on once their rows are narrow; 0: it runs every pair to its end
    uint32_t  wide_exact;            // ... 1: every round of the wide rows takes the exact per-cell path (tests: the packed two-diagonals-per-register path against it)
    uint32_t  compact_fmt;           // compact arena layout (CompactView): 0 = rows + directory; no directory: 1 = 64 words
                                     // per score, diagonal k at slot k & 63; 3 = tiles of 8 scores x 64 diagonals;
                                     // 4 = 256 words per score, slot k & 255; 5 = 32 words per score, slot k & 31
};

}  // namespace wfa
