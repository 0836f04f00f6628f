// wfa_ctx.hpp -- what the host-side translation units of libwfahip.so share: the context, its device buffers, the error
// macros of the C-ABI, and the declarations of the functions one unit defines for the others.
//   wfa_host.hip   the router: context life cycle, options, wfahip_align_batch_device -- the sub-wave passes and the
//                  long-pair ladder behind it (align_device)
//   wfa_entry.hip  the host entries: wfahip_align_batch (sliced upload / alignment / download), packed input,
//                  wfahip_align_pair, submit / collect, the results cache
//   wfa_debug.hip  parity and measurement aids: wavefront dumps, the compact arenas, the device-side generator, the clock probe
#pragma once
#include "../../include/wfa_hip.h"
#include "wfa_common.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <malloc.h>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <optional>
#include <sys/file.h>
#include <unistd.h>


#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            std::snprintf(ctx->last_error, sizeof ctx->last_error, "%s:%d %s -> %s", __FILE__, __LINE__, \
                          #expr, hipGetErrorString(_e));                                               \
            return (_e == hipErrorOutOfMemory) ? WFAHIP_ERR_OOM : WFAHIP_ERR_HIP;                      \
        }                                                                                              \
    } while (0)

// No C++ exception may cross the C-ABI (a cgo / ctypes caller cannot unwind): every extern "C" body with an
// allocation in it runs inside this guard.
#define WFAHIP_GUARD(expr)                  \
    try {                                   \
        return (expr);                      \
    } catch (const std::bad_alloc &) {      \
        if (std::getenv("WFAHIP_DEBUG_TIMING")) std::fprintf(stderr, "[wfahip] std::bad_alloc caught at the C-ABI (%s:%d)\n", __FILE__, __LINE__); \
        return WFAHIP_ERR_OOM;              \
    } catch (...) {                         \
        if (std::getenv("WFAHIP_DEBUG_TIMING")) std::fprintf(stderr, "[wfahip] exception caught at the C-ABI (%s:%d)\n", __FILE__, __LINE__); \
        return WFAHIP_ERR_INTERNAL;         \
    }


struct DevBuf {
    void  *p     = nullptr;
    size_t bytes = 0;
};

// ctrl words (device): [0] queue_head [1] redo_count [2,3] ops_cursor (u64) [4,5] debug_info
constexpr int CTRL_WORDS = 8;
constexpr int BLK_BATCH  = 8;  // pairs a group of the blocked kernel stages at a time (short reads)
// Pinned host block of a context (uint32 words).  Copies between pageable memory and the device are staged by the
// runtime and cost 0.3-0.5 ms each even for a few bytes; the retry ladder sits on the critical path of a batch.
constexpr size_t HPIN_CTRL = 0, HPIN_REDO = 16, HPIN_REDO_ENT = 4096, HPIN_WORK = HPIN_REDO + 2 * HPIN_REDO_ENT,
                 HPIN_WORK_IDS = 16384, HPIN_WORDS = HPIN_WORK + HPIN_WORK_IDS;


struct wfahip_ctx {
    int           device     = 0;
    int           num_cus    = 256;
    size_t        total_mem  = 0;
    hipStream_t   stream     = nullptr;
    hipStream_t   stream2    = nullptr;  // backtrace kernels of chunk c overlap the forward kernel of chunk c+1
    std::vector<hipEvent_t> evpool;
    hipEvent_t    ev0 = nullptr, ev1 = nullptr, evA = nullptr, evB = nullptr, evC = nullptr;
    DevBuf        arena, ctrl, redo, work, meta;
    DevBuf        fin;                       // device-side result arrays of the host entry (wfa_finalize.hpp)
    void         *pin[2]     = {nullptr, nullptr};  // pinned staging for result downloads
    hipEvent_t    pin_ev[2]  = {nullptr, nullptr};
    uint32_t     *hpin       = nullptr;  // small pinned block: control words, head of the redo list, work-list staging
    hipStream_t   stream_up  = nullptr;  // host entry: the blob upload runs ahead of the alignment of earlier pairs
    hipStream_t   stream_dn  = nullptr;  // host entry: the results of a slice are downloaded beside the next slice's alignment
    std::vector<hipEvent_t> ev_up;
    DevBuf        team_ctl;                  // barrier counters / reduction sets of the team kernel
    DevBuf        arena2, meta2;             // retry passes run beside the first pass's backtrace kernel
    DevBuf        doneq;                     // streamed backtrace: 256 bytes of counters + one 16-byte entry per pair
    hipEvent_t    evBtA = nullptr, evBtB = nullptr;
    bool          bt_pending = false;        // the first pass's backtrace kernel is still running on stream2
    DevBuf        in_blob, in_qoff, in_qlen, in_toff, in_tlen, out_rec, out_ops;  // host-entry staging
    DevBuf        prepack;                   // 2-bit packed sequences of the current chunk (wfa_prepack_kernel)
    DevBuf        in_small;                  // host entry, small batches: blob + offset / length arrays as one image
    char         *one_pin    = nullptr;      // wfahip_align_pair: mapped pinned block (input image, record, ops) the kernels read and write directly
    char         *one_dev    = nullptr;      // ... its device address
    std::vector<uint64_t> pack_qw, pack_tw;  // ... and the word offsets of the sequences in it
    uint32_t     *pack_pin   = nullptr;      // host entry: page-locked home of the 2-bit words it packs itself, slice by slice
    size_t        pack_pin_bytes = 0;
    int64_t       opt_autopack             = 1;   // 1: wfahip_align_batch 2-bit packs large pure-ACGT batches on host threads while earlier slices upload
    const uint32_t *pk_words = nullptr;      // host entries with packed input, while they call the alignment: the uploaded 2-bit words (device); the byte
                                             // blob they stand for is filled in only for the pairs a pass reads as bytes (wfa_unpack_pairs_kernel)
    bool          one_ctl_clean = false;     // ... whose control words the last call's kernel left zeroed
    DevBuf        one_ctl;                   // ... and its control words: queue head / redo count / ops cursor, then the done queue of the streamed backtrace
    int64_t       opt_arena_budget_pct     = 60;  // long-pair ladder: percent of device memory its arenas may take (80 / 85 / 90: five or six slots
                                                  // instead of four for the configs[4] pairs -- the main launch of 32 pairs 2 875 -> 1 949 / 2 074 / 1 744 ms --
                                                  // but 2 / 1 / 3 of them then outgrow the smaller slots and their re-run takes 1.1 s: no gain, measured)
    int64_t       opt_long_wave_bt_pairs   = 0;   // chunks of at most this many long pairs are walked by a wave per pair (0: twenty per CU -- the walk keeps its state in scalar registers since round 4: 2e4 x 50 kbp, the 4 186 leftovers: 6.2 -> 5.3 ms; all 2e4 that way: +13 ms)
    int64_t       opt_long_mid_lone        = 1;   // long reads handed on for their band, when they are few (<= 6 per SIMD): the lone-wave 128-diagonal instance takes them
    int64_t       opt_pair_lds             = 1;   // wfahip_align_pair's lone-pair instance keeps the pair's arena rows in LDS (0: in global memory); a pair that needs more
                                                 // rows than 160 KB hold is re-run by the global-memory instance, and the next calls start there
    uint32_t      one_lds_skip             = 0;   // calls left that skip the LDS instance (after a pair that did not fit it)
    int64_t       opt_pair_fast            = 1;   // wfahip_align_pair, when the pair allows it: 1 = one launch of the lone-pair instance (a lane per diagonal, the wave walks its
                                                  // own backtrace); 3 = round 3's one launch of the four-pairs-per-wave streaming instance; 2 = that kernel + the backtrace kernel; 0 = the batch entry
    DevBuf        in_packed;                 // host entry with pre-packed input: the 2-bit words as uploaded (unpacked into in_blob on the device)
    // wfahip_submit / wfahip_collect: pairs handed in one at a time, aligned as one batch
    std::vector<uint8_t>  sub_blob;
    std::vector<uint64_t> sub_qoff, sub_toff;
    std::vector<uint32_t> sub_qlen, sub_tlen;
    // options (0 = automatic)
    int64_t       opt_arena_bytes_per_slot = 0;
    int64_t       opt_slots                = 0;
    int64_t       opt_threads_per_pair     = 0;
    int64_t       opt_packed               = 1;  // 0: never use the packed (sub-wave) kernels
    int64_t       opt_reg                  = 1;  // 0: never use the register-window kernel
    int64_t       opt_blk                  = 16; // blocked register-window kernel: lanes per pair (16 or 8), 0 = off
    int64_t       opt_bt_stream            = 96; // > 0: that many waves of the first pass's launch backtrace finished pairs while the others go on
    int64_t       opt_bt_stream_min        = 393216; // ... for chunks of at least this many pairs
    int64_t       opt_bt_stream_single     = 0;      // 1: stream the backtrace (off by default: the backtrace kernel of a single chunk runs beside the
                                                     // retry passes, and on passes of several chunks the streaming instance's write-through stores cost
                                                     // more than the kernel they save)
    int64_t       opt_bt_stream_wait_us    = 20000;  // a streaming wave gives up on a queue entry after this long
    int64_t       opt_blk_narrow           = 1;  // 1: reads under 200 bases start on the 8-lanes-per-pair instance (32-diagonal window, 8 pairs per wave)
    int64_t       opt_blk_wide             = 1;  // 1: pairs leaving the 64-diagonal window retry on the wave-per-pair blocked kernel (256 diagonals)
    int64_t       opt_blk_batch            = 1;  // short reads: stage up to BLK_BATCH pairs per group at a time (1 = automatic count, 2..8 = that many, 0 = off)
    int64_t       opt_packed_arena_bytes   = 0;  // per pair, 0 = automatic
    int64_t       opt_chunk_pairs          = 0;  // 0 = automatic
    int64_t       opt_packed_waves_per_cu  = 0;  // 0 = automatic
    int64_t       opt_team_min_len         = 8192;  // pairs at least this long use the team kernel (several workgroups
                                                    // per pair) in the generic ladder; 0 = never
    int64_t       opt_team_wgs             = 0;     // workgroups per team, 0 = automatic
    int64_t       opt_team_solo_max        = 4096;           // (wfa_team.hpp: TEAM_SOLO_MAX)  // rows up to this width are done by one workgroup
    bool          opt_team_solo_max_set    = false;          // (wfa_teamc_kernel takes 512 unless the option was set: its pipelined team rows cost ~6 us, a row of workgroup 0 alone
                                                             // ~8 -- configs[4] x 32 pairs, two of which spend 3.5e4 rows between 65 and 4 096 diagonals: 30.8 -> 32.5 pairs/s)
    int64_t       opt_team_wave            = 1;              // rows up to 64 diagonals are done by one wave (LDS ring)
    int64_t       opt_team_strict          = 1;              // agent-scope release in every team barrier (0: see wfa_team.hpp)
    int64_t       opt_unpack_all           = 0;              // 1: host entries with packed input expand ALL of it to bytes on the device first (rounds 2-3)
    int64_t       opt_team_compact         = 1;              // 1: wide wavefronts on wfa_teamc_kernel (round 5: one backtrace word per diagonal in the arena, the rows the next
                                                             // steps source in LDS stripes, reductions travelling with the barrier: wfa_teamc.hpp); 0: wfa_team_kernel
    int64_t       opt_team_fast            = 1;              // ... 1: its stripe modes run their steady state in the short step (0: every step takes the general one; tests compare the two)
    int64_t       opt_team_pipe            = 1;              // ... 1: its team stripe steps are pipelined (a row's exchanges beside the next row's cells)
    int64_t       opt_team_scout           = 0;              // ... 1: batches of more than two pairs per team first run with ONE workgroup per pair, which hands on (ST_REDO_WIDE) the pairs
                                                             // whose band stays wider than a stripe: the others no longer park 31 CUs each (2: whatever the batch size; 0: off).
                                                             // Off by default: 26 of configs[4]'s first 32 pairs keep a wide band, the six others cost a team 0.09 s each -- 32 pairs
                                                             // take 1.25 s with the pass and without it (round 5)
    int64_t       opt_team_order           = 1;              // ... 1: the pairs that will keep a wide band are queued first (a scheduling hint)
    int64_t       opt_team_slack           = 1024;           // ... diagonals of room on either side when its stripes are positioned (tests: a few, so that the axis moves often)
    bool          dbg_teamc                = false;          // wfahip_debug_team_compact is running: the one-pair debug launch takes wfa_teamc_kernel
    DevBuf        xbuf;                                      // ... its exchange rows
    int64_t       opt_team_paged           = 1;              // 1: the teams share one pool of arena pages (a pair holds what it needs) instead of a slot each
    DevBuf        page_ctl;                                  // ... its free-page stack and the page lists of the teams
    int64_t       opt_team_xcd             = 2;              // 1: teams of one XCD's CUs (blockIdx % 8); 2 (default): ... and a team that finds itself on one XCD keeps
                                                             // its rows in that XCD's L2 (plain stores, no release in its barriers): 548 -> 512 ms per 8 x 100 kbp with
                                                             // eight teams; 310 passes of that sample over a poisoned pool without a deviation (profiles/r04_team_xcd_soak.txt)
    int64_t       opt_arena_poison         = 0;              // tests: fill the arena with a pattern before every long-pair launch
    int64_t       opt_pilot                = 1;  // 1: a 4 096-pair pilot decides whether a large batch uses the sub-wave kernels
    int64_t       opt_tail_overlap         = 1;  // 1: retry passes overlap the backtrace kernel of the first pass
    int64_t       opt_overlap              = 0;  // 1: backtrace of chunk c on a second stream beside the forward kernel of chunk c+1 (measured: no gain)
    int64_t       opt_fail_pass            = 0;   // test aid (fault injection): the sub-wave pass of this kind reports WFAHIP_ERR_OOM
    int64_t       opt_prepack              = 0;   // 1: a chunk's sequences are 2-bit packed by a kernel of their own before the 16-lane forward kernel
                                                  // (measured: forward 19.96 -> 19.54 ms per 1e6 x 1 kbp pairs, but the packing kernel takes 0.9 ms: off)
    int64_t       opt_narrow_long          = 0;   // experiment: reads of any length start on the 8-lanes-per-pair instance (32-diagonal windows)
    DevBuf        wide_ckpt;                      // wfa_wide_kernel: what its first launch hands its second, per pair of the chunk
    int64_t       opt_wide                 = 1;   // semi-global batches of reads up to 2 047 bases (penalties of one of the sub-wave shapes) start on wfa_wide_kernel (round 6: a workgroup per
                                                  // pair, the rows in 16-bit LDS rings of any width, two launches per chunk under wf-adaptive); 3: one launch per chunk, every pair
                                                  // runs to its end in the wide rings; 0: on the generic ladder
    int64_t       opt_wide_max_len         = 2047;// ... of reads up to this length.  1e6 x 1 kbp: 250 ms against the generic kernel's 403; 1e5 x 300 bp: 8.0 against 12.7;
                                                  // 2e4 x 1 kbp without wf-adaptive: 30 against 71
    int           opt_wide_waves           = 0;   // ... waves per pair in its first phase: 0 = by the rings' size (four above 12 KB: a 1 kbp pair's 25 KB leave a SIMD a wave and a
                                                  // half otherwise), 1 or 4 forced (tests)
    bool          opt_wide_exact           = false; // ... 1: no packed fast path in the wide rows (tests)
    int64_t       opt_wide_min_pairs       = 64;  // ... from this many pairs on (fewer: the one-workgroup-per-pair kernel, whose workgroup is larger than a wave)
    int64_t       opt_duo                  = 1;   // reads of 240+ bases start on wfa_duo_kernel (8 or 16 lanes per pair, changing while the pair runs):
                                                  // 0 never, 1 for batches of at least opt_duo_min_pairs (below that its start-up -- a wave takes one new pair
                                                  // per step -- costs more than the fuller rows give: 1e5 pairs 2.6 vs 2.3 ms), 2 always
    int64_t       opt_duo_min_pairs        = 200000;
    int64_t       opt_duo_short            = 0;   // 1 / 2: batches of short reads (<= 240 bases) use it too, with eight pairs per fetch.  Off: measured
                                                  // SLOWER than the batched 8-lane instance (1e5 x 150 bp: forward 0.317 vs 0.249 ms, 1e6: 1.60 vs 1.43 ms, plus the
                                                  // packing kernel) -- a 150-base pair lives ten steps, so a wave restructures on nearly every step
    int64_t       opt_duo_short_min_pairs  = 50000;
    bool          ctrl_clean               = false;  // the control words are zero: the last call zeroed them on ctrl_clean_stream as it left
    hipStream_t   ctrl_clean_stream        = nullptr;
    hipEvent_t    ctrl_clean_ev            = nullptr;  // ... recorded behind that memset: a call on ANOTHER stream waits for it before it touches them
    bool          redo_was_empty           = false;  // the last pass that asked for its redo list found it empty
    int64_t       opt_compact_call_bases   = 50000000;  // first passes over at most this many bases (pairs x longest read) keep their backtrace
                                                        // kernel on the call's stream (no event wait on the second one): 0 = never
    int64_t       opt_lane                 = 1;   // reads of at most 240 bases start on wfa_lane_kernel (a lane per pair): 0 never, 1 for batches of
                                                  // at least opt_lane_min_pairs, 2 always
    int64_t       opt_lane_pack            = 1;   // 1: the lanes of wfa_lane_kernel pack the bytes of their pairs themselves, 0: wfa_prepack_kernel before it
    int64_t       opt_lane_min_pairs       = 32768;  // (below ~30 000 pairs a generation of 64 pairs per wave leaves most of the GPU idle for as
                                                     // long as its slowest pair runs: 16 000 x 150 bases 0.192 ms against 0.157 on the 8-lane kernel)
    int64_t       opt_long                 = 1;   // 1: global pairs longer than opt_long_min_len (penalties shaped 2:4:1) take the sub-wave kernels with sliding
                                                  // sequence windows (wfa_blk_kernel<.., LONG>): four pairs per wave at any read length
    int64_t       opt_long_min_len         = 4000;   // (below it both whole sequences of a pair fit the plain instances' LDS at full occupancy)
    int64_t       opt_long_window_words    = 240; // packed words per sequence window: 3 840 bases, 7.5 KB of LDS per wave of four pairs -- twenty waves per CU
                                                  // (256 words: nineteen fit, and 2e4 x 50 kbp pairs -- 5 000 waves -- ran a second round: forward 31.5 against 28.0 ms)
    int64_t       opt_long_first           = 0;   // 0: by batch size; 11 / 12 / 13: long reads start on the 64- / 128- / 256-diagonal instance
    int64_t       opt_long_wave_bt         = 1;   // 1: the backtrace of those pairs is walked by a wave per pair (0: a lane per pair, like short pairs)
    int64_t       opt_census               = 0;   // 1: the sub-wave forward kernels count the wavefront words they store (REC_CELLS, timing.cells_stored)
    int64_t       opt_learn                = 1;   // 1: long pairs start on the arena level the previous call of the same kind ended on
    uint64_t      learn_key                = 0;   // workload class of the last call that used the team kernel
    int           learn_level              = 0;   // ... and the level by which 90 % of its long pairs had finished
    uint32_t      learn_calls              = 0;
    // rows per pair of the blocked kernels' arenas: the default holds scores up to half the read length (error rates up
    // to ~8 % at 4/6/2); a class of batches whose pairs ran out of rows gets twice / four times / eight times as many
    // from its next call on (the call that finds out re-runs those pairs on the same kernel with four times the rows)
    uint64_t      rows_key                 = 0;
    uint32_t      rows_scale               = 1;
    // ... and the window such a class starts on: 0 = the 64-diagonal first pass, 9 = wfa_blk_kernel<32,1> (128 diagonals),
    // 5 = wfa_blk_kernel<64,1> (256), learned when most pairs of a call were handed on because of their band
    uint64_t      band_key                 = 0;
    int           band_kind                = 0;
    uint32_t      band_calls               = 0;   // (every sixteenth call of the class starts on its natural first pass again: data changes)
    int64_t       opt_blk_mid              = 1;   // 1: band failures of the 64-diagonal kernels try the 128-diagonal instance before the 256-diagonal one   // calls of that class since the level was learned (every 4th one probes one level lower)
    int64_t       opt_mem_limit            = 0;   // tests: pretend the device has this many bytes (arena budgets follow)
    int           force_mode               = -1;  // debug: start the ladder in this mode
    // debug / parity aid (wfahip_debug_compact_arena): where the first chunk of the most recent first pass left its arena
    const uint32_t *dbg_arena = nullptr;
    const uint4    *dbg_meta  = nullptr;
    uint64_t        dbg_words = 0, dbg_first = 0, dbg_n = 0;
    uint32_t        dbg_fmt = 0, dbg_g = 1;
    wfahip_timing timing{};
    char          last_error[256] = {0};
};


inline int ensure(wfahip_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return WFAHIP_OK;
    if (b.p) {
        HIP_TRY(hipFree(b.p));
        b.p = nullptr, b.bytes = 0;
    }
    // (wfahip_debug_compact_arena keeps pointers into the first pass's arena and meta buffers: a later retry or ladder
    // pass that re-allocates either of them ends that snapshot instead of leaving it dangling)
    if (&b == &ctx->arena || &b == &ctx->meta) ctx->dbg_arena = nullptr, ctx->dbg_meta = nullptr, ctx->dbg_n = 0;
    size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return WFAHIP_OK;
}

inline void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr, b.bytes = 0;
}

inline uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) {
        uint32_t t = a % b;
        a          = b;
        b          = t;
    }
    return a;
}

inline int check_params(const wfahip_params *p) {
    if (!p) return WFAHIP_ERR_BAD_ARG;
    // Mismatch == 0: the reference's own loop does not terminate when the first bases differ (the seed is then a
    // Mismatch cell at score 0 whose source M[s - 0][k] is itself).  GapOpen + GapExt == 0: M[s-o-e] is the row being
    // written.  GapExt == 0 alone is aligned (by the generic kernel: the I row of a score becomes a serial scan).
    if (p->mismatch == 0 || p->gap_open + p->gap_ext == 0) return WFAHIP_ERR_UNSUPPORTED;
    if (p->adaptive && p->min_wf_len == 0) return WFAHIP_ERR_BAD_ARG;  // AdaptiveReduction rejects it (wfa.go:134-137)
    return WFAHIP_OK;
}

constexpr size_t LDS_MAX_BYTES = 160 * 1024;

// ---- defined in wfa_entry.hip
void results_zero(wfahip_results *r);
// records + ops of n pairs (as the kernels leave them) -> the malloc'd arrays of a wfahip_results
int unpack_results(const std::vector<uint32_t> &rec, const std::vector<uint64_t> &ops, uint64_t n, wfahip_results *out, uint64_t *cells_total);

// ---- defined in wfa_host.hip
// the device-resident core behind every entry (exception-safe: the host entry calls it while its upload / download threads are joinable)
int align_device(wfahip_ctx *ctx, const wfahip_params *p, const void *d_blob, uint64_t blob_bytes, const void *d_q_off, const void *d_q_len,
                 const void *d_t_off, const void *d_t_len, uint64_t n_pairs, uint32_t max_len, void *d_rec, void *d_ops, uint64_t ops_cap,
                 uint64_t *ops_needed, hipStream_t st, bool debug_single, uint64_t ops_cursor0 = 0);
// one workgroup of wfa_backtrace_kernel behind a lone-pair forward launch (wfahip_align_pair)
hipError_t wfa_launch_backtrace_one(const wfa::KParams &P, hipStream_t st);

