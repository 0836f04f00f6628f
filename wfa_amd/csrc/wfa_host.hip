// wfa_host.hip -- the C-ABI of libwfahip.so (include/wfa_hip.h): context, workspaces, launch
// configuration, retry ladder (bigger arena / byte-compare path) and result unpacking.
//
// There is NO CPU fallback here: every alignment is produced by the HIP kernels.  Without a GPU the
// entry points return WFAHIP_ERR_NO_DEVICE.
#include "../../include/wfa_hip.h"
// (the kernels' headers for their constants and device functions; the forward kernels are instantiated per penalty shape in
// wfa_fwd_s*.hip, wfa_duo_kernel in wfa_duo.hip, the long-pair kernels in wfa_long.hip: this unit keeps the router, the
// non-template kernels of the pipeline -- packing, backtrace, result assembly, the generator -- and the two first-generation
// forward kernels that are only selectable by option)
#include "wfa_generic.hpp"
#include "wfa_packed.hpp"
#include "wfa_reg.hpp"
#include "wfa_blk.hpp"
#include "wfa_duo_cfg.hpp"
#include "wfa_lane.hpp"
#include "wfa_team.hpp"
#include "wfa_teamc.hpp"
#include "wfa_fwd.hpp"
#include "wfa_long.hpp"
#include "wfa_finalize.hpp"
#include "wfa_gen_dev.hpp"

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

using namespace wfa;

// the per-shape launchers behind one switch (wfa_fwd.hpp)
namespace wfa {
hipError_t wfa_launch_fwd(int shape, int kind, uint32_t flags, const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st) {
    switch (shape) {
    case 0: return wfa_launch_fwd_s24(kind, flags, P, grid, lds_bytes, st);
    case 1: return wfa_launch_fwd_s13(kind, flags, P, grid, lds_bytes, st);
    case 2: return wfa_launch_fwd_s12(kind, flags, P, grid, lds_bytes, st);
    case 3: return wfa_launch_fwd_s23(kind, flags, P, grid, lds_bytes, st);
    case 4: return wfa_launch_fwd_s22(kind, flags, P, grid, lds_bytes, st);
    case 5: return wfa_launch_fwd_s33(kind, flags, P, grid, lds_bytes, st);
    }
    return hipErrorInvalidValue;
}
hipError_t wfa_launch_pair(int shape, bool lds_arena, const KParams &P, size_t lds_bytes, hipStream_t st) {
    switch (shape) {
    case 0: return wfa_launch_pair_s24(lds_arena, P, lds_bytes, st);
    case 1: return wfa_launch_pair_s13(lds_arena, P, lds_bytes, st);
    case 2: return wfa_launch_pair_s12(lds_arena, P, lds_bytes, st);
    case 3: return wfa_launch_pair_s23(lds_arena, P, lds_bytes, st);
    case 4: return wfa_launch_pair_s22(lds_arena, P, lds_bytes, st);
    case 5: return wfa_launch_pair_s33(lds_arena, P, lds_bytes, st);
    }
    return hipErrorInvalidValue;
}
}  // namespace wfa

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

namespace {

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

}  // namespace

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
    int64_t       opt_team_solo_max        = TEAM_SOLO_MAX;  // rows up to this width are done by one workgroup
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

namespace {

int ensure(wfahip_ctx *ctx, DevBuf &b, size_t bytes) {
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

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr, b.bytes = 0;
}

uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) {
        uint32_t t = a % b;
        a          = b;
        b          = t;
    }
    return a;
}

// Share of device memory the arenas of the long-pair ladder may take: 0.6 -> four 43 GiB slots for the hard 100 kbp
// semi-global pairs (a team of workgroups each).  Option arena_budget_pct; more slots were measured and lose to the
// re-runs of the pairs that outgrow them (see the option).
inline double ladder_budget(const wfahip_ctx *ctx) { return std::min(0.9, std::max(0.1, (double)ctx->opt_arena_budget_pct / 100.0)); }

// wfa_team_kernel / wfa_teamc_kernel synchronise their workgroups with barriers they spin on: every workgroup of a launch
// must be resident, which holds for ONE such launch on a GPU (the grid is sized by its CUs) and not for two -- two contexts
// on one device (wfahip_create_multi with repeated ids, bench.py --share-gpus, two processes of one job) would each hold part
// of the GPU and wait for the rest until the barrier timeout reports WFAHIP_ERR_INTERNAL.  So team launches on a device
// are serialised: a mutex per device inside the process, an advisory file lock per device (named by its PCI bus id)
// between processes.  Held from the launch to the stream synchronisation behind it; the sub-wave kernels never take it.
struct TeamLaunchLock {
    std::mutex *mx = nullptr;
    int         fd = -1;
    explicit TeamLaunchLock(int device) {
        static std::mutex per_device[64];
        mx = &per_device[device & 63];
        mx->lock();
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) != hipSuccess) std::snprintf(bus, sizeof bus, "dev%d", device);
        for (char *c = bus; *c; c++)
            if (*c == ':' || *c == '/' || *c == '.') *c = '_';
        const char *tmp = std::getenv("TMPDIR");
        char        path[256];
        std::snprintf(path, sizeof path, "%s/wfahip_team_%s.lock", (tmp && *tmp) ? tmp : "/tmp", bus);
        fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd >= 0 && flock(fd, LOCK_EX) != 0) close(fd), fd = -1;  // (no lock file: the process-wide mutex alone)
    }
    ~TeamLaunchLock() {
        if (fd >= 0) (void)flock(fd, LOCK_UN), close(fd);
        if (mx) mx->unlock();
    }
    TeamLaunchLock(const TeamLaunchLock &) = delete;
    TeamLaunchLock &operator=(const TeamLaunchLock &) = delete;
};

struct LaunchCfg {
    int      waves;        // 1, 4 or 16 waves per pair
    int      mode;         // 0 = 2-bit LDS, 1 = bytes in global memory
    uint32_t lds_seq_words;
    size_t   lds_bytes;
    uint64_t arena_words;  // per slot
    uint32_t slots;
};

hipError_t launch_generic(const KParams &P, const LaunchCfg &c, hipStream_t st) { return wfa_launch_generic(P, c.waves, c.mode, c.slots, c.lds_bytes, st); }

struct Job {
    int                   mode;
    int                   level;  // arena size: make_cfg's ladder (x8, x8, x2 .., then one slot fewer per level)
    bool                  all;    // identity work list over all pairs
    std::vector<uint32_t> pairs;
    uint32_t              max_len = 0;  // length bound of these pairs (0 = the batch's)
    bool                  hint    = false;  // `level` came from an earlier call of the class (learn), not from a failed level below it
    bool                  scout   = false;  // long pairs, first launch: may run as the team kernel's scout pass (one workgroup per pair; wide pairs are handed on)
};

constexpr size_t LDS_MAX_BYTES = 160 * 1024;

// Launch configuration for one job.
int make_cfg(wfahip_ctx *ctx, uint32_t max_len, int mode, int level, uint64_t n_work, bool semi_global, LaunchCfg &c) {
    int waves = max_len <= 4096 ? 1 : (max_len <= 65536 ? 4 : 16);
    if (ctx->opt_threads_per_pair > 0) {
        int64_t t = ctx->opt_threads_per_pair;
        waves     = t <= 64 ? 1 : (t <= 256 ? 4 : 16);
    }
    c.waves         = waves;
    c.mode          = mode;
    c.lds_seq_words = (mode == 0) ? ((max_len + 15) / 16 + 1) : 0;
    c.lds_bytes     = (2ull * c.lds_seq_words + GEN_LDS_EXTRA_WORDS) * 4ull;
    if (c.lds_bytes > LDS_MAX_BYTES) return 1;  // caller must use mode 1

    // semi-global rows are n+m-1 wide until wf-adaptive collapses the band (a few dozen scores): ~4x the words
    uint64_t base_words = std::max<uint64_t>(64 * 1024, (semi_global ? 384ull : 96ull) * max_len);
    if (ctx->opt_arena_bytes_per_slot > 0) base_words = std::max<uint64_t>(4096, ctx->opt_arena_bytes_per_slot / 4);
    // the ladder: x8, x8, then x2 per level -- a slot of a long pair is tens of GB by then, and every doubling
    // halves the number of pairs that can be in flight (level 3 of a 100 kbp semi-global pair: 21.6 GB)
    // Once at most six slots fit the budget (tens of GB per pair), a level is "one slot fewer", and a slot takes
    // its whole share of the budget: 5 x 46 GB, 4 x 57, 3 x 76, 2 x 115, 1 x 230 on a 288 GB device with the budget at
    // 80 % (round 2, 60 %: 4 x 43, 3 x 57, 2 x 86, 1 x 172) -- every slot dropped is a team of workgroups less in flight.
    const uint64_t budget_words = (uint64_t)((double)ctx->total_mem * ladder_budget(ctx)) / 4ull;
    uint64_t       words        = base_words;
    auto snap = [&](uint64_t w) {
        const uint64_t fit = w ? budget_words / w : 0;
        return (fit >= 1 && fit <= 6 && ctx->opt_arena_bytes_per_slot <= 0) ? budget_words / fit : w;  // (not an explicit size)
    };
    for (int i = 0; i < level; i++) {
        const uint64_t fit = budget_words / words;
        if (fit >= 2 && fit <= 6)
            words = budget_words / (fit - 1);
        else
            words *= (i < 2 ? 8 : 2);
    }
    words         = snap(words) & ~7ull;  // directory entries are 32-byte aligned from the slot end
    c.arena_words = words;

    // resident workgroups per CU: 32 wave slots, LDS, and keep <= 8 blocks of >=256 threads
    uint32_t per_cu = 32 / waves;
    per_cu          = std::min<uint32_t>(per_cu, (uint32_t)(LDS_MAX_BYTES / std::max<size_t>(c.lds_bytes, 1)));
    per_cu          = std::max<uint32_t>(per_cu, 1);
    uint64_t slots  = (uint64_t)ctx->num_cus * per_cu;
    if (ctx->opt_slots > 0) slots = (uint64_t)ctx->opt_slots;
    // arena budget: at most ~60 % of device memory
    uint64_t budget = budget_words * 4ull;
    uint64_t fit    = budget / (words * 4ull);
    if (fit == 0) return 2;  // even one slot does not fit
    slots   = std::min<uint64_t>(slots, fit);
    slots   = std::min<uint64_t>(slots, std::max<uint64_t>(n_work, 1));
    c.slots = (uint32_t)slots;
    return 0;
}

}  // namespace

// Pre-packed input without the detour through bytes (round 4).  The host entries used to expand the whole upload to bytes
// (wfa_unpack_kernel) only for the first pass to pack its chunk again (wfa_prepack_kernel): 5 GB of HBM traffic and 1.3 ms per
// 1e6 x 1 kbp pairs.  Now a pass that fetches from pre-packed slots has them COPIED from the uploaded words
// (wfa_prepack_words_kernel: a wave per pair), and a pass that reads bytes -- the retry rungs, the byte path, the long-pair
// kernels: a thousandth of the pairs -- has exactly its pairs expanded first (wfa_unpack_pairs_kernel).  Offsets are the
// byte offsets of the blob the words stand for (multiples of 16: every sequence starts at a word).
__global__ __launch_bounds__(256) void wfa_prepack_words_kernel(const KParams P, const uint32_t *__restrict__ words, uint32_t *__restrict__ out,
                                                                uint32_t SW, uint32_t PW) {
    const uint32_t lane = threadIdx.x & 63u, wi = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wi >= P.chunk_n) return;
    const uint32_t pr = P.work ? P.work[wi] : P.chunk_first + wi;
    const uint32_t nq = P.q_len[pr], mt = P.t_len[pr];
    uint32_t       status = ST_PENDING;
    if (nq == 0 || mt == 0)
        status = ST_EMPTY;  // wfa.go:204-206
    else if (nq > 0x1FFFFFFFu || mt > 0x1FFFFFFFu)
        status = ST_TOO_LONG;  // wfa.go:207-209
    else if (((nq > mt ? nq : mt) + 15u) / 16u + 1u > SW)
        status = ST_REDO_LDS;
    uint32_t *const slot = out + (uint64_t)wi * PW;
    if (status == ST_PENDING) {
        const uint32_t *const qw = words + P.q_off[pr] / 16u, *const tw = words + P.t_off[pr] / 16u;
        const uint32_t nwq = (nq + 15u) >> 4, nwt = (mt + 15u) >> 4;
        for (uint32_t v = lane; v < 2u * SW; v += 64u) {
            const bool     isq = v < SW;
            const uint32_t j   = isq ? v : v - SW;
            slot[4u + v]       = j < (isq ? nwq : nwt) ? (isq ? qw : tw)[j] : 0u;
        }
    }
    if (lane == 0u) slot[0] = nq, slot[1] = mt, slot[2] = status, slot[3] = 0u;
}

__global__ __launch_bounds__(256) void wfa_unpack_pairs_kernel(const KParams P, const uint32_t *__restrict__ words, uint8_t *__restrict__ blob) {
    const uint32_t lane = threadIdx.x & 63u, wi = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wi >= P.chunk_n) return;
    const uint32_t pr = P.work ? P.work[wi] : P.chunk_first + wi;
    const uint32_t len[2] = {P.q_len[pr], P.t_len[pr]};
    const uint64_t off[2] = {P.q_off[pr], P.t_off[pr]};
    if (len[0] == 0u || len[1] == 0u || len[0] > 0x1FFFFFFFu || len[1] > 0x1FFFFFFFu) return;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t nw = (len[q] + 15u) >> 4;
        for (uint32_t j = lane; j < nw; j += 64u) {
            const uint32_t w = words[off[q] / 16u + j];
            uint32_t       o[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) v |= ((0x47544341u >> (8u * ((w >> (2 * (4 * d + b))) & 3u))) & 0xFFu) << (8 * b);  // "ACTG"[code]
                o[d] = v;
            }
            *reinterpret_cast<uint4 *>(blob + off[q] + 16ull * j) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}


// ------------------------------------------------------------------------------------------ C-ABI
extern "C" int wfahip_version(void) { return WFAHIP_VERSION; }

extern "C" const char *wfahip_strerror(int code) {
    switch (code) {
    case WFAHIP_OK: return "ok";
    case WFAHIP_ERR_NO_DEVICE: return "no HIP device available";
    case WFAHIP_ERR_BAD_ARG: return "bad argument";
    case WFAHIP_ERR_OOM: return "out of memory (device or ops buffer)";
    case WFAHIP_ERR_HIP: return "HIP runtime error";
    case WFAHIP_ERR_UNSUPPORTED: return "unsupported penalties (mismatch and gap_open + gap_ext must be > 0) or input";
    case WFAHIP_ERR_INTERNAL: return "internal error";
    }
    return "unknown error";
}

extern "C" int wfahip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_impl(int device_id, wfahip_ctx **out) {
    if (!out) return WFAHIP_ERR_BAD_ARG;
    *out = nullptr;
    int n = wfahip_device_count();
    if (n <= 0) return WFAHIP_ERR_NO_DEVICE;
    int dev = device_id;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return WFAHIP_ERR_NO_DEVICE;
    if (dev >= n) return WFAHIP_ERR_BAD_ARG;
    wfahip_ctx *ctx = new wfahip_ctx();
    ctx->device     = dev;
    if (hipSetDevice(dev) != hipSuccess) {
        delete ctx;
        return WFAHIP_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        ctx->num_cus   = prop.multiProcessorCount;
        ctx->total_mem = prop.totalGlobalMem;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreate(&ctx->evA) != hipSuccess || hipEventCreate(&ctx->evB) != hipSuccess ||
        hipEventCreate(&ctx->evC) != hipSuccess || hipEventCreate(&ctx->evBtA) != hipSuccess ||
        hipEventCreate(&ctx->evBtB) != hipSuccess || hipEventCreateWithFlags(&ctx->ctrl_clean_ev, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&ctx->hpin), HPIN_WORDS * 4, hipHostMallocDefault) != hipSuccess) {
        wfahip_destroy(ctx);  // (releases whichever streams / events were created)
        return WFAHIP_ERR_HIP;
    }
    *out = ctx;
    return WFAHIP_OK;
}

extern "C" int wfahip_create(int device_id, wfahip_ctx **out) { WFAHIP_GUARD(create_impl(device_id, out)) }

extern "C" void wfahip_destroy(wfahip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
    if (ctx->one_pin) (void)hipHostFree(ctx->one_pin);
    if (ctx->pack_pin) (void)hipHostFree(ctx->pack_pin);
    if (ctx->stream_up) (void)hipStreamDestroy(ctx->stream_up);
    if (ctx->stream_dn) (void)hipStreamDestroy(ctx->stream_dn);
    for (hipEvent_t e : ctx->ev_up) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]);
        if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
    }
    for (DevBuf *b : {&ctx->arena, &ctx->fin, &ctx->team_ctl, &ctx->arena2, &ctx->meta2, &ctx->doneq, &ctx->ctrl, &ctx->redo, &ctx->work, &ctx->meta, &ctx->in_blob, &ctx->in_qoff, &ctx->in_qlen,
                      &ctx->in_toff, &ctx->in_tlen, &ctx->out_rec, &ctx->out_ops, &ctx->in_packed, &ctx->in_small, &ctx->prepack, &ctx->one_ctl, &ctx->page_ctl, &ctx->xbuf})
        release(*b);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evA) (void)hipEventDestroy(ctx->evA);
    if (ctx->evB) (void)hipEventDestroy(ctx->evB);
    if (ctx->evC) (void)hipEventDestroy(ctx->evC);
    if (ctx->evBtA) (void)hipEventDestroy(ctx->evBtA);
    if (ctx->evBtB) (void)hipEventDestroy(ctx->evBtB);
    if (ctx->ctrl_clean_ev) (void)hipEventDestroy(ctx->ctrl_clean_ev);
    for (hipEvent_t e : ctx->evpool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    delete ctx;
}

// The options a deployment may set.  Every other key is a routing experiment, a test aid or a knob of one kernel family
// (include/wfa_hip.h lists them): those are refused unless WFAHIP_DEBUG=1 is in the environment -- some change what is safe
// ("team_strict" 0 drops a release the team kernel needs; "fail_pass" injects a failure), none belongs in production.
static bool public_option(const std::string &k) {
    static const char *const keys[] = {"census", "learn", "mem_limit", "arena_budget_pct", "autopack", "pair_fast", "pair_lds"};
    for (const char *p : keys)
        if (k == p) return true;
    return false;
}

static int set_option_impl(wfahip_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return WFAHIP_ERR_BAD_ARG;
    std::string k(key);
    if (!public_option(k)) {
        const char *dbg = std::getenv("WFAHIP_DEBUG");
        if (!dbg || !*dbg || std::strcmp(dbg, "0") == 0) {
            std::snprintf(ctx->last_error, sizeof ctx->last_error, "option \"%.60s\" is a debug / experiment knob: set WFAHIP_DEBUG=1 in the environment to use it", key);
            return WFAHIP_ERR_UNSUPPORTED;
        }
    }
    if (k == "arena_bytes_per_slot")
        ctx->opt_arena_bytes_per_slot = value;
    else if (k == "slots")
        ctx->opt_slots = value;
    else if (k == "threads_per_pair")
        ctx->opt_threads_per_pair = value;
    else if (k == "packed")
        ctx->opt_packed = value;
    else if (k == "reg")
        ctx->opt_reg = value;
    else if (k == "blk")
        ctx->opt_blk = (value == 8 || value == 16) ? value : 0;
    else if (k == "blk_batch")
        ctx->opt_blk_batch = value;
    else if (k == "packed_arena_bytes")
        ctx->opt_packed_arena_bytes = value;
    else if (k == "chunk_pairs")
        ctx->opt_chunk_pairs = value;
    else if (k == "packed_waves_per_cu")
        ctx->opt_packed_waves_per_cu = value;
    else if (k == "overlap")
        ctx->opt_overlap = value;
    else if (k == "tail_overlap")
        ctx->opt_tail_overlap = value;
    else if (k == "blk_wide")
        ctx->opt_blk_wide = value;
    else if (k == "blk_narrow")
        ctx->opt_blk_narrow = value;
    else if (k == "bt_stream")
        ctx->opt_bt_stream = value;
    else if (k == "bt_stream_min")
        ctx->opt_bt_stream_min = value;
    else if (k == "bt_stream_single")
        ctx->opt_bt_stream_single = value;
    else if (k == "bt_stream_wait_us")
        ctx->opt_bt_stream_wait_us = value;
    else if (k == "pilot")
        ctx->opt_pilot = value;
    else if (k == "team_min_len")
        ctx->opt_team_min_len = value;
    else if (k == "team_wgs")
        ctx->opt_team_wgs = value;
    else if (k == "team_solo_max")
        ctx->opt_team_solo_max = value, ctx->opt_team_solo_max_set = true;
    else if (k == "team_wave")
        ctx->opt_team_wave = value;
    else if (k == "team_strict")
        ctx->opt_team_strict = value;
    else if (k == "team_xcd")
        ctx->opt_team_xcd = value;
    else if (k == "unpack_all")
        ctx->opt_unpack_all = value;
    else if (k == "team_paged")
        ctx->opt_team_paged = value;
    else if (k == "team_compact")
        ctx->opt_team_compact = value;
    else if (k == "team_fast")
        ctx->opt_team_fast = value;
    else if (k == "team_pipe")
        ctx->opt_team_pipe = value;
    else if (k == "team_scout")
        ctx->opt_team_scout = value;
    else if (k == "team_order")
        ctx->opt_team_order = value;
    else if (k == "team_slack")
        ctx->opt_team_slack = value > 0 ? value : 1;
    else if (k == "arena_poison")
        ctx->opt_arena_poison = value;
    else if (k == "fail_pass")
        ctx->opt_fail_pass = value;
    else if (k == "prepack")
        ctx->opt_prepack = value;
    else if (k == "narrow_long")
        ctx->opt_narrow_long = value;
    else if (k == "census")
        ctx->opt_census = value;
    else if (k == "long")
        ctx->opt_long = value;
    else if (k == "long_first")
        ctx->opt_long_first = value;
    else if (k == "long_wave_bt_pairs")
        ctx->opt_long_wave_bt_pairs = value;
    else if (k == "long_mid_lone")
        ctx->opt_long_mid_lone = value;
    else if (k == "pair_lds")
        ctx->opt_pair_lds = value, ctx->one_lds_skip = 0;
    else if (k == "long_wave_bt")
        ctx->opt_long_wave_bt = value;
    else if (k == "long_min_len")
        ctx->opt_long_min_len = value;
    else if (k == "long_window_words")
        ctx->opt_long_window_words = (value >= 64 && value <= 4096 && value % 4 == 0) ? value : 240;  // (64 words: a window every ~500 bases -- tests)
    else if (k == "duo")
        ctx->opt_duo = value;
    else if (k == "duo_min_pairs")
        ctx->opt_duo_min_pairs = value;
    else if (k == "compact_call_bases")
        ctx->opt_compact_call_bases = value;
    else if (k == "lane")
        ctx->opt_lane = value;
    else if (k == "lane_pack")
        ctx->opt_lane_pack = value;
    else if (k == "lane_min_pairs")
        ctx->opt_lane_min_pairs = value;
    else if (k == "duo_short")
        ctx->opt_duo_short = value;
    else if (k == "duo_short_min_pairs")
        ctx->opt_duo_short_min_pairs = value;
    else if (k == "blk_mid")
        ctx->opt_blk_mid = value;
    else if (k == "pair_fast")
        ctx->opt_pair_fast = value;
    else if (k == "arena_budget_pct")
        ctx->opt_arena_budget_pct = value;
    else if (k == "autopack")
        ctx->opt_autopack = value;
    else if (k == "learn")
        ctx->opt_learn = value, ctx->learn_key = 0;
    else if (k == "mem_limit") {
        ctx->opt_mem_limit = value;
        hipDeviceProp_t prop;
        if (value > 0)
            ctx->total_mem = (size_t)value;
        else if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess)
            ctx->total_mem = prop.totalGlobalMem;
    }
    else
        return WFAHIP_ERR_BAD_ARG;
    return WFAHIP_OK;
}

extern "C" int wfahip_set_option(wfahip_ctx *ctx, const char *key, int64_t value) { WFAHIP_GUARD(set_option_impl(ctx, key, value)) }

extern "C" int wfahip_last_timing(const wfahip_ctx *ctx, wfahip_timing *out) {
    if (!ctx || !out) return WFAHIP_ERR_BAD_ARG;
    *out = ctx->timing;
    return WFAHIP_OK;
}

extern "C" void wfahip_free(void *p) { std::free(p); }

extern "C" const char *wfahip_last_error(const wfahip_ctx *ctx) { return ctx ? ctx->last_error : ""; }

static int check_params(const wfahip_params *p) {
    if (!p) return WFAHIP_ERR_BAD_ARG;
    // Mismatch == 0: the reference's own loop does not terminate when the first bases differ (the seed is then a
    // Mismatch cell at score 0 whose source M[s - 0][k] is itself).  GapOpen + GapExt == 0: M[s-o-e] is the row being
    // written.  GapExt == 0 alone is aligned (by the generic kernel: the I row of a score becomes a serial scan).
    if (p->mismatch == 0 || p->gap_open + p->gap_ext == 0) return WFAHIP_ERR_UNSUPPORTED;
    if (p->adaptive && p->min_wf_len == 0) return WFAHIP_ERR_BAD_ARG;  // AdaptiveReduction rejects it (wfa.go:134-137)
    return WFAHIP_OK;
}

// Core: everything device-resident.  keep_debug: run with one slot and keep ctrl debug words.
static int align_device_impl(wfahip_ctx *ctx, const wfahip_params *p, const void *d_blob, uint64_t blob_bytes,
                             const void *d_q_off, const void *d_q_len, const void *d_t_off, const void *d_t_len,
                             uint64_t n_pairs, uint32_t max_len, void *d_rec, void *d_ops, uint64_t ops_cap,
                             uint64_t *ops_needed, hipStream_t st, bool debug_single, uint64_t ops_cursor0) {
    // (ops_cursor0: where this call's ops start in d_ops -- the host entry aligns a batch in several calls that
    // share one op buffer)
    int rc = check_params(p);
    if (rc != WFAHIP_OK) return rc;
    if (n_pairs > 0xFFFFFFF0ull) return WFAHIP_ERR_BAD_ARG;
    ctx->timing = wfahip_timing{};
    ctx->dbg_arena = nullptr, ctx->dbg_n = 0;
    if (ctx->bt_pending) {  // a previous call failed half-way: let its backtrace kernel drain before buffers are reused
        (void)hipStreamSynchronize(ctx->stream2);
        ctx->bt_pending = false;
    }
    if (ops_needed) *ops_needed = 0;
    if (n_pairs == 0) return WFAHIP_OK;
    if (!d_q_off || !d_q_len || !d_t_off || !d_t_len || !d_rec || (!d_ops && ops_cap)) return WFAHIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!st) st = ctx->stream;

    if (max_len == 0) {  // compute the bound from the device-resident length arrays
        std::vector<uint32_t> ql(n_pairs), tl(n_pairs);
        HIP_TRY(hipMemcpyAsync(ql.data(), d_q_len, n_pairs * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(tl.data(), d_t_len, n_pairs * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < n_pairs; i++) {
            if (ql[i] <= WFAHIP_MAX_SEQ_LEN) max_len = std::max(max_len, ql[i]);
            if (tl[i] <= WFAHIP_MAX_SEQ_LEN) max_len = std::max(max_len, tl[i]);
        }
        if (max_len == 0) max_len = 1;
    }

    rc = ensure(ctx, ctx->ctrl, CTRL_WORDS * 4);
    if (rc) return rc;
    rc = ensure(ctx, ctx->redo, n_pairs * 8);
    if (rc) return rc;
    uint32_t *d_ctrl = static_cast<uint32_t *>(ctx->ctrl.p);
    // control words (+ the redo list, sorted by pair, when `ent` is given) in one round trip through pinned memory
    // Small calls are bound by the gaps between their few GPU operations (1e5 x 150-base pairs: a third of the call), so they
    // shed what they can: the control words are zeroed once (ctrl_zeroed: nothing has touched them since the memset at
    // the start), the backtrace kernel follows the forward kernel on the same stream instead of waiting for an event on
    // the second one, and the control words fetched behind it serve as the call's final ones when nothing ran after them
    // (ctrl_fresh; hc_last).
    bool     ctrl_zeroed = true, ctrl_fresh = false;
    uint32_t hc_last[CTRL_WORDS] = {0};
    const auto fetch_ctrl = [&](uint32_t *hc, std::vector<uint64_t> *ent) -> int {
        HIP_TRY(hipMemcpyAsync(ctx->hpin + HPIN_CTRL, d_ctrl, CTRL_WORDS * 4, hipMemcpyDeviceToHost, st));
        const size_t head = std::min<size_t>(HPIN_REDO_ENT * 8, ctx->redo.bytes);
        // (the head of the redo list rides along -- unless the class of batches handed nothing on the last time: then it is
        // fetched only if this call did, and a clean call has one copy less to wait for)
        const bool with_head = ent && !ctx->redo_was_empty;
        if (with_head) HIP_TRY(hipMemcpyAsync(ctx->hpin + HPIN_REDO, ctx->redo.p, head, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::memcpy(hc, ctx->hpin + HPIN_CTRL, CTRL_WORDS * 4);
        if (ent) {
            if (!with_head && hc[1] != 0u) {
                HIP_TRY(hipMemcpyAsync(ctx->hpin + HPIN_REDO, ctx->redo.p, head, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            ctx->redo_was_empty = hc[1] == 0u;
            const size_t n = hc[1], nh = std::min<size_t>(n, head / 8);
            ent->assign(n, 0);
            if (nh) std::memcpy(ent->data(), ctx->hpin + HPIN_REDO, nh * 8);
            if (n > nh)
                HIP_TRY(hipMemcpy(ent->data() + nh, static_cast<const char *>(ctx->redo.p) + nh * 8, (n - nh) * 8, hipMemcpyDeviceToHost));
            std::sort(ent->begin(), ent->end(), [](uint64_t a, uint64_t b) { return (uint32_t)a < (uint32_t)b; });
        }
        return WFAHIP_OK;
    };
    // work list of a retry pass -> ctx->work (short lists go through pinned memory; the caller syncs `st` before the next upload)
    const auto upload_work = [&](const uint32_t *ids, size_t n) -> int {
        int r = ensure(ctx, ctx->work, n * 4);
        if (r) return r;
        const void *src = ids;
        if (n <= HPIN_WORK_IDS) {
            std::memcpy(ctx->hpin + HPIN_WORK, ids, n * 4);
            src = ctx->hpin + HPIN_WORK;
        }
        HIP_TRY(hipMemcpyAsync(ctx->work.p, src, n * 4, hipMemcpyHostToDevice, st));
        return WFAHIP_OK;
    };

    KParams P{};
    P.blob = static_cast<const uint8_t *>(d_blob), P.blob_bytes = blob_bytes;
    P.q_off = static_cast<const uint64_t *>(d_q_off), P.q_len = static_cast<const uint32_t *>(d_q_len);
    P.t_off = static_cast<const uint64_t *>(d_t_off), P.t_len = static_cast<const uint32_t *>(d_t_len);
    P.queue_head = d_ctrl + 0;
    P.redo_count = d_ctrl + 1;
    P.ops_cursor = reinterpret_cast<unsigned long long *>(d_ctrl + 2);
    P.debug_info = debug_single ? d_ctrl + 4 : nullptr;
    P.redo_list  = static_cast<uint32_t *>(ctx->redo.p);
    P.x = p->mismatch, P.o = p->gap_open, P.e = p->gap_ext, P.oe = p->gap_open + p->gap_ext;
    P.g                = gcd_u32(gcd_u32(P.x, P.oe), P.e);
    P.global_alignment = p->global_alignment ? 1 : 0;
    P.adaptive = p->adaptive ? 1 : 0, P.min_wf_len = p->min_wf_len, P.max_dist_diff = p->max_dist_diff;
    P.census = ctx->opt_census ? 1u : 0u;
    P.rec = static_cast<uint32_t *>(d_rec);
    P.ops = static_cast<uint64_t *>(d_ops), P.ops_cap = ops_cap;

    // (a call that ended well leaves the control words zeroed for the next one -- a memset the GPU runs while the host is
    // on its way back to the caller, instead of one the first kernel of the next call waits for)
    // (the next call on the SAME stream finds them zero.  On another stream -- the host entry on the context's stream, then the
    // device entry on a caller's -- that memset may still be pending: the new stream waits for it first, or it could land in
    // the middle of this call and reset the pair queue and the ops cursor under the kernels)
    if (ctx->ctrl_clean && ctx->ctrl_clean_stream != st) HIP_TRY(hipStreamWaitEvent(st, ctx->ctrl_clean_ev, 0));
    if (!(ctx->ctrl_clean && ctx->ctrl_clean_stream == st)) HIP_TRY(hipMemsetAsync(d_ctrl, 0, CTRL_WORDS * 4, st));
    ctx->ctrl_clean = false;
    if (ops_cursor0) {
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_ctrl + 2), (int)(uint32_t)ops_cursor0, 1, st));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_ctrl + 3), (int)(uint32_t)(ops_cursor0 >> 32), 1, st));
    }
    HIP_TRY(hipEventRecord(ctx->ev0, st));

    std::deque<Job> jobs;
    std::vector<uint32_t> no_memory;
    std::vector<uint32_t> h_len;   // max(q_len, t_len) per pair, only when the batch mixes short and long pairs
    uint32_t              sub_len_used = 0;
    const int             max_level = 12;
    bool                  first     = true, after_scout = false;

    // ---- pass 1: sub-wave forward kernels + lane-per-pair backtrace kernel, chunk by chunk.
    //      kind 2 = register-window kernel (4 pairs per wave), kind 1 = LDS-ring packed kernel (2 pairs per wave).
    bool packed_done = false;
    if (ctx->opt_packed && !debug_single && ctx->force_mode < 0 && P.global_alignment && P.e != 0u) {
        const uint32_t dx = P.x / P.g, doe = P.oe / P.g, de = P.e / P.g;
        const uint32_t dm = std::max(dx, doe) + 1, di = de + 1;
        // Mixed lengths: the sub-wave kernels keep both sequences of a pair in a few KB of LDS.  When the longest
        // pair of the batch does not fit but most pairs do, the pipeline is sized for the longest pair that fits;
        // the kernels hand the longer ones on themselves (ST_REDO_LDS) and the generic / team kernels take them.
        constexpr uint32_t SUB_LEN_LIMIT = 10200;
        uint32_t           sub_len       = max_len;
        // long reads: the blocked kernels with sliding sequence windows (kinds 11 / 12 / 13 = 64 / 128 / 256 diagonals)
        // the penalty shape the register-ring kernels are instantiated for (wfa_fwd.hpp); -1: none, the LDS-ring kernel takes the batch
        const int  shape    = fwd_shape(dx, doe, de);
        const bool can_long = ctx->opt_long != 0 && ctx->opt_blk == 16 && shape >= 0 && (int64_t)max_len > ctx->opt_long_min_len;
        // (a batch of mostly short pairs with a few long ones keeps the short pairs' pipeline -- slots, arenas and windows of the
        // long instances are sized by the longest pair -- and the long ones get a pass of their own behind it: long_first below)
        if (max_len > SUB_LEN_LIMIT && n_pairs >= 256) {
            std::vector<uint32_t> ql(n_pairs), tl(n_pairs);
            HIP_TRY(hipMemcpyAsync(ql.data(), d_q_len, n_pairs * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(tl.data(), d_t_len, n_pairs * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            uint64_t n_fit = 0;
            uint32_t best  = 0;
            h_len.resize(n_pairs);
            for (uint64_t i = 0; i < n_pairs; i++) {
                const uint32_t l = std::max(ql[i], tl[i]);
                h_len[i]         = l;
                if (l <= SUB_LEN_LIMIT) n_fit++, best = std::max(best, l);
            }
            if (n_fit * 10 >= n_pairs * 9 && best > 0)
                sub_len = best, sub_len_used = best;  // at least 90 % of the pairs fit
            else
                h_len.clear();
        }
        const bool     long_first = can_long && sub_len_used == 0;  // the whole batch starts on the sliding-window instances
        const uint32_t seq_words = (sub_len + 15) / 16 + 1;
        const uint64_t sub_words = packed_sub_lds_words(seq_words, dm, di);
        // what forward_pass sizes slots and arenas by: the class of pairs it is given (the batch's; the long pairs' in their own pass)
        uint32_t fp_seq_words = seq_words;
        const size_t   lds_b     = (size_t)sub_words * 2 * 4;       // packed kernel: two halves
        const size_t   lds_c     = (size_t)seq_words * 2 * 4 * 4;   // register kernel: four rows, sequences only
        const size_t   lds_d     = (size_t)seq_words * 2 * 4 * (ctx->opt_blk == 8 ? 8 : 4) + 16;  // blocked kernel
        const bool     can_b     = lds_b <= 20 * 1024;
        const bool     can_c     = ctx->opt_reg && dx == 2 && doe == 4 && de == 1 && lds_c <= 20 * 1024;
        const bool     can_d     = long_first || (ctx->opt_blk && shape >= 0 && (shape == 0 || ctx->opt_blk == 16) && sub_len < 32768 &&
                                                lds_d <= (ctx->opt_blk == 8 ? 40 : 20) * 1024);
        const auto words_dir_of = [&](uint32_t len) {
            uint64_t wd = std::max<uint64_t>(1024, 8ull * len);  // compact rows: 1 word per diagonal
            if (ctx->opt_packed_arena_bytes > 0) wd = std::max<uint64_t>(1024, ctx->opt_packed_arena_bytes / 4);
            return (wd + 7) & ~7ull;
        };
        uint64_t fp_words_dir = words_dir_of(sub_len);
        P.arena_words   = fp_words_dir;
        P.dx = dx, P.doe = doe, P.de = de, P.dm = dm, P.di = di;
        P.lds_seq_words = seq_words;
        P.sub_lds_words = (uint32_t)sub_words;
        P.min_xe        = std::min(P.x, P.e);
#ifdef WFA_STAMPS
        static DevBuf stampbuf;
        if ((rc = ensure(ctx, stampbuf, 128))) return rc;
        HIP_TRY(hipMemsetAsync(stampbuf.p, 0, 128, st));
        P.debug_info = static_cast<uint32_t *>(stampbuf.p);
#endif
        // one pass over `count` pairs (identity range when list == nullptr); returns the {pair,status} redo entries
        // detach_bt: (first pass, one chunk) the backtrace kernel goes to stream2 and is only waited for at the very
        // end of the call, so the retry passes -- which use the second arena -- run beside it.
        // (workload class of the learned row count: length bucket, penalties, wf-adaptive)
        uint64_t rkey = 0;
        {
            uint32_t lb = 0;
            while ((2u << lb) <= max_len) lb++;
            rkey = 1ull | ((uint64_t)lb << 1) | ((uint64_t)P.adaptive << 9) | ((uint64_t)(P.x & 0xFFF) << 12) | ((uint64_t)(P.oe & 0xFFF) << 24) |
                   ((uint64_t)(P.e & 0xFFF) << 36) | ((uint64_t)(P.max_dist_diff & 0xFFFF) << 48);
        }
        uint64_t arena_mult = (ctx->rows_key == rkey && ctx->opt_packed_arena_bytes <= 0) ? ctx->rows_scale : 1;  // rows per pair, in units of the default
        auto forward_pass = [&](int kind, const std::vector<uint32_t> *list, uint64_t first_pair, uint64_t count,
                                std::vector<uint64_t> &redo_out, bool detach_bt) -> int {
            // (retry passes take the second pair of buffers: the first pass's backtrace may still be reading the first -- and
            // wfahip_debug_compact_arena shows what the first pass left.  Unless the first pair is large and free: two large
            // arenas side by side are for overlap, not for a snapshot)
            const uint32_t seq_words = fp_seq_words;  // (shadow the batch's: this pass's class of pairs)
            const uint64_t words_dir = fp_words_dir;
            const bool second = ctx->bt_pending || (list && ctx->arena.bytes <= ctx->total_mem / 10);
            DevBuf &arena_buf = second ? ctx->arena2 : ctx->arena;
            DevBuf &meta_buf  = second ? ctx->meta2 : ctx->meta;
            // short reads: the blocked kernel stages BLK_BATCH pairs per group at a time
            // (kind 6: eight pairs per wave, 32-diagonal window; only with the batched refill)
            const bool     blk_batch    = (kind == 3 || kind == 6) && seq_words <= 16 && ctx->opt_blk_batch != 0;
            if (kind == 6 && !blk_batch && ctx->opt_narrow_long == 0) return WFAHIP_ERR_INTERNAL;
            if (ctx->opt_fail_pass == kind) return WFAHIP_ERR_OOM;  // (fault injection, tests only)
            // kind 8 (wfa_duo_kernel): sequences come pre-packed, slot = 4 header words + 2 x (even) words per sequence
            const uint32_t duo_sw       = (seq_words + 1u) & ~1u, duo_pw = 4u + 2u * duo_sw;
            if (kind == 8 && (duo_pw > 256u || blk_batch)) return WFAHIP_ERR_INTERNAL;
            if (kind == 10 && (seq_words > (uint32_t)LN_SEQ_WORDS || list)) return WFAHIP_ERR_INTERNAL;
            const uint32_t lane_sw      = (seq_words + 1u) & ~1u;  // kind 10 (wfa_lane_kernel): words per sequence in its slots and in LDS
            // kinds 11 / 12 / 13: kinds 3 / 9 / 5 with sliding sequence windows (long reads): pre-packed slots of long_sw words per
            // sequence in HBM, long_cw words of each in LDS
            // kinds 14 / 15: the whole wave on ONE pair with one / two diagonals per lane (64 / 128 diagonals): batches too small to fill the GPU
            const bool     is_long      = kind >= 11 && kind <= 15;
            const uint32_t long_sw      = (seq_words + 3u) & ~3u, long_cw = (uint32_t)ctx->opt_long_window_words;
            if (is_long && !can_long) return WFAHIP_ERR_INTERNAL;
            if (is_long && (uint64_t)seq_words * 16u < (uint64_t)max_len && !list) return WFAHIP_ERR_INTERNAL;  // (slots sized for the short class)
            const int      bkind        = (kind == 11 || kind == 14) ? 3 : (kind == 12 || kind == 15) ? 9 : kind == 13 ? 5 : kind;  // (arena format)
            const size_t   lds_bytes    = is_long ? (size_t)(kind == 11 ? 4 : kind == 12 ? 2 : 1) * 2 * long_cw * 4
                                          : kind == 10 ? (size_t)64 * lane_stride_words(lane_sw) * 4
                                          : kind == 8 ? (size_t)duo_lds_words(duo_pw) * 4
                                          : kind == 9 ? (size_t)seq_words * 2 * 4 * 2 + 16
                                          : blk_batch ? (size_t)(kind == 6 ? 8 : 4) * BLK_BATCH * (2 * seq_words + 8) * 4 + 16
                                          : kind == 6 ? (size_t)seq_words * 2 * 4 * 8 + 16
                                          : kind == 5 ? (size_t)seq_words * 2 * 4 + 16
                                                      : (kind >= 3 ? lds_d : (kind == 2 ? lds_c : lds_b));
            const uint32_t pairs_wave   = kind >= 13 ? 1 : kind == 10 ? 64 : bkind == 5 ? 1 : bkind == 9 ? 2 : (kind == 4 || kind == 6 || kind == 8 ? 8 : (kind >= 2 ? 4 : 2));
            // blocked kernels: fixed-pitch arena, no directory.  64-diagonal window: 16 words per base = 250 scores at
            // 1 kbp; 256-diagonal window (kind 5, the retry rung): 128 words per base = 500 scores at 1 kbp
            const uint64_t words        = bkind == 5   ? std::max<uint64_t>((words_dir * 16 * arena_mult + 511) & ~511ull, 8192)
                                          : bkind == 9 ? std::max<uint64_t>((words_dir * 4 * arena_mult + 511) & ~511ull, 4096)
                                          : kind == 8 ? std::max<uint64_t>((words_dir * arena_mult + 511) & ~511ull, 1024)  // 16-bit words
                                          : kind == 10 ? std::max<uint64_t>((words_dir * arena_mult + 511) & ~511ull, 1024)  // rows of 32 x 16 bit
                                          : kind >= 3 ? std::max<uint64_t>((words_dir * 2 * arena_mult + 511) & ~511ull, 2048)
                                                      : words_dir;
            P.arena_words = words, P.compact_fmt = kind == 10 ? 8u : kind == 8 ? DUO_ARENA_FMT : kind == 6 ? 5u : bkind == 5 ? 4u : bkind == 9 ? 6u : (kind >= 3 ? (WFA_BLK_TILED ? 3u : 1u) : 0u);
            const uint32_t waves_lds    = (uint32_t)std::min<size_t>(32, LDS_MAX_BYTES / lds_bytes);
            const bool     overlap      = ctx->opt_overlap != 0;
            uint32_t       waves_per_cu = is_long ? std::min<uint32_t>(waves_lds, (kind == 11 && !P.census) ? 4 * WFA_BLK_WAVES : 16)
                                          : kind == 10 ? std::min<uint32_t>(waves_lds, 8)
                                          : kind == 8 ? std::min<uint32_t>(waves_lds, 4 * WFA_DUO_WAVES)
                                          : kind == 4 ? std::min<uint32_t>(waves_lds, 12)
                                          : kind >= 2 ? std::min<uint32_t>(waves_lds, 20)
                                                      : (overlap ? std::min<uint32_t>(waves_lds, 24) : waves_lds);
            if (ctx->opt_packed_waves_per_cu > 0)
                waves_per_cu = std::min<uint32_t>(waves_lds, (uint32_t)ctx->opt_packed_waves_per_cu);
            // Chunking: every pair of a chunk owns an arena until its backtrace has run.  Two chunk buffers
            // alternate so the (latency-bound) backtrace kernel of chunk c runs on a second stream beside the
            // (issue-bound) forward kernel of chunk c+1.
            const uint64_t resident = (uint64_t)pairs_wave * ctx->num_cus * waves_per_cu;  // pairs in flight
            uint64_t budget = (uint64_t)((double)ctx->total_mem * 0.35) / (overlap ? 2 : 1);
            uint64_t chunk  = std::max<uint64_t>(1, std::min<uint64_t>(count, budget / (words * 4ull)));
            if (overlap && count >= 8 * resident) chunk = std::min<uint64_t>(chunk, (count + 7) / 8);
            if (ctx->opt_chunk_pairs > 0) chunk = std::min<uint64_t>(chunk, (uint64_t)ctx->opt_chunk_pairs);
            const uint64_t n_chunks = (count + chunk - 1) / chunk;
            const uint32_t n_buf    = (overlap && n_chunks > 1) ? 2 : 1;
            // (retry passes: how many pairs are handed on varies a little from call to call -- which pairs share a wave
            // is a matter of timing -- so their buffers get a quarter of headroom instead of being re-allocated, tens
            // of milliseconds for a few GB, whenever a call needs a few pairs more than the one before)
            // (the headroom stays inside the budget the chunk was sized by, and an allocation that fails with it is tried again
            // at the exact size: a budget-bound retry pass must not fail where the plain size would have fitted)
            const uint64_t chunk_alloc = list ? std::max<uint64_t>(chunk, std::min<uint64_t>(chunk + chunk / 4 + 64, budget / (words * 4ull))) : chunk;
            int rc2 = WFAHIP_OK;
            if (arena_buf.bytes < (size_t)(words * 4ull * chunk * n_buf)) {
                rc2 = ensure(ctx, arena_buf, (size_t)(words * 4ull * chunk_alloc * n_buf));
                if (rc2 == WFAHIP_ERR_OOM && chunk_alloc > chunk) rc2 = ensure(ctx, arena_buf, (size_t)(words * 4ull * chunk * n_buf));
            }
            if (rc2) return rc2;
            if (meta_buf.bytes < chunk * 16 * n_buf && (rc2 = ensure(ctx, meta_buf, chunk_alloc * 16 * n_buf))) return rc2;
            detach_bt = detach_bt && n_chunks == 1 && ctx->opt_tail_overlap != 0 &&
                        (count * (uint64_t)max_len > (uint64_t)ctx->opt_compact_call_bases || ctx->opt_compact_call_bases <= 0);
            ctrl_fresh = false;
            // streamed backtrace: a few waves walk finished pairs while the forward kernel is still running
            // (off unless asked for since round 2: with the forward pass at 20 ms per 1e6 pairs the write-through row
            // stores of the streaming instance cost more than the backtrace kernel they save -- 3e6 x 1 kbp pairs in two
            // chunks: 65.2 ms with a backtrace kernel per chunk, 70.5 ms streamed)
            const bool stream_bt = kind == 3 && shape == 0 && !blk_batch && n_buf == 1 && ctx->opt_bt_stream > 0 && (int64_t)chunk >= ctx->opt_bt_stream_min &&
                                   ctx->opt_bt_stream_single != 0;
            P.done_q = nullptr, P.done_ctl = nullptr, P.n_stream_wgs = 0;
            if (stream_bt) {
                if ((rc2 = ensure(ctx, ctx->doneq, 256 + 16 * chunk))) return rc2;
                P.done_ctl = static_cast<uint32_t *>(ctx->doneq.p);
                P.done_q   = reinterpret_cast<uint4 *>(static_cast<char *>(ctx->doneq.p) + 256);
            }
            ctx->timing.arena_bytes = std::max<uint64_t>(ctx->timing.arena_bytes, words * 4ull * chunk * n_buf);
            if (list && (rc2 = upload_work(list->data(), count))) return rc2;
            while (ctx->evpool.size() < 4 * n_chunks) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                ctx->evpool.push_back(e);
            }
            if (!ctrl_zeroed) HIP_TRY(hipMemsetAsync(d_ctrl, 0, 8, st));  // queue_head, redo_count
            ctrl_zeroed = false;
            hipStream_t st_bt = (n_buf == 2 || detach_bt) ? ctx->stream2 : st;
            for (uint64_t c = 0; c < n_chunks; c++) {
                const uint64_t c0 = c * chunk, cn = std::min<uint64_t>(chunk, count - c0);
                hipEvent_t evFa = ctx->evpool[4 * c], evFb = ctx->evpool[4 * c + 1];
                hipEvent_t evBa = detach_bt ? ctx->evBtA : ctx->evpool[4 * c + 2];
                hipEvent_t evBb = detach_bt ? ctx->evBtB : ctx->evpool[4 * c + 3];
                const uint32_t buf = (uint32_t)(c % n_buf);
                P.arena       = static_cast<uint32_t *>(arena_buf.p) + (uint64_t)buf * chunk * words;
                P.pair_meta   = static_cast<uint4 *>(meta_buf.p) + (uint64_t)buf * chunk;
                P.chunk_first = (uint32_t)(first_pair + c0), P.chunk_n = (uint32_t)cn;
                P.work        = list ? static_cast<const uint32_t *>(ctx->work.p) + c0 : nullptr;
                if (!list && c == 0 && first_pair == 0)
                    ctx->dbg_arena = P.arena, ctx->dbg_meta = P.pair_meta, ctx->dbg_words = words, ctx->dbg_first = first_pair,
                    ctx->dbg_n = cn, ctx->dbg_fmt = P.compact_fmt, ctx->dbg_g = P.g;
                uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)ctx->num_cus * waves_per_cu,
                                                            (cn + pairs_wave - 1) / pairs_wave);
                const uint32_t n_bt = stream_bt ? (uint32_t)std::min<int64_t>(ctx->opt_bt_stream, grid / 4) : 0u;
                if (stream_bt) {
                    // the first n_bt workgroups of the launch only backtrace; they take the place of forward waves when
                    // the launch fills the GPU
                    if (grid < (uint32_t)ctx->num_cus * waves_per_cu) grid = std::min<uint32_t>(grid + n_bt, (uint32_t)ctx->num_cus * waves_per_cu);
                    P.n_stream_wgs = n_bt;
                    P.stream_wait  = (uint32_t)std::min<int64_t>(std::max<int64_t>(ctx->opt_bt_stream_wait_us, 0) * 100, 0x7FFFFFFF);
                    HIP_TRY(hipMemsetAsync(ctx->doneq.p, 0, 256 + 16 * cn, st));
                    HIP_TRY(hipMemsetAsync(P.pair_meta, 0xFF, 16 * cn, st));  // ST_PENDING: only pairs without a backtrace get a status
                }
                // the chunk's sequences 2-bit packed up front (unbatched 16-lane first pass over a range of pairs)
                P.prepack = nullptr, P.prepack_words = 0;
                P.lds_seq_words = is_long ? long_sw : kind == 10 ? lane_sw : kind == 8 ? duo_sw : seq_words;
                if (kind == 10) P.sub_lds_words = lane_stride_words(lane_sw);
                if ((kind == 3 && !blk_batch && !list && ctx->opt_prepack != 0) || kind == 8 || (kind == 10 && ctx->opt_lane_pack == 0) || is_long) {
                    const uint32_t pw = 4u + 2u * P.lds_seq_words;
                    if ((rc2 = ensure(ctx, ctx->prepack, (size_t)chunk * pw * 4))) return rc2;
                    if (ctx->pk_words)  // (packed input: the slots are copied from the uploaded words, no bytes in between)
                        hipLaunchKernelGGL(wfa_prepack_words_kernel, dim3((uint32_t)((cn + 3) / 4)), dim3(256), 0, st, P, ctx->pk_words,
                                           static_cast<uint32_t *>(ctx->prepack.p), P.lds_seq_words, pw);
                    else {
                        // (a chunk that gives fewer than four waves per SIMD at four pairs per wave: a wave per pair; short slots pack
                        // two pairs side by side in a wave: they keep four)
                        const uint32_t ppw = (cn <= (uint64_t)ctx->num_cus * 64 && 2u * P.lds_seq_words > 32u) ? 1u : (uint32_t)PREPACK_PAIRS;
                        hipLaunchKernelGGL(wfa_prepack_kernel, dim3((uint32_t)((cn + 4 * ppw - 1) / (4 * ppw))), dim3(256), 0, st, P,
                                           static_cast<uint32_t *>(ctx->prepack.p), P.lds_seq_words, pw, ppw);
                    }
                    HIP_TRY(hipGetLastError());
                    P.prepack = static_cast<const uint32_t *>(ctx->prepack.p), P.prepack_words = pw;
                } else if (ctx->pk_words) {  // (packed input, a pass that reads bytes: exactly its pairs are expanded first)
                    hipLaunchKernelGGL(wfa_unpack_pairs_kernel, dim3((uint32_t)((cn + 3) / 4)), dim3(256), 0, st, P, ctx->pk_words,
                                       const_cast<uint8_t *>(P.blob));
                    HIP_TRY(hipGetLastError());
                }
                if (is_long) P.lds_seq_words = long_cw;  // (the forward kernel's sequence windows; the slots hold long_sw words per sequence)
                if (n_buf == 2 && c >= 2) HIP_TRY(hipStreamWaitEvent(st, ctx->evpool[4 * (c - 2) + 3], 0));  // buffer free
                if (c > 0) HIP_TRY(hipMemsetAsync(d_ctrl, 0, 4, st));  // queue_head only (chunk 0: cleared with redo_count above)
                // (tests: the sub-wave kernels zero nothing -- no word the backtrace reads may be one they did not write)
                if (ctx->opt_arena_poison) HIP_TRY(hipMemsetAsync(P.arena, 0xA5, (size_t)(words * 4ull * cn), st));
                HIP_TRY(hipEventRecord(evFa, st));
                if (kind == 8) {
                    HIP_TRY(wfa_launch_duo(shape, P, grid, lds_bytes, st, P.census != 0));
                } else if (kind >= 3) {
                    uint32_t fl = (P.census ? (uint32_t)FWD_CENSUS : 0u) | (P.adaptive ? (uint32_t)FWD_ADAPTIVE : 0u);
                    if (kind == 3 && stream_bt) fl |= FWD_STREAM;
                    if (blk_batch) {
                        // entries per grab: the chunk's share of one resident group, cut into the fewest rounds of <= 8
                        const uint64_t groups = (uint64_t)ctx->num_cus * 16 * pairs_wave;  // 4 waves per SIMD x 4 (8) pairs
                        const uint64_t share  = (cn + groups - 1) / groups;
                        const uint64_t rounds = (share + BLK_BATCH - 1) / BLK_BATCH;
                        P.blk_batch_n = (uint32_t)std::min<uint64_t>(BLK_BATCH, std::max<uint64_t>(1, (share + rounds - 1) / std::max<uint64_t>(1, rounds)));
                        if (ctx->opt_blk_batch > 1) P.blk_batch_n = (uint32_t)std::min<int64_t>(BLK_BATCH, ctx->opt_blk_batch);
                        fl |= FWD_BATCH;
                    }
                    // (instances without the census of stored words exist where a first pass runs them: the others count always)
                    HIP_TRY(wfa_launch_fwd(shape, kind, fl, P, grid, lds_bytes, st));
                } else if (kind == 2)
                    hipLaunchKernelGGL((wfa_reg_kernel<2, 4, 1>), dim3(grid), dim3(64), lds_bytes, st, P);
                else
                    hipLaunchKernelGGL(wfa_packed_kernel, dim3(grid), dim3(64), lds_bytes, st, P);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(evFb, st));
                if (st_bt != st) HIP_TRY(hipStreamWaitEvent(st_bt, evFb, 0));
                if (st_bt != st) HIP_TRY(hipEventRecord(evBa, st_bt));  // (same stream: the backtrace starts where the forward kernel ends)
                // long pairs, up to two waves per SIMD of them: a wave per pair walks an LDS region of the arena (3.2 against 5.7 ms
                // for 500 x 50 kbp).  Beyond that the lane-per-pair kernel wins: the walk is bound by instructions, and there one
                // instruction serves 64 pairs (2e4 x 50 kbp: 15 against 47 ms)
                if (is_long && (ctx->opt_long_wave_bt >= 2 || (ctx->opt_long_wave_bt == 1 && cn <= (ctx->opt_long_wave_bt_pairs ? (uint64_t)ctx->opt_long_wave_bt_pairs : (uint64_t)ctx->num_cus * 20))))
                    hipLaunchKernelGGL(wfa_backtrace_wave_kernel, dim3((uint32_t)((cn + BTW_WAVES - 1) / BTW_WAVES)), dim3(64 * BTW_WAVES), 0, st_bt, P);
                else
                    hipLaunchKernelGGL(wfa_backtrace_kernel, dim3((uint32_t)((cn + BT_THREADS - 1) / BT_THREADS)), dim3(BT_THREADS), 0, st_bt, P);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(evBb, st_bt));
            }
            if (st_bt != st && !detach_bt)
                for (uint64_t c = (n_chunks >= 2 ? n_chunks - 2 : 0); c < n_chunks; c++)
                    HIP_TRY(hipStreamWaitEvent(st, ctx->evpool[4 * c + 3], 0));
            uint32_t hc[CTRL_WORDS];
            if (!detach_bt) HIP_TRY(hipEventRecord(ctx->ev1, st));  // (the end of the call if nothing follows)
            if ((rc2 = fetch_ctrl(hc, &redo_out))) return rc2;  // (detach_bt: the forward kernel is done, the backtrace may still run)
            if (detach_bt) ctx->bt_pending = true;
            else ctrl_fresh = true, std::memcpy(hc_last, hc, sizeof hc_last);
            for (uint64_t c = 0; c < n_chunks; c++) {
                float msF = 0, msB = 0;
                HIP_TRY(hipEventElapsedTime(&msF, ctx->evpool[4 * c], ctx->evpool[4 * c + 1]));
                if (!detach_bt) HIP_TRY(hipEventElapsedTime(&msB, ctx->evpool[4 * c + (st_bt != st ? 2 : 1)], ctx->evpool[4 * c + 3]));
                ctx->timing.kernel_ms += msF + msB;
                if (!list) ctx->timing.main_kernel_ms += msF, ctx->timing.n_main_launches++;
                ctx->timing.n_launches += 2;
            }
            P.work = nullptr;
            P.lds_seq_words = seq_words;
            P.sub_lds_words = (uint32_t)sub_words;
            return WFAHIP_OK;
        };

        if (can_b || can_c || can_d) {
            std::vector<uint64_t> redo1, redo2;
            // short reads (< 200 bases, batched refill): eight pairs per wave in 32-diagonal windows first; what outgrows
            // them retries on the 16-lane instance below
            const bool narrow1 = can_d && ctx->opt_blk == 16 && ctx->opt_blk_narrow != 0 &&
                                 ((ctx->opt_blk_batch != 0 && max_len < 200 && seq_words <= 16) ||
                                  (ctx->opt_narrow_long != 0 && shape == 0 && (size_t)seq_words * 2 * 4 * 8 + 16 <= 8 * 1024));
            // reads of 240+ bases: the variable-lanes kernel (its slots hold at most 126 packed words per sequence)
            const bool duo_long  = !narrow1 && seq_words > 16 &&
                                   (ctx->opt_duo >= 2 || (ctx->opt_duo == 1 && (int64_t)n_pairs >= ctx->opt_duo_min_pairs));
            const bool duo_short = seq_words <= 16 && ctx->opt_duo != 0 &&
                                   (ctx->opt_duo_short >= 2 || (ctx->opt_duo_short == 1 && (int64_t)n_pairs >= ctx->opt_duo_short_min_pairs));
            const bool duo1    = can_d && ctx->opt_blk == 16 && (duo_long || duo_short) && 4u + 2u * ((seq_words + 1u) & ~1u) <= 256u;
            // short reads (at most 240 bases): a lane per pair
            const bool lane1   = can_d && ctx->opt_blk == 16 && seq_words <= (uint32_t)LN_SEQ_WORDS &&
                                 (ctx->opt_lane >= 2 || (ctx->opt_lane == 1 && (int64_t)n_pairs >= ctx->opt_lane_min_pairs));
            const int  kind1   = long_first ? 11 : (duo1 && duo_short) ? 8 : lane1 ? 10 : duo1 ? 8 : narrow1 ? 6 : (can_d ? (ctx->opt_blk == 8 ? 4 : 3) : (can_c ? 2 : 1));
            // the rungs above the 64-diagonal first pass: 128 and 256 diagonals (long reads: the same with sliding sequence windows)
            const int  kind_mid = long_first ? 12 : 9, kind_wide = long_first ? 13 : 5, kind_64 = long_first ? 11 : 3;
            // Pilot: on a large batch with wf-adaptive off the first 4 096 pairs go first.  When most of them leave the
            // 64-diagonal window the rest does not start there only to be handed on: it goes straight to the
            // wave-per-pair kernel (256 diagonals) if that one takes most of the pilot's leftovers, else to the
            // generic ladder.  (wf-adaptive or short reads: narrow bands, no pilot.)
            uint64_t done_pairs = 0;
            bool     skip_rest  = false;
            int      kind_rest  = kind1;
            const bool wide_ok  = kind1 >= 3 && ctx->opt_blk_wide != 0;
            std::vector<uint64_t> redo_w;  // handed on by the 256-diagonal kernel, or not eligible for it
            // band failures of `from` -> kind 9 (128 diagonals, two pairs per wave; unless `from` comes from there), its band
            // failures -> kind 5 (256 diagonals, a wave per pair); everything else -> redo_w
            uint64_t mid_in = 0, mid_fail = 0;  // what the 128-diagonal instance was given / handed on (learned routing below)
            const auto wide_pass = [&](std::vector<uint64_t> &from, bool from_mid = false) -> int {
                std::vector<uint32_t> lst;
                for (uint64_t e : from) {
                    if ((uint32_t)(e >> 32) == ST_REDO_BAND) lst.push_back((uint32_t)e);
                    else redo_w.push_back(e);
                }
                from.clear();
                if (lst.empty()) return 0;
                const size_t n_in = lst.size();
                std::vector<uint64_t> r2;
                if (!from_mid && ctx->opt_blk_mid != 0) {
                    // (long reads, leftovers that are a few waves per SIMD anyway: a wave per pair with two diagonals per lane steps
                    // in half the instructions of two pairs per wave with four -- 2e4 x 50 kbp: 53.1 -> 47.4 ms per step, the 4 186 leftovers' pass 20.6 -> ~15 ms)
                    const int km = (long_first && ctx->opt_long_mid_lone != 0 && lst.size() <= 6ull * (uint64_t)ctx->num_cus * 4ull) ? 15 : kind_mid;
                    const int rcm = forward_pass(km, &lst, 0, lst.size(), r2, false);
                    if (rcm) return rcm;
                    ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - r2.size());
                    mid_in += lst.size();
                    lst.clear();
                    for (uint64_t e : r2) {
                        if ((uint32_t)(e >> 32) == ST_REDO_BAND) lst.push_back((uint32_t)e);
                        else redo_w.push_back(e);
                    }
                    mid_fail += lst.size();
                    r2.clear();
                    if (lst.empty()) return 1;
                }
                const int rcw = forward_pass(kind_wide, &lst, 0, lst.size(), r2, false);
                if (rcw) return rcw;  // WFAHIP_ERR_* (negative): the whole call fails, no pair is silently dropped
                ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - r2.size());
                redo_w.insert(redo_w.end(), r2.begin(), r2.end());
                return r2.size() * 2 > n_in ? 2 : 1;  // 2: the wide kernels do not take most of them either
            };
            uint64_t n_first_fail = 0;  // pairs the first-pass kernel(s) handed on
            if (n_pairs >= 65536 && !P.adaptive && max_len >= 200 && ctx->opt_pilot != 0) {  // (the pilot costs one extra pass, ~0.1 ms)
                const uint64_t pilot = 4096;
                if ((rc = forward_pass(kind1, nullptr, 0, pilot, redo1, false))) return rc;
                done_pairs   = pilot;
                n_first_fail = redo1.size();
                if (redo1.size() * 2 > pilot) {
                    const int wv = wide_ok ? wide_pass(redo1) : 2;
                    if (wv < 0) return wv;
                    if (wv == 1) kind_rest = kind_wide;
                    else skip_rest = true;
                }
            }
            // a class of batches whose pairs were mostly handed on for their band the last time starts on the window that
            // took them (1 kbp at 20 % error: every pair needs ~100 diagonals)
            // (band_kind is kept in terms of the plain instances -- 3 / 9 / 5 -- and mapped to this call's rungs)
            if (wide_ok && done_pairs == 0 && ctx->band_key == rkey && ctx->band_kind != 0 && (kind1 == 3 || kind1 == 8 || kind1 == 11) &&
                (ctx->band_kind == 5 || ctx->band_kind == 3 || ctx->opt_blk_mid != 0) &&
                !(ctx->band_kind == 3 && ctx->opt_duo >= 2) &&  // (option duo = 2: the variable-lanes kernel whatever was learned)
                (++ctx->band_calls & 15u) != 0u)
                kind_rest = ctx->band_kind == 5 ? kind_wide : ctx->band_kind == 9 ? kind_mid : kind_64;
            // Long reads in batches too small to fill the GPU start on the wider windows: 500 pairs are 500 waves of one pair each
            // (a wave alone on its SIMD steps in the same ~1.5 us whether it holds one pair or four), and nothing is handed on for
            // its band -- a fifth of 50 kbp pairs at 5 % error leave a 64-diagonal window at some score
            if (long_first && done_pairs == 0 && kind_rest == 11 && wide_ok) {
                const uint64_t simds = (uint64_t)ctx->num_cus * 4;
                if (ctx->opt_long_first >= 11 && ctx->opt_long_first <= 15) kind_rest = (int)ctx->opt_long_first;
                // (a wave per pair, two diagonals per lane: 128 diagonals at half the instructions of a step.  3 000 / 6 000 pairs of
                // 50 kbp: 14.2 / 23.1 ms against 18.1 / 25.3 with two pairs per wave and 26.6 with four; 2e4 pairs are issue-bound and
                // four pairs per wave share a step's instructions)
                else if (n_pairs <= 6 * simds) kind_rest = 15;
            }
            if (!skip_rest) {
                std::vector<uint64_t> more;
                if ((rc = forward_pass(kind_rest, nullptr, done_pairs, n_pairs - done_pairs, more, true))) return rc;
                n_first_fail += more.size();
                uint64_t n_band = 0;
                for (uint64_t e : more) n_band += (uint32_t)(e >> 32) == ST_REDO_BAND;
                if (kind_rest == kind_mid || kind_rest == 15) {
                    mid_in += n_pairs - done_pairs, mid_fail += n_band;
                    if ((rc = wide_pass(more, true)) < 0) return rc;  // -> the 256-diagonal instance
                } else {
                    (kind_rest == kind_wide ? redo_w : redo1).insert((kind_rest == kind_wide ? redo_w : redo1).end(), more.begin(), more.end());
                }
                if ((kind_rest == 3 || kind_rest == 8 || kind_rest == 11) && P.adaptive && n_band * 2 > n_pairs - done_pairs) ctx->band_key = rkey, ctx->band_kind = 9;
                else if (kind_rest == 15 && P.adaptive && n_band * 2 > n_pairs - done_pairs) ctx->band_key = rkey, ctx->band_kind = 5;  // (128 diagonals were not enough)
                // (bands mostly wider than 32 diagonals: the variable-lanes kernel then runs its pairs wide, parks and resumes for
                // nothing and hands on more than the plain 64-diagonal kernel would -- 1e6 x 1 kbp @8 %: 53-71 ms against 46)
                else if (kind_rest == 8 && n_band * 50 > n_pairs - done_pairs) ctx->band_key = rkey, ctx->band_kind = 3;
                else if (kind_rest == 8 && ctx->band_key == rkey) ctx->band_kind = 0;
                else if (kind_rest == kind_64 && kind1 == kind_64 && ctx->band_key == rkey) ctx->band_kind = 0;
                done_pairs = n_pairs;
            }
            if (std::getenv("WFAHIP_DEBUG_TIMING") && P.done_ctl) {
                uint32_t dc[2];
                HIP_TRY(hipStreamSynchronize(ctx->stream2));
                HIP_TRY(hipMemcpy(dc, P.done_ctl, sizeof dc, hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[wfahip] streamed backtrace: %u entries pushed, %u ticketed\n", dc[0], dc[1]);
            }
            if (std::getenv("WFAHIP_DEBUG_TIMING")) {
                uint64_t cnt[4] = {0, 0, 0, 0};
                for (uint64_t e : redo1) cnt[std::min<uint32_t>(3, (uint32_t)(e >> 32) - ST_REDO_BYTES)]++;
                std::fprintf(stderr, "[wfahip] handed on by the first pass: bytes %llu, arena %llu, lds %llu, band %llu\n",
                             (unsigned long long)cnt[0], (unsigned long long)cnt[1], (unsigned long long)cnt[2],
                             (unsigned long long)cnt[3]);
            }
            ctx->timing.main_kernel_kind = (uint32_t)kind_rest;
            ctx->timing.n_packed_pairs += (uint32_t)(done_pairs - n_first_fail);
            ctx->timing.n_retried_pairs += (uint32_t)n_first_fail;
            // Mixed lengths: the long pairs the short pairs' kernels handed on for their length (ST_REDO_LDS) take the
            // sliding-window instances now -- slots, windows and arenas sized for the batch's longest pair -- by their number:
            // a wave per pair up to one per SIMD; what leaves its band there tries 256 diagonals; the rest goes down the ladder
            if (can_long && sub_len_used != 0) {
                std::vector<uint32_t> lst;
                std::vector<uint64_t> keep, r2;
                for (uint64_t e : redo1) ((uint32_t)(e >> 32) == ST_REDO_LDS ? (void)lst.push_back((uint32_t)e) : (void)keep.push_back(e));
                if (!lst.empty()) {
                    std::sort(lst.begin(), lst.end());
                    const uint32_t keep_sw = fp_seq_words;
                    const uint64_t keep_wd = fp_words_dir, simds = (uint64_t)ctx->num_cus * 4;
                    fp_seq_words = (max_len + 15) / 16 + 1, fp_words_dir = words_dir_of(max_len);
                    const int kl = lst.size() <= simds ? 15 : lst.size() <= 4 * simds ? 12 : 11;
                    rc = forward_pass(kl, &lst, 0, lst.size(), r2, false);
                    if (rc == WFAHIP_OK) {
                        ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - r2.size());
                        lst.clear();
                        for (uint64_t e : r2) ((uint32_t)(e >> 32) == ST_REDO_BAND ? (void)lst.push_back((uint32_t)e) : (void)keep.push_back(e));
                        r2.clear();
                        if (!lst.empty() && (rc = forward_pass(13, &lst, 0, lst.size(), r2, false)) == WFAHIP_OK) {
                            ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - r2.size());
                            keep.insert(keep.end(), r2.begin(), r2.end());
                        }
                    }
                    fp_seq_words = keep_sw, fp_words_dir = keep_wd;
                    if (rc) return rc;
                    redo1.swap(keep);
                }
            }
            Job jb, ja;
            jb.mode = 1, jb.level = 0, jb.all = false;
            ja.mode = 0, ja.level = 0, ja.all = false;
            // A few thousand leftovers finish sooner as one generic launch (one wave per pair, all of them resident at
            // once, backtrace included) than through another forward + backtrace pass; beyond that the LDS-ring
            // kernel's throughput wins.
            const uint64_t resident_generic = (uint64_t)ctx->num_cus * 32;
            if (kind1 == 6 || kind1 == 10) {  // band / arena failures of the 32-diagonal instance -> the 64-diagonal one.  (What the variable-lanes
                               // kernel hands on -- a band wider than a row, rarely no park record free: 0.08 % of 1 kbp pairs --
                               // goes straight to the 128-diagonal instance below: one retry pass instead of two.)
                std::vector<uint32_t> lst;
                std::vector<uint64_t> keep, r2;
                for (uint64_t e : redo1) {
                    const uint32_t stw = (uint32_t)(e >> 32);
                    if (stw == ST_REDO_BAND || stw == ST_REDO_ARENA) lst.push_back((uint32_t)e);
                    else keep.push_back(e);
                }
                if (!lst.empty()) {
                    if ((rc = forward_pass(3, &lst, 0, lst.size(), r2, false))) return rc;
                    ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - r2.size());
                    keep.insert(keep.end(), r2.begin(), r2.end());
                    redo1.swap(keep);
                }
            }
            if (wide_ok) {
                // pairs whose band outgrew the 64-diagonal window: the same kernel with a wave per pair (256 diagonals)
                const int wv = wide_pass(redo1);
                if (wv < 0) return wv;
            }
            redo1.insert(redo1.end(), redo_w.begin(), redo_w.end());
            redo_w.clear();
            if (kind1 >= 3) {
                // pairs that ran out of arena rows (a score above half the read length: error rates beyond ~8 %): the same
                // 64-diagonal kernel with four times the rows; and the class starts with more rows next time
                std::vector<uint32_t> lst;
                std::vector<uint64_t> keep, r2;
                for (uint64_t e : redo1) ((uint32_t)(e >> 32) == ST_REDO_ARENA ? (void)lst.push_back((uint32_t)e) : (void)keep.push_back(e));
                if (lst.size() * 64 > n_pairs && ctx->opt_packed_arena_bytes <= 0) {
                    const uint32_t nxt = (uint32_t)std::min<uint64_t>(8, arena_mult * 2);
                    if (ctx->rows_key != rkey || ctx->rows_scale < nxt) ctx->rows_key = rkey, ctx->rows_scale = nxt;
                }
                if (!lst.empty() && arena_mult <= 8 && ctx->opt_packed_arena_bytes <= 0) {
                    const uint64_t keep_mult = arena_mult;
                    arena_mult *= 4;
                    rc = forward_pass((kind_rest == kind_wide || kind_rest == kind_mid || kind_rest >= 14) ? kind_rest : kind_64, &lst, 0, lst.size(), r2, false);
                    arena_mult = keep_mult;
                    if (rc) return rc;
                    ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - r2.size());
                    keep.insert(keep.end(), r2.begin(), r2.end());
                    redo1.swap(keep);
                    if (wide_ok) {  // (what outgrows the 64-diagonal window on the way)
                        const int wv = wide_pass(redo1);
                        if (wv < 0) return wv;
                    }
                }
            }
            redo1.insert(redo1.end(), redo_w.begin(), redo_w.end());
            redo_w.clear();
            if (kind1 >= 2 && can_b && !wide_ok && redo1.size() > resident_generic) {
                // second chance on the LDS-ring kernel (64-diagonal bands at any alignment) for band/arena misses
                std::vector<uint32_t> lst;
                for (uint64_t e : redo1) {
                    const uint32_t stw = (uint32_t)(e >> 32);
                    if (stw == ST_REDO_BYTES) jb.pairs.push_back((uint32_t)e);
                    else if (stw == ST_REDO_BAND) lst.push_back((uint32_t)e);
                    else ja.pairs.push_back((uint32_t)e);
                }
                if (!lst.empty()) {
                    if ((rc = forward_pass(1, &lst, 0, lst.size(), redo2, false))) return rc;
                    ctx->timing.n_packed_pairs += (uint32_t)(lst.size() - redo2.size());
                    for (uint64_t e : redo2) ((uint32_t)(e >> 32) == ST_REDO_BYTES ? jb : ja).pairs.push_back((uint32_t)e);
                }
            } else {
                for (uint64_t e : redo1) ((uint32_t)(e >> 32) == ST_REDO_BYTES ? jb : ja).pairs.push_back((uint32_t)e);
            }
            if (skip_rest)
                for (uint64_t i = done_pairs; i < n_pairs; i++) ja.pairs.push_back((uint32_t)i);
            std::sort(ja.pairs.begin(), ja.pairs.end());
            for (Job *jp : {&ja, &jb}) {
                if (jp->pairs.empty()) continue;
                if (sub_len_used == 0) {
                    jobs.push_back(std::move(*jp));
                    continue;
                }
                // mixed lengths: the leftovers that fit the short class keep its (small) arenas and LDS budget
                Job js = *jp, jl = *jp;
                js.pairs.clear(), jl.pairs.clear();
                js.max_len = sub_len_used;
                for (uint32_t pid : jp->pairs) (h_len[pid] <= sub_len_used ? js : jl).pairs.push_back(pid);
                if (!js.pairs.empty()) jobs.push_back(std::move(js));
                if (!jl.pairs.empty()) jobs.push_back(std::move(jl));
            }
#ifdef WFA_STAMPS
            {
                unsigned long long acc[16];
                HIP_TRY(hipMemcpy(acc, P.debug_info, 128, hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[events] wave-steps %llu  slow %llu  hit %llu  reduce %llu  found %llu  found pair-steps %llu  "
                             "continuation iterations %llu  running pair-steps %llu\n", acc[8], acc[9], acc[10], acc[11], acc[12],
                             acc[13], acc[14], acc[15]);
                unsigned long long tot = 0;
                for (int i = 0; i < 6; i++) tot += acc[i];
                const char *nm[6] = {"refill", "next", "extend", "ranges+reduce", "stores", "ring+finish+window"};
                for (int i = 0; i < 6; i++)
                    std::fprintf(stderr, "[stamps] %-22s %6.2f %%  (%llu cyc)\n", nm[i], 100.0 * acc[i] / (double)tot, acc[i]);
                P.debug_info = nullptr;
            }
#endif
            if (ctx->band_key == rkey && ctx->band_kind == 9 && mid_in > 0 && mid_fail * 2 > mid_in) ctx->band_kind = 5;
            first       = false;
            packed_done = true;
        }
    }
    // Long pairs (team-kernel regime): the arena a pair needs is only known once it has been aligned -- a 100 kbp
    // semi-global pair takes anything from 0.2 to 13 GB -- and climbing the ladder from the bottom costs a launch per
    // level.  A context remembers, per workload class (mode, wf-adaptive, penalties, length bucket), the lowest level by
    // which 90 % of the long pairs had finished, and starts the next call of that class there (the slots of the last
    // levels are tens of GB: a few outliers must not size everybody's arena).
    uint64_t  lkey = 0;
    {
        uint32_t lb = 0;
        while ((2u << lb) <= max_len) lb++;
        lkey = 1ull | ((uint64_t)lb << 1) | ((uint64_t)(ctx->opt_team_compact != 0) << 7) | ((uint64_t)P.global_alignment << 8) | ((uint64_t)P.adaptive << 9) | ((uint64_t)(P.x & 0xFFF) << 12) |
               ((uint64_t)(P.oe & 0xFFF) << 24) | ((uint64_t)(P.e & 0xFFF) << 36) | ((uint64_t)(P.max_dist_diff & 0xFFFF) << 48);
    }
    int      learned_now = -1;
    uint64_t team_total = 0, team_done = 0;
    // (tracked for every batch of long pairs that goes straight to this ladder -- team kernel or one workgroup per pair:
    // 500 x 50 kbp global pairs at 20 % error all need the second level, 72 ms of a 172 ms call were spent finding that out)
    const bool learn_track = !packed_done && ctx->opt_team_min_len > 0 && max_len >= (uint64_t)ctx->opt_team_min_len;
    if (!packed_done) {
        Job j;
        j.mode = ctx->force_mode == 1 ? 1 : 0, j.level = 0, j.all = true;
        // The learned level is a HINT about speed, never about results: a call may start there, but (a) every sixteenth call of
        // the class starts one level lower, so that one hard batch does not pin the class to large slots and few teams
        // for ever (a start level can only be confirmed or raised by the call that uses it), and (b) a start level whose
        // slot does not fit this call's lengths is stepped down below instead of failing the pairs.
        if (ctx->opt_learn && !debug_single && ctx->learn_key == lkey && ctx->opt_team_min_len > 0 && max_len >= (uint64_t)ctx->opt_team_min_len) {
            j.level = ctx->learn_level;
            if (j.level > 0 && (++ctx->learn_calls & 15u) == 0u) j.level -= 1;  // (a probe that fails costs a launch: 72 ms of a 94 ms call on 500 x 50 kbp)
            j.hint = j.level > 0;
        }
        j.scout = true;
        ctx->timing.ladder_start_level = (uint32_t)j.level;  // (start level of the long-pair ladder: tests of the learned hint read it)
        jobs.push_back(std::move(j));
    }

    while (!jobs.empty()) {
        ctrl_fresh = false, ctrl_zeroed = false;
        Job job = std::move(jobs.front());
        jobs.pop_front();
        const uint64_t n_work = job.all ? n_pairs : job.pairs.size();
        if (n_work == 0) continue;
        const uint32_t max_len_all = max_len;
        const uint32_t max_len     = job.max_len ? job.max_len : max_len_all;  // (shadows the batch's bound for this job)
        LaunchCfg cfg;
        int       cr = make_cfg(ctx, max_len, job.mode, job.level, n_work, !P.global_alignment, cfg);
        if (cr == 1) {  // sequences do not fit LDS: byte path for the whole job
            job.mode = 1;
            cr       = make_cfg(ctx, max_len, 1, job.level, n_work, !P.global_alignment, cfg);
        }
        // a learned start level whose slot does not fit (the class buckets lengths by powers of two, slots scale with the
        // length): climb down to the largest level that does, never straight to "no memory"
        while (cr == 2 && job.hint && job.level > 0) {
            job.level -= 1;
            cr = make_cfg(ctx, max_len, job.mode, job.level, n_work, !P.global_alignment, cfg);
        }
        // (the paged team kernel has levels beyond "one slot of this level fits": the pool is the whole budget there and a level
        // halves the teams that share it -- the launch configuration is then that of the last level whose slot fitted)
        const bool paged_capable = ctx->opt_team_paged != 0 && !debug_single && ctx->opt_team_wgs == 0 && ctx->opt_arena_bytes_per_slot <= 0 &&
                                   ctx->opt_team_min_len > 0 && max_len >= (uint64_t)ctx->opt_team_min_len && P.e != 0u;
        const bool no_slot_fits = cr == 2;  // (if the paged launch does not happen after all, the job ends as "no memory" as it used to)
        for (int lv = job.level; cr == 2 && paged_capable && lv > 0;) cr = make_cfg(ctx, max_len, job.mode, --lv, n_work, !P.global_alignment, cfg);
        if (debug_single) cfg.slots = 1;
        // Wide wavefronts: a team of workgroups per pair (wfa_team_kernel) instead of one workgroup per pair.
        uint32_t team_T = 0, team_n = 0, team_wave_rows = 0;
        // It pays when one workgroup per pair cannot fill the GPU: few pairs, or arenas so large that only a few
        // fit (cfg.slots is the number of pairs the generic kernel could run at once).
        if (cr == 0 && (!debug_single || ctx->opt_team_wgs > 0) && ctx->opt_team_min_len > 0 && max_len >= (uint64_t)ctx->opt_team_min_len &&
            (cfg.slots < (uint32_t)std::max(1, ctx->num_cus / 2) || ctx->opt_team_wgs > 0) && P.e != 0u &&
            std::max(P.x, std::max(P.oe, P.e)) / P.g < (uint32_t)TEAM_RING) {
            const uint32_t cus = (uint32_t)std::max(1, ctx->num_cus);
            uint32_t t0 = (uint32_t)std::min<uint64_t>(cus, std::max<uint64_t>(2, (2ull * max_len + 8191) / 8192));
            if (ctx->opt_team_wgs > 0) t0 = (uint32_t)std::min<int64_t>(cus, ctx->opt_team_wgs);
            team_n = (uint32_t)std::min<uint64_t>(n_work, std::max<uint32_t>(1, cus / t0));
            team_T = ctx->opt_team_wgs > 0 ? t0 : cus / team_n;
            // one arena per team
            const uint64_t budget = (uint64_t)((double)ctx->total_mem * ladder_budget(ctx));
            while (team_n > 1 && (uint64_t)team_n * cfg.arena_words * 4ull > budget) team_n--;
            if ((uint64_t)team_n * cfg.arena_words * 4ull > budget) cr = 2;
            if (ctx->opt_team_wgs == 0) team_T = std::min<uint32_t>(cus / team_n, 2 * t0);  // ~2 cells per thread and stripe
            cfg.slots         = team_n;
            cfg.lds_seq_words = (cfg.lds_seq_words + 1u) & ~1u;
            cfg.lds_bytes     = (2ull * cfg.lds_seq_words + 16 + TEAM_RING * (sizeof(DirEnt) / 4)) * 4ull;
            if (cfg.lds_bytes > LDS_MAX_BYTES) team_T = 0;  // (cannot happen: make_cfg already bounded the sequences)
            // wave mode: an LDS ring of the last rows (a power of two above the farthest source), if it fits
            team_wave_rows = 0;
            if (ctx->opt_team_wave) {
                uint32_t rows = 2;
                while (rows <= std::max(P.x, std::max(P.oe, P.e)) / P.g) rows *= 2;
                const size_t ring_bytes = (size_t)rows * 3 * 64 * 4;
                if (rows <= (uint32_t)TEAM_RING && cfg.lds_bytes + ring_bytes <= LDS_MAX_BYTES) {
                    team_wave_rows = rows;
                    cfg.lds_bytes += ring_bytes;
                }
            }
        }
        // wfa_teamc_kernel (round 5) instead of wfa_team_kernel when its LDS rings fit beside the sequences: rows of max(x, o+e)/g +
        // 2 e/g x 4 098 words (the default penalties: six rows, 96 KB; a 100 kbp pair's packed sequences: 50 KB)
        bool   team_c = false;
        size_t lds_c  = 0;
        if (team_T > 0 && ctx->opt_team_compact != 0 && (!debug_single || ctx->dbg_teamc) && max_len < (1u << 27)) {
            const uint32_t rm = std::max(P.x, P.oe) / P.g, re = P.e / P.g;
            lds_c = (2ull * cfg.lds_seq_words + TC_RED + TEAM_RING * (sizeof(DirEnt) / 4)) * 4ull + (size_t)team_wave_rows * 3 * 64 * 4 +
                    (size_t)64 * TC_U * 4 + (size_t)(rm + 2 * re) * TC_ROWW * 4;
            team_c = lds_c <= LDS_MAX_BYTES;
        }
        if (debug_single && ctx->dbg_teamc && !team_c) return WFAHIP_ERR_UNSUPPORTED;
        // Paged arena of the team kernel: ONE pool for all teams, a pair takes pages as its rows grow.  The ladder level sizes the
        // pool (as many slot sizes as there are teams) until that reaches the budget; from there a level halves the number of
        // teams that share it -- down to one team with the whole pool.
        bool     paged = false, scout_now = false;
        uint64_t pool_words = 0, dir_words = 0;
        uint32_t page_log = 0, n_pages = 0;
        if (team_T > 0 && paged_capable) {
            const uint32_t cus = (uint32_t)std::max(1, ctx->num_cus);
            const uint32_t t0  = (uint32_t)std::min<uint64_t>(cus, std::max<uint64_t>(2, (2ull * max_len + 8191) / 8192));
            uint32_t teams = (uint32_t)std::min<uint64_t>(n_work, std::max<uint32_t>(1, std::min<uint32_t>(8u, cus / t0)));
            // scout pass (wfa_teamc_kernel only): a team is ONE workgroup, as many teams as CUs; it finishes the pairs whose band collapses and hands the
            // others on.  Worth a launch of its own when the teams would otherwise take several pairs each.
            scout_now = job.scout && team_c && ctx->opt_team_scout != 0 && P.adaptive && (ctx->opt_team_scout >= 2 || n_work > 2ull * teams);
            const uint32_t teams_full = teams;
            if (scout_now) teams = (uint32_t)std::min<uint64_t>(n_work, cus);
            const uint64_t budget_w = (uint64_t)((double)ctx->total_mem * ladder_budget(ctx)) / 4ull;
            // the first level at which `teams` slots no longer fit the budget, and how far this job is beyond it
            int over = 0;
            for (int lv = 0; lv <= job.level; lv++) {
                LaunchCfg c2;
                const int r2 = make_cfg(ctx, max_len, job.mode, lv, n_work, !P.global_alignment, c2);
                if (r2 == 2 || (uint64_t)teams * c2.arena_words > budget_w) over++;
            }
            // (over = 1: the first level whose slots no longer fit -- all teams, the whole budget; every further level halves the teams)
            const bool spent = over > 1 && (teams >> (over - 2)) <= 1u;  // the level before already ran ONE team with the whole pool
            if (over > 1) teams = std::max<uint32_t>(1, teams >> (over - 1));
            LaunchCfg c0;
            const int r0 = make_cfg(ctx, max_len, job.mode, job.level, n_work, !P.global_alignment, c0);
            // a directory entry per score index up to the worst score two sequences of this length can reach (every base a
            // mismatch or part of one long gap), per team, behind the pages
            const uint64_t worst_idx = ((uint64_t)(P.x + P.e) * max_len + 2ull * P.oe) / P.g + 64;
            dir_words = ((uint64_t)DIR_WORDS * worst_idx + 4095) & ~4095ull;
            const uint64_t dirs = (uint64_t)teams * dir_words;
            uint64_t rows_words = (r0 == 2 || over > 0) ? (budget_w > dirs ? budget_w - dirs : 0) : std::min<uint64_t>(budget_w, (uint64_t)teams * c0.arena_words);
            // pages of 1/64 of the rows' share, between "a row and then some" and 256 MB
            uint64_t pw = 1ull << 20;
            while (pw < 8ull * max_len) pw <<= 1;
            while (pw < (64ull << 20) && pw * 64 < rows_words) pw <<= 1;
            while ((1ull << page_log) < pw) page_log++;
            n_pages    = (uint32_t)std::min<uint64_t>(rows_words >> page_log, 1u << 20);
            pool_words = ((uint64_t)n_pages << page_log) + dirs;
            if (spent) {
                cr = 2;
            } else if (n_pages >= teams) {
                paged = true, team_n = teams, cr = 0;
                if (ctx->opt_team_wgs == 0) team_T = scout_now ? 1u : std::min<uint32_t>(cus / team_n, 2 * t0);
            } else if (scout_now && n_pages >= teams_full) {  // (too few pages for a team per CU: as many scouts as there are pages)
                paged = true, team_n = std::min<uint32_t>(teams, n_pages), cr = 0;
                team_T = 1u;
            }
            if (!paged) scout_now = false;
        }
        if (no_slot_fits && !paged) cr = 2;  // (the configuration of a lower level was only borrowed for the paged launch)
        if (cr == 2 || job.level > max_level) {
            if (job.all) {
                no_memory.resize(n_pairs);
                std::iota(no_memory.begin(), no_memory.end(), 0u);
            } else {
                no_memory.insert(no_memory.end(), job.pairs.begin(), job.pairs.end());
            }
            continue;
        }
        // (after a sub-wave first pass the ladder takes the second arena: that pass's backtrace kernel may still be reading
        // the first, and wfahip_debug_compact_arena shows what it left)
        DevBuf &jarena = (ctx->bt_pending || (packed_done && ctx->arena.bytes <= ctx->total_mem / 10)) ? ctx->arena2 : ctx->arena;
        // Long pairs climbing the ladder: take the whole arena budget once instead of freeing and re-allocating a
        // bigger buffer at every level (hipMalloc / hipFree of tens of GB cost more than the alignments).
        if (team_T > 0 && job.level >= 2 && jarena.bytes < (size_t)((double)ctx->total_mem * ladder_budget(ctx)))
            (void)ensure(ctx, jarena, (size_t)((double)ctx->total_mem * ladder_budget(ctx)));
        if (paged) {
            cfg.slots = team_n, cfg.arena_words = pool_words;
            rc = ensure(ctx, jarena, (size_t)pool_words * 4ull);
        } else {
            rc = ensure(ctx, jarena, (size_t)cfg.arena_words * 4ull * cfg.slots);
            if (rc == WFAHIP_ERR_OOM && cfg.slots > 1) {  // shrink once
                cfg.slots = std::max<uint32_t>(1, cfg.slots / 4);
                rc        = ensure(ctx, jarena, (size_t)cfg.arena_words * 4ull * cfg.slots);
            }
        }
        if (rc) return rc;
        ctx->timing.arena_bytes = std::max<uint64_t>(ctx->timing.arena_bytes, paged ? pool_words * 4ull : (uint64_t)cfg.arena_words * 4ull * cfg.slots);

        P.arena = static_cast<uint32_t *>(jarena.p), P.arena_words = cfg.arena_words;
        P.page_ctl = nullptr, P.page_words_log2 = 0, P.n_pages = 0, P.dir_region_words = 0;
        if (paged) {
            const size_t pc_words = 4u + (size_t)n_pages + (size_t)team_n * TEAM_MAX_PAGES;
            if ((rc = ensure(ctx, ctx->page_ctl, pc_words * 4))) return rc;
            std::vector<uint32_t> init(4u + n_pages);
            init[0] = 0u, init[1] = n_pages, init[2] = 0u, init[3] = 0u;
            for (uint32_t i = 0; i < n_pages; i++) init[4u + i] = n_pages - 1u - i;  // (page 0 on top of the stack)
            HIP_TRY(hipMemcpyAsync(ctx->page_ctl.p, init.data(), init.size() * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));  // (`init` is a pageable temporary)
            P.page_ctl = static_cast<uint32_t *>(ctx->page_ctl.p), P.page_words_log2 = page_log, P.n_pages = n_pages, P.dir_region_words = dir_words;
        }
        P.lds_seq_words = cfg.lds_seq_words;
        P.n_work        = (uint32_t)n_work;
        // Teams take pairs from one queue, and a pair costs a team anything from 0.1 s to 0.4 s (configs[4]): the expensive ones go
        // first, so that no team starts one when the others are about to finish.  Under wf-adaptive a pair whose lengths differ by
        // more than MaxDistDiff keeps a wide band for most of its scores (the first reduce cuts the final diagonal off; DESIGN.md
        // section 4d) -- a scheduling hint only, results do not depend on the order.
        std::optional<TeamLaunchLock> team_lock;  // (released at the end of this job, behind the synchronisation that follows its launch)
        if (team_T > 0) team_lock.emplace(ctx->device);
        std::vector<uint32_t> team_order;
        // (the hint reads the batch's length arrays: not for a handful of long pairs out of millions of short ones)
        if (team_T > 0 && ctx->opt_team_order != 0 && P.adaptive && n_work > team_n && n_work <= (1u << 24) && (n_pairs <= (1u << 20) || n_pairs <= 64 * n_work)) {
            std::vector<uint32_t> ql(n_pairs), tl(n_pairs);
            HIP_TRY(hipMemcpyAsync(ql.data(), d_q_len, n_pairs * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(tl.data(), d_t_len, n_pairs * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            team_order.resize(n_work);
            if (job.all) std::iota(team_order.begin(), team_order.end(), 0u);
            else std::copy(job.pairs.begin(), job.pairs.end(), team_order.begin());
            const uint32_t mdd = P.max_dist_diff;
            const auto wide = [&](uint32_t i) { return (ql[i] > tl[i] ? ql[i] - tl[i] : tl[i] - ql[i]) > mdd; };
            std::stable_sort(team_order.begin(), team_order.end(), [&](uint32_t a, uint32_t b) {
                const bool wa = wide(a), wb = wide(b);
                if (wa != wb) return wa;
                return std::max(ql[a], tl[a]) > std::max(ql[b], tl[b]);
            });
        }
        if (!team_order.empty()) {
            if ((rc = upload_work(team_order.data(), n_work))) return rc;
            HIP_TRY(hipStreamSynchronize(st));  // (`team_order` is a pageable temporary when it is longer than the pinned block)
            P.work = static_cast<const uint32_t *>(ctx->work.p);
        } else if (job.all) {
            P.work = nullptr;
        } else {
            if ((rc = upload_work(job.pairs.data(), n_work))) return rc;
            P.work = static_cast<const uint32_t *>(ctx->work.p);
        }
        HIP_TRY(hipMemsetAsync(d_ctrl, 0, 8, st));  // queue_head, redo_count
        if (ctx->pk_words) {  // (packed input: the long-pair kernels read bytes -- this job's pairs are expanded first)
            P.chunk_first = 0, P.chunk_n = (uint32_t)n_work;
            hipLaunchKernelGGL(wfa_unpack_pairs_kernel, dim3((uint32_t)((n_work + 3) / 4)), dim3(256), 0, st, P, ctx->pk_words, const_cast<uint8_t *>(P.blob));
            HIP_TRY(hipGetLastError());
        }
        // (tests: what an earlier launch left in the arena must never be read -- with the same batch run twice a stale
        // read returns the right value and hides itself)
        if (ctx->opt_arena_poison)
            HIP_TRY(hipMemsetAsync(jarena.p, 0xA5, paged ? (size_t)pool_words * 4ull : (size_t)cfg.arena_words * 4ull * cfg.slots, st));
        HIP_TRY(hipEventRecord(ctx->evA, st));
        if (team_T > 0) {
            const size_t ctl_words = team_c ? (size_t)TC_CTL_WORDS : (size_t)TEAM_CTL_WORDS;
            if ((rc = ensure(ctx, ctx->team_ctl, (size_t)team_n * ctl_words * 4))) return rc;
            HIP_TRY(hipMemsetAsync(ctx->team_ctl.p, 0, (size_t)team_n * ctl_words * 4, st));
            // teams of one XCD's CUs (team = blockIdx % 8) when at most eight teams run and the CUs divide by eight
            const bool     xmap   = ctx->opt_team_xcd != 0 && ctx->opt_team_wgs == 0 && team_n <= 8 && ctx->num_cus % 8 == 0 && ctx->num_cus >= 16;
            const uint32_t grid_t = xmap ? (uint32_t)ctx->num_cus : team_n * team_T;
            if (xmap) team_T = (uint32_t)ctx->num_cus / 8u;
            if (team_c && team_T <= (uint32_t)TC_MAX_T) {
                TcArgs X{};
                const uint32_t rm = std::max(P.x, P.oe) / P.g, re = P.e / P.g;
                X.xw         = (2u * max_len + 64u + 63u) & ~63u;
                X.xbuf_words = (uint64_t)((rm + 1) + 2 * (re + 1)) * X.xw;
                if ((rc = ensure(ctx, ctx->xbuf, (size_t)team_n * X.xbuf_words * 4))) return rc;
                if (ctx->opt_arena_poison) HIP_TRY(hipMemsetAsync(ctx->xbuf.p, 0xA5, (size_t)team_n * X.xbuf_words * 4, st));
                X.team_ctl = static_cast<uint32_t *>(ctx->team_ctl.p), X.xbuf = static_cast<uint32_t *>(ctx->xbuf.p);
                X.T = team_T, X.n_teams = team_n, X.tpx = xmap ? 1u : 0u;
                X.solo_max = (uint32_t)std::max<int64_t>(0, ctx->opt_team_solo_max_set ? ctx->opt_team_solo_max : std::min<int64_t>(ctx->opt_team_solo_max, 512)), X.wave_rows = team_wave_rows;
                X.strict = (uint32_t)(ctx->opt_team_strict != 0) | (xmap && ctx->opt_team_xcd >= 2 ? 4u : 0u);
                X.slack  = (uint32_t)std::min<int64_t>(std::max<int64_t>(1, ctx->opt_team_slack), 1 << 20);
                X.fast   = ctx->opt_team_fast != 0 ? 1u : 0u;
                X.scout  = scout_now ? 1u : 0u;
                X.pipe   = ctx->opt_team_pipe != 0 ? 1u : 0u;
                X.dbg    = debug_single ? d_ctrl + 4 : nullptr;
                HIP_TRY(wfa_launch_teamc(P, X, job.mode, grid_t, lds_c, st));
            } else {
                team_c = false;
                HIP_TRY(wfa_launch_team(P, job.mode, grid_t, cfg.lds_bytes, st, static_cast<uint32_t *>(ctx->team_ctl.p), team_T,
                                        (uint32_t)std::max<int64_t>(0, ctx->opt_team_solo_max), team_wave_rows,
                                        (uint32_t)(ctx->opt_team_strict != 0) | (xmap ? 2u : 0u) | (xmap && ctx->opt_team_xcd >= 2 ? 4u : 0u) | (team_n << 16)));
            }
        } else {
            // wave mode of the generic kernel: directory ring + ring of the last rows in LDS, if they fit
            P.wave_rows = 0, P.wave_bt = 0;
            if (ctx->opt_team_wave && std::max(P.x, std::max(P.oe, P.e)) / P.g < (uint32_t)WAVE_DIR_RING) {
                const size_t dir_bytes = 16 + (size_t)WAVE_DIR_RING * sizeof(DirEnt);
                if (cfg.lds_bytes + dir_bytes <= LDS_MAX_BYTES) {
                    P.wave_bt = 1, cfg.lds_bytes += dir_bytes;
                    uint32_t rows = 2;
                    while (rows <= std::max(P.x, std::max(P.oe, P.e)) / P.g) rows *= 2;
                    const size_t ring_bytes = (size_t)rows * 3 * 64 * 4;
                    if (P.e != 0u && cfg.lds_bytes + ring_bytes <= LDS_MAX_BYTES) P.wave_rows = rows, cfg.lds_bytes += ring_bytes;
                }
            }
            HIP_TRY(launch_generic(P, cfg, st));
        }
        HIP_TRY(hipEventRecord(ctx->evB, st));
        uint32_t              hctrl[CTRL_WORDS];
        std::vector<uint64_t> ent;  // {pair, status}, sorted by pair
        if ((rc = fetch_ctrl(hctrl, &ent))) return rc;
        if (team_T > 0) {  // a team barrier that ran into its spin bound
            const size_t TEAM_CTL_STRIDE = team_c ? (size_t)TC_CTL_WORDS : (size_t)TEAM_CTL_WORDS;
            std::vector<uint32_t> tc((size_t)team_n * TEAM_CTL_STRIDE);
            HIP_TRY(hipMemcpy(tc.data(), ctx->team_ctl.p, tc.size() * 4, hipMemcpyDeviceToHost));
#ifdef WFA_TEAM_STAMPS
            for (uint32_t t = 0; team_c && t < team_n; t++) {  // wfa_teamc_kernel's phases (its own numbering, wfa_teamc.hpp)
                const unsigned long long *a = reinterpret_cast<const unsigned long long *>(&tc[(size_t)t * TEAM_CTL_STRIDE + 64]);
                const double steps = (double)std::max<unsigned long long>(1, a[16] + a[17] + a[18]);
                std::fprintf(stderr, "[teamc %u] steps: stripe(team) %llu xbuf %llu stripe(solo) %llu, ring loads %llu | us: head %.0f cells %.0f wave-red %.0f wait-wg %.0f rings+edges %.0f "
                             "exchange1 %.0f band-ends %.0f exchange2 %.0f tail %.0f | wave mode %.0f backtrace %.0f | per wide step %.2f us\n", t, a[16], a[17], a[18], a[19],
                             a[0] / 100.0, a[1] / 100.0, a[2] / 100.0, a[3] / 100.0, a[4] / 100.0, a[5] / 100.0, a[6] / 100.0, a[7] / 100.0, a[8] / 100.0, a[9] / 100.0, a[10] / 100.0,
                             (a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + a[8] + a[11] + a[12] + a[13] + a[14]) / 100.0 / steps);
                {
                    const unsigned long long *m = reinterpret_cast<const unsigned long long *>(&tc[(size_t)t * TEAM_CTL_STRIDE + TC_TRACE_OFF]);
                    std::fprintf(stderr, "[teamc %u] the workgroup in the middle of the team, us: head %.0f cells %.0f wave-red %.0f wait-wg %.0f rings+edges %.0f exchange1 %.0f band-ends %.0f exchange2 %.0f tail %.0f\n",
                                 t, m[0] / 100.0, m[1] / 100.0, m[2] / 100.0, m[3] / 100.0, m[4] / 100.0, m[5] / 100.0, m[6] / 100.0, m[7] / 100.0, m[8] / 100.0);
                }
                {
                    const unsigned long long *x = reinterpret_cast<const unsigned long long *>(&tc[(size_t)t * TEAM_CTL_STRIDE + TC_TRACE_OFF + 64]) - 24;
                    std::fprintf(stderr, "[teamc %u] exchange 1 in detail (inside the figure above), us: rows into the rings + slot stored %.0f, polled %.0f (%llu polls), reduced %.0f\n", t,
                                 x[24] / 100.0, x[25] / 100.0, x[27], x[26] / 100.0);
                }
                {
                    const unsigned long long *x = reinterpret_cast<const unsigned long long *>(&tc[(size_t)t * TEAM_CTL_STRIDE + TC_TRACE_OFF + 64]) - 24;
                    std::fprintf(stderr, "[teamc %u] pipelined steps: committed %llu, left (row not eligible) %llu, deep %llu, complex %llu, not entered %llu | us of wave 0: head %.0f, row S + late cells %.0f, "
                                 "wait for the cell waves %.0f, commit %.0f\n", t, a[20], a[21], a[22], a[23], x[28], a[0] / 100.0, x[29] / 100.0, x[30] / 100.0, x[31] / 100.0);
                }
                std::fprintf(stderr, "[teamc %u] head in detail, us: ring entries + ranges %.0f, room %.0f, first barrier %.0f, mode + scratch %.0f, second barrier %.0f\n", t, a[11] / 100.0,
                             a[12] / 100.0, a[13] / 100.0, a[14] / 100.0, a[0] / 100.0);
            }
            for (uint32_t t = 0; !team_c && t < team_n; t++) {
                const unsigned long long *a = reinterpret_cast<const unsigned long long *>(&tc[(size_t)t * TEAM_CTL_STRIDE + 64]);
                std::fprintf(stderr, "[team %u] us: P1 %.0f  barriers %.0f  P2 %.0f  P3 %.0f  tail(team) %.0f  solo steps %.0f  end search %.0f  backtrace %.0f  wave mode %.0f | steps: wave %llu solo %llu team %llu\n", t,
                             a[0] / 100.0, a[1] / 100.0, a[2] / 100.0, a[3] / 100.0, a[4] / 100.0, a[5] / 100.0, a[6] / 100.0, a[7] / 100.0,
                             a[8] / 100.0, a[9], a[10], a[11]);
                std::fprintf(stderr, "[team %u] band ends: nothing fails %llu, both within 64 cells %llu, within the 512-cell windows %llu, full passes %llu; mean trim low %.1f high %.1f\n",
                             t, a[12], a[13], a[14], a[15], (double)a[16] / (double)std::max<unsigned long long>(1, a[13] + a[14]),
                             (double)a[17] / (double)std::max<unsigned long long>(1, a[13] + a[14]));
                std::fprintf(stderr, "[team %u] inside P1 (wave 0 of workgroup 0), us: cells %.0f  wave reductions %.0f  wait for the other waves %.0f  (rest = team atomics); stripe-mode ring loads %llu\n",
                             t, a[18] / 100.0, a[19] / 100.0, a[20] / 100.0, a[21]);
            }
#endif
            if (std::getenv("WFAHIP_DEBUG_TIMING")) {
                uint32_t n_xl = 0;
                for (uint32_t t = 0; t < team_n; t++) n_xl += tc[(size_t)t * TEAM_CTL_STRIDE + 11];
                for (uint32_t t = 0; t < team_n; t++) std::fprintf(stderr, "[wfahip]   team %u: XCC ids seen, as a mask: 0x%x\n", t, tc[(size_t)t * TEAM_CTL_STRIDE + 3]);
                std::fprintf(stderr, "[wfahip] team kernel: %u teams of %u workgroups, %u of them on one XCD each\n", team_n, team_T, n_xl);
            }
            for (uint32_t t = 0; t < team_n; t++)
                if (tc[(size_t)t * TEAM_CTL_STRIDE + 1] != 0u) {
                    if (team_c && std::getenv("WFAHIP_DEBUG_TIMING")) {  // (wfa_teamc_kernel: where every workgroup of the team was)
                        const uint32_t *tr = &tc[(size_t)t * TEAM_CTL_STRIDE + TC_TRACE_OFF];
                        for (uint32_t w = 0; w < team_T && w < (uint32_t)TC_MAX_T; w++) {
                            std::fprintf(stderr, "[wfahip]   team %u workgroup %u: last mode change at score %u (%u -> %u), last park wake at %u (cmd %u), last wave exit at %u (flags %u), "
                                         "pair left at %u (mode %u done %u overflow %u alone %u); ctl: cmd %u score %u flags %u\n", t, w, tr[2 * TC_MAX_T + w] >> 8, tr[2 * TC_MAX_T + w] & 15u,
                                         (tr[2 * TC_MAX_T + w] >> 4) & 15u, tr[3 * TC_MAX_T + w] >> 8, tr[3 * TC_MAX_T + w] & 255u, tr[4 * TC_MAX_T + w] >> 8, tr[4 * TC_MAX_T + w] & 255u,
                                         tr[5 * TC_MAX_T + w] >> 8, tr[5 * TC_MAX_T + w] & 15u, (tr[5 * TC_MAX_T + w] >> 4) & 1u, (tr[5 * TC_MAX_T + w] >> 5) & 1u, (tr[5 * TC_MAX_T + w] >> 6) & 1u,
                                         tc[(size_t)t * TEAM_CTL_STRIDE + 4], tc[(size_t)t * TEAM_CTL_STRIDE + 5], tc[(size_t)t * TEAM_CTL_STRIDE + 10]);
                            if ((tr[w] >> 24) == 0xABu)
                                std::fprintf(stderr, "[wfahip]   team %u workgroup %u: left after an aborted barrier at wfa_teamc.hpp:%u, exchanges %u\n", t, w, tr[w] & 0xFFFFFFu, tr[TC_MAX_T + w]);
                            else
                                std::fprintf(stderr, "[wfahip]   team %u workgroup %u: score %u phase %u mode %u, exchanges %u, barrier count %u\n", t, w, tr[w] >> 8,
                                             tr[w] & 15u, (tr[w] >> 4) & 15u, tr[TC_MAX_T + w], tc[(size_t)t * TEAM_CTL_STRIDE]);
                        }
                    }
                    std::snprintf(ctx->last_error, sizeof ctx->last_error, "team kernel: barrier timeout in team %u", t);
                    return WFAHIP_ERR_INTERNAL;
                }
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->evA, ctx->evB));
        ctx->timing.kernel_ms += ms;
        if (first) {
            ctx->timing.main_kernel_ms = ms, ctx->timing.n_main_launches = 1, first = false;
            if (team_T > 0) ctx->timing.main_kernel_kind = team_c ? 17 : 7;  // wfa_teamc_kernel / wfa_team_kernel (bench.py names the dominant kernel by this)
            after_scout = scout_now;
        } else if (after_scout && team_T > 0 && team_c) {  // (the scout pass and the teams behind it are one kernel: its launches add up)
            ctx->timing.main_kernel_ms += ms, ctx->timing.n_main_launches++;
        }
        ctx->timing.n_launches++;

        const uint32_t n_redo = hctrl[1];
        if (learn_track) {  // the lowest level by which 90 % of the long pairs of this call have finished
            if (team_total == 0) team_total = n_work;
            team_done += n_work - n_redo;
            if (learned_now < 0 && 10 * team_done >= 9 * team_total) learned_now = job.level;
        }
        if (n_redo) {
            Job jb, ja, jw;
            jb.mode = 1, jb.level = job.level, jb.all = false, jb.max_len = job.max_len;
            ja.mode = job.mode, ja.level = job.level + 1, ja.all = false, ja.max_len = job.max_len;
            jw.mode = job.mode, jw.level = job.level, jw.all = false, jw.max_len = job.max_len, jw.hint = job.hint;  // (handed on by the scout pass: a team's work, same level)
            for (uint32_t i = 0; i < n_redo; i++) {
                const uint32_t stw = (uint32_t)(ent[i] >> 32);
                (stw == ST_REDO_WIDE ? jw : (stw == ST_REDO_BYTES || stw == ST_REDO_LDS ? jb : ja)).pairs.push_back((uint32_t)ent[i]);
            }
            ctx->timing.n_retried_pairs += n_redo - (uint32_t)jw.pairs.size();  // (a pair the scouts hand on has not failed anything)
            if (!jw.pairs.empty()) jobs.push_front(std::move(jw));
            if (!jb.pairs.empty()) jobs.push_back(std::move(jb));
            if (!ja.pairs.empty()) jobs.push_back(std::move(ja));
        }
        if (debug_single) break;
    }
    if (learned_now >= 0) {
        if (ctx->learn_key != lkey || ctx->learn_level != learned_now) ctx->learn_calls = 0;
        ctx->learn_key = lkey, ctx->learn_level = learned_now;
    }
    if (ctx->bt_pending) HIP_TRY(hipStreamWaitEvent(st, ctx->evBtB, 0));
    if (!no_memory.empty() || ctx->bt_pending) ctrl_fresh = false;
    if (!ctrl_fresh) HIP_TRY(hipEventRecord(ctx->ev1, st));

    for (uint32_t pid : no_memory) {
        uint32_t recw[REC_WORDS] = {0};
        recw[REC_STATUS]         = ST_NO_MEMORY;
        HIP_TRY(hipMemcpyAsync(P.rec + (uint64_t)pid * REC_WORDS, recw, sizeof recw, hipMemcpyHostToDevice, st));
    }
    uint32_t hctrl[CTRL_WORDS];
    if (ctrl_fresh) std::memcpy(hctrl, hc_last, sizeof hctrl);
    else if ((rc = fetch_ctrl(hctrl, nullptr))) return rc;
    float ms = 0;
    if (ctx->bt_pending) {
        HIP_TRY(hipEventElapsedTime(&ms, ctx->evBtA, ctx->evBtB));
        ctx->timing.kernel_ms += ms;
        ctx->bt_pending = false;
    }
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->timing.total_ms    = ms;
    const uint64_t cursor   = (uint64_t)hctrl[2] | ((uint64_t)hctrl[3] << 32);
    ctx->timing.ops_written = cursor;
    if (ops_needed) *ops_needed = cursor;
    if (!debug_single && hipMemsetAsync(d_ctrl, 0, CTRL_WORDS * 4, st) == hipSuccess && hipEventRecord(ctx->ctrl_clean_ev, st) == hipSuccess)
        ctx->ctrl_clean = true, ctx->ctrl_clean_stream = st;
    else if (!debug_single)
        (void)hipStreamSynchronize(st);  // (a memset without its event must not stay pending)
    if (cursor > ops_cap) {
        if (std::getenv("WFAHIP_DEBUG_TIMING")) std::fprintf(stderr, "[wfahip] CIGAR op buffer too small: %llu needed, %llu there\n", (unsigned long long)cursor, (unsigned long long)ops_cap);
        return WFAHIP_ERR_OOM;
    }
    return WFAHIP_OK;
}

// (exception-safe: the host entry calls this while its upload / download threads are joinable)
static int align_device(wfahip_ctx *ctx, const wfahip_params *p, const void *d_blob, uint64_t blob_bytes,
                        const void *d_q_off, const void *d_q_len, const void *d_t_off, const void *d_t_len,
                        uint64_t n_pairs, uint32_t max_len, void *d_rec, void *d_ops, uint64_t ops_cap,
                        uint64_t *ops_needed, hipStream_t st, bool debug_single, uint64_t ops_cursor0 = 0) {
    WFAHIP_GUARD(align_device_impl(ctx, p, d_blob, blob_bytes, d_q_off, d_q_len, d_t_off, d_t_len, n_pairs, max_len, d_rec, d_ops,
                                   ops_cap, ops_needed, st, debug_single, ops_cursor0))
}

extern "C" int wfahip_align_batch_device(wfahip_ctx *ctx, const wfahip_params *p, const void *d_seq_blob,
                                         uint64_t blob_bytes, const void *d_q_off, const void *d_q_len,
                                         const void *d_t_off, const void *d_t_len, uint64_t n_pairs,
                                         uint32_t max_len, void *d_rec, void *d_ops, uint64_t ops_cap,
                                         uint64_t *ops_needed, void *stream) {
    if (!ctx) return WFAHIP_ERR_BAD_ARG;
    return align_device(ctx, p, d_seq_blob, blob_bytes, d_q_off, d_q_len, d_t_off, d_t_len, n_pairs, max_len,
                        d_rec, d_ops, ops_cap, ops_needed, static_cast<hipStream_t>(stream), false);
}

static void results_zero(wfahip_results *r) { std::memset(r, 0, sizeof *r); }

// Result arrays are malloc blocks OWNED BY THE LIBRARY: a binding must hand them back through wfahip_results_free and
// never free() them itself -- blocks that circulate through the cache below are page-locked (hipHostRegister), and
// freeing a registered block behind the runtime's back leaves a stale registration.  wfahip_results_free keeps the
// large ones for the next call instead of returning them to the system: a fresh 0.7 GB ops array costs its download
// twice over in first-touch page faults (the reference recycles its results the same way, wfa_cigar.go:92).
namespace {
struct ResBlock { void *p; size_t bytes; };
std::mutex            g_res_mu;
std::vector<ResBlock> g_res_cache;
size_t                g_res_cached_bytes = 0;
constexpr size_t      RES_CACHE_MIN = 1u << 20, RES_CACHE_MAX_BYTES = 4ull << 30, RES_CACHE_MAX_BLOCKS = 32;

// Blocks that come back from the cache are page-locked (hipHostRegister, once per block): the result download then
// goes straight into them at link rate -- through pinned staging plus copy-out threads 0.8 GB of results took 36 ms of
// a 44 ms call.  They stay registered while they circulate between wfahip_results_free and the next call.
std::vector<ResBlock> g_res_pinned;  // (guarded by g_res_mu)
bool res_is_pinned(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_res_mu);
    for (const ResBlock &b : g_res_pinned)
        if (p >= b.p && static_cast<const char *>(p) + bytes <= static_cast<const char *>(b.p) + b.bytes) return true;
    return false;
}
void *res_alloc(size_t bytes) {
    if (bytes >= RES_CACHE_MIN) {
        std::lock_guard<std::mutex> lk(g_res_mu);
        size_t best = g_res_cache.size();
        for (size_t i = 0; i < g_res_cache.size(); i++)
            if (g_res_cache[i].bytes >= bytes && g_res_cache[i].bytes <= 2 * bytes &&
                (best == g_res_cache.size() || g_res_cache[i].bytes < g_res_cache[best].bytes))
                best = i;
        if (best != g_res_cache.size()) {
            void *p = g_res_cache[best].p;
            const size_t cap = g_res_cache[best].bytes;
            g_res_cached_bytes -= cap;
            g_res_cache.erase(g_res_cache.begin() + (long)best);
            bool pinned = false;
            for (const ResBlock &b : g_res_pinned) pinned = pinned || b.p == p;
            if (!pinned && !std::getenv("WFAHIP_NO_PINNED_RESULTS") &&
                hipHostRegister(p, cap, hipHostRegisterPortable) == hipSuccess)
                g_res_pinned.push_back({p, cap});
            else if (!pinned)
                (void)hipGetLastError();
            return p;
        }
    }
    return std::malloc(bytes);
}
void res_release(void *p) {
    if (!p) return;
    const size_t bytes = malloc_usable_size(p);  // (the block's real capacity, whatever the caller did to n / n_ops)
    if (bytes >= RES_CACHE_MIN) {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (g_res_cache.size() < RES_CACHE_MAX_BLOCKS && g_res_cached_bytes + bytes <= RES_CACHE_MAX_BYTES) {
            g_res_cache.push_back({p, bytes});
            g_res_cached_bytes += bytes;
            return;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        for (size_t i = 0; i < g_res_pinned.size(); i++)
            if (g_res_pinned[i].p == p) {
                (void)hipHostUnregister(p);
                g_res_pinned.erase(g_res_pinned.begin() + (long)i);
                break;
            }
    }
    std::free(p);
}
}  // namespace

extern "C" void wfahip_results_free(wfahip_results *r) {
    if (!r) return;
    for (void *p : {(void *)r->status, (void *)r->score, (void *)r->tbegin, (void *)r->tend, (void *)r->qbegin, (void *)r->qend,
                    (void *)r->align_len, (void *)r->matches, (void *)r->gaps, (void *)r->gap_regions, (void *)r->ops_len,
                    (void *)r->ops_off, (void *)r->ops})
        res_release(p);
    results_zero(r);
}

static int unpack_results(const std::vector<uint32_t> &rec, const std::vector<uint64_t> &ops, uint64_t n,
                          wfahip_results *out, uint64_t *cells_total) {
    results_zero(out);
    out->n = n;
    size_t cnt = std::max<uint64_t>(n, 1);
#define ALLOC(field, type)                                            \
    out->field = static_cast<type *>(std::calloc(cnt, sizeof(type))); \
    if (!out->field) return WFAHIP_ERR_OOM;
    ALLOC(status, int32_t) ALLOC(score, uint32_t) ALLOC(tbegin, int32_t) ALLOC(tend, int32_t)
    ALLOC(qbegin, int32_t) ALLOC(qend, int32_t) ALLOC(align_len, uint32_t) ALLOC(matches, uint32_t)
    ALLOC(gaps, uint32_t) ALLOC(gap_regions, uint32_t) ALLOC(ops_off, uint64_t) ALLOC(ops_len, uint32_t)
#undef ALLOC
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++)
        if (rec[i * REC_WORDS + REC_STATUS] == ST_OK) total += rec[i * REC_WORDS + REC_OPS_LEN];
    out->ops = static_cast<uint64_t *>(std::malloc(std::max<uint64_t>(total, 1) * 8));
    if (!out->ops) return WFAHIP_ERR_OOM;
    uint64_t pos = 0, cells = 0;
    for (uint64_t i = 0; i < n; i++) {  // ops are re-packed in pair order (device order is completion order)
        const uint32_t *r  = &rec[i * REC_WORDS];
        uint32_t        st = r[REC_STATUS];
        out->status[i]     = (st == ST_OK || st == ST_EMPTY || st == ST_TOO_LONG) ? (int32_t)st : WFAHIP_PAIR_NO_MEMORY;
        if (st != ST_OK) continue;
        out->score[i]       = r[REC_SCORE];
        out->tbegin[i]      = (int32_t)r[REC_TBEGIN];
        out->tend[i]        = (int32_t)r[REC_TEND];
        out->qbegin[i]      = (int32_t)r[REC_QBEGIN];
        out->qend[i]        = (int32_t)r[REC_QEND];
        out->align_len[i]   = r[REC_ALIGN_LEN];
        out->matches[i]     = r[REC_MATCHES];
        out->gaps[i]        = r[REC_GAPS];
        out->gap_regions[i] = r[REC_GAP_REGIONS];
        out->ops_len[i]     = r[REC_OPS_LEN];
        out->ops_off[i]     = pos;
        uint64_t src        = (uint64_t)r[REC_OPS_OFF_LO] | ((uint64_t)r[REC_OPS_OFF_HI] << 32);
        std::memcpy(out->ops + pos, ops.data() + src, (size_t)r[REC_OPS_LEN] * 8);
        pos += r[REC_OPS_LEN];
        cells += (uint64_t)r[REC_CELLS_LO] | ((uint64_t)r[REC_CELLS_HI] << 32);
    }
    out->n_ops = pos;
    if (cells_total) *cells_total = cells;
    return WFAHIP_OK;
}

namespace {

constexpr size_t PIN_CHUNK = 32u << 20;

// Device -> pageable host memory: 32 MB pieces through two pinned buffers (full PCIe rate), copied out to their
// destination by a few host threads while the next piece is in flight (first-touch page faults of freshly
// malloc'd result arrays are what limits a plain hipMemcpy here).
int download(wfahip_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return WFAHIP_OK;
    if (res_is_pinned(dst, bytes)) {  // a recycled, page-locked result block: one copy at link rate
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return WFAHIP_OK;
    }
    for (int i = 0; i < 2; i++) {
        if (!ctx->pin[i]) HIP_TRY(hipHostMalloc(&ctx->pin[i], PIN_CHUNK, hipHostMallocDefault));
        if (!ctx->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
    }
    // (measured on the 256-thread GPU box, 0.8 GB of results: 8 threads 72 ms, 16 threads 53 ms, 32 threads 60 ms)
    unsigned n_thr = std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
    if (const char *e = std::getenv("WFAHIP_DL_THREADS")) n_thr = (unsigned)std::max(1, std::atoi(e));
    const size_t   n_chk = (bytes + PIN_CHUNK - 1) / PIN_CHUNK;
    auto issue = [&](size_t c) -> hipError_t {
        const size_t off = c * PIN_CHUNK, sz = std::min(PIN_CHUNK, bytes - off);
        hipError_t   e   = hipMemcpyAsync(ctx->pin[c & 1], static_cast<const char *>(src) + off, sz, hipMemcpyDeviceToHost, st);
        return e != hipSuccess ? e : hipEventRecord(ctx->pin_ev[c & 1], st);
    };
    HIP_TRY(issue(0));
    for (size_t c = 0; c < n_chk; c++) {
        HIP_TRY(hipEventSynchronize(ctx->pin_ev[c & 1]));
        if (c + 1 < n_chk) HIP_TRY(issue(c + 1));
        const size_t off = c * PIN_CHUNK, sz = std::min(PIN_CHUNK, bytes - off);
        char        *d = static_cast<char *>(dst) + off;
        const char  *p = static_cast<const char *>(ctx->pin[c & 1]);
        if (sz < (4u << 20) || n_thr == 1) {
            std::memcpy(d, p, sz);
        } else {
            std::vector<std::thread> th;
            const size_t             part = ((sz / n_thr) + 4095) & ~size_t(4095);
            for (unsigned t = 0; t < n_thr; t++) {
                const size_t a = std::min(sz, (size_t)t * part), b = std::min(sz, a + part);
                if (b <= a) continue;
                try {
                    th.emplace_back([=] { std::memcpy(d + a, p + a, b - a); });
                } catch (...) {  // no more threads: this part is copied here
                    std::memcpy(d + a, p + a, b - a);
                }
            }
            for (auto &t : th) t.join();
        }
    }
    return WFAHIP_OK;
}

}  // namespace

// Pre-packed input (wfahip_align_batch_packed): 2-bit words -> the byte blob the kernels read, on the device.  One
// thread per word: 16 bases = one 16-byte store.  Code -> letter is the inverse of the kernels' (c >> 1) & 3.
__global__ __launch_bounds__(256) void wfa_unpack_kernel(const uint32_t *__restrict__ words, uint4 *__restrict__ bytes, uint64_t w0,
                                                         uint64_t w1) {
    const uint64_t i = w0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= w1) return;
    const uint32_t w = words[i];
    uint32_t       o[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) v |= ((0x47544341u >> (8u * ((w >> (2 * (4 * d + b))) & 3u))) & 0xFFu) << (8 * b);  // "ACTG"[code]
        o[d] = v;
    }
    bytes[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void wfa_scale_offsets_kernel(uint64_t *q_off, uint64_t *t_off, uint64_t n) {  // words -> bytes
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) q_off[i] *= 16, t_off[i] *= 16;
}

struct PackedFacts {
    uint32_t max_len;
    uint64_t sum_len;
};
// packed != nullptr: the sequences arrive 2-bit packed (word i of `packed` = bytes [16 i, 16 i + 16) of the blob the
// offsets refer to); seq_blob is not read.
static int align_batch_impl(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *seq_blob,
                            uint64_t blob_bytes, const uint64_t *q_off, const uint32_t *q_len,
                            const uint64_t *t_off, const uint32_t *t_len, uint64_t n_pairs,
                            wfahip_results *out, const uint32_t *packed = nullptr,
                            const std::function<int(uint64_t, uint64_t)> *lazy_pack = nullptr, const PackedFacts *facts = nullptr) {
    // (facts: the caller is the library itself -- it laid the packed words out pair after pair and has already validated the
    // caller's offsets and summed the lengths: the three passes over a million pairs this function would make are 3 ms of a 38 ms call)
    // (lazy_pack: `packed` is the library's own buffer and is only filled as the pipeline gets to a range of pairs --
    // lazy_pack(first, last) packs pairs [first, last) and returns 0, or 2 when it meets a byte outside ACGT)
    if (!ctx || !out) return WFAHIP_ERR_BAD_ARG;
    // (pre-packed input: q_off / t_off arrive in WORDS of 16 bases; they are uploaded as they are and scaled to byte
    // offsets on the device -- a second pair of host arrays would cost more in page faults than the alignment of a slice)
    const uint64_t osc = packed ? 16 : 1;
    results_zero(out);
    int rc = check_params(p);
    if (rc) return rc;
    if (n_pairs == 0) return WFAHIP_OK;
    if (!q_off || !q_len || !t_off || !t_len || (!seq_blob && !packed && blob_bytes)) return WFAHIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    // the bytes [lo, hi) of the blob -> in_blob on stream `s` (packed input: the words that hold them, then the unpack kernel)
    const auto upload_range = [&](uint64_t lo, uint64_t hi, hipStream_t s) -> hipError_t {
        if (hi <= lo) return hipSuccess;
        if (!packed) return hipMemcpyAsync(static_cast<char *>(ctx->in_blob.p) + lo, seq_blob + lo, hi - lo, hipMemcpyHostToDevice, s);
        // (the words stay words: the alignment copies them into the first pass's slots and expands only the pairs a pass reads
        // as bytes -- ctx->pk_words below; wfa_unpack_kernel, which expanded everything, is kept for option "unpack_all")
        const uint64_t w0 = lo / 16, w1 = (hi + 15) / 16;
        hipError_t     e  = hipMemcpyAsync(static_cast<uint32_t *>(ctx->in_packed.p) + w0, packed + w0, (w1 - w0) * 4, hipMemcpyHostToDevice, s);
        if (e != hipSuccess || ctx->opt_unpack_all == 0) return e;
        hipLaunchKernelGGL(wfa_unpack_kernel, dim3((uint32_t)((w1 - w0 + 255) / 256)), dim3(256), 0, s,
                           static_cast<const uint32_t *>(ctx->in_packed.p), static_cast<uint4 *>(ctx->in_blob.p), w0, w1);
        return hipGetLastError();
    };
    struct PkGuard {  // the packed words are the alignment's input for the duration of this call only
        wfahip_ctx *c;
        ~PkGuard() { c->pk_words = nullptr; }
    } pk_guard{ctx};
    // (ctx->pk_words is set once in_packed is allocated, below)

    uint32_t max_len = facts ? std::max(1u, facts->max_len) : 1;
    uint64_t sum_len = facts ? facts->sum_len : 0;
    for (uint64_t i = 0; i < n_pairs && !facts; i++) {
        if (q_len[i] <= WFAHIP_MAX_SEQ_LEN && t_len[i] <= WFAHIP_MAX_SEQ_LEN && q_len[i] && t_len[i]) {
            // (written so that a hostile 64-bit offset cannot wrap the sum around)
            if (q_off[i] > blob_bytes / osc || q_len[i] > blob_bytes - q_off[i] * osc || t_off[i] > blob_bytes / osc ||
                t_len[i] > blob_bytes - t_off[i] * osc)
                return WFAHIP_ERR_BAD_ARG;
            max_len = std::max(max_len, std::max(q_len[i], t_len[i]));
            sum_len += (uint64_t)q_len[i] + t_len[i];
        }
    }
    hipStream_t st = ctx->stream;
    // device staging (+16 bytes so aligned dword loads at the tail stay inside the allocation)
    if ((rc = ensure(ctx, ctx->in_blob, blob_bytes + 32))) return rc;
    if (packed && (rc = ensure(ctx, ctx->in_packed, (blob_bytes + 15) / 16 * 4 + 16))) return rc;
    if (packed && ctx->opt_unpack_all == 0) ctx->pk_words = static_cast<const uint32_t *>(ctx->in_packed.p);
    if ((rc = ensure(ctx, ctx->in_qoff, n_pairs * 8))) return rc;
    if ((rc = ensure(ctx, ctx->in_toff, n_pairs * 8))) return rc;
    if ((rc = ensure(ctx, ctx->in_qlen, n_pairs * 4))) return rc;
    if ((rc = ensure(ctx, ctx->in_tlen, n_pairs * 4))) return rc;
    if ((rc = ensure(ctx, ctx->out_rec, n_pairs * REC_WORDS * 4))) return rc;
    const bool dbg_t = std::getenv("WFAHIP_DEBUG_TIMING") != nullptr;
    auto       now   = [] { return std::chrono::steady_clock::now(); };
    auto       ms_of = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_h2d = now();
    // Large batches: the pairs are aligned in a few slices, each as soon as the part of the blob it refers to has
    // arrived (an uploader thread feeds a copy stream), so most of the alignment time hides behind the upload.
    // Needs the blob ranges of consecutive slices to be disjoint enough (pairs laid out in order, the usual case).
    constexpr int     UP_SLICES = 6;  // (at most; packed input -- a quarter of the bytes -- is cut into three: a slice costs ~2 ms of its own)
    // (measured, 1e6 x 1 kbp packed on the fly: 4 slices 35.8 ms, 5 slices 34.8, 3 slices 39; six fall under the pair count at
    // which the variable-lanes kernel takes a pass: 46)
    int               n_sl = (packed && !lazy_pack) ? 3 : (lazy_pack && n_pairs >= 900000 ? 5 : 4);
    if (const char *e = std::getenv("WFAHIP_SLICES")) n_sl = std::max(1, std::min(UP_SLICES, std::atoi(e)));
    uint64_t          sl_first[UP_SLICES + 1], sl_lo[UP_SLICES], sl_hi[UP_SLICES];
    bool              sliced = n_pairs >= 200000 && blob_bytes >= (packed ? (256u << 20) : (64u << 20)) && !std::getenv("WFAHIP_NO_UPLOAD_OVERLAP");
    if (sliced) {
        uint64_t covered = 0;
        for (int k = 0; k <= UP_SLICES; k++) sl_first[k] = k <= n_sl ? n_pairs * k / n_sl : n_pairs;
        // (packing on the fly: nothing can be uploaded before the first slice is packed, so the first slice is a small one; and
        // the last one too: its results are downloaded with nothing left to hide them behind)
        if (lazy_pack && n_sl == 4) sl_first[1] = n_pairs * 12 / 100, sl_first[2] = n_pairs * 46 / 100, sl_first[3] = n_pairs * 80 / 100;
        if (lazy_pack && n_sl == 3) sl_first[1] = n_pairs * 14 / 100, sl_first[2] = n_pairs * 62 / 100;
        if (lazy_pack && n_sl >= 5) {  // a small first slice, a smaller last one, equal ones between
            sl_first[1] = n_pairs * 10 / 100;
            for (int k = 2; k < n_sl; k++) sl_first[k] = n_pairs * (10 + (k - 1) * 74 / (n_sl - 2)) / 100;
        }
        for (int k = 0; k < n_sl && sliced; k++) {
            uint64_t lo = blob_bytes, hi = 0;
            if (facts && sl_first[k + 1] > sl_first[k]) {  // (pair after pair: a slice's words are one range, query of its first pair .. target of its last)
                const uint64_t a = sl_first[k], b = sl_first[k + 1] - 1;
                lo = q_off[a] * osc, hi = (t_off[b] + wfahip_packed_words(t_len[b] <= WFAHIP_MAX_SEQ_LEN && q_len[b] && t_len[b] && q_len[b] <= WFAHIP_MAX_SEQ_LEN ? t_len[b] : 0)) * osc;
            }
            for (uint64_t i = sl_first[k]; i < sl_first[k + 1] && !facts; i++) {
                if (!(q_len[i] <= WFAHIP_MAX_SEQ_LEN && t_len[i] <= WFAHIP_MAX_SEQ_LEN && q_len[i] && t_len[i])) continue;
                lo = std::min(lo, std::min(q_off[i], t_off[i]) * osc);
                hi = std::max(hi, std::max(q_off[i] * osc + q_len[i], t_off[i] * osc + t_len[i]));
            }
            if (hi <= lo) lo = hi = 0;
            lo &= ~15ull;  // (whole aligned dwords of the first sequence; the tail padding of in_blob covers the end)
            sl_lo[k] = lo, sl_hi[k] = hi, covered += hi - lo;
        }
        sliced = covered <= blob_bytes + blob_bytes / 4;
    }
    if (sliced) {
        if (!ctx->stream_up) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_up, hipStreamNonBlocking));
        if (!ctx->stream_dn) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_dn, hipStreamNonBlocking));
        while (ctx->ev_up.size() < (size_t)UP_SLICES) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_up.push_back(e);
        }
    }
    // Small batches (a caller that cannot batch: Align = a batch of one): every copy between pageable memory and the
    // device is a staged transfer of 15-25 us, and a call makes 5 of them on the way in and 13 on the way out.  Here
    // the inputs travel as ONE image through the context's pinned buffer and the result arrays come back as one.
    const uint64_t small_img = ((blob_bytes + 15) & ~15ull) + 24 * n_pairs + 64;
    const bool     small     = !sliced && !packed && n_pairs <= 4096 && small_img <= (4u << 20);
    void *d_blob = ctx->in_blob.p, *d_qoff = ctx->in_qoff.p, *d_toff = ctx->in_toff.p, *d_qlen = ctx->in_qlen.p, *d_tlen = ctx->in_tlen.p;
    if (lazy_pack && !sliced) {
        const int e = (*lazy_pack)(0, n_pairs);
        if (e) return e == 2 ? WFAHIP_ERR_UNSUPPORTED : WFAHIP_ERR_HIP;
    }
    if (small) {
        if (!ctx->pin[0]) HIP_TRY(hipHostMalloc(&ctx->pin[0], PIN_CHUNK, hipHostMallocDefault));
        if ((rc = ensure(ctx, ctx->in_small, small_img))) return rc;
        char *const    img = static_cast<char *>(ctx->pin[0]);
        const uint64_t o1 = (blob_bytes + 15) & ~15ull, o2 = o1 + 8 * n_pairs, o3 = o2 + 8 * n_pairs, o4 = o3 + 4 * n_pairs;
        if (blob_bytes) std::memcpy(img, seq_blob, blob_bytes);
        std::memcpy(img + o1, q_off, 8 * n_pairs), std::memcpy(img + o2, t_off, 8 * n_pairs);
        std::memcpy(img + o3, q_len, 4 * n_pairs), std::memcpy(img + o4, t_len, 4 * n_pairs);
        HIP_TRY(hipMemcpyAsync(ctx->in_small.p, img, o4 + 4 * n_pairs, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // (the pinned buffer is reused for the results)
        char *const d = static_cast<char *>(ctx->in_small.p);
        d_blob = d, d_qoff = d + o1, d_toff = d + o2, d_qlen = d + o3, d_tlen = d + o4;
    } else {
        if (blob_bytes && !sliced) HIP_TRY(upload_range(0, blob_bytes, st));
        HIP_TRY(hipMemcpyAsync(ctx->in_qoff.p, q_off, n_pairs * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->in_toff.p, t_off, n_pairs * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->in_qlen.p, q_len, n_pairs * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->in_tlen.p, t_len, n_pairs * 4, hipMemcpyHostToDevice, st));
        if (packed) {
            hipLaunchKernelGGL(wfa_scale_offsets_kernel, dim3((uint32_t)((n_pairs + 255) / 256)), dim3(256), 0, st,
                               static_cast<uint64_t *>(ctx->in_qoff.p), static_cast<uint64_t *>(ctx->in_toff.p), n_pairs);
            HIP_TRY(hipGetLastError());
        }
    }

    if (dbg_t) HIP_TRY(hipStreamSynchronize(st));
    const auto t_dev = now();
    // CIGAR ops are merged runs: a first guess of (n+m)/4 + 8 per pair, grown on demand (at most n+m+2 each)
    uint64_t ops_cap = sum_len / 4 + 8 * n_pairs + 1024;
    for (int attempt = 0; attempt < 6; attempt++) {
        if ((rc = ensure(ctx, ctx->out_ops, ops_cap * 8))) return rc;
        uint64_t needed = 0;
        if (sliced && attempt == 0) {
            std::atomic<int> recorded{0}, up_err{0};
            const auto upload_all = [&] {
                if (hipSetDevice(ctx->device) != hipSuccess) up_err = 1;
                for (int k = 0; k < n_sl; k++) {
                    // (host-side packing of slice k runs here, beside the upload of slice k-1 and the alignment of earlier ones)
                    if (!up_err && lazy_pack) {
                        const int e = (*lazy_pack)(sl_first[k], sl_first[k + 1]);
                        if (e) up_err = e == 2 ? 2 : 1;
                    }
                    if (!up_err && sl_hi[k] > sl_lo[k] && upload_range(sl_lo[k], sl_hi[k], ctx->stream_up) != hipSuccess) up_err = 1;
                    if (hipEventRecord(ctx->ev_up[k], ctx->stream_up) != hipSuccess) up_err = 1;
                    recorded = k + 1;
                }
            };
            std::thread uploader;
            try {
                uploader = std::thread(upload_all);
            } catch (...) {  // no thread to be had: upload here, then align (no overlap, same result)
                upload_all();
            }
            wfahip_timing acc{};
            uint64_t      cursor = 0;
            // ---- per-slice assembly (wfa_finalize.hpp) and download beside the next slice's alignment
            const uint64_t n = n_pairs;
            const uint64_t per_blk = (uint64_t)FIN_BLOCK * FIN_ITEMS;
            auto           al8s    = [](uint64_t v) { return (v + 7) & ~7ull; };
            uint64_t       foff    = 0;
            auto           takes   = [&](uint64_t bytes) {
                const uint64_t o = foff;
                foff += al8s(bytes);
                return o;
            };
            const uint64_t so_tot = takes(16 * UP_SLICES), so_blk = takes(8ull * (n / per_blk + UP_SLICES + 1)), so_ooff = takes(8ull * n),
                           so_loc = takes(8ull * n);
            uint64_t so_f[11];
            for (int i = 0; i < 11; i++) so_f[i] = takes(4ull * n);
            const uint64_t so_ops = takes(8ull * ops_cap);
            if ((rc = ensure(ctx, ctx->fin, foff))) {
                if (uploader.joinable()) uploader.join();
                return rc;
            }
            char *const sfb = static_cast<char *>(ctx->fin.p);
            results_zero(out);
            out->n = n;
            bool alloc_ok = true;
            {
                const size_t cnt = std::max<uint64_t>(n, 1);
#define ALLOCS(field, type) alloc_ok = alloc_ok && (out->field = static_cast<type *>(res_alloc(cnt * sizeof(type)))) != nullptr;
                ALLOCS(status, int32_t) ALLOCS(score, uint32_t) ALLOCS(tbegin, int32_t) ALLOCS(tend, int32_t)
                ALLOCS(qbegin, int32_t) ALLOCS(qend, int32_t) ALLOCS(align_len, uint32_t) ALLOCS(matches, uint32_t)
                ALLOCS(gaps, uint32_t) ALLOCS(gap_regions, uint32_t) ALLOCS(ops_off, uint64_t) ALLOCS(ops_len, uint32_t)
#undef ALLOCS
            }
            uint64_t         ops_done = 0, cells_done = 0, host_ops_cap = 0, blk_done = 0;
            uint64_t         deferred_at = ~0ull;  // ops from this offset on did not fit the estimated host array: downloaded at the end
            // one downloader thread takes the slices in order as the main thread marks them ready (state 1 = ready,
            // 2 = nothing to download, skip)
            struct DlTask { uint64_t k0, nk, o0, ops_k; bool with_ops; };
            DlTask           dl_tasks[UP_SLICES] = {};
            std::atomic<int> dl_state[UP_SLICES];
            for (auto &a : dl_state) a = 0;
            std::atomic<int> dl_err{0};
            const auto dl_run = [&] {
                if (hipSetDevice(ctx->device) != hipSuccess) dl_err = 1;
                for (int k = 0; k < n_sl; k++) {
                    while (dl_state[k].load() == 0) std::this_thread::yield();
                    if (dl_state[k].load() == 2 || dl_err) continue;
                    const DlTask t       = dl_tasks[k];
                    void *const dsts[11] = {out->status, out->score, out->tbegin, out->tend, out->qbegin, out->qend,
                                            out->align_len, out->matches, out->gaps, out->gap_regions, out->ops_len};
                    int r2 = WFAHIP_OK;
                    for (int i = 0; i < 11 && r2 == WFAHIP_OK; i++)
                        r2 = download(ctx, static_cast<char *>(dsts[i]) + 4 * t.k0, sfb + so_f[i] + 4 * t.k0, 4ull * t.nk, ctx->stream_dn);
                    if (r2 == WFAHIP_OK) r2 = download(ctx, out->ops_off + t.k0, sfb + so_ooff + 8 * t.k0, 8ull * t.nk, ctx->stream_dn);
                    if (r2 == WFAHIP_OK && t.with_ops) r2 = download(ctx, out->ops + t.o0, sfb + so_ops + 8 * t.o0, 8ull * t.ops_k, ctx->stream_dn);
                    if (r2 != WFAHIP_OK) dl_err = 1;
                }
            };
            std::thread dl_thread;
            bool        dl_inline = false;
            try {
                dl_thread = std::thread(dl_run);
            } catch (...) {
                dl_inline = true;  // no thread: everything is downloaded after the last slice
            }
            const auto dl_finish = [&] {  // slices never marked (early exit) are skipped; then the downloads are waited for / run here
                for (auto &a : dl_state) {
                    int z = 0;
                    a.compare_exchange_strong(z, 2);
                }
                if (dl_thread.joinable()) dl_thread.join();
                else if (dl_inline) dl_run();
            };
            if (!alloc_ok) rc = WFAHIP_ERR_OOM;
            for (int k = 0; k < n_sl && rc == WFAHIP_OK; k++) {
                while (recorded.load() <= k) std::this_thread::yield();  // (the event must have been recorded before the wait)
                if (up_err) {
                    rc = up_err == 2 ? WFAHIP_ERR_UNSUPPORTED : WFAHIP_ERR_HIP;  // (2: a byte outside ACGT -- the caller falls back to the byte path)
                    break;
                }
                if (hipStreamWaitEvent(st, ctx->ev_up[k], 0) != hipSuccess) {
                    rc = WFAHIP_ERR_HIP;
                    break;
                }
                const uint64_t k0 = sl_first[k], nk = sl_first[k + 1] - k0;
                if (nk == 0) {
                    dl_state[k] = 2;
                    continue;
                }
                if (dbg_t) {
                    (void)hipEventSynchronize(ctx->ev_up[k]);
                    std::fprintf(stderr, "[wfahip]   slice %d: blob part here at +%.1f ms\n", k, ms_of(t_dev, now()));
                }
                rc = align_device(ctx, p, ctx->in_blob.p, blob_bytes, static_cast<char *>(ctx->in_qoff.p) + 8 * k0,
                                  static_cast<char *>(ctx->in_qlen.p) + 4 * k0, static_cast<char *>(ctx->in_toff.p) + 8 * k0,
                                  static_cast<char *>(ctx->in_tlen.p) + 4 * k0, nk, max_len,
                                  static_cast<char *>(ctx->out_rec.p) + (size_t)REC_WORDS * 4 * k0, ctx->out_ops.p, ops_cap, &needed, st,
                                  false, cursor);
                cursor = ctx->timing.ops_written;
                acc.kernel_ms += ctx->timing.kernel_ms, acc.total_ms += ctx->timing.total_ms, acc.n_launches += ctx->timing.n_launches;
                acc.n_retried_pairs += ctx->timing.n_retried_pairs, acc.main_kernel_ms += ctx->timing.main_kernel_ms;
                acc.n_main_launches += ctx->timing.n_main_launches, acc.n_packed_pairs += ctx->timing.n_packed_pairs;
                acc.arena_bytes      = std::max(acc.arena_bytes, ctx->timing.arena_bytes);
                acc.main_kernel_kind = ctx->timing.main_kernel_kind;
                if (rc != WFAHIP_OK) break;
                // assemble this slice's part of the result arrays
                const uint32_t nb_k = (uint32_t)((nk + per_blk - 1) / per_blk);
                FinParams F{};
                F.rec = static_cast<const uint32_t *>(ctx->out_rec.p) + (size_t)REC_WORDS * k0;
                F.ops = static_cast<const uint64_t *>(ctx->out_ops.p), F.n = nk;
                F.totals  = reinterpret_cast<unsigned long long *>(sfb + so_tot) + 2 * k;
                F.blk_sum = reinterpret_cast<uint64_t *>(sfb + so_blk) + blk_done;
                F.loc_off = reinterpret_cast<uint64_t *>(sfb + so_loc) + k0;
                F.ops_off = reinterpret_cast<uint64_t *>(sfb + so_ooff) + k0;
                F.status = reinterpret_cast<int32_t *>(sfb + so_f[0]) + k0, F.score = reinterpret_cast<uint32_t *>(sfb + so_f[1]) + k0;
                F.tbegin = reinterpret_cast<int32_t *>(sfb + so_f[2]) + k0, F.tend = reinterpret_cast<int32_t *>(sfb + so_f[3]) + k0;
                F.qbegin = reinterpret_cast<int32_t *>(sfb + so_f[4]) + k0, F.qend = reinterpret_cast<int32_t *>(sfb + so_f[5]) + k0;
                F.align_len = reinterpret_cast<uint32_t *>(sfb + so_f[6]) + k0, F.matches = reinterpret_cast<uint32_t *>(sfb + so_f[7]) + k0;
                F.gaps = reinterpret_cast<uint32_t *>(sfb + so_f[8]) + k0, F.gap_regions = reinterpret_cast<uint32_t *>(sfb + so_f[9]) + k0;
                F.ops_len = reinterpret_cast<uint32_t *>(sfb + so_f[10]) + k0;
                F.ops_out = reinterpret_cast<uint64_t *>(sfb + so_ops) + ops_done, F.ops_base = ops_done;
                bool hip_ok = hipMemsetAsync(F.totals, 0, 16, st) == hipSuccess;
                hipLaunchKernelGGL(fin_scan_blocks, dim3(nb_k), dim3(FIN_BLOCK), 0, st, F);
                hipLaunchKernelGGL(fin_scan_sums, dim3(1), dim3(1024), 0, st, F, nb_k);
                hipLaunchKernelGGL(fin_gather, dim3((uint32_t)((nk + 3) / 4)), dim3(256), 0, st, F);
                hip_ok = hip_ok && hipGetLastError() == hipSuccess &&
                         hipMemcpyAsync(ctx->hpin + HPIN_CTRL + CTRL_WORDS, F.totals, 16, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipStreamSynchronize(st) == hipSuccess;
                if (!hip_ok) {
                    rc = WFAHIP_ERR_HIP;
                    break;
                }
                unsigned long long tk[2];
                std::memcpy(tk, ctx->hpin + HPIN_CTRL + CTRL_WORDS, 16);
                const uint64_t ops_k = tk[0];
                cells_done += tk[1], blk_done += nb_k;
                if (!out->ops) {  // size the host op array from the first slice: ops per pair x pairs + 15 %
                    host_ops_cap = std::min<uint64_t>(ops_cap, (uint64_t)((double)ops_k / (double)nk * (double)n * 1.15) + 65536);
                    out->ops     = static_cast<uint64_t *>(res_alloc(std::max<uint64_t>(host_ops_cap, 1) * 8));
                    if (!out->ops) {
                        rc = WFAHIP_ERR_OOM;
                        break;
                    }
                }
                if (deferred_at == ~0ull && ops_done + ops_k > host_ops_cap) deferred_at = ops_done;
                if (dbg_t) std::fprintf(stderr, "[wfahip]   slice %d: aligned + assembled at +%.1f ms\n", k, ms_of(t_dev, now()));
                dl_tasks[k] = DlTask{k0, nk, ops_done, ops_k, deferred_at == ~0ull};
                dl_state[k] = 1;
                ops_done += ops_k;
            }
            if (uploader.joinable()) uploader.join();
            dl_finish();
            if (rc == WFAHIP_OK && dl_err) rc = WFAHIP_ERR_HIP;
            if (rc == WFAHIP_OK || rc == WFAHIP_ERR_OOM) {
                acc.ops_written = ctx->timing.ops_written;
                ctx->timing     = acc;
            }
            if (rc == WFAHIP_ERR_OOM && alloc_ok && needed > ops_cap) {  // the op buffer was too small: the whole blob is resident now, one plain call redoes it
                wfahip_results_free(out);
                HIP_TRY(hipStreamSynchronize(ctx->stream_up));
                ops_cap = std::max(needed, ops_cap) + ops_cap / 2 + 1024;
                continue;
            }
            if (rc == WFAHIP_OK && deferred_at != ~0ull) {  // more ops than estimated: an array of the exact size takes what is there + the rest
                uint64_t *full = static_cast<uint64_t *>(res_alloc(std::max<uint64_t>(ops_done, 1) * 8));
                if (!full) {
                    rc = WFAHIP_ERR_OOM;
                } else {
                    std::memcpy(full, out->ops, deferred_at * 8);
                    res_release(out->ops);
                    out->ops = full;
                    rc = download(ctx, out->ops + deferred_at, sfb + so_ops + 8 * deferred_at, 8ull * (ops_done - deferred_at), st);
                }
            }
            if (rc != WFAHIP_OK) {
                wfahip_results_free(out);
                return rc;
            }
            out->n_ops               = ops_done;
            ctx->timing.cells_stored = cells_done;
            if (dbg_t)
                std::fprintf(stderr, "[wfahip] host entry (sliced): small arrays %.1f ms, upload + alignment + assembly + download %.1f ms\n",
                             ms_of(t_h2d, t_dev), ms_of(t_dev, now()));
            return WFAHIP_OK;
        }
        rc = align_device(ctx, p, d_blob, blob_bytes, d_qoff, d_qlen, d_toff, d_tlen, n_pairs, max_len, ctx->out_rec.p, ctx->out_ops.p,
                          ops_cap, &needed, st, false);
        if (rc == WFAHIP_ERR_OOM && needed > ops_cap) {
            // (with headroom: what a call needs is not the same on every attempt -- a context learns the rows and the window a
            // class of pairs needs while it runs, and the retry passes it takes, each reserving op slots, change with that)
            ops_cap = needed + needed / 3 + 1024;
            continue;
        }
        break;
    }
    if (rc) return rc;

    const auto t_d2h = now();
    // ---- assemble wfahip_results on the device (ops packed in pair order, one array per field), then download
    const uint64_t n         = n_pairs;
    const uint32_t n_blocks  = (uint32_t)((n + (uint64_t)FIN_BLOCK * FIN_ITEMS - 1) / ((uint64_t)FIN_BLOCK * FIN_ITEMS));
    const uint64_t ops_total_cap = std::max<uint64_t>(ctx->timing.ops_written, 1);
    auto           al8       = [](uint64_t v) { return (v + 7) & ~7ull; };
    // layout of ctx->fin (bytes): totals[2] | blk_sum[n_blocks] | ops_off[n] | loc_off[n] | 11 x u32[n] | ops_out
    uint64_t off = 0;
    auto     take = [&](uint64_t bytes) {
        const uint64_t o = off;
        off += al8(bytes);
        return o;
    };
    const uint64_t o_tot = take(16), o_blk = take(8ull * n_blocks), o_ooff = take(8ull * n), o_loc = take(8ull * n);
    uint64_t       o_f[11];
    for (int i = 0; i < 11; i++) o_f[i] = take(4ull * n);
    const uint64_t o_ops = take(8ull * ops_total_cap);
    if ((rc = ensure(ctx, ctx->fin, off))) return rc;
    char *fb = static_cast<char *>(ctx->fin.p);
    FinParams F{};
    F.rec = static_cast<const uint32_t *>(ctx->out_rec.p), F.ops = static_cast<const uint64_t *>(ctx->out_ops.p), F.n = n;
    F.totals  = reinterpret_cast<unsigned long long *>(fb + o_tot);
    F.blk_sum = reinterpret_cast<uint64_t *>(fb + o_blk), F.ops_off = reinterpret_cast<uint64_t *>(fb + o_ooff);
    F.loc_off = reinterpret_cast<uint64_t *>(fb + o_loc);
    F.status = reinterpret_cast<int32_t *>(fb + o_f[0]), F.score = reinterpret_cast<uint32_t *>(fb + o_f[1]);
    F.tbegin = reinterpret_cast<int32_t *>(fb + o_f[2]), F.tend = reinterpret_cast<int32_t *>(fb + o_f[3]);
    F.qbegin = reinterpret_cast<int32_t *>(fb + o_f[4]), F.qend = reinterpret_cast<int32_t *>(fb + o_f[5]);
    F.align_len = reinterpret_cast<uint32_t *>(fb + o_f[6]), F.matches = reinterpret_cast<uint32_t *>(fb + o_f[7]);
    F.gaps = reinterpret_cast<uint32_t *>(fb + o_f[8]), F.gap_regions = reinterpret_cast<uint32_t *>(fb + o_f[9]);
    F.ops_len = reinterpret_cast<uint32_t *>(fb + o_f[10]), F.ops_out = reinterpret_cast<uint64_t *>(fb + o_ops);
    HIP_TRY(hipMemsetAsync(F.totals, 0, 16, st));
    hipLaunchKernelGGL(fin_scan_blocks, dim3(n_blocks), dim3(FIN_BLOCK), 0, st, F);
    hipLaunchKernelGGL(fin_scan_sums, dim3(1), dim3(1024), 0, st, F, n_blocks);
    hipLaunchKernelGGL(fin_gather, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, F);
    HIP_TRY(hipGetLastError());
    unsigned long long totals[2] = {0, 0};
    const bool         small_out = small && off <= PIN_CHUNK;  // every result array in ONE copy
    if (small_out) {
        HIP_TRY(hipMemcpyAsync(ctx->pin[0], fb, off, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::memcpy(totals, static_cast<char *>(ctx->pin[0]) + o_tot, 16);
    } else {
        HIP_TRY(hipMemcpyAsync(ctx->hpin + HPIN_CTRL + CTRL_WORDS, F.totals, 16, hipMemcpyDeviceToHost, st));  // (pinned)
        HIP_TRY(hipStreamSynchronize(st));
        std::memcpy(totals, ctx->hpin + HPIN_CTRL + CTRL_WORDS, 16);
    }
    const auto t_unp = now();

    results_zero(out);
    out->n = n;
    {
        const size_t cnt = std::max<uint64_t>(n, 1);
#define ALLOC(field, type)                                          \
    out->field = static_cast<type *>(res_alloc(cnt * sizeof(type))); \
    if (!out->field) {                                               \
        wfahip_results_free(out);                                    \
        return WFAHIP_ERR_OOM;                                       \
    }
        ALLOC(status, int32_t) ALLOC(score, uint32_t) ALLOC(tbegin, int32_t) ALLOC(tend, int32_t)
        ALLOC(qbegin, int32_t) ALLOC(qend, int32_t) ALLOC(align_len, uint32_t) ALLOC(matches, uint32_t)
        ALLOC(gaps, uint32_t) ALLOC(gap_regions, uint32_t) ALLOC(ops_off, uint64_t) ALLOC(ops_len, uint32_t)
#undef ALLOC
        out->n_ops = totals[0];
        out->ops   = static_cast<uint64_t *>(res_alloc(std::max<uint64_t>(totals[0], 1) * 8));
        if (!out->ops) {
            wfahip_results_free(out);
            return WFAHIP_ERR_OOM;
        }
    }
    void *const       dsts[11] = {out->status, out->score, out->tbegin, out->tend, out->qbegin, out->qend,
                                  out->align_len, out->matches, out->gaps, out->gap_regions, out->ops_len};
    if (small_out) {
        const char *const img = static_cast<const char *>(ctx->pin[0]);
        for (int i = 0; i < 11; i++) std::memcpy(dsts[i], img + o_f[i], 4ull * n);
        std::memcpy(out->ops_off, img + o_ooff, 8ull * n);
        if (totals[0]) std::memcpy(out->ops, img + o_ops, 8ull * totals[0]);
    } else {
        for (int i = 0; i < 11 && rc == WFAHIP_OK; i++) rc = download(ctx, dsts[i], fb + o_f[i], 4ull * n, st);
        if (rc == WFAHIP_OK) rc = download(ctx, out->ops_off, fb + o_ooff, 8ull * n, st);
        if (rc == WFAHIP_OK) rc = download(ctx, out->ops, fb + o_ops, 8ull * totals[0], st);
    }
    const uint64_t cells = totals[1];
    if (dbg_t)
        std::fprintf(stderr, "[wfahip] host entry: H2D %.1f ms, device %.1f ms, finalize %.1f ms, download %.1f ms\n",
                     ms_of(t_h2d, t_dev), ms_of(t_dev, t_d2h), ms_of(t_d2h, t_unp), ms_of(t_unp, now()));
    ctx->timing.cells_stored = cells;
    if (rc) wfahip_results_free(out);
    return rc;
}

// 16 bases -> one word, eight bytes at a time: the codes are (byte >> 1) & 3, gathered by three shift-or steps; a byte
// outside ACGT shows as a difference between the byte and the canonical letter of its code (0x41 + 2 code, + 15 for T).
namespace {
inline uint32_t pack8(uint64_t w, uint64_t &bad) {
    const uint64_t x = (w >> 1) & 0x0303030303030303ull;
    const uint64_t t = (x >> 1) & ~x & 0x0101010101010101ull;  // code 2 = 'T'
    bad |= (0x4141414141414141ull + 2 * x + 15 * t) ^ w;
    uint64_t y = (x | (x >> 6)) & 0x000F000F000F000Full;
    y          = (y | (y >> 12)) & 0x000000FF000000FFull;
    return (uint32_t)((y | (y >> 24)) & 0xFFFFull);
}
// one sequence -> dst[0 .. (len + 15) / 16] (the last word is the zero pad word); returns true on a byte outside ACGT
bool pack_seq_fast(const uint8_t *s, uint32_t len, uint32_t *dst) {
    uint64_t       bad = 0;
    const uint32_t nw = len / 16;
    for (uint32_t w = 0; w < nw; w++) {
        uint64_t a, b;
        std::memcpy(&a, s + 16 * w, 8), std::memcpy(&b, s + 16 * w + 8, 8);
        dst[w] = pack8(a, bad) | (pack8(b, bad) << 16);
    }
    const uint32_t rem = len - 16 * nw;
    if (rem) {
        uint8_t tail[16];
        std::memset(tail, 'A', 16);
        std::memcpy(tail, s + 16 * nw, rem);
        uint64_t a, b;
        std::memcpy(&a, tail, 8), std::memcpy(&b, tail + 8, 8);
        dst[nw] = pack8(a, bad) | (pack8(b, bad) << 16);
        dst[nw + 1] = 0;
    } else {
        dst[nw] = 0;
    }
    return bad != 0;
}
}  // namespace

// wfahip_align_batch on a large batch: a quarter of the bytes cross PCIe.  The sequences are 2-bit packed by host threads,
// a slice at a time, into a page-locked buffer of the context -- slice k is packed while slice k-1 uploads and earlier
// slices are being aligned -- and expanded again on the device (wfa_unpack_kernel), where bandwidth is free.  A byte outside
// ACGT anywhere (the reference compares raw bytes, wfa.go:408-454) ends the attempt and the batch takes the byte path.
static int align_batch_autopack(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *seq_blob, uint64_t blob_bytes,
                                const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off, const uint32_t *t_len,
                                uint64_t n_pairs, wfahip_results *out) {
    if (!ctx || !out || !q_off || !q_len || !t_off || !t_len || !seq_blob) return WFAHIP_ERR_BAD_ARG;
    std::vector<uint64_t> &q_woff = ctx->pack_qw, &t_woff = ctx->pack_tw;  // (kept between calls: fresh pages cost more than the sums)
    q_woff.resize(n_pairs), t_woff.resize(n_pairs);
    uint64_t    pos = 0;
    PackedFacts facts{1u, 0ull};
    for (uint64_t i = 0; i < n_pairs; i++) {
        // (a pair the alignment rejects -- too long, or empty -- is not validated and packs as nothing: exactly the byte entry's rule)
        const bool v = q_len[i] <= WFAHIP_MAX_SEQ_LEN && t_len[i] <= WFAHIP_MAX_SEQ_LEN && q_len[i] && t_len[i];
        if (v && (q_off[i] > blob_bytes || q_len[i] > blob_bytes - q_off[i] || t_off[i] > blob_bytes || t_len[i] > blob_bytes - t_off[i]))
            return WFAHIP_ERR_BAD_ARG;
        if (v) facts.max_len = std::max(facts.max_len, std::max(q_len[i], t_len[i])), facts.sum_len += (uint64_t)q_len[i] + t_len[i];
        q_woff[i] = pos, pos += wfahip_packed_words(v ? q_len[i] : 0);
        t_woff[i] = pos, pos += wfahip_packed_words(v ? t_len[i] : 0);
    }
    // (pairs are packed one by one: a batch whose pairs SHARE sequences -- one target against many queries -- would carry every
    // copy over PCIe; beyond a quarter more than the blob itself the byte path is the cheaper one.  Sharing shows in the BASES
    // the pairs name against the bytes of the blob -- not in the packed words, whose pad word and 16-base rounding per
    // sequence alone are a quarter of a tightly laid-out batch of 100-base reads)
    if (facts.sum_len > blob_bytes + blob_bytes / 4) return WFAHIP_ERR_UNSUPPORTED;
    const size_t need = (size_t)(pos + 4) * 4;
    if (ctx->pack_pin_bytes < need) {
        HIP_TRY(hipSetDevice(ctx->device));
        if (ctx->pack_pin) (void)hipHostFree(ctx->pack_pin);
        ctx->pack_pin = nullptr, ctx->pack_pin_bytes = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&ctx->pack_pin), need + need / 8, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->pack_pin = nullptr;
            return WFAHIP_ERR_UNSUPPORTED;  // (no page-locked memory for the packed words: the byte path needs none)
        }
        ctx->pack_pin_bytes = need + need / 8;
    }
    uint32_t *const packed = ctx->pack_pin;
    unsigned n_thr = std::max(1u, std::min(64u, std::thread::hardware_concurrency() / 2));
    if (const char *e = std::getenv("WFAHIP_PACK_THREADS")) n_thr = (unsigned)std::max(1, std::atoi(e));
    const std::function<int(uint64_t, uint64_t)> lazy = [&](uint64_t first, uint64_t last) -> int {
        std::atomic<int> bad{0};
        const auto range = [&](uint64_t a, uint64_t b) {
            bool bd = false;
            for (uint64_t i = a; i < b; i++) {
                if (!(q_len[i] <= WFAHIP_MAX_SEQ_LEN && t_len[i] <= WFAHIP_MAX_SEQ_LEN && q_len[i] && t_len[i])) {
                    packed[q_woff[i]] = 0, packed[t_woff[i]] = 0;
                    continue;
                }
                bd |= pack_seq_fast(seq_blob + q_off[i], q_len[i], packed + q_woff[i]);
                bd |= pack_seq_fast(seq_blob + t_off[i], t_len[i], packed + t_woff[i]);
            }
            if (bd) bad = 1;
        };
        const uint64_t cnt = last - first;
        const unsigned nt  = (unsigned)std::min<uint64_t>(n_thr, cnt / 2048 + 1);
        std::vector<std::thread> th;
        const uint64_t per = (cnt + nt - 1) / nt;
        for (unsigned t = 0; t < nt; t++) {
            const uint64_t a = std::min<uint64_t>(last, first + (uint64_t)t * per), b = std::min<uint64_t>(last, a + per);
            bool inl = nt == 1;
            if (!inl) {
                try {
                    th.emplace_back(range, a, b);
                } catch (...) {
                    inl = true;
                }
            }
            if (inl) range(a, b);
        }
        for (auto &t : th) t.join();
        return bad ? 2 : 0;
    };
    return align_batch_impl(ctx, p, nullptr, pos * 16, q_woff.data(), q_len, t_woff.data(), t_len, n_pairs, out, packed, &lazy, &facts);
}

static int align_batch_entry(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *seq_blob, uint64_t blob_bytes,
                             const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off, const uint32_t *t_len,
                             uint64_t n_pairs, wfahip_results *out) {
    if (ctx && out && ctx->opt_autopack != 0 && seq_blob && n_pairs >= 200000 && blob_bytes >= (256u << 20) &&
        !std::getenv("WFAHIP_NO_AUTOPACK") && !std::getenv("WFAHIP_NO_UPLOAD_OVERLAP")) {
        const int rc = align_batch_autopack(ctx, p, seq_blob, blob_bytes, q_off, q_len, t_off, t_len, n_pairs, out);
        if (rc != WFAHIP_ERR_UNSUPPORTED || check_params(p) != WFAHIP_OK) return rc;
        // (a byte outside ACGT somewhere in the batch: the byte path takes all of it)
    }
    return align_batch_impl(ctx, p, seq_blob, blob_bytes, q_off, q_len, t_off, t_len, n_pairs, out);
}

extern "C" int wfahip_align_batch(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *seq_blob,
                                  uint64_t blob_bytes, const uint64_t *q_off, const uint32_t *q_len,
                                  const uint64_t *t_off, const uint32_t *t_len, uint64_t n_pairs,
                                  wfahip_results *out) {
    WFAHIP_GUARD(align_batch_entry(ctx, p, seq_blob, blob_bytes, q_off, q_len, t_off, t_len, n_pairs, out))
}

// ---- pre-packed input (a quarter of the bytes cross PCIe)
extern "C" int wfahip_align_batch_packed(wfahip_ctx *ctx, const wfahip_params *p, const uint32_t *packed, uint64_t n_words,
                                         const uint64_t *q_woff, const uint32_t *q_len, const uint64_t *t_woff,
                                         const uint32_t *t_len, uint64_t n_pairs, wfahip_results *out) {
    if (!ctx || !out || (!packed && n_words)) return WFAHIP_ERR_BAD_ARG;
    try {
        if (n_pairs && (!q_woff || !t_woff)) return WFAHIP_ERR_BAD_ARG;
        static const uint32_t no_words[4] = {0, 0, 0, 0};
        return align_batch_impl(ctx, p, nullptr, n_words * 16, q_woff, q_len, t_woff, t_len, n_pairs, out, packed ? packed : no_words);
    } catch (const std::bad_alloc &) {
        return WFAHIP_ERR_OOM;
    } catch (...) {
        return WFAHIP_ERR_INTERNAL;
    }
}

extern "C" uint64_t wfahip_packed_words(uint32_t len) { return ((uint64_t)len + 15) / 16 + 1; }  // (+1: the pad word the kernels' 16-base windows may read)

// Host-side packer: n_pairs (query, target) byte sequences -> 2-bit words, every sequence at a word boundary, in pair
// order (query then target).  Returns WFAHIP_ERR_UNSUPPORTED when a byte outside {A,C,G,T} is found (such a batch
// must use the byte entry: the reference compares raw bytes, wfa.go:408-454).  packed must hold
// sum(wfahip_packed_words(q_len[i]) + wfahip_packed_words(t_len[i])) words.
static int pack_pairs_impl(const uint8_t *seq_blob, const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off,
                           const uint32_t *t_len, uint64_t n_pairs, int n_threads, uint32_t *packed, uint64_t *q_woff,
                           uint64_t *t_woff, uint64_t *n_words) {
    if (!q_off || !q_len || !t_off || !t_len || !packed || !q_woff || !t_woff || (!seq_blob && n_pairs)) return WFAHIP_ERR_BAD_ARG;
    uint64_t pos = 0;
    for (uint64_t i = 0; i < n_pairs; i++) {
        q_woff[i] = pos, pos += wfahip_packed_words(q_len[i] <= WFAHIP_MAX_SEQ_LEN ? q_len[i] : 0);
        t_woff[i] = pos, pos += wfahip_packed_words(t_len[i] <= WFAHIP_MAX_SEQ_LEN ? t_len[i] : 0);
    }
    if (n_words) *n_words = pos;
    std::atomic<int> bad{0};
    const auto       one = [&](const uint8_t *s, uint32_t len, uint32_t *dst) {
        if (len > WFAHIP_MAX_SEQ_LEN) len = 0;  // (rejected per pair by the alignment itself)
        if (pack_seq_fast(s, len, dst)) bad = 1;  // (eight bytes at a time: round 2's byte loop packed 2 GB in 34 ms on 32 threads)
    };
    const auto range = [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            one(seq_blob + q_off[i], q_len[i], packed + q_woff[i]);
            one(seq_blob + t_off[i], t_len[i], packed + t_woff[i]);
        }
    };
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_pairs / 1024 + 1) n_threads = (int)(n_pairs / 1024 + 1);
    std::vector<std::thread> th;
    const uint64_t           per = (n_pairs + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; t++) {
        const uint64_t a = std::min<uint64_t>(n_pairs, (uint64_t)t * per), b = std::min<uint64_t>(n_pairs, a + per);
        bool inl = n_threads == 1;
        if (!inl) {
            try {
                th.emplace_back(range, a, b);
            } catch (...) {
                inl = true;
            }
        }
        if (inl) range(a, b);
    }
    for (auto &t : th) t.join();
    return bad ? WFAHIP_ERR_UNSUPPORTED : WFAHIP_OK;
}

extern "C" int wfahip_pack_pairs(const uint8_t *seq_blob, const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off,
                                 const uint32_t *t_len, uint64_t n_pairs, int n_threads, uint32_t *packed, uint64_t *q_woff,
                                 uint64_t *t_woff, uint64_t *n_words) {
    WFAHIP_GUARD(pack_pairs_impl(seq_blob, q_off, q_len, t_off, t_len, n_pairs, n_threads, packed, q_woff, t_woff, n_words))
}

// ---- Align (wfa.go:196) for ONE pair without the batch plumbing.
// A caller that cannot batch pays per call, not per base: through wfahip_align_batch a 1 kbp pair took 0.45-0.75 ms, of
// which the kernels were a fifth -- five launches, a dozen staged copies, thirteen result arrays.  Here the pair is
// written into a page-locked block that is mapped into the GPU's address space; the forward kernel reads the two
// sequences from there (2 KB over PCIe, inside its refill), the backtrace kernel writes the record and the CIGAR ops
// back into it, and the host waits for the stream once: one memset of the control words, two launches, no copy.
// Taken by global alignments with penalties shaped 2 : 4 : 1 (4/6/2, the default) whose sequences fit the blocked
// kernel's LDS budget and whose worst-case CIGAR fits the block; everything else -- and a pair the blocked kernel hands
// on (band wider than 64 diagonals, a byte outside ACGT, arena rows) -- goes through wfahip_align_batch: same results.
namespace {
constexpr size_t ONE_PIN_BYTES = 1u << 20, ONE_IMG_MAX = 64u << 10, ONE_REC_OFF = ONE_IMG_MAX, ONE_OPS_OFF = ONE_IMG_MAX + 256;
}

static int align_pair_impl(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m,
                           uint32_t *rec_out, uint64_t *ops_out, uint64_t ops_cap, uint64_t *n_ops) {
    if (!ctx || !p || !rec_out || !n_ops || (!ops_out && ops_cap)) return WFAHIP_ERR_BAD_ARG;
    int rc = check_params(p);
    if (rc != WFAHIP_OK) return rc;
    std::memset(rec_out, 0, REC_WORDS * 4);
    *n_ops = 0;
    if (n == 0 || m == 0) {
        rec_out[REC_STATUS] = ST_EMPTY;  // wfa.go:204-206
        return WFAHIP_OK;
    }
    if (n > WFAHIP_MAX_SEQ_LEN || m > WFAHIP_MAX_SEQ_LEN) {
        rec_out[REC_STATUS] = ST_TOO_LONG;  // wfa.go:207-209
        return WFAHIP_OK;
    }
    if (!q || !t) return WFAHIP_ERR_BAD_ARG;
    const uint32_t max_len = std::max(n, m), min_len = std::min(n, m);
    const uint32_t g = gcd_u32(gcd_u32(p->mismatch, p->gap_open + p->gap_ext), p->gap_ext ? p->gap_ext : p->mismatch);
    const uint32_t seq_words = (max_len + 15) / 16 + 1;
    const uint64_t q_cap = ((uint64_t)n + 15) & ~15ull, img = q_cap + (((uint64_t)m + 15) & ~15ull) + 64;
    // worst-case score of a global alignment under the reference's rules (first cell consumed as (mis)match):
    // every shared base a mismatch, the overhang one gap, plus one more gap for the first-cell quirk
    const uint64_t worst = (uint64_t)p->mismatch * min_len + 2ull * (p->gap_open + p->gap_ext) + (uint64_t)p->gap_ext * (max_len - min_len + 2);
    const uint32_t min_xe = std::min(p->mismatch, p->gap_ext ? p->gap_ext : p->mismatch);
    const uint64_t ops_bound = 2 * (worst / std::max(1u, min_xe)) + 64;
    // (the lone-pair instance exists for every penalty shape of wfa_fwd.hpp; round 3's paths, pair_fast = 2 / 3, for the default one)
    const int  shape = p->gap_ext != 0 ? fwd_shape(p->mismatch / g, (p->gap_open + p->gap_ext) / g, p->gap_ext / g) : -1;
    const bool fast = ctx->opt_pair_fast != 0 && ctx->opt_packed != 0 && ctx->opt_blk == 16 && ctx->force_mode < 0 && p->global_alignment &&
                      shape >= 0 && (shape == 0 || ctx->opt_pair_fast == 1) &&
                      (size_t)seq_words * 2 * 4 * (ctx->opt_pair_fast == 1 ? 1 : 4) + 16 <= (ctx->opt_pair_fast == 1 ? 64 : 20) * 1024 && img <= ONE_IMG_MAX &&
                      ONE_OPS_OFF + ops_bound * 8 <= ONE_PIN_BYTES;
    if (fast) {
        HIP_TRY(hipSetDevice(ctx->device));
        if (ctx->bt_pending) {
            (void)hipStreamSynchronize(ctx->stream2);
            ctx->bt_pending = false;
        }
        if (!ctx->one_pin) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->one_pin), ONE_PIN_BYTES, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->one_dev), ctx->one_pin, 0));
        }
        hipStream_t st = ctx->stream;
        // arena: rows for the worst-case score (64 words per score index), capped: what does not fit retries the usual way
        const uint64_t words = std::min<uint64_t>(((worst / g + 16) * 64 + 511) & ~511ull, std::max<uint64_t>(2048, (32ull * max_len + 511) & ~511ull));
        if ((rc = ensure(ctx, ctx->one_ctl, 1024))) return rc;
        if ((rc = ensure(ctx, ctx->redo, 64))) return rc;
        if ((rc = ensure(ctx, ctx->meta, 64))) return rc;
        if (ctx->arena.bytes < words * 4 && (rc = ensure(ctx, ctx->arena, words * 4))) return rc;
        char *const h = ctx->one_pin;
        std::memcpy(h, q, n);
        std::memcpy(h + q_cap, t, m);
        uint64_t *const offs = reinterpret_cast<uint64_t *>(h + img - 64);  // {q_off, t_off} {q_len, t_len}
        offs[0] = 0, offs[1] = q_cap;
        uint32_t *const lens = reinterpret_cast<uint32_t *>(offs + 2);
        lens[0] = n, lens[1] = m;
        uint32_t *const hrec = reinterpret_cast<uint32_t *>(h + ONE_REC_OFF);
        hrec[REC_STATUS] = ST_PENDING;
        uint32_t *const d_ctrl = static_cast<uint32_t *>(ctx->one_ctl.p);
        KParams P{};
        char *const d = ctx->one_dev;
        P.blob = reinterpret_cast<const uint8_t *>(d), P.blob_bytes = img - 64;
        P.q_off = reinterpret_cast<const uint64_t *>(d + img - 64), P.t_off = P.q_off + 1;
        P.q_len = reinterpret_cast<const uint32_t *>(d + img - 48), P.t_len = P.q_len + 1;
        P.queue_head = d_ctrl + 0, P.redo_count = d_ctrl + 1, P.ops_cursor = reinterpret_cast<unsigned long long *>(d_ctrl + 2);
        P.redo_list = static_cast<uint32_t *>(ctx->redo.p);
        P.x = p->mismatch, P.o = p->gap_open, P.e = p->gap_ext, P.oe = p->gap_open + p->gap_ext, P.g = g;
        P.global_alignment = 1, P.adaptive = p->adaptive ? 1 : 0, P.min_wf_len = p->min_wf_len, P.max_dist_diff = p->max_dist_diff;
        P.rec = reinterpret_cast<uint32_t *>(d + ONE_REC_OFF);
        P.ops = reinterpret_cast<uint64_t *>(d + ONE_OPS_OFF), P.ops_cap = (ONE_PIN_BYTES - ONE_OPS_OFF) / 8;
        P.arena = static_cast<uint32_t *>(ctx->arena.p), P.arena_words = words, P.compact_fmt = WFA_BLK_TILED ? 3u : 1u;
        P.pair_meta = static_cast<uint4 *>(ctx->meta.p);
        P.dx = P.x / g, P.doe = P.oe / g, P.de = 1, P.dm = std::max(P.dx, P.doe) + 1, P.di = 2, P.min_xe = min_xe;
        P.lds_seq_words = seq_words, P.chunk_first = 0, P.chunk_n = 1, P.n_work = 1;
        // Round 4, the lone-pair instance with the arena rows in LDS first (pair_lds): rows behind the sequences, as many as the
        // worst-case score needs or 160 KB hold; a pair that runs out of them (ST_REDO_ARENA in its record) is run again by the
        // global-memory instance, and the next calls start there (one_lds_skip).
        const uint32_t lds_off   = ((uint32_t)seq_words * 2u + 4u + 31u) & ~31u;  // words; tiles want their 128-byte lines
        const uint64_t lds_rows  = std::min<uint64_t>((worst / g + 16 + 7) & ~7ull, ((160u * 1024u - lds_off * 4u) / 256u) & ~7u);
        bool           use_lds   = ctx->opt_pair_fast == 1 && ctx->opt_pair_lds != 0 && ctx->one_lds_skip == 0 && lds_rows >= 64;
        if (ctx->one_lds_skip) ctx->one_lds_skip--;
        uint32_t launches = 0;
        for (;;) {
            hrec[REC_STATUS] = ST_PENDING;
            // (the lone-pair instance leaves its control words zeroed: only the first call, or one after another path, clears them)
            if (!(ctx->opt_pair_fast == 1 && ctx->one_ctl_clean)) HIP_TRY(hipMemsetAsync(d_ctrl, 0, 1024, st));
            ctx->one_ctl_clean = false;
            if (ctx->opt_pair_fast == 2) {  // forward kernel, then the backtrace kernel (kept for comparison: 305 us per 1 kbp pair)
                HIP_TRY(wfa_launch_fwd(0, 3, 0u, P, 1u, (size_t)seq_words * 2 * 4 * 4 + 16, st));
                hipLaunchKernelGGL(wfa_backtrace_kernel, dim3(1), dim3(256), 0, st, P);
                launches += 2;
            } else if (ctx->opt_pair_fast == 1) {
                // ONE launch of the lone-pair instance: the whole wave on the pair, a lane per diagonal (a quarter of the
                // instructions of a step of the four-pairs-per-wave kernel -- a lone wave's step is the latency of its own instruction
                // stream), and the same wave walks the backtrace when the forward pass is done -- from the rows in LDS, or from an
                // LDS region of the global arena.
                P.fuse_bt = 1;
                if (use_lds) {
                    KParams PL = P;
                    PL.arena_words = lds_rows * 64, PL.lds_arena_off = lds_off, PL.one_n = n, PL.one_m = m;
                    HIP_TRY(wfa_launch_pair(shape, true, PL, (size_t)lds_off * 4 + (size_t)lds_rows * 256, st));
                } else {
                    HIP_TRY(wfa_launch_pair(shape, false, P, std::max<size_t>((size_t)seq_words * 2 * 4 + 16, (size_t)CompactViewWave::WORDS * 4 + 16), st));
                }
                launches++;
            } else {
                // ONE launch: the streaming instance of the forward kernel -- the wave pushes its finished pair to the done
                // queue and, once the pair queue is empty, walks it itself (stream_backtrace at the end of the kernel): one
                // launch and its gap less than forward kernel + backtrace kernel (298 against 304 us for a 1 kbp pair).  The walk
                // itself is 66-77 us either way: one lane, ~250 instructions per CIGAR op -- not its reads (walking a copy of the
                // rows in LDS took as long).
                P.done_ctl = d_ctrl + 64, P.done_q = reinterpret_cast<uint4 *>(d_ctrl + 128), P.n_stream_wgs = 0;
                P.stream_wait = 2000000;  // 20 ms of the 100 MHz clock
                HIP_TRY(wfa_launch_fwd(0, 3, (uint32_t)FWD_STREAM, P, 1u, (size_t)seq_words * 2 * 4 * 4 + 16, st));
                launches++;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            ctx->one_ctl_clean = ctx->opt_pair_fast == 1;
            if (use_lds && hrec[REC_STATUS] == ST_REDO_ARENA) {  // more rows than LDS holds: the global-memory instance
                use_lds = false, ctx->one_lds_skip = 64;
                continue;
            }
            break;
        }
        ctx->timing = wfahip_timing{};
        ctx->timing.n_launches = launches, ctx->timing.main_kernel_kind = ctx->opt_pair_fast == 1 ? 16 : 3;  // (the batch entry never makes fewer than two launches)
        if (hrec[REC_STATUS] == ST_OK) {
            const uint64_t off = (uint64_t)hrec[REC_OPS_OFF_LO] | ((uint64_t)hrec[REC_OPS_OFF_HI] << 32);
            const uint32_t len = hrec[REC_OPS_LEN];
            if (off + len <= P.ops_cap) {
                *n_ops = len;
                if (len > ops_cap) return WFAHIP_ERR_OOM;  // (*n_ops says how many the caller's buffer must hold)
                std::memcpy(rec_out, hrec, REC_WORDS * 4);
                rec_out[REC_OPS_OFF_LO] = rec_out[REC_OPS_OFF_HI] = 0;
                std::memcpy(ops_out, reinterpret_cast<const uint64_t *>(h + ONE_OPS_OFF) + off, (size_t)len * 8);
                return WFAHIP_OK;
            }
        }
        // handed on (band / bytes / arena rows): the batch entry finishes it
    }
    const uint64_t qo = 0, to = ((uint64_t)n + 15) & ~15ull;
    std::vector<uint8_t> blob(to + m);
    std::memcpy(blob.data(), q, n);
    std::memcpy(blob.data() + to, t, m);
    wfahip_results res;
    rc = wfahip_align_batch(ctx, p, blob.data(), blob.size(), &qo, &n, &to, &m, 1, &res);
    if (rc != WFAHIP_OK) return rc;
    rec_out[REC_STATUS] = (uint32_t)res.status[0];
    if (res.status[0] == WFAHIP_PAIR_OK) {
        rec_out[REC_SCORE] = res.score[0], rec_out[REC_TBEGIN] = (uint32_t)res.tbegin[0], rec_out[REC_TEND] = (uint32_t)res.tend[0];
        rec_out[REC_QBEGIN] = (uint32_t)res.qbegin[0], rec_out[REC_QEND] = (uint32_t)res.qend[0];
        rec_out[REC_ALIGN_LEN] = res.align_len[0], rec_out[REC_MATCHES] = res.matches[0], rec_out[REC_GAPS] = res.gaps[0];
        rec_out[REC_GAP_REGIONS] = res.gap_regions[0], rec_out[REC_OPS_LEN] = res.ops_len[0];
        *n_ops = res.ops_len[0];
        if (res.ops_len[0] > ops_cap) rc = WFAHIP_ERR_OOM;
        else std::memcpy(ops_out, res.ops + res.ops_off[0], (size_t)res.ops_len[0] * 8);
    }
    wfahip_results_free(&res);
    return rc;
}

extern "C" int wfahip_align_pair(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m,
                                 uint32_t *rec, uint64_t *ops, uint64_t ops_cap, uint64_t *n_ops) {
    WFAHIP_GUARD(align_pair_impl(ctx, p, q, n, t, m, rec, ops, ops_cap, n_ops))
}

// ---- one pair at a time behind the batch: submit copies the pair, collect aligns everything submitted so far
static int submit_impl(wfahip_ctx *ctx, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m, uint64_t *ticket) {
    if (!ctx || (!q && n) || (!t && m)) return WFAHIP_ERR_BAD_ARG;
    const uint64_t pos  = ctx->sub_blob.size();
    const uint64_t qcap = ((uint64_t)std::min<uint32_t>(n, WFAHIP_MAX_SEQ_LEN + 1u) + 15) & ~15ull;
    const uint64_t tcap = ((uint64_t)std::min<uint32_t>(m, WFAHIP_MAX_SEQ_LEN + 1u) + 15) & ~15ull;
    const bool     keep = n <= WFAHIP_MAX_SEQ_LEN && m <= WFAHIP_MAX_SEQ_LEN;  // (too long: rejected per pair, nothing to copy)
    ctx->sub_blob.resize(pos + (keep ? qcap + tcap : 0));
    if (keep && n) std::memcpy(ctx->sub_blob.data() + pos, q, n);
    if (keep && m) std::memcpy(ctx->sub_blob.data() + pos + qcap, t, m);
    if (ticket) *ticket = ctx->sub_qlen.size();
    ctx->sub_qoff.push_back(pos), ctx->sub_toff.push_back(pos + qcap);
    ctx->sub_qlen.push_back(n), ctx->sub_tlen.push_back(m);
    return WFAHIP_OK;
}

extern "C" int wfahip_submit(wfahip_ctx *ctx, const uint8_t *q, uint32_t n, const uint8_t *t, uint32_t m, uint64_t *ticket) {
    WFAHIP_GUARD(submit_impl(ctx, q, n, t, m, ticket))
}

extern "C" uint64_t wfahip_pending(const wfahip_ctx *ctx) { return ctx ? ctx->sub_qlen.size() : 0; }

static int collect_impl(wfahip_ctx *ctx, const wfahip_params *p, wfahip_results *out) {
    if (!ctx || !out) return WFAHIP_ERR_BAD_ARG;
    const uint64_t n  = ctx->sub_qlen.size();
    const int      rc = align_batch_impl(ctx, p, ctx->sub_blob.data(), ctx->sub_blob.size(), ctx->sub_qoff.data(), ctx->sub_qlen.data(),
                                         ctx->sub_toff.data(), ctx->sub_tlen.data(), n, out);
    if (rc == WFAHIP_OK) {  // (on failure the submissions stay: the caller may collect again, e.g. with other parameters)
        ctx->sub_blob.clear(), ctx->sub_qoff.clear(), ctx->sub_toff.clear(), ctx->sub_qlen.clear(), ctx->sub_tlen.clear();
    }
    return rc;
}

extern "C" int wfahip_collect(wfahip_ctx *ctx, const wfahip_params *p, wfahip_results *out) { WFAHIP_GUARD(collect_impl(ctx, p, out)) }

extern "C" int wfahip_debug_compact_arena(wfahip_ctx *ctx, uint64_t pair, uint32_t **words, uint64_t *n_words, uint32_t *fmt,
                                          uint32_t *meta4) {
    if (!ctx || !words || !n_words) return WFAHIP_ERR_BAD_ARG;
    *words = nullptr, *n_words = 0;
    if (!ctx->dbg_arena || pair < ctx->dbg_first || pair >= ctx->dbg_first + ctx->dbg_n) return WFAHIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    uint32_t *w = static_cast<uint32_t *>(std::malloc((size_t)ctx->dbg_words * 4));
    if (!w) return WFAHIP_ERR_OOM;
    const uint64_t slot = pair - ctx->dbg_first;
    if (hipMemcpy(w, ctx->dbg_arena + slot * ctx->dbg_words, (size_t)ctx->dbg_words * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (meta4 && hipMemcpy(meta4, ctx->dbg_meta + slot, 16, hipMemcpyDeviceToHost) != hipSuccess)) {
        std::free(w);
        return WFAHIP_ERR_HIP;
    }
    *words = w, *n_words = ctx->dbg_words;
    if (fmt) *fmt = ctx->dbg_fmt;
    return WFAHIP_OK;
}

// ---- the synthetic dataset generated where it is used (wfa_gen_dev.hpp): no host generation, no upload
extern "C" int wfahip_generate_pairs_device(wfahip_ctx *ctx, uint64_t seed, uint64_t first_index, uint64_t n_pairs, uint32_t length,
                                            double error_rate, void *d_blob, void *d_q_off, void *d_q_len, void *d_t_off, void *d_t_len,
                                            void *stream) {
    if (!ctx || !d_blob || !d_q_off || !d_q_len || !d_t_off || !d_t_len || length == 0 || error_rate < 0.0) return WFAHIP_ERR_BAD_ARG;
    if (n_pairs == 0) return WFAHIP_OK;
    if (n_pairs > 0x7FFFFFFFull) return WFAHIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t stride = wfahip_gen_stride(length, error_rate);
    const uint32_t edits  = (uint32_t)std::llround((double)length * error_rate);
    const size_t   lds    = (size_t)length + edits + 16;
    if (lds > LDS_MAX_BYTES) return WFAHIP_ERR_UNSUPPORTED;  // (the text of a pair is edited in LDS)
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wfa_gen_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    hipLaunchKernelGGL(wfa_gen_kernel, dim3((uint32_t)n_pairs), dim3(GEN_THREADS), lds, st, seed, first_index, n_pairs, length, edits, stride,
                       static_cast<uint8_t *>(d_blob), static_cast<uint64_t *>(d_q_off), static_cast<uint32_t *>(d_q_len),
                       static_cast<uint64_t *>(d_t_off), static_cast<uint32_t *>(d_t_len));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return WFAHIP_OK;
}

// ---------------------------------------------------------------------------------------------- clock probe (bench)
// Every wave runs a chain of dependent integer max / add instructions (the forward kernels' mix) and reads both clocks
// around it: s_memtime counts shader cycles, s_memrealtime the constant 100 MHz reference.
__global__ __launch_bounds__(256) void wfa_clock_probe_kernel(unsigned long long *out, uint32_t iters) {
    unsigned long long t0, t1, r0, r1;
    uint32_t           a = threadIdx.x, b = blockIdx.x | 1u;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) a = (a > b ? a : b) + (uint32_t)u, b = (b > a ? b : a) ^ a;
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(a), "v"(b) : "memory");
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0, out[2 * w + 1] = r1 - r0;
    }
}

static int debug_clock_impl(wfahip_ctx *ctx, double *mhz, double *mhz_min, double *mhz_max) {
    if (!ctx || !mhz) return WFAHIP_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t wgs = (uint32_t)ctx->num_cus * 4u, waves = wgs * 4u;  // four waves per SIMD
    DevBuf buf;
    int rc = ensure(ctx, buf, (size_t)waves * 16);
    if (rc) return rc;
    std::vector<unsigned long long> h((size_t)waves * 2);
    for (int pass = 0; pass < 2; pass++) {  // (the first pass brings the clock up; the second one is read)
        hipLaunchKernelGGL(wfa_clock_probe_kernel, dim3(wgs), dim3(256), 0, ctx->stream, static_cast<unsigned long long *>(buf.p), 6000u);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            release(buf);
            return WFAHIP_ERR_HIP;
        }
    }
    const hipError_t e = hipMemcpy(h.data(), buf.p, h.size() * 8, hipMemcpyDeviceToHost);
    release(buf);
    if (e != hipSuccess) return WFAHIP_ERR_HIP;
    double sum = 0, lo = 1e30, hi = 0;
    uint32_t cnt = 0;
    for (uint32_t w = 0; w < waves; w++) {
        if (h[2 * w + 1] == 0) continue;
        const double f = (double)h[2 * w] / (double)h[2 * w + 1] * 100.0;  // cycles per tick of the 100 MHz clock -> MHz
        sum += f, lo = std::min(lo, f), hi = std::max(hi, f), cnt++;
    }
    if (cnt == 0) return WFAHIP_ERR_INTERNAL;
    *mhz = sum / cnt;
    if (mhz_min) *mhz_min = lo;
    if (mhz_max) *mhz_max = hi;
    return WFAHIP_OK;
}

extern "C" int wfahip_debug_clock(wfahip_ctx *ctx, double *mhz, double *mhz_min, double *mhz_max) { WFAHIP_GUARD(debug_clock_impl(ctx, mhz, mhz_min, mhz_max)) }

// compact: wfahip_debug_team_compact -- the pair runs on wfa_teamc_kernel (option team_wgs must name the team's size) and a row is
// its ONE backtrace word per diagonal (blk_word(), wfa_device.hpp) instead of the M, I and D words
static int debug_wavefronts_impl(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n,
                                 const uint8_t *t, uint32_t m, wfahip_row **rows, uint64_t *n_rows,
                                 uint32_t **words, uint64_t *n_words, wfahip_results *res, bool compact = false) {
    if (!ctx || !rows || !n_rows || !words || !n_words || !q || !t || n == 0 || m == 0) return WFAHIP_ERR_BAD_ARG;
    *rows = nullptr, *words = nullptr, *n_rows = 0, *n_words = 0;
    if (res) results_zero(res);
    int rc = check_params(p);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    bool acgt = true;
    for (uint32_t i = 0; i < n && acgt; i++) acgt = q[i] == 'A' || q[i] == 'C' || q[i] == 'G' || q[i] == 'T';
    for (uint32_t i = 0; i < m && acgt; i++) acgt = t[i] == 'A' || t[i] == 'C' || t[i] == 'G' || t[i] == 'T';

    std::vector<uint8_t> blob((size_t)n + m);
    std::memcpy(blob.data(), q, n);
    std::memcpy(blob.data() + n, t, m);
    uint64_t qo = 0, to = n;
    if ((rc = ensure(ctx, ctx->in_blob, blob.size() + 16))) return rc;
    if ((rc = ensure(ctx, ctx->in_qoff, 8))) return rc;
    if ((rc = ensure(ctx, ctx->in_toff, 8))) return rc;
    if ((rc = ensure(ctx, ctx->in_qlen, 4))) return rc;
    if ((rc = ensure(ctx, ctx->in_tlen, 4))) return rc;
    if ((rc = ensure(ctx, ctx->out_rec, REC_WORDS * 4))) return rc;
    uint64_t ops_cap = (uint64_t)n + m + 16;
    if ((rc = ensure(ctx, ctx->out_ops, ops_cap * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->in_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->in_qoff.p, &qo, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->in_toff.p, &to, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->in_qlen.p, &n, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->in_tlen.p, &m, 4, hipMemcpyHostToDevice, st));

    // grow the single slot until the pair fits; the byte path is chosen up front for non-ACGT input
    const int64_t saved = ctx->opt_arena_bytes_per_slot;
    int64_t       bytes = saved > 0 ? saved : (int64_t)std::max<uint64_t>(256 * 1024, 384ull * std::max(n, m));
    uint32_t      recw[REC_WORDS];
    uint32_t      hctrl[CTRL_WORDS];
    for (int attempt = 0;; attempt++) {
        {
            // (the three fields steer the one debug launch; they are restored on every way out of this scope)
            struct Restore {
                wfahip_ctx *c;
                int64_t     saved;
                ~Restore() { c->dbg_teamc = false, c->force_mode = -1, c->opt_arena_bytes_per_slot = saved; }
            } restore{ctx, saved};
            ctx->opt_arena_bytes_per_slot = bytes;
            // debug_single stops after one launch, so the byte path is chosen up front for non-ACGT input
            ctx->force_mode = acgt ? 0 : 1;
            ctx->dbg_teamc  = compact;
            rc = align_device(ctx, p, ctx->in_blob.p, blob.size(), ctx->in_qoff.p, ctx->in_qlen.p, ctx->in_toff.p,
                              ctx->in_tlen.p, 1, std::max(n, m), ctx->out_rec.p, ctx->out_ops.p, ops_cap, nullptr, st,
                              true);
        }
        if (rc) return rc;
        HIP_TRY(hipMemcpy(recw, ctx->out_rec.p, sizeof recw, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hctrl, ctx->ctrl.p, sizeof hctrl, hipMemcpyDeviceToHost));
        if (recw[REC_STATUS] == ST_REDO_ARENA && attempt < 8) {
            bytes *= 8;
            continue;
        }
        break;
    }
    if (recw[REC_STATUS] != ST_OK) return WFAHIP_ERR_INTERNAL;

    const uint64_t cap   = (uint64_t)(((bytes / 4) + 7) & ~7ll);
    const uint32_t n_ent = hctrl[4];
    std::vector<uint32_t> dir((size_t)n_ent * DIR_WORDS);
    HIP_TRY(hipMemcpy(dir.data(), static_cast<uint32_t *>(ctx->arena.p) + cap - (uint64_t)DIR_WORDS * n_ent,
                      dir.size() * 4, hipMemcpyDeviceToHost));
    // entry i (score i*g) sits DIR_WORDS*(i+1) words below the slot end: {base_lo, base_hi, lo, w, stride, ...}
    auto entry = [&](uint32_t i) { return &dir[(size_t)(n_ent - 1 - i) * DIR_WORDS]; };
    uint64_t total = 0, nr = 0;
    const uint64_t ncomp = compact ? 1ull : 3ull;
    for (uint32_t i = 0; i < n_ent; i++)
        if ((int32_t)entry(i)[3] > 0) total += ncomp * entry(i)[3], nr++;
    *rows  = static_cast<wfahip_row *>(std::malloc(std::max<uint64_t>(nr, 1) * sizeof(wfahip_row)));
    *words = static_cast<uint32_t *>(std::malloc(std::max<uint64_t>(total, 1) * 4));
    if (!*rows || !*words) return WFAHIP_ERR_OOM;
    const uint32_t g = gcd_u32(gcd_u32(p->mismatch, p->gap_open + p->gap_ext), p->gap_ext);
    uint64_t       pos = 0, ri = 0;
    for (uint32_t i = 0; i < n_ent; i++) {
        const uint32_t *e = entry(i);
        const uint32_t  w = e[3], stride = e[4];
        if ((int32_t)w <= 0) continue;
        const uint64_t base = (uint64_t)e[0] | ((uint64_t)e[1] << 32);
        for (int c = 0; c < (int)ncomp; c++)  // M, I, D rows are `stride` words apart
            HIP_TRY(hipMemcpy(*words + pos + (uint64_t)c * w,
                              static_cast<uint32_t *>(ctx->arena.p) + base + (uint64_t)c * stride, 4ull * w,
                              hipMemcpyDeviceToHost));
        (*rows)[ri++] = wfahip_row{i * g, (int32_t)e[2], w, pos};
        pos += ncomp * w;
    }
    *n_rows = nr, *n_words = total;
    if (res) {
        std::vector<uint32_t> rec(recw, recw + REC_WORDS);
        std::vector<uint64_t> ops(std::max<uint64_t>(ctx->timing.ops_written, 1));
        if (ctx->timing.ops_written)
            HIP_TRY(hipMemcpy(ops.data(), ctx->out_ops.p, ctx->timing.ops_written * 8, hipMemcpyDeviceToHost));
        rc = unpack_results(rec, ops, 1, res, nullptr);
    }
    return rc;
}

extern "C" int wfahip_debug_wavefronts(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n,
                                       const uint8_t *t, uint32_t m, wfahip_row **rows, uint64_t *n_rows,
                                       uint32_t **words, uint64_t *n_words, wfahip_results *res) {
    WFAHIP_GUARD(debug_wavefronts_impl(ctx, p, q, n, t, m, rows, n_rows, words, n_words, res))
}

extern "C" int wfahip_debug_team_compact(wfahip_ctx *ctx, const wfahip_params *p, const uint8_t *q, uint32_t n,
                                         const uint8_t *t, uint32_t m, wfahip_row **rows, uint64_t *n_rows,
                                         uint32_t **words, uint64_t *n_words, wfahip_results *res) {
    WFAHIP_GUARD(debug_wavefronts_impl(ctx, p, q, n, t, m, rows, n_rows, words, n_words, res, true))
}
