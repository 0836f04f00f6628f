// wfa_host.hip -- the C-ABI of libwfahip.so (include/wfa_hip.h): context, workspaces, launch
// configuration, retry ladder (bigger arena / byte-compare path) and result unpacking.
//
// There is NO CPU fallback here: every alignment is produced by the HIP kernels.  Without a GPU the
// entry points return WFAHIP_ERR_NO_DEVICE.
#include <map>
#include "wfa_ctx.hpp"
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
#include "wfa_wide.hpp"

using namespace wfa;
static_assert(TEAM_SOLO_MAX == 4096, "wfa_ctx.hpp: default of opt_team_solo_max");

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
// wfa_wide_kernel of penalty shape `shape` (wfa_fwd.hpp: fwd_shape()); phase 0 with `waves` waves per pair (1 or 4), phase 1 with one
hipError_t wfa_launch_wide(int shape, int phase, int waves, const KParams &P, uint32_t grid, size_t lds_bytes, hipStream_t st) {
    const auto go = [&](auto kern, int nw) -> hipError_t {
        if (lds_bytes > 64 * 1024) {  // (per device: the attribute belongs to the kernel as loaded there)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds_bytes, st, P);
        return hipGetLastError();
    };
#define WFA_WIDE_SHAPE(I, DX_, DOE_)                                                                            \
    case 3 * I: return go(wfa_wide_kernel<DX_, DOE_, 0, 1>, 1);                                                 \
    case 3 * I + 1: return go(wfa_wide_kernel<DX_, DOE_, 0, 4>, 4);                                             \
    case 3 * I + 2: return go(wfa_wide_kernel<DX_, DOE_, 1, 1>, 1);
    switch (shape * 3 + (phase ? 2 : (waves > 1 ? 1 : 0))) {
        WFA_WIDE_SHAPE(0, 2, 4)
        WFA_WIDE_SHAPE(1, 1, 3)
        WFA_WIDE_SHAPE(2, 1, 2)
        WFA_WIDE_SHAPE(3, 2, 3)
        WFA_WIDE_SHAPE(4, 2, 2)
        WFA_WIDE_SHAPE(5, 3, 3)
    }
#undef WFA_WIDE_SHAPE
    return hipErrorInvalidValue;
}
}  // namespace wfa

namespace {

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
                      &ctx->in_toff, &ctx->in_tlen, &ctx->out_rec, &ctx->out_ops, &ctx->in_packed, &ctx->in_small, &ctx->prepack, &ctx->one_ctl, &ctx->page_ctl, &ctx->xbuf, &ctx->wide_ckpt})
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
    else if (k == "wide")
        ctx->opt_wide = value;
    else if (k == "wide_min_pairs")
        ctx->opt_wide_min_pairs = value;
    else if (k == "wide_max_len")
        ctx->opt_wide_max_len = value;
    else if (k == "wide_exact")
        ctx->opt_wide_exact = value != 0;
    else if (k == "wide_waves")
        ctx->opt_wide_waves = (value == 1 || value == 4) ? (int)value : 0;
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

    bool packed_done = false;
    // ---- semi-global short reads (round 6): wfa_wide_kernel -- a wave per pair, the last rows as 16-bit offsets in LDS rings of
    //      any width, one 16-bit backtrace word per diagonal in the arena -- + the lane-per-pair backtrace kernel, chunk by chunk.
    //      What it cannot hold (bytes outside ACGT, an arena that overflows) goes to the ladder below, like the leftovers of pass 1.
    if (ctx->opt_packed && ctx->opt_wide != 0 && !debug_single && ctx->force_mode < 0 && !P.global_alignment && P.e != 0u && max_len <= WIDE_MAX_LEN &&
        (int64_t)max_len <= ctx->opt_wide_max_len && (int64_t)n_pairs >= ctx->opt_wide_min_pairs && !ctx->pk_words) {
        const uint32_t dx = P.x / P.g, doe = P.oe / P.g, de = P.e / P.g;
        const int      shape = fwd_shape(dx, doe, de);
        if (shape >= 0) {
            const uint32_t seq_words = (max_len + 15) / 16 + 1;
            const uint32_t row_hw    = wide_row_hw(max_len);
            const size_t   lds_bytes = (size_t)wide_lds_words(seq_words, max_len) * 4;
            // rows: with wf-adaptive the n + m - 1 seeds live for a dozen score steps, then a band of a few dozen diagonals; without it
            // every row keeps them.  (A pair that needs more is re-run by the ladder: ST_REDO_ARENA.)
            uint64_t words = P.adaptive ? 36ull * max_len + 4096 : 2ull * max_len * (uint64_t)(max_len / 3 + 64) / 2 + 8192;
            if (ctx->opt_packed_arena_bytes > 0) words = std::max<uint64_t>(1024, ctx->opt_packed_arena_bytes / 4);
            words = (words + 511) & ~511ull;
            const uint64_t budget = (uint64_t)((double)ctx->total_mem * 0.35);
            const uint64_t chunk  = std::max<uint64_t>(1, std::min<uint64_t>(n_pairs, std::min<uint64_t>(budget / (words * 4ull),
                                                                             ctx->opt_chunk_pairs > 0 ? (uint64_t)ctx->opt_chunk_pairs : ~0ull)));
            if ((rc = ensure(ctx, ctx->arena, (size_t)(words * 4ull * chunk)))) return rc;
            if ((rc = ensure(ctx, ctx->meta, chunk * 16))) return rc;
            // two launches per chunk when wf-adaptive narrows the rows: the first runs the wide rows, the second -- rings of 256
            // diagonals, the CU's full complement of waves -- the rest (wfa_wide.hpp)
            const bool two_phase = P.adaptive != 0 && ctx->opt_wide >= 1 && ctx->opt_wide != 3;
            if (two_phase && (rc = ensure(ctx, ctx->wide_ckpt, (size_t)chunk * WIDE_CKPT_WORDS * 4))) return rc;
            P.wide_ckpt = static_cast<uint32_t *>(ctx->wide_ckpt.p), P.wide_ckpt_on = two_phase ? 1u : 0u, P.wide_exact = ctx->opt_wide_exact ? 1u : 0u;
            const size_t lds_narrow = (size_t)wide_lds_words_narrow(seq_words) * 4;
            // waves per pair in the first phase: rings above 12 KB leave a SIMD fewer than three one-wave workgroups -- four waves share them
            const int wide_waves = ctx->opt_wide_waves > 0 ? ctx->opt_wide_waves : (lds_bytes > 12 * 1024 ? 4 : 1);
            ctx->timing.arena_bytes = std::max<uint64_t>(ctx->timing.arena_bytes, words * 4ull * chunk);
            P.dx = dx, P.doe = doe, P.de = de, P.dm = std::max(dx, doe) + 1, P.di = de + 1;
            P.lds_seq_words = seq_words, P.sub_lds_words = row_hw, P.min_xe = std::min(P.x, P.e);
            P.arena_words = words, P.compact_fmt = WIDE_FMT;
            P.prepack = nullptr, P.prepack_words = 0, P.done_q = nullptr, P.done_ctl = nullptr, P.n_stream_wgs = 0, P.work = nullptr;
            const uint64_t n_chunks = (n_pairs + chunk - 1) / chunk;
            while (ctx->evpool.size() < 4 * n_chunks) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                ctx->evpool.push_back(e);
            }
            ctrl_zeroed = false;
            for (uint64_t c = 0; c < n_chunks; c++) {
                const uint64_t c0 = c * chunk, cn = std::min<uint64_t>(chunk, n_pairs - c0);
                P.arena = static_cast<uint32_t *>(ctx->arena.p), P.pair_meta = static_cast<uint4 *>(ctx->meta.p);
                P.chunk_first = (uint32_t)c0, P.chunk_n = (uint32_t)cn;
                if (c > 0) HIP_TRY(hipMemsetAsync(d_ctrl, 0, 4, st));  // queue_head
                if (ctx->opt_arena_poison) HIP_TRY(hipMemsetAsync(P.arena, 0xA5, (size_t)(words * 4ull * cn), st));
                const uint32_t grid = (uint32_t)cn;  // (a workgroup -- one wave -- per pair)
                HIP_TRY(hipEventRecord(ctx->evpool[4 * c], st));
                HIP_TRY(wfa_launch_wide(shape, 0, wide_waves, P, grid, lds_bytes, st));
                if (two_phase) HIP_TRY(wfa_launch_wide(shape, 1, 1, P, grid, lds_narrow, st));
                HIP_TRY(hipEventRecord(ctx->evpool[4 * c + 1], st));
                hipLaunchKernelGGL(wfa_backtrace_kernel, dim3((uint32_t)((cn + BT_THREADS - 1) / BT_THREADS)), dim3(BT_THREADS), 0, st, P);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(ctx->evpool[4 * c + 3], st));
            }
            uint32_t              hc[CTRL_WORDS];
            std::vector<uint64_t> redo;
            HIP_TRY(hipEventRecord(ctx->ev1, st));
            if ((rc = fetch_ctrl(hc, &redo))) return rc;
            ctrl_fresh = true, std::memcpy(hc_last, hc, sizeof hc_last);
            for (uint64_t c = 0; c < n_chunks; c++) {
                float msF = 0, msB = 0;
                HIP_TRY(hipEventElapsedTime(&msF, ctx->evpool[4 * c], ctx->evpool[4 * c + 1]));
                HIP_TRY(hipEventElapsedTime(&msB, ctx->evpool[4 * c + 1], ctx->evpool[4 * c + 3]));
                ctx->timing.kernel_ms += msF + msB, ctx->timing.main_kernel_ms += msF, ctx->timing.n_main_launches++, ctx->timing.n_launches += 2;
            }
            if (std::getenv("WFAHIP_WIDE_TRACE")) {  // (debug aid: why the pairs of the last chunk were handed on -- the kernel leaves {status, score index, reason, span})
                const uint64_t cn = n_pairs - (n_chunks - 1) * chunk;
                std::vector<uint32_t> hm(4 * cn);
                HIP_TRY(hipMemcpy(hm.data(), ctx->meta.p, 16 * cn, hipMemcpyDeviceToHost));
                std::map<uint64_t, uint32_t> tally;
                uint32_t shown = 0;
                for (uint64_t i = 0; i < cn; i++)
                    if (hm[4 * i] >= ST_REDO_BYTES) {
                        tally[((uint64_t)hm[4 * i] << 32) | hm[4 * i + 2]]++;
                        if (shown++ < 12) std::fprintf(stderr, "[wfahip] wide: slot %llu status %u si %u reason %u span %u\n", (unsigned long long)i, hm[4 * i], hm[4 * i + 1], hm[4 * i + 2], hm[4 * i + 3]);
                    }
                for (const auto &kv : tally) std::fprintf(stderr, "[wfahip] wide: status %u reason %u: %u pairs of %llu\n", (uint32_t)(kv.first >> 32), (uint32_t)kv.first, kv.second, (unsigned long long)cn);
            }
            ctx->timing.main_kernel_kind = 18;
            ctx->timing.n_packed_pairs   = (uint32_t)(n_pairs - redo.size());
            ctx->timing.n_retried_pairs += (uint32_t)redo.size();
            Job jb, ja;
            jb.mode = 1, jb.level = 0, jb.all = false;
            ja.mode = 0, ja.level = 0, ja.all = false;
            for (uint64_t e : redo) ((uint32_t)(e >> 32) == ST_REDO_BYTES ? jb : ja).pairs.push_back((uint32_t)e);
            std::sort(ja.pairs.begin(), ja.pairs.end());
            for (Job *jp : {&ja, &jb})
                if (!jp->pairs.empty()) jobs.push_back(std::move(*jp));
            first = false, packed_done = true;
        }
    }
    // ---- pass 1: sub-wave forward kernels + lane-per-pair backtrace kernel, chunk by chunk.
    //      kind 2 = register-window kernel (4 pairs per wave), kind 1 = LDS-ring packed kernel (2 pairs per wave).
    if (!packed_done && ctx->opt_packed && !debug_single && ctx->force_mode < 0 && P.global_alignment && P.e != 0u) {
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
        // more than MaxDistDiff keeps a wide band for most of its scores (the first reduce cuts the final diagonal off; KERNELS.md
        // 4d) -- a scheduling hint only, results do not depend on the order.
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
int align_device(wfahip_ctx *ctx, const wfahip_params *p, const void *d_blob, uint64_t blob_bytes,
                        const void *d_q_off, const void *d_q_len, const void *d_t_off, const void *d_t_len,
                        uint64_t n_pairs, uint32_t max_len, void *d_rec, void *d_ops, uint64_t ops_cap,
                        uint64_t *ops_needed, hipStream_t st, bool debug_single, uint64_t ops_cursor0) {
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

hipError_t wfa_launch_backtrace_one(const wfa::KParams &P, hipStream_t st) {
    hipLaunchKernelGGL(wfa_backtrace_kernel, dim3(1), dim3(256), 0, st, P);
    return hipGetLastError();
}

