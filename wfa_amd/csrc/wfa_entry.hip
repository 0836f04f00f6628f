// wfa_entry.hip -- the host entries of the C-ABI (include/wfa_hip.h): wfahip_align_batch -- host blobs in, host result arrays out,
// with the upload, the alignment (align_device, wfa_host.hip) and the download of consecutive slices overlapped --, pre-packed
// input, wfahip_align_pair, submit / collect, and the cache the result arrays circulate through.
#define WFA_NO_AUX_KERNELS 1  // (device functions and constants of the kernels' headers only: the kernels are launched by wfa_host.hip)
#include "wfa_ctx.hpp"
#include "wfa_generic.hpp"
#include "wfa_packed.hpp"
#include "wfa_blk.hpp"
#include "wfa_duo_cfg.hpp"
#include "wfa_lane.hpp"
#include "wfa_fwd.hpp"
#include "wfa_long.hpp"
#include "wfa_finalize.hpp"

using namespace wfa;

void results_zero(wfahip_results *r) { std::memset(r, 0, sizeof *r); }

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

int unpack_results(const std::vector<uint32_t> &rec, const std::vector<uint64_t> &ops, uint64_t n,
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
                HIP_TRY(wfa_launch_backtrace_one(P, st));
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

