// wfa_debug.hip -- parity and measurement aids of the C-ABI (include/wfa_hip.h): the compact arena of a pair, every stored
// wavefront row of one pair, the device-side dataset generator, the clock probe.
#define WFA_NO_AUX_KERNELS 1  // (device functions and constants of the kernels' headers only: the kernels are launched by wfa_host.hip)
#include "wfa_ctx.hpp"
#include "wfa_generic.hpp"
#include "wfa_packed.hpp"
#include "wfa_blk.hpp"
#include "wfa_duo_cfg.hpp"
#include "wfa_lane.hpp"
#include "wfa_fwd.hpp"
#include "wfa_long.hpp"
#include "wfa_team.hpp"
#include "wfa_teamc.hpp"
#include "wfa_gen_dev.hpp"

using namespace wfa;

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

