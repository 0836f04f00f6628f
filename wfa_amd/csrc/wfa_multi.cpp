// wfa_multi.cpp -- a set of contexts (one per GPU) behind one call: wfahip_create_multi / wfahip_align_batch_multi.
//
// Built on the public C-ABI only (include/wfa_hip.h).  Alignments share no state -- the reference's own model is one
// Aligner per goroutine (wfa.go:73-78) and its CLI walks the pairs one by one (wfa-go/wfa-go.go:166-178) -- so a batch
// is cut into contiguous shards of pairs, balanced by sequence bytes, one per context; every shard is aligned by its
// own host thread on its own GPU (wfahip_align_batch: upload, kernels, assembly, download) and the shard results are
// copied into one wfahip_results in pair order.  There is no exchange between GPUs.
#include "../../include/wfa_hip.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

struct wfahip_multi {
    std::vector<wfahip_ctx *> ctx;
};

extern "C" void wfahip_destroy_multi(wfahip_multi *m) {
    if (!m) return;
    for (wfahip_ctx *c : m->ctx) wfahip_destroy(c);
    delete m;
}

static int create_multi_impl(const int *device_ids, int n_devices, wfahip_multi **out) {
    if (!out) return WFAHIP_ERR_BAD_ARG;
    *out = nullptr;
    const int have = wfahip_device_count();
    if (have <= 0) return WFAHIP_ERR_NO_DEVICE;
    if (n_devices <= 0) {
        if (device_ids) return WFAHIP_ERR_BAD_ARG;
        n_devices = have;
    }
    wfahip_multi *m = new wfahip_multi();
    for (int i = 0; i < n_devices; i++) {
        const int   id = device_ids ? device_ids[i] : i;
        wfahip_ctx *c  = nullptr;
        const int   rc = (id < 0 || id >= have) ? WFAHIP_ERR_BAD_ARG : wfahip_create(id, &c);
        if (rc != WFAHIP_OK) {
            wfahip_destroy_multi(m);
            return rc;
        }
        m->ctx.push_back(c);
    }
    *out = m;
    return WFAHIP_OK;
}

extern "C" int wfahip_create_multi(const int *device_ids, int n_devices, wfahip_multi **out) {
    try {
        return create_multi_impl(device_ids, n_devices, out);
    } catch (const std::bad_alloc &) {
        return WFAHIP_ERR_OOM;
    } catch (...) {
        return WFAHIP_ERR_INTERNAL;
    }
}

extern "C" int wfahip_multi_size(const wfahip_multi *m) { return m ? (int)m->ctx.size() : 0; }

extern "C" wfahip_ctx *wfahip_multi_ctx(wfahip_multi *m, int i) {
    return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr;
}

namespace {

struct Shard {
    uint64_t              first = 0, count = 0;  // pairs [first, first + count)
    uint64_t              lo = 0, hi = 0;        // blob bytes the shard refers to
    std::vector<uint64_t> q_off, t_off;          // rebased to lo
    wfahip_results        res{};
    int                   rc = WFAHIP_OK;
};

inline bool valid_pair(uint32_t n, uint32_t m) { return n && m && n <= WFAHIP_MAX_SEQ_LEN && m <= WFAHIP_MAX_SEQ_LEN; }

template <class F>
void run_threads(size_t n, F &&fn) {  // fn(i) for i in [0, n): one thread each; a thread that cannot start runs here
    std::vector<std::thread> th;
    th.reserve(n);
    for (size_t i = 1; i < n; i++) {
        try {
            th.emplace_back(fn, i);
        } catch (...) {
            fn(i);
        }
    }
    if (n) fn(0);
    for (auto &t : th) t.join();
}

int align_multi_impl(wfahip_multi *m, const wfahip_params *p, const uint8_t *seq_blob, uint64_t blob_bytes,
                     const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off, const uint32_t *t_len,
                     uint64_t n_pairs, wfahip_results *out) {
    if (!m || m->ctx.empty() || !out) return WFAHIP_ERR_BAD_ARG;
    std::memset(out, 0, sizeof *out);
    if (m->ctx.size() == 1 || n_pairs < 2 * m->ctx.size())
        return wfahip_align_batch(m->ctx[0], p, seq_blob, blob_bytes, q_off, q_len, t_off, t_len, n_pairs, out);
    if (!q_off || !q_len || !t_off || !t_len || (!seq_blob && blob_bytes)) return WFAHIP_ERR_BAD_ARG;

    // ---- contiguous shards with about the same number of sequence bytes (+ a constant per pair)
    const size_t          S = m->ctx.size();
    std::vector<uint64_t> pre(n_pairs + 1, 0);
    for (uint64_t i = 0; i < n_pairs; i++)
        pre[i + 1] = pre[i] + 64 + (valid_pair(q_len[i], t_len[i]) ? (uint64_t)q_len[i] + t_len[i] : 0);
    std::vector<Shard> sh(S);
    uint64_t           at = 0;
    for (size_t k = 0; k < S; k++) {
        const uint64_t goal = pre[n_pairs] / S * (k + 1);
        uint64_t       end  = k + 1 == S ? n_pairs : (uint64_t)(std::lower_bound(pre.begin() + (long)at, pre.end(), goal) - pre.begin());
        end                 = std::min<uint64_t>(std::max<uint64_t>(end, at), n_pairs);
        sh[k].first = at, sh[k].count = end - at;
        at = end;
    }
    // Layout assumption: the pairs of a batch lie in the blob in pair order (make_blob, the generator, a cgo binding
    // that appends pair after pair), so a shard's sequences span about its share of the blob and only blob[lo, hi) is
    // uploaded to its GPU.  Batches that share sequences between pairs (one target against many queries) or store
    // them out of order are still aligned correctly, but every GPU may then receive most of the blob.
    for (Shard &s : sh) {
        uint64_t lo = blob_bytes, hi = 0;
        for (uint64_t i = s.first; i < s.first + s.count; i++) {
            if (!valid_pair(q_len[i], t_len[i])) continue;
            if (q_off[i] > blob_bytes || q_len[i] > blob_bytes - q_off[i] || t_off[i] > blob_bytes || t_len[i] > blob_bytes - t_off[i])
                return WFAHIP_ERR_BAD_ARG;  // (no sum that a hostile 64-bit offset could wrap)
            lo = std::min(lo, std::min(q_off[i], t_off[i]));
            hi = std::max(hi, std::max(q_off[i] + q_len[i], t_off[i] + t_len[i]));
        }
        if (hi <= lo) lo = hi = 0;
        lo &= ~15ull;  // keeps every sequence's alignment inside the blob
        s.lo = lo, s.hi = hi;
        s.q_off.resize(s.count), s.t_off.resize(s.count);
        for (uint64_t i = 0; i < s.count; i++) {
            const bool v = valid_pair(q_len[s.first + i], t_len[s.first + i]);
            s.q_off[i]   = v ? q_off[s.first + i] - lo : 0;
            s.t_off[i]   = v ? t_off[s.first + i] - lo : 0;
        }
    }

    // ---- every shard on its own GPU, from its own thread
    run_threads(S, [&](size_t k) {
        Shard &s = sh[k];
        if (s.count == 0) return;
        s.rc = wfahip_align_batch(m->ctx[k], p, seq_blob + s.lo, s.hi - s.lo, s.q_off.data(), q_len + s.first, s.t_off.data(),
                                  t_len + s.first, s.count, &s.res);
    });
    int rc = WFAHIP_OK;
    for (Shard &s : sh)
        if (s.rc != WFAHIP_OK && rc == WFAHIP_OK) rc = s.rc;

    // ---- merge in pair order
    uint64_t n_ops = 0;
    std::vector<uint64_t> ops_base(S, 0);
    for (size_t k = 0; k < S; k++) ops_base[k] = n_ops, n_ops += sh[k].res.n_ops;
    if (rc == WFAHIP_OK) {
        const size_t cnt = (size_t)std::max<uint64_t>(n_pairs, 1);
        out->n = n_pairs, out->n_ops = n_ops;
#define ALLOC(field, type) out->field = static_cast<type *>(std::malloc(cnt * sizeof(type)));
        ALLOC(status, int32_t) ALLOC(score, uint32_t) ALLOC(tbegin, int32_t) ALLOC(tend, int32_t) ALLOC(qbegin, int32_t)
        ALLOC(qend, int32_t) ALLOC(align_len, uint32_t) ALLOC(matches, uint32_t) ALLOC(gaps, uint32_t)
        ALLOC(gap_regions, uint32_t) ALLOC(ops_off, uint64_t) ALLOC(ops_len, uint32_t)
#undef ALLOC
        out->ops = static_cast<uint64_t *>(std::malloc((size_t)std::max<uint64_t>(n_ops, 1) * 8));
        if (!out->status || !out->score || !out->tbegin || !out->tend || !out->qbegin || !out->qend || !out->align_len ||
            !out->matches || !out->gaps || !out->gap_regions || !out->ops_off || !out->ops_len || !out->ops)
            rc = WFAHIP_ERR_OOM;
    }
    if (rc == WFAHIP_OK) {
        run_threads(S, [&](size_t k) {
            const Shard &s = sh[k];
            if (s.count == 0) return;
            const wfahip_results &r = s.res;
#define COPY(field) std::memcpy(out->field + s.first, r.field, (size_t)s.count * sizeof *r.field);
            COPY(status) COPY(score) COPY(tbegin) COPY(tend) COPY(qbegin) COPY(qend) COPY(align_len) COPY(matches) COPY(gaps)
            COPY(gap_regions) COPY(ops_len)
#undef COPY
            for (uint64_t i = 0; i < s.count; i++) out->ops_off[s.first + i] = r.status[i] == WFAHIP_PAIR_OK ? r.ops_off[i] + ops_base[k] : 0;
            if (r.n_ops) std::memcpy(out->ops + ops_base[k], r.ops, (size_t)r.n_ops * 8);
        });
    }
    for (Shard &s : sh) wfahip_results_free(&s.res);
    if (rc != WFAHIP_OK) wfahip_results_free(out);
    return rc;
}

}  // namespace

extern "C" int wfahip_align_batch_multi(wfahip_multi *m, const wfahip_params *p, const uint8_t *seq_blob, uint64_t blob_bytes,
                                        const uint64_t *q_off, const uint32_t *q_len, const uint64_t *t_off, const uint32_t *t_len,
                                        uint64_t n_pairs, wfahip_results *out) {
    try {
        return align_multi_impl(m, p, seq_blob, blob_bytes, q_off, q_len, t_off, t_len, n_pairs, out);
    } catch (const std::bad_alloc &) {
        return WFAHIP_ERR_OOM;
    } catch (...) {
        return WFAHIP_ERR_INTERNAL;
    }
}
