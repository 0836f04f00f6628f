// wfa_gen.cpp -- seeded synthetic DNA-pair generator (host side of the C-ABI).
//
// Mirrors the datasets the reference's benchmark used (README.md:298-306: WFA's
// `generate_dataset -n N -l L -e E`, whose source is not part of the reference checkout): a random
// ACGT pattern of length L and a text made from it by round(L*e) random edits (mismatch /
// insertion / deletion, uniformly chosen, at uniformly chosen positions).  query = pattern,
// target = text, the order of the ">" / "<" lines the CLI reads (wfa-go/wfa-go.go:166-178).
//
// Deterministic: pair i depends only on (seed, first_index + i).
#include "../../include/wfa_hip.h"

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct SplitMix64 {
    uint64_t s;
    inline uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    inline uint32_t below(uint32_t n) {  // uniform in [0, n)
        return (uint32_t)(((unsigned __int128)next() * n) >> 64);
    }
};

inline uint32_t n_edits(uint32_t length, double e) { return (uint32_t)std::llround((double)length * e); }

void gen_range(uint64_t seed, uint64_t first, uint64_t begin, uint64_t end, uint32_t L, double er,
               uint64_t stride, uint8_t *blob, uint64_t *q_off, uint32_t *q_len, uint64_t *t_off,
               uint32_t *t_len) {
    static const char B[4] = {'A', 'C', 'G', 'T'};
    const uint32_t    E    = n_edits(L, er);
    const uint64_t    qcap = ((uint64_t)L + 15) & ~15ull;
    for (uint64_t i = begin; i < end; i++) {
        SplitMix64 r{seed ^ ((first + i) * 0x9E3779B97F4A7C15ull)};
        uint8_t   *q = blob + i * stride;
        uint8_t   *t = q + qcap;
        uint64_t   bits = 0;
        for (uint32_t j = 0; j < L; j++) {
            if ((j & 31) == 0) bits = r.next();
            q[j] = (uint8_t)B[bits & 3];
            bits >>= 2;
        }
        std::memcpy(t, q, L);
        uint32_t len = L;
        for (uint32_t k = 0; k < E; k++) {
            uint32_t type = r.below(3);
            if (type == 0) {  // mismatch: one of the three other bases
                uint32_t pos = r.below(len);
                uint32_t old = (t[pos] == 'A') ? 0 : (t[pos] == 'C') ? 1 : (t[pos] == 'G') ? 2 : 3;
                t[pos]       = (uint8_t)B[(old + 1 + r.below(3)) & 3];
            } else if (type == 1) {  // insertion before position pos (pos == len appends)
                uint32_t pos = r.below(len + 1);
                uint8_t  b   = (uint8_t)B[r.below(4)];
                std::memmove(t + pos + 1, t + pos, len - pos);
                t[pos] = b;
                len++;
            } else {  // deletion (kept non-empty)
                uint32_t pos = r.below(len);
                if (len > 1) {
                    std::memmove(t + pos, t + pos + 1, len - pos - 1);
                    len--;
                }
            }
        }
        q_off[i] = i * stride;
        q_len[i] = L;
        t_off[i] = i * stride + qcap;
        t_len[i] = len;
    }
}

}  // namespace

extern "C" uint64_t wfahip_gen_stride(uint32_t length, double error_rate) {
    const uint64_t qcap = ((uint64_t)length + 15) & ~15ull;
    const uint64_t tcap = ((uint64_t)length + n_edits(length, error_rate) + 15) & ~15ull;
    return qcap + tcap;
}

extern "C" int wfahip_generate_pairs(uint64_t seed, uint64_t first_index, uint64_t n_pairs, uint32_t length,
                                     double error_rate, int n_threads, uint8_t *blob, uint64_t *q_off,
                                     uint32_t *q_len, uint64_t *t_off, uint32_t *t_len) {
    if (!blob || !q_off || !q_len || !t_off || !t_len || length == 0 || error_rate < 0.0) return WFAHIP_ERR_BAD_ARG;
    const uint64_t stride = wfahip_gen_stride(length, error_rate);
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_pairs) n_threads = n_pairs ? (int)n_pairs : 1;
    std::vector<std::thread> th;
    try {
        th.reserve((size_t)n_threads);
    } catch (...) {
        n_threads = 1;
    }
    const uint64_t           per = (n_pairs + n_threads - 1) / n_threads;
    for (int i = 0; i < n_threads; i++) {
        uint64_t b = std::min<uint64_t>(n_pairs, (uint64_t)i * per), e = std::min<uint64_t>(n_pairs, b + per);
        bool inline_run = n_threads == 1;
        if (!inline_run) {
            try {
                th.emplace_back(gen_range, seed, first_index, b, e, length, error_rate, stride, blob, q_off, q_len,
                                t_off, t_len);
            } catch (...) {  // no thread to be had: this range is generated here (no exception crosses the C-ABI,
                inline_run = true;  // and no joinable thread is destroyed)
            }
        }
        if (inline_run) gen_range(seed, first_index, b, e, length, error_rate, stride, blob, q_off, q_len, t_off, t_len);
    }
    for (auto &t : th) t.join();
    return WFAHIP_OK;
}
