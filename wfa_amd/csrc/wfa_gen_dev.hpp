// wfa_gen_dev.hpp -- the seeded synthetic DNA-pair generator of wfa_gen.cpp, on the device.
//
// Same dataset, byte for byte: pair i depends only on (seed, first_index + i); the pattern is L uniform bases, the
// text is the pattern after E = round(L e) sequential edits (mismatch / insertion / deletion at uniform positions of
// the CURRENT text).  One workgroup per pair.  The edits are sequential by definition, but everything inside an edit
// is not: splitmix64 is a counter-based generator (call c returns mix(s0 + (c + 1) G)), so every thread knows every
// random number without communication, and the memmove of an insertion / deletion is done by all threads on the text
// held in LDS -- each thread shifts its own contiguous run, after saving the one byte a neighbour would overwrite.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wfa {

constexpr int GEN_THREADS = 256;

__device__ __forceinline__ uint64_t gen_mix(uint64_t s0, uint64_t call) {  // the (call+1)-th SplitMix64::next() of a pair
    uint64_t z = s0 + (call + 1ull) * 0x9E3779B97F4A7C15ull;
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t gen_below(uint64_t r, uint32_t n) { return (uint32_t)__umul64hi(r, (uint64_t)n); }  // uniform in [0, n)

__global__ __launch_bounds__(GEN_THREADS) void wfa_gen_kernel(uint64_t seed, uint64_t first, uint64_t n_pairs, uint32_t L, uint32_t E,
                                                              uint64_t stride, uint8_t *blob, uint64_t *q_off, uint32_t *q_len,
                                                              uint64_t *t_off, uint32_t *t_len) {
    extern __shared__ uint8_t txt[];  // the text being edited: L + E + 16 bytes
    const uint64_t i = blockIdx.x;
    if (i >= n_pairs) return;
    const int      tid  = threadIdx.x;
    const uint64_t s0   = seed ^ ((first + i) * 0x9E3779B97F4A7C15ull);
    const uint64_t qcap = ((uint64_t)L + 15) & ~15ull;
    uint8_t *const q    = blob + i * stride;
    uint8_t *const t    = q + qcap;
    const uint32_t B    = 0x54474341u;  // "ACGT"

    // ---- pattern: base j comes from bits 2 (j % 32) of call j / 32
    for (uint32_t j = tid; j < L; j += GEN_THREADS) {
        const uint8_t b = (uint8_t)(B >> (8u * (uint32_t)((gen_mix(s0, j >> 5) >> (2u * (j & 31u))) & 3ull)));
        q[j]   = b;
        txt[j] = b;
    }
    __syncthreads();
    uint64_t call = ((uint64_t)L + 31) / 32;  // calls used so far
    uint32_t len  = L;
    for (uint32_t k = 0; k < E; k++) {  // every thread follows the same scalar sequence
        const uint32_t type = gen_below(gen_mix(s0, call++), 3);
        if (type == 0) {  // mismatch: one of the three other bases
            const uint32_t pos = gen_below(gen_mix(s0, call++), len);
            const uint8_t  c   = txt[pos];
            const uint32_t old = (c == 'A') ? 0u : (c == 'C') ? 1u : (c == 'G') ? 2u : 3u;
            const uint32_t nw  = (old + 1u + gen_below(gen_mix(s0, call++), 3)) & 3u;
            __syncthreads();  // (everybody has read txt[pos])
            if (tid == 0) txt[pos] = (uint8_t)(B >> (8u * nw));
            __syncthreads();
        } else if (type == 1) {  // insertion before position pos (pos == len appends)
            const uint32_t pos = gen_below(gen_mix(s0, call++), len + 1u);
            const uint8_t  b   = (uint8_t)(B >> (8u * gen_below(gen_mix(s0, call++), 4)));
            // new[i + 1] = old[i] for i in [pos, len): thread T owns the run [a, e) of source positions
            const uint32_t span = len - pos, run = (span + GEN_THREADS - 1) / GEN_THREADS;
            const uint32_t a = pos + min(span, (uint32_t)tid * run), e = pos + min(span, ((uint32_t)tid + 1u) * run);
            const uint8_t  keep = e > a ? txt[a] : 0;  // the byte the thread below overwrites first
            __syncthreads();
            for (uint32_t p = e; p > a + 1u; p--) txt[p] = txt[p - 1u];  // top down, inside the own run
            if (e > a) txt[a + 1u] = keep;
            if (tid == 0) txt[pos] = b;
            len++;
            __syncthreads();
        } else {  // deletion (the text is kept non-empty)
            const uint32_t pos = gen_below(gen_mix(s0, call++), len);
            if (len > 1u) {
                // new[i - 1] = old[i] for i in (pos, len)
                const uint32_t lo = pos + 1u, span = len - lo, run = (span + GEN_THREADS - 1) / GEN_THREADS;
                const uint32_t a = lo + min(span, (uint32_t)tid * run), e = lo + min(span, ((uint32_t)tid + 1u) * run);
                const uint8_t  keep = e > a ? txt[e - 1u] : 0;  // the byte the thread above overwrites first
                __syncthreads();
                for (uint32_t p = a; p + 1u < e; p++) txt[p - 1u] = txt[p];  // bottom up, inside the own run
                if (e > a) txt[e - 2u] = keep;
                len--;
                __syncthreads();
            }
        }
    }
    for (uint32_t j = tid; j < len; j += GEN_THREADS) t[j] = txt[j];
    if (tid == 0) {
        q_off[i] = i * stride, q_len[i] = L;
        t_off[i] = i * stride + qcap, t_len[i] = len;
    }
}

}  // namespace wfa
