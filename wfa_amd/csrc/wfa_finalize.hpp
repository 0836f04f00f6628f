// wfa_finalize.hpp -- device-side assembly of wfahip_results for the host entry (wfahip_align_batch).
//
// The alignment kernels leave one 64-byte record per pair and the CIGAR ops of a pair wherever its backtrace got
// room in the shared op buffer (completion order, with slack between pairs).  AlignmentResult-as-arrays wants
// the ops packed in pair order and one array per field (include/wfa_hip.h: wfahip_results).  Doing that on the
// host cost more than the alignments (1e6 pairs: 0.7 s of D2H into fresh pages + per-pair memcpy, against 29 ms of
// kernels), so it is done here: an exclusive scan of ops_len in pair order (block scan, scan of the block sums),
// then one wave per pair copies its ops to their final place and one lane writes the field arrays.  The host only
// copies finished arrays.
#pragma once
#include "wfa_common.hpp"

namespace wfa {

constexpr int FIN_BLOCK = 1024, FIN_ITEMS = 4;  // scan: pairs per block = FIN_BLOCK * FIN_ITEMS

struct FinParams {
    const uint32_t *rec;      // [n][REC_WORDS]
    const uint64_t *ops;      // shared op buffer of the alignment kernels
    uint64_t        n;
    uint64_t       *loc_off;  // [n] exclusive offset inside the pair's scan block (64-bit: 4 096 long pairs can hold > 2^32 ops)
    uint64_t       *blk_sum;  // [n_blocks] ops per scan block, then exclusive bases
    // outputs (struct of arrays, device)
    int32_t  *status;
    uint32_t *score;
    int32_t  *tbegin, *tend, *qbegin, *qend;
    uint32_t *align_len, *matches, *gaps, *gap_regions;
    uint64_t *ops_off;
    uint32_t *ops_len;
    uint64_t *ops_out;
    uint64_t  ops_base;          // added to every ops_off (a slice of a batch: ops_out points ops_base entries into the final array)
    unsigned long long *totals;  // [0] total ops, [1] total cells
};

__device__ __forceinline__ uint32_t fin_len(const FinParams &F, uint64_t i) {
    const uint32_t *r = F.rec + i * REC_WORDS;
    return (i < F.n && r[REC_STATUS] == ST_OK) ? r[REC_OPS_LEN] : 0u;
}

// pass 1: per-pair offsets inside a block of FIN_BLOCK*FIN_ITEMS pairs + the block's total
__global__ __launch_bounds__(FIN_BLOCK) void fin_scan_blocks(const FinParams F) {
    __shared__ unsigned long long wsum[FIN_BLOCK / 64];
    const int      tid  = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t i0   = ((uint64_t)blockIdx.x * FIN_BLOCK + tid) * FIN_ITEMS;
    uint32_t           v[FIN_ITEMS];
    unsigned long long mine = 0ull, cells = 0ull;
#pragma unroll
    for (int k = 0; k < FIN_ITEMS; k++) {
        v[k] = fin_len(F, i0 + k), mine += v[k];
        if (i0 + k < F.n && F.rec[(i0 + k) * REC_WORDS + REC_STATUS] == ST_OK)
            cells += (unsigned long long)F.rec[(i0 + k) * REC_WORDS + REC_CELLS_LO] |
                     ((unsigned long long)F.rec[(i0 + k) * REC_WORDS + REC_CELLS_HI] << 32);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o, 64);
    if (lane == 0 && cells != 0ull) atomicAdd(&F.totals[1], cells);
    unsigned long long incl = mine;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long wbase = 0ull;
    for (int w = 0; w < wv; w++) wbase += wsum[w];
    unsigned long long run = wbase + incl - mine;
#pragma unroll
    for (int k = 0; k < FIN_ITEMS; k++) {
        if (i0 + k < F.n) F.loc_off[i0 + k] = run;
        run += v[k];
    }
    if (tid == FIN_BLOCK - 1) F.blk_sum[blockIdx.x] = run;
}

// pass 2: exclusive scan of the block sums (one block; the totals land in F.totals[0])
__global__ __launch_bounds__(1024) void fin_scan_sums(const FinParams F, uint32_t n_blocks) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0ull;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const uint32_t     b    = b0 + tid;
        unsigned long long mine = b < n_blocks ? F.blk_sum[b] : 0ull, incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned long long wbase = carry;
        for (int w = 0; w < wv; w++) wbase += wsum[w];
        if (b < n_blocks) F.blk_sum[b] = wbase + incl - mine;
        __syncthreads();
        if (tid == 1023) carry = wbase + incl;
        __syncthreads();
    }
    if (tid == 0) F.totals[0] = carry;
}

// pass 3: one wave per pair -- field arrays (lane 0) and the pair's ops copied to their final place
__global__ __launch_bounds__(256) void fin_gather(const FinParams F) {
    const uint64_t i    = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int      lane = threadIdx.x & 63;
    if (i >= F.n) return;
    const uint32_t *r  = F.rec + i * REC_WORDS;
    const uint32_t  st = r[REC_STATUS];
    const bool      ok = st == ST_OK;
    const uint64_t  dst = F.blk_sum[i / (FIN_BLOCK * FIN_ITEMS)] + F.loc_off[i];
    const uint32_t  len = ok ? r[REC_OPS_LEN] : 0u;
    if (lane == 0) {
        F.status[i]      = (st == ST_OK || st == ST_EMPTY || st == ST_TOO_LONG) ? (int32_t)st : 4 /* WFAHIP_PAIR_NO_MEMORY */;
        F.score[i]       = ok ? r[REC_SCORE] : 0u;
        F.tbegin[i]      = ok ? (int32_t)r[REC_TBEGIN] : 0;
        F.tend[i]        = ok ? (int32_t)r[REC_TEND] : 0;
        F.qbegin[i]      = ok ? (int32_t)r[REC_QBEGIN] : 0;
        F.qend[i]        = ok ? (int32_t)r[REC_QEND] : 0;
        F.align_len[i]   = ok ? r[REC_ALIGN_LEN] : 0u;
        F.matches[i]     = ok ? r[REC_MATCHES] : 0u;
        F.gaps[i]        = ok ? r[REC_GAPS] : 0u;
        F.gap_regions[i] = ok ? r[REC_GAP_REGIONS] : 0u;
        F.ops_len[i]     = len;
        F.ops_off[i]     = ok ? F.ops_base + dst : 0ull;
    }
    if (len == 0u) return;
    const uint64_t src = (uint64_t)r[REC_OPS_OFF_LO] | ((uint64_t)r[REC_OPS_OFF_HI] << 32);
    for (uint32_t k = lane; k < len; k += 64) F.ops_out[dst + k] = F.ops[src + k];
}

}  // namespace wfa
