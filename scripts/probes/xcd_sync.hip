// Probe: can the workgroups that share an XCD exchange data and synchronise through that XCD's L2 (plain stores,
// L1 invalidate + plain loads, workgroup-scope atomics), and what does a barrier + exchange step cost there against
// the memory-side protocol of wfa_team.hpp (agent-scope atomics, sc1 loads / stores)?
// Build: hipcc -O2 --offload-arch=gfx950 -o /tmp/xcd_sync scripts/probes/xcd_sync.hip ; run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int MAXT = 64;
constexpr uint32_t SPIN_MAX = 1u << 22;
__device__ inline uint32_t l2_atomic_add(uint32_t *p, uint32_t v) {  // a returning atomic, never folded into a load
    uint32_t r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
struct Team { uint32_t count; uint32_t members; uint32_t bar; uint32_t pad[13]; uint32_t slot[MAXT * 32]; };  // slot: one 128 B line per member

// mode 0: memory side (agent scope); mode 1: XCD-local (workgroup-scope atomics, plain stores, buffer_inv sc0 + plain
// loads); mode 2: like 1 without the L1 invalidate (must show stale reads if the test can see them)
template <int MODE>
__global__ void probe(Team *teams, uint32_t *xcc_of_block, uint32_t iters, unsigned long long *out_ticks, uint32_t *errors, uint32_t T_expected) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    const int tid = threadIdx.x;
    __shared__ uint32_t sh_rank, sh_T;
    Team *tm = teams + xcc;
    if (tid == 0) {
        xcc_of_block[blockIdx.x] = xcc;
        sh_rank = atomicAdd(&tm->members, 1u);
    }
    __syncthreads();
    const uint32_t rank = sh_rank;
    // wait until all blocks have registered (grid-wide, memory side)
    if (tid == 0) {
        atomicAdd(&teams[8].count, 1u);
        for (uint32_t sp = 0; __hip_atomic_load(&teams[8].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && sp < SPIN_MAX; sp++) __builtin_amdgcn_s_sleep(8);
        sh_T = __hip_atomic_load(&tm->members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint32_t T = sh_T;
    if (T != T_expected && tid == 0 && rank == 0) atomicAdd(&errors[1], 1u);
    uint32_t bar_target = 0, nerr = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 1; it <= iters; it++) {
        // every lane of the block writes one word of the block's line
        uint32_t *mine = tm->slot + rank * 32 + (tid & 31);
        const uint32_t val = (it << 8) | rank;
        if (MODE == 0) __hip_atomic_store(mine, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *reinterpret_cast<volatile uint32_t *>(mine) = val;
        __syncthreads();  // (s_waitcnt vmcnt(0) + s_barrier)
        if (tid == 0) {
            bar_target += T;
            if (MODE == 0) {
                if (atomicAdd(&tm->bar, 1u) + 1u < bar_target) {
                    uint32_t sp = 0;
                    while (__hip_atomic_load(&tm->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bar_target && ++sp < SPIN_MAX) __builtin_amdgcn_s_sleep(2);
                    if (sp >= SPIN_MAX) { atomicAdd(&errors[1], 1000u); it = iters; }
                }
            } else {
                if (l2_atomic_add(&tm->bar, 1u) + 1u < bar_target) {
                    uint32_t sp = 0;
                    while (l2_atomic_add(&tm->bar, 0u) < bar_target && ++sp < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
                    if (sp >= SPIN_MAX) { atomicAdd(&errors[1], 1000u); it = iters; }
                }
            }
        }
        __syncthreads();
        if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
        const uint32_t other = (rank + 1 + (it % (T > 1 ? T - 1 : 1))) % T;
        const uint32_t *theirs = tm->slot + other * 32 + (tid & 31);
        uint32_t got;
        if (MODE == 0) got = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else got = *reinterpret_cast<const volatile uint32_t *>(theirs);
        if (got != ((it << 8) | other)) nerr++;
        // second barrier so that nobody overwrites a line that is still being read
        __syncthreads();
        if (tid == 0) {
            bar_target += T;
            if (MODE == 0) {
                if (atomicAdd(&tm->bar, 1u) + 1u < bar_target) {
                    uint32_t sp = 0;
                    while (__hip_atomic_load(&tm->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bar_target && ++sp < SPIN_MAX) __builtin_amdgcn_s_sleep(2);
                    if (sp >= SPIN_MAX) { atomicAdd(&errors[1], 1000u); it = iters; }
                }
            } else {
                if (l2_atomic_add(&tm->bar, 1u) + 1u < bar_target) {
                    uint32_t sp = 0;
                    while (l2_atomic_add(&tm->bar, 0u) < bar_target && ++sp < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
                    if (sp >= SPIN_MAX) { atomicAdd(&errors[1], 1000u); it = iters; }
                }
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (nerr) atomicAdd(&errors[0], nerr);
    if (tid == 0 && rank == 0) out_ticks[xcc] = t1 - t0;
}

int main() {
    Team *teams; uint32_t *xcc_of_block, *errors; unsigned long long *ticks;
    const int grid = 256, iters = 5000;
    CK(hipMalloc(&teams, sizeof(Team) * 9)); CK(hipMalloc(&xcc_of_block, grid * 4)); CK(hipMalloc(&errors, 8)); CK(hipMalloc(&ticks, 8 * 8));
    for (int mode = 0; mode < 3; mode++) {
        CK(hipMemset(teams, 0, sizeof(Team) * 9)); CK(hipMemset(errors, 0, 8)); CK(hipMemset(ticks, 0, 64));
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, teams, xcc_of_block, iters, ticks, errors, 32u);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, teams, xcc_of_block, iters, ticks, errors, 32u);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, teams, xcc_of_block, iters, ticks, errors, 32u);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> xb(grid); uint32_t err[2]; unsigned long long tk[8];
        CK(hipMemcpy(xb.data(), xcc_of_block, grid * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(err, errors, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(tk, ticks, 64, hipMemcpyDeviceToHost));
        int rr = 0; for (int b = 0; b < grid; b++) rr += (xb[b] == (uint32_t)(b % 8));
        std::printf("mode %d (%s): blocks on XCC blockIdx%%8: %d of %d; wrong reads %u; teams not of 32: %u; us per exchange step (2 barriers):",
                    mode, mode == 0 ? "memory side" : mode == 1 ? "XCD-local, L1 invalidate" : "XCD-local, no invalidate", rr, grid, err[0], err[1]);
        for (int x = 0; x < 8; x++) std::printf(" %.2f", tk[x] / 100.0 / iters);
        std::printf("\n");
        std::fflush(stdout);
    }
    return 0;
}
