// Litmus probe for the hand-over the team kernel relies on (wfa_team.hpp, team_barrier with fenced = false):
//   producer workgroup: data words with agent-scope relaxed atomic STORES (write-through, sc1)
//                       -> __syncthreads -> [fence] -> one lane: agent-scope atomic add on a counter
//   consumer workgroup: one lane spins on the counter with agent-scope relaxed atomic loads -> __syncthreads
//                       -> data words with agent-scope relaxed atomic LOADS (sc1), NO acquire fence, NO invalidate
// on DIFFERENT XCDs (per-XCD L2s are not coherent).  Four variants of the two sides:
//   A  release fence (agent) before the arrive, sc1 loads                     = what the kernel ships (team_strict = 1)
//   B  no fence before the arrive (only __syncthreads = s_waitcnt vmcnt(0)), sc1 loads   = round 1's barrier
//   C  release fence, PLAIN cached loads on the consumer side                  = why the loads must be sc1
//   D  release fence + acquire fence after the spin, plain loads               = the textbook form (the kernel's "fenced" barrier)
// Every round the producer writes a new value pattern into one of two 64 KB windows (so that a stale line from an
// earlier round is a WRONG value, not the right one, and -- with few pairs -- is still in the consumer's L2); the consumer counts words that are not this round's value.
// Expected: A = 0 and D = 0 stale words; B > 0 (the store's vmcnt returns before the data is at the memory side); C > 0.
// Output: stale words per variant over ROUNDS x PAIRS x WORDS checks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

constexpr int THREADS = 1024, WINDOW = 2, U = 16;  // each round uses one of WINDOW disjoint sets of U x THREADS words (64 KB: more than a wave's stores in flight)

template <int VARIANT>
__global__ __launch_bounds__(THREADS) void litmus(uint32_t *data, uint32_t *ctl, unsigned long long *stale, int rounds, int delay) {
    // workgroups 2p (producer) and 2p + 1 (consumer) form pair p; consecutive workgroups land on different XCDs
    const int pair = blockIdx.x >> 1, tid = threadIdx.x;
    const bool producer = (blockIdx.x & 1) == 0;
    uint32_t *const d = data + (size_t)pair * THREADS * WINDOW * U;
    uint32_t *const ready = ctl + pair * 32, *const ack = ctl + pair * 32 + 16;
    unsigned long long bad = 0;
    __shared__ volatile int dead_s;
    volatile int *const dead = &dead_s;
    if (tid == 0) dead_s = 0;
    __syncthreads();
    for (int r = 1; r <= rounds; r++) {
        uint32_t *const w = d + (size_t)(r % WINDOW) * THREADS * U;
        const uint32_t val = (uint32_t)r * 2654435761u + (uint32_t)tid;
        if (producer) {
#pragma unroll
            for (int u = 0; u < U; u++) __hip_atomic_store(w + u * THREADS + tid, val + (uint32_t)u * 7919u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid == 0) {
                if (VARIANT != 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t spins = 0;
                while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)r && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 22)) *dead = 1;  // (bounded: a probe must never hang the device)
            }
            __syncthreads();
            if (*dead) return;
        } else {
            if (tid == 0) {
                uint32_t spins = 0;
                while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)r && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 22)) *dead = 1;
                if (VARIANT == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (*dead) return;
            for (int i = 0; i < delay; i++) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int u = U - 1; u >= 0; u--) {  // (the words stored last first)
                uint32_t got;
                if (VARIANT == 2 || VARIANT == 3) got = w[u * THREADS + tid];  // plain (cached) load
                else got = __hip_atomic_load(w + u * THREADS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad += got != val + (uint32_t)u * 7919u;
            }
            __syncthreads();
            if (tid == 0) __hip_atomic_store(ack, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!producer) {
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
        if ((tid & 63) == 0 && bad) atomicAdd(stale, bad);
    }
}

int main(int argc, char **argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 100, rounds = argc > 2 ? atoi(argv[2]) : 20000, mask = argc > 3 ? atoi(argv[3]) : 15;
    uint32_t *data, *ctl;
    unsigned long long *stale;
    CHK(hipMalloc(&data, (size_t)pairs * THREADS * WINDOW * U * 4));
    CHK(hipMalloc(&ctl, (size_t)pairs * 32 * 4));
    CHK(hipMalloc(&stale, 8));
    const char *names[4] = {"A release before the arrive, sc1 loads (shipped)", "B no release, sc1 loads (round 1's barrier)",
                            "C release, plain cached loads", "D release + acquire, plain loads (fenced barrier)"};
    printf("# %d producer/consumer pairs of %d-thread workgroups on different XCDs, %d rounds each, %d words per round: %.3g checks per variant\n",
           pairs, THREADS, rounds, THREADS * U, (double)pairs * rounds * THREADS * U);
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 4; v++) {
            if (!((mask >> v) & 1)) continue;
            CHK(hipMemset(data, 0xA5, (size_t)pairs * THREADS * WINDOW * U * 4));
            CHK(hipMemset(ctl, 0, (size_t)pairs * 32 * 4));
            CHK(hipMemset(stale, 0, 8));
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            CHK(hipEventRecord(e0, 0));
            switch (v) {
            case 0: hipLaunchKernelGGL(litmus<0>, dim3(2 * pairs), dim3(THREADS), 0, 0, data, ctl, stale, rounds, 0); break;
            case 1: hipLaunchKernelGGL(litmus<1>, dim3(2 * pairs), dim3(THREADS), 0, 0, data, ctl, stale, rounds, 0); break;
            case 2: hipLaunchKernelGGL(litmus<2>, dim3(2 * pairs), dim3(THREADS), 0, 0, data, ctl, stale, rounds, 0); break;
            default: hipLaunchKernelGGL(litmus<3>, dim3(2 * pairs), dim3(THREADS), 0, 0, data, ctl, stale, rounds, 0); break;
            }
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h = 0;
            CHK(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
            printf("run %d  %-52s stale words %10llu   (%.1f ms, %.2f us per round trip)\n", rep, names[v], h, ms, ms * 1e3 / rounds);
            fflush(stdout);
        }
    return 0;
}
