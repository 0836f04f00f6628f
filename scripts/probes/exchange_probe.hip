// exchange_probe.hip -- the barrier-with-reduction of wfa_teamc_kernel (wfa_teamc.hpp: exchange()) on its own: T workgroups of
// 1 024 threads, N rounds; every workgroup hands in eight values derived from (round, workgroup) and checks the minima it gets back.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/exchange_probe scripts/probes/exchange_probe.hip ; run: /tmp/exchange_probe [T] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
constexpr int MAX_T = 64, SLOT = 8;
__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
__global__ __launch_bounds__(1024) void probe(unsigned long long *slots, uint32_t *ctl, uint32_t T, uint32_t rounds, uint32_t *errors) {
    __shared__ int red[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t b = blockIdx.x;
    uint32_t xseq = 0, nerr = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        if (tid == 0)
            for (int f = 0; f < SLOT; f++) red[16 + f] = (int)((b * 7919u + r * 104729u + f * 13u) % 1000003u);
        __syncthreads();
        xseq += 1u;
        if (tid < 64) {
            unsigned long long *const set = slots + (size_t)(xseq & 1u) * MAX_T * SLOT;
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                for (int f = 0; f < SLOT; f++)
                    __hip_atomic_store(set + (size_t)b * SLOT + f, ((unsigned long long)xseq << 32) | (uint32_t)red[16 + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int v[SLOT];
            uint32_t spins = 0;
            bool bad = false;
            for (;;) {
                bool ok = true;
                for (int f = 0; f < SLOT; f++) {
                    const unsigned long long w = (uint32_t)lane < T ? __hip_atomic_load(set + (size_t)lane * SLOT + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                                    : ((unsigned long long)xseq << 32) | 0x7FFFFFFFull;
                    ok   = ok && (uint32_t)(w >> 32) == xseq;
                    v[f] = (int)(uint32_t)w;
                }
                if (__ballot(!ok) == 0ull) break;
                if ((++spins & 255u) == 0u && (spins > (1u << 20) || __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    atomicExch(&ctl[1], 1u);
                    bad = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            for (int f = 0; f < SLOT; f++) {
                const int m = wave_min(v[f]);
                if (lane == 0) red[16 + f] = m;
            }
            if (lane == 0) red[31] = bad;
        }
        __syncthreads();
        if (red[31]) {
            if (tid == 0) atomicAdd(errors + 1, 1u), errors[2] = r;
            return;
        }
        if (tid == 0)
            for (int f = 0; f < SLOT; f++) {
                int want = 0x7FFFFFFF;
                for (uint32_t bb = 0; bb < T; bb++) {
                    const int x = (int)((bb * 7919u + r * 104729u + f * 13u) % 1000003u);
                    want = x < want ? x : want;
                }
                nerr += red[16 + f] != want;
            }
        __syncthreads();
    }
    if (tid == 0 && nerr) atomicAdd(errors, nerr);
}
int main(int argc, char **argv) {
    const uint32_t T = argc > 1 ? atoi(argv[1]) : 32, rounds = argc > 2 ? atoi(argv[2]) : 100000;
    unsigned long long *slots;
    uint32_t *ctl, *err;
    hipMalloc(&slots, 2 * MAX_T * SLOT * 8), hipMemset(slots, 0, 2 * MAX_T * SLOT * 8);
    hipMalloc(&ctl, 64), hipMemset(ctl, 0, 64);
    hipMalloc(&err, 64), hipMemset(err, 0, 64);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(T), dim3(1024), 0, 0, slots, ctl, T, rounds, err);
    hipEventRecord(b);
    const hipError_t e = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    uint32_t h[3];
    hipMemcpy(h, err, 12, hipMemcpyDeviceToHost);
    printf("exchange probe: T = %u, %u rounds: %s, wrong minima %u, timeouts %u (round %u), %.3f us per exchange\n", T, rounds, hipGetErrorString(e), h[0], h[1], h[2],
           ms * 1e3 / rounds);
    return 0;
}
