// exchange_probe.hip -- the barrier-with-reduction of wfa_teamc_kernel (wfa_teamc.hpp: exchange()) on its own, in several forms:
// `teams` teams of T workgroups of 1 024 threads (team = blockIdx % teams: with eight teams each sits on one XCD), N rounds; every
// workgroup hands in NF values derived from (round, workgroup) and checks the minima it gets back.
//   variant 0: slots, plain 8-byte stores, NF agent-scope 8-byte loads per lane and poll      (the kernel's, teams on one XCD)
//   variant 1: slots, plain 8-byte stores, NF/2 agent-scope 16-byte loads per lane and poll
//   variant 2: atomicMin into NF accumulators + arrival counter, poll the counter, read the accumulators  (wfa_team_kernel's)
//   variant 3: slots, agent-scope (write-through) stores, 8-byte loads                        (teams spread over XCDs)
//   variant 4: slots, plain stores; poll ONE word per slot (the last stored), then read the rest once with 16-byte loads
//   variant 9: latency of one dependent agent-scope load (lane 0 of every workgroup chases a pointer)
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/exchange_probe scripts/probes/exchange_probe.hip
// Run:   /tmp/exchange_probe [T] [rounds] [variant] [teams] [NF] [sleep]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
constexpr int MAX_T = 64, SLOT = 16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ u32x4 ld16(const void *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int NF>
__global__ __launch_bounds__(1024) void probe(unsigned long long *slots_all, uint32_t *ctl_all, uint32_t T, uint32_t teams, uint32_t rounds, int variant, int slp,
                                              uint32_t *errors) {
    __shared__ int red[64];
    const int      tid = threadIdx.x, lane = tid & 63;
    const uint32_t team = blockIdx.x % teams, b = blockIdx.x / teams;
    unsigned long long *const slots = slots_all + (size_t)team * 4 * MAX_T * SLOT;
    uint32_t *const           ctl   = ctl_all + (size_t)team * 256;
    uint32_t *const           acc   = ctl + 64;  // 3 sets of 16
    uint32_t                  xseq = 0, nerr = 0, target = 0;
    if (variant == 9) {
        if (tid == 0) {
            uint32_t p = b;
            for (uint32_t r = 0; r < rounds; r++) p = __hip_atomic_load(ctl + 128 + (p & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + b;
            if (p == 0xFFFFFFFFu) errors[3] = p;
        }
        return;
    }
    for (uint32_t r = 0; r < rounds; r++) {
        if (tid < NF) red[16 + tid] = (int)(r * 31u + (uint32_t)tid + ((b * 7u + r) % T));
        __syncthreads();
        xseq += 1u;
        if (tid < 64) {
            bool bad = false;
            int  v[NF];
            if (variant == 2) {
                uint32_t *const a = acc + (xseq % 3u) * 16u;
                if (b == 0 && lane < NF) {
                    acc[((xseq + 1u) % 3u) * 16u + lane] = 0x7FFFFFFFu;
                    wait_vm();
                }
                if (lane < NF) atomicMin((int *)a + lane, red[16 + lane]);
                wait_vm();
                target += T;
                if (lane == 0) {
                    uint32_t spins = 0;
                    if ((int32_t)(atomicAdd(&ctl[0], 1u) + 1u - target) < 0)
                        while ((int32_t)(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                            if (++spins > (1u << 22)) {
                                bad = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                }
                const int got = lane < NF ? (int)__hip_atomic_load(a + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                if (lane < NF) red[16 + lane] = got;
            } else {
                unsigned long long *const set = slots + (size_t)(xseq & 1u) * MAX_T * SLOT;
                if (tid == 0) {
                    if (variant == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        const unsigned long long wv = ((unsigned long long)xseq << 32) | (uint32_t)red[16 + f];
                        if (variant == 3) __hip_atomic_store(set + (size_t)b * SLOT + f, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else set[(size_t)b * SLOT + f] = wv;
                    }
                }
                uint32_t spins = 0;
                for (;;) {
                    bool ok = true;
                    if (variant == 1) {
                        u32x4 w[NF / 2];
#pragma unroll
                        for (int f = 0; f < NF / 2; f++)
                            if ((uint32_t)lane < T) w[f] = ld16(set + (size_t)lane * SLOT + 2 * f);
                        wait_vm();
#pragma unroll
                        for (int f = 0; f < NF / 2; f++) {
                            if ((uint32_t)lane < T) {
                                ok       = ok && w[f].y == xseq && w[f].w == xseq;
                                v[2 * f] = (int)w[f].x, v[2 * f + 1] = (int)w[f].z;
                            } else v[2 * f] = v[2 * f + 1] = 0x7FFFFFFF;
                        }
                    } else if (variant == 4) {
                        const unsigned long long w0 =
                            (uint32_t)lane < T ? __hip_atomic_load(set + (size_t)lane * SLOT + NF - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)xseq << 32);
                        ok = (uint32_t)(w0 >> 32) == xseq;
                        if (__ballot(!ok) == 0ull) {
                            u32x4 w[NF / 2];
#pragma unroll
                            for (int f = 0; f < NF / 2; f++)
                                if ((uint32_t)lane < T) w[f] = ld16(set + (size_t)lane * SLOT + 2 * f);
                            wait_vm();
#pragma unroll
                            for (int f = 0; f < NF / 2; f++) {
                                if ((uint32_t)lane < T) {
                                    ok       = ok && w[f].y == xseq && w[f].w == xseq;
                                    v[2 * f] = (int)w[f].x, v[2 * f + 1] = (int)w[f].z;
                                } else v[2 * f] = v[2 * f + 1] = 0x7FFFFFFF;
                            }
                        }
                    } else {
#pragma unroll
                        for (int f = 0; f < NF; f++) {
                            const unsigned long long w = (uint32_t)lane < T ? __hip_atomic_load(set + (size_t)lane * SLOT + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                                            : ((unsigned long long)xseq << 32) | 0x7FFFFFFFull;
                            ok   = ok && (uint32_t)(w >> 32) == xseq;
                            v[f] = (int)(uint32_t)w;
                        }
                    }
                    if (__ballot(!ok) == 0ull) break;
                    if (++spins > (1u << 22)) {
                        bad = true;
                        break;
                    }
                    if (slp == 1) __builtin_amdgcn_s_sleep(1);
                    else if (slp == 2) __builtin_amdgcn_s_sleep(2);
                    else if (slp == 4) __builtin_amdgcn_s_sleep(4);
                }
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const int m = wave_min(v[f]);
                    if (lane == 0) red[16 + f] = m;
                }
            }
            if (__ballot(bad) != 0ull && lane == 0) red[31 + 16] = 1;
            else if (lane == 0) red[31 + 16] = 0;
        }
        __syncthreads();
        if (red[31 + 16]) {
            if (tid == 0) atomicAdd(errors + 1, 1u), errors[2] = r;
            return;
        }
        if (tid < NF) nerr += red[16 + tid] != (int)(r * 31u + (uint32_t)tid);
        __syncthreads();
    }
    if (nerr) atomicAdd(errors, nerr);
}
int main(int argc, char **argv) {
    const uint32_t T = argc > 1 ? atoi(argv[1]) : 32, rounds = argc > 2 ? atoi(argv[2]) : 100000;
    const int      variant = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t teams = argc > 4 ? atoi(argv[4]) : 8;
    const int      nf = argc > 5 ? atoi(argv[5]) : 8, slp = argc > 6 ? atoi(argv[6]) : 1;
    unsigned long long *slots;
    uint32_t           *ctl, *err;
    const size_t        sb = (size_t)teams * 4 * MAX_T * SLOT * 8, cb = (size_t)teams * 256 * 4;
    hipMalloc(&slots, sb), hipMemset(slots, 0, sb);
    hipMalloc(&ctl, cb);
    std::vector<uint32_t> h0(teams * 256, 0u);
    for (uint32_t t = 0; t < teams; t++)
        for (int i = 0; i < 48; i++) h0[t * 256 + 64 + i] = 0x7FFFFFFFu;
    for (uint32_t t = 0; t < teams; t++)
        for (int i = 0; i < 64; i++) h0[t * 256 + 128 + i] = (uint32_t)((i * 37 + 11) & 63);
    hipMemcpy(ctl, h0.data(), cb, hipMemcpyHostToDevice);
    hipMalloc(&err, 64), hipMemset(err, 0, 64);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipEventRecord(a);
    if (nf == 8) hipLaunchKernelGGL(probe<8>, dim3(T * teams), dim3(1024), 0, 0, slots, ctl, T, teams, rounds, variant, slp, err);
    else if (nf == 4) hipLaunchKernelGGL(probe<4>, dim3(T * teams), dim3(1024), 0, 0, slots, ctl, T, teams, rounds, variant, slp, err);
    else hipLaunchKernelGGL(probe<14>, dim3(T * teams), dim3(1024), 0, 0, slots, ctl, T, teams, rounds, variant, slp, err);
    hipEventRecord(b);
    const hipError_t e = hipDeviceSynchronize();
    float            ms = 0;
    hipEventElapsedTime(&ms, a, b);
    uint32_t h[4];
    hipMemcpy(h, err, 16, hipMemcpyDeviceToHost);
    printf("exchange probe: variant %d, %u teams of T = %u, NF = %d, sleep %d, %u rounds: %s, wrong minima %u, timeouts %u (round %u), %.3f us per %s\n", variant, teams, T, nf, slp,
           rounds, hipGetErrorString(e), h[0], h[1], h[2], ms * 1e3 / rounds, variant == 9 ? "load" : "exchange");
    return 0;
}
