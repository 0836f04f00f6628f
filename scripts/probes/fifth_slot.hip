// Probe (round 6): can a small-register kernel run BESIDE a persistent 4-waves-per-SIMD kernel?
// wfa_duo_kernel holds 4 waves x 128 VGPRs = the whole 512-register file of every SIMD, so wfa_prepack_kernel (28 VGPRs) and
// wfa_backtrace_kernel (36) can only run before / after it.  If the forward kernel used at most 120 registers, 32 would be
// left per SIMD: a fifth wave of a kernel with <= 32 VGPRs fits.  This probe measures whether the dispatcher really
// co-schedules such a wave from another stream, and what it costs the big kernel:
//   A<V>: persistent ALU kernel, one wave64 per workgroup, 4 096 workgroups, V registers allocated (120 or 128), 9 KB LDS
//   B   : streaming copy kernel (256-thread workgroups, <= 32 VGPRs), 2 GB read + 0.5 GB written (the packing kernel's traffic)
//   C   : pointer-chasing kernel (<= 32 VGPRs): 1e6 lanes x 60 dependent 64-byte-line reads over a 16 GB arena (the backtrace's pattern)
// Times: A alone, B alone, C alone, A || B, A || C, A || B then C (two side streams).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

template <int V>
__global__ __launch_bounds__(64, 4) void k_big(unsigned *out, int iters) {
    extern __shared__ unsigned lds[];
    unsigned a0 = threadIdx.x + 100, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    lds[threadIdx.x] = a0;
    if (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else asm volatile("v_mov_b32 v119, 0" ::: "v119");
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_u32 %0, %1, %2\n v_max_u32 %3, %4, %5\n v_add_u32 %6, %7, %0\n v_max_u32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_max_u32 %7, %0, %1\n v_add_u32 %2, %3, %4\n v_max_u32 %5, %6, %7\n "
                     "v_add_u32 %0, %1, %2\n v_max_u32 %3, %4, %5\n v_add_u32 %6, %7, %0\n v_max_u32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_max_u32 %7, %0, %1\n v_add_u32 %2, %3, %4\n v_max_u32 %5, %6, %7\n "
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if ((i & 31) == 0) a0 += lds[(threadIdx.x + i) & 63];
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = 1;
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *in, unsigned *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = in[i];
        out[i] = v.x ^ v.y ^ v.z ^ v.w;
    }
}

__global__ __launch_bounds__(512) void k_chase(const unsigned *arena, size_t lines, unsigned *out, int hops) {
    size_t p = ((size_t)blockIdx.x * 512 + threadIdx.x) * 0x9E3779B97F4A7C15ull % lines;
    unsigned acc = 0;
    for (int h = 0; h < hops; h++) {
        const unsigned v = arena[p * 16];
        acc += v;
        p = (p * 6364136223846793005ull + v + 1442695040888963407ull) % lines;
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    hipStream_t sa, sb, sc;
    CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    const size_t n16 = (size_t)2 << 30 >> 4;  // 2 GB of uint4
    uint4 *d_in; unsigned *d_out, *d_flag, *d_arena, *d_acc;
    const size_t lines = (size_t)16 << 30 >> 6;
    CHK(hipMalloc(&d_in, n16 * 16)); CHK(hipMalloc(&d_out, n16 * 4)); CHK(hipMalloc(&d_flag, 64));
    CHK(hipMalloc(&d_arena, lines * 64)); CHK(hipMalloc(&d_acc, (size_t)1 << 22));
    CHK(hipMemset(d_in, 1, n16 * 16)); CHK(hipMemset(d_arena, 0, lines * 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 60000;
    auto big = [&](int V, hipStream_t s) {
        if (V == 128) hipLaunchKernelGGL(k_big<128>, dim3(4096), dim3(64), 9216, s, d_flag, iters);
        else hipLaunchKernelGGL(k_big<120>, dim3(4096), dim3(64), 9216, s, d_flag, iters);
    };
    auto copy = [&](hipStream_t s) { hipLaunchKernelGGL(k_copy, dim3(62500), dim3(256), 0, s, d_in, d_out, n16); };
    auto chase = [&](hipStream_t s) { hipLaunchKernelGGL(k_chase, dim3(1954), dim3(512), 0, s, d_arena, lines, d_acc, 60); };
    auto timed = [&](const char *name, auto fn) {
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            CHK(hipDeviceSynchronize());
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%-44s %8.3f ms\n", name, ms);
        }
    };
    hipFuncAttributes fa;
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_big<120>))); printf("k_big<120> regs %d\n", fa.numRegs);
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_big<128>))); printf("k_big<128> regs %d\n", fa.numRegs);
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_copy))); printf("k_copy regs %d\n", fa.numRegs);
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_chase))); printf("k_chase regs %d\n", fa.numRegs);
    for (int V : {128, 120}) {
        printf("== big kernel with %d VGPRs\n", V);
        timed("A alone", [&] { big(V, sa); });
        timed("B (copy) alone", [&] { copy(sb); });
        timed("C (chase) alone", [&] { chase(sc); });
        timed("A || B", [&] { big(V, sa); copy(sb); });
        timed("A || C", [&] { big(V, sa); chase(sc); });
        timed("A || (B ; C ; B ; C)", [&] { big(V, sa); copy(sb); chase(sb); copy(sb); chase(sb); });
        timed("A || B || C", [&] { big(V, sa); copy(sb); chase(sc); });
    }
    return 0;
}
