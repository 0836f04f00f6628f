// Probe (round 3): calibration of scripts/probes/valu_rate2.hip.  Round 2 reported s_memtime ticks per instruction and
// found class-A forms at 1.45 ticks per SIMD instruction with 8 waves per SIMD -- below the 2-cycle floor of a wave64
// instruction on a SIMD -- so either a tick is not a core cycle or the waves were not where the probe assumed.  This
// version checks both:
//   * every wave records s_memtime AND s_memrealtime (the constant 100 MHz counter) around its loop, and the host times
//     the launch with HIP events: ticks per microsecond = the frequency s_memtime really counts at;
//   * every wave records HW_REG_HW_ID and HW_REG_XCC_ID: the host counts the waves per (XCC, SE, SH, CU, SIMD) and prints
//     the histogram -- "w waves on every SIMD" is asserted, not assumed;
//   * rates are printed in nanoseconds per SIMD instruction (no clock assumption) and in core cycles at the clock the
//     run sustained, measured with a chain of dependent v_add_u32 of one wave per SIMD (4-cycle issue-to-issue on wave64:
//     cycles = chain length x 4), which ties the tick to the core clock without trusting a nominal figure.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#define REP8(x) x x x x x x x x
#define CHK(e) do { if ((e) != hipSuccess) { printf("hip error line %d\n", __LINE__); exit(2); } } while (0)
#define I3(ins) ins " %0, %1, %2\n " ins " %3, %4, %5\n " ins " %6, %7, %0\n " ins " %1, %2, %3\n " ins " %4, %5, %6\n " ins " %7, %0, %1\n " ins " %2, %3, %4\n " ins " %5, %6, %7\n "
#define IDPP(ins, ctl) ins " %0, %4, %0 " ctl "\n " ins " %1, %5, %1 " ctl "\n " ins " %2, %6, %2 " ctl "\n " ins " %3, %7, %3 " ctl "\n " ins " %4, %0, %4 " ctl "\n " ins " %5, %1, %5 " ctl "\n " ins " %6, %2, %6 " ctl "\n " ins " %7, %3, %7 " ctl "\n "
#define DEP8(ins) ins " %0, %0, %1\n " ins " %0, %0, %1\n " ins " %0, %0, %1\n " ins " %0, %0, %1\n " ins " %0, %0, %1\n " ins " %0, %0, %1\n " ins " %0, %0, %1\n " ins " %0, %0, %1\n "

struct Rec { unsigned long long dt_mem, dt_real; unsigned hw_id, xcc_id; };

#define KERNEL(NAME, BODY)                                                                                                   \
    __global__ __launch_bounds__(64) void NAME(Rec *out, int iters) {                                                        \
        unsigned a0 = threadIdx.x + 100, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();                        \
        for (int i = 0; i < iters; i++)                                                                                      \
            asm volatile(REP8(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"vcc", "s40", "s41", "s42", "s43"); \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
        unsigned hw, xcc;                                                                                                    \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));   \
        if (threadIdx.x == 0) out[blockIdx.x] = Rec{t1 - t0, r1 - r0, hw, xcc};                                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0].hw_id = 1;                                              \
    }
#define CTL_XOR1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
KERNEL(k_dep_add, DEP8("v_add_u32"))
KERNEL(k_add_u32, I3("v_add_u32"))
KERNEL(k_xor, I3("v_xor_b32"))
KERNEL(k_max_u32, I3("v_max_u32"))
KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %4, 3, %5\n v_lshlrev_b32 %6, 3, %7\n v_lshlrev_b32 %1, 3, %0\n v_lshlrev_b32 %3, 3, %2\n v_lshlrev_b32 %5, 3, %4\n v_lshlrev_b32 %7, 3, %6\n ")
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_lt_u32 vcc, %5, %6\n v_cndmask_b32 %7, %1, %2, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %5, %6, %0, vcc\n v_cmp_lt_u32 vcc, %7, %1\n v_cndmask_b32 %2, %3, %4, vcc\n ")
KERNEL(k_max_i32_dpp, IDPP("v_max_i32_dpp", CTL_XOR1))
KERNEL(k_mix_add_max, "v_add_u32 %0, %1, %2\n v_max_u32 %3, %4, %5\n v_add_u32 %6, %7, %0\n v_max_u32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_max_u32 %7, %0, %1\n v_add_u32 %2, %3, %4\n v_max_u32 %5, %6, %7\n ")

struct Ent { const char *name; void (*fn)(Rec *, int); };
#define E(n) {#n, n}
static Ent ents[] = {E(k_dep_add), E(k_add_u32), E(k_xor), E(k_max_u32), E(k_lshlrev), E(k_cmp_cnd), E(k_max_i32_dpp), E(k_mix_add_max)};

int main() {
    const int iters = 2000;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs, clockRate %d kHz (nominal peak)\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    double cyc_per_tick = 0;  // from the dependent chain at one wave per SIMD
    for (int w : {1, 2, 4, 5, 8}) {
        for (const Ent &e : ents) {
            const int blocks = prop.multiProcessorCount * 4 * w;
            Rec *d;
            CHK(hipMalloc(&d, blocks * sizeof(Rec)));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d, iters);  // warm-up (clocks, code)
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d, iters);
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Rec> h(blocks);
            CHK(hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
            CHK(hipFree(d));
            // waves per SIMD actually seen: key = xcc, se, sh, cu, simd (HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
            std::map<unsigned, int> per_simd;
            for (const Rec &r : h) per_simd[((r.xcc_id & 0xF) << 16) | (r.hw_id & 0xFF30)]++;
            int mn = 1 << 30, mx = 0;
            for (auto &kv : per_simd) mn = std::min(mn, kv.second), mx = std::max(mx, kv.second);
            std::vector<double> tm(blocks), tr(blocks);
            for (int i = 0; i < blocks; i++) tm[i] = (double)h[i].dt_mem, tr[i] = (double)h[i].dt_real;
            std::sort(tm.begin(), tm.end());
            std::sort(tr.begin(), tr.end());
            const double n_ins = (double)iters * 64;
            const double med_t = tm[blocks / 2], med_r = tr[blocks / 2];
            const double tick_mhz = med_t / (med_r / 100.0);  // memtime ticks per microsecond (memrealtime = 100 MHz)
            const double ns_wave  = med_r * 10.0 / n_ins;     // ns per instruction of one wave
            if (w == 1 && !strcmp(e.name, "k_dep_add")) cyc_per_tick = 4.0 / (med_t / n_ins);  // dependent wave64 VALU: 4 cycles issue to issue
            printf("%-14s target waves/SIMD %d  SIMDs used %4zu  waves/SIMD seen min %d max %d | memtime %.1f MHz | kernel %.3f ms, longest wave %.3f ms | "
                   "ns/instr/wave %6.3f  ns per SIMD instr %6.3f  ticks per SIMD instr %6.3f  core cycles per SIMD instr %6.3f\n",
                   e.name + 2, w, per_simd.size(), mn, mx, tick_mhz, ms, tr[blocks - 1] * 1e-5, ns_wave, ns_wave / w, med_t / n_ins / w,
                   cyc_per_tick > 0 ? med_t / n_ins / w * cyc_per_tick : 0.0);
            fflush(stdout);
        }
    }
    printf("# core cycles per s_memtime tick (dependent v_add_u32 chain, one wave per SIMD, 4 cycles per instruction): %.4f\n", cyc_per_tick);
    return 0;
}
