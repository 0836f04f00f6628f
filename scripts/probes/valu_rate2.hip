// Probe: issue cost of the vector-instruction forms the alignment kernels use (or could use), gfx950.
// Every kernel runs long runs of INDEPENDENT instructions of one form; W waves per SIMD (W = 2, 4, 8; the blocked
// forward kernel runs at 4).  Output: s_memtime ticks per instruction of ONE wave and per instruction of the SIMD
// (= the former / W), and the latter relative to v_add_u32 at the same occupancy.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define REP8(x) x x x x x x x x
#define CHK(e) do { if ((e) != hipSuccess) { printf("hip error line %d\n", __LINE__); exit(2); } } while (0)
// eight instructions over eight registers, each reading two others: no instruction depends on the one before it
#define I3(ins) ins " %0, %1, %2\n " ins " %3, %4, %5\n " ins " %6, %7, %0\n " ins " %1, %2, %3\n " ins " %4, %5, %6\n " ins " %7, %0, %1\n " ins " %2, %3, %4\n " ins " %5, %6, %7\n "
#define I4(ins) ins " %0, %1, %2, %3\n " ins " %3, %4, %5, %6\n " ins " %6, %7, %0, %1\n " ins " %1, %2, %3, %4\n " ins " %4, %5, %6, %7\n " ins " %7, %0, %1, %2\n " ins " %2, %3, %4, %5\n " ins " %5, %6, %7, %0\n "
#define I2(ins) ins " %0, %1\n " ins " %2, %3\n " ins " %4, %5\n " ins " %6, %7\n " ins " %1, %0\n " ins " %3, %2\n " ins " %5, %4\n " ins " %7, %6\n "
#define ISH(ins) ins " %0, 3, %1\n " ins " %2, 3, %3\n " ins " %4, 3, %5\n " ins " %6, 3, %7\n " ins " %1, 3, %0\n " ins " %3, 3, %2\n " ins " %5, 3, %4\n " ins " %7, 3, %6\n "
#define ICMP32(ins) ins " vcc, %1, %2\n " ins " vcc, %4, %5\n " ins " vcc, %7, %0\n " ins " vcc, %2, %3\n " ins " vcc, %5, %6\n " ins " vcc, %0, %1\n " ins " vcc, %3, %4\n " ins " vcc, %6, %7\n "
#define ICMP64(ins) ins " s[40:41], %1, %2\n " ins " s[42:43], %4, %5\n " ins " s[40:41], %7, %0\n " ins " s[42:43], %2, %3\n " ins " s[40:41], %5, %6\n " ins " s[42:43], %0, %1\n " ins " s[40:41], %3, %4\n " ins " s[42:43], %6, %7\n "
#define IDPP(ins, ctl) ins " %0, %4, %0 " ctl "\n " ins " %1, %5, %1 " ctl "\n " ins " %2, %6, %2 " ctl "\n " ins " %3, %7, %3 " ctl "\n " ins " %4, %0, %4 " ctl "\n " ins " %5, %1, %5 " ctl "\n " ins " %6, %2, %6 " ctl "\n " ins " %7, %3, %7 " ctl "\n "
#define IMOVDPP(ctl) "v_mov_b32_dpp %0, %4 " ctl "\n v_mov_b32_dpp %1, %5 " ctl "\n v_mov_b32_dpp %2, %6 " ctl "\n v_mov_b32_dpp %3, %7 " ctl "\n v_mov_b32_dpp %4, %0 " ctl "\n v_mov_b32_dpp %5, %1 " ctl "\n v_mov_b32_dpp %6, %2 " ctl "\n v_mov_b32_dpp %7, %3 " ctl "\n "

#define KERNEL(NAME, BODY)                                                                                                   \
    __global__ __launch_bounds__(64) void NAME(unsigned long long *out, int iters) {                                         \
        unsigned a0 = threadIdx.x + 100, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int i = 0; i < iters; i++)                                                                                      \
            asm volatile(REP8(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"vcc", "s40", "s41", "s42", "s43"); \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = 1;                                                    \
    }
#define CTL_SHR "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define CTL_XOR1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
KERNEL(k_add_u32, I3("v_add_u32"))
KERNEL(k_sub_u32, I3("v_sub_u32"))
KERNEL(k_and, I3("v_and_b32"))
KERNEL(k_or, I3("v_or_b32"))
KERNEL(k_xor, I3("v_xor_b32"))
KERNEL(k_mov, I2("v_mov_b32"))
KERNEL(k_lshlrev, ISH("v_lshlrev_b32"))
KERNEL(k_lshrrev, ISH("v_lshrrev_b32"))
KERNEL(k_ashrrev, ISH("v_ashrrev_i32"))
KERNEL(k_max_u32, I3("v_max_u32"))
KERNEL(k_min_u32, I3("v_min_u32"))
KERNEL(k_max_i32, I3("v_max_i32"))
KERNEL(k_max_f32, I3("v_max_f32"))
KERNEL(k_min_f32, I3("v_min_f32"))
KERNEL(k_add_f32, I3("v_add_f32"))
KERNEL(k_mul_f32, I3("v_mul_f32"))
KERNEL(k_fma_f32, I4("v_fma_f32"))
KERNEL(k_max3_f32, I4("v_max3_f32"))
KERNEL(k_min3_f32, I4("v_min3_f32"))
KERNEL(k_med3_f32, I4("v_med3_f32"))
KERNEL(k_max3_u32, I4("v_max3_u32"))
KERNEL(k_min3_u32, I4("v_min3_u32"))
KERNEL(k_med3_i32, I4("v_med3_i32"))
KERNEL(k_add3_u32, I4("v_add3_u32"))
KERNEL(k_or3, I4("v_or3_b32"))
KERNEL(k_and_or, I4("v_and_or_b32"))
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %1, 3, %2\n v_lshl_or_b32 %3, %4, 3, %5\n v_lshl_or_b32 %6, %7, 3, %0\n v_lshl_or_b32 %1, %2, 3, %3\n v_lshl_or_b32 %4, %5, 3, %6\n v_lshl_or_b32 %7, %0, 3, %1\n v_lshl_or_b32 %2, %3, 3, %4\n v_lshl_or_b32 %5, %6, 3, %7\n ")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 2, %2\n v_lshl_add_u32 %3, %4, 2, %5\n v_lshl_add_u32 %6, %7, 2, %0\n v_lshl_add_u32 %1, %2, 2, %3\n v_lshl_add_u32 %4, %5, 2, %6\n v_lshl_add_u32 %7, %0, 2, %1\n v_lshl_add_u32 %2, %3, 2, %4\n v_lshl_add_u32 %5, %6, 2, %7\n ")
KERNEL(k_xad, I4("v_xad_u32"))
KERNEL(k_alignbit, I4("v_alignbit_b32"))
KERNEL(k_perm, I4("v_perm_b32"))
KERNEL(k_bfe, "v_bfe_u32 %0, %1, 3, 5\n v_bfe_u32 %2, %3, 3, 5\n v_bfe_u32 %4, %5, 3, 5\n v_bfe_u32 %6, %7, 3, 5\n v_bfe_u32 %1, %0, 3, 5\n v_bfe_u32 %3, %2, 3, 5\n v_bfe_u32 %5, %4, 3, 5\n v_bfe_u32 %7, %6, 3, 5\n ")
KERNEL(k_ffbl, I2("v_ffbl_b32"))
KERNEL(k_ffbh, I2("v_ffbh_u32"))
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %1, 1, %2\n v_mad_u32_u24 %3, %4, 1, %5\n v_mad_u32_u24 %6, %7, 1, %0\n v_mad_u32_u24 %1, %2, 1, %3\n v_mad_u32_u24 %4, %5, 1, %6\n v_mad_u32_u24 %7, %0, 1, %1\n v_mad_u32_u24 %2, %3, 1, %4\n v_mad_u32_u24 %5, %6, 1, %7\n ")
KERNEL(k_mul_u24, I3("v_mul_u32_u24"))
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %1, %2\n v_add_co_u32 %3, vcc, %4, %5\n v_add_co_u32 %6, vcc, %7, %0\n v_add_co_u32 %1, vcc, %2, %3\n v_add_co_u32 %4, vcc, %5, %6\n v_add_co_u32 %7, vcc, %0, %1\n v_add_co_u32 %2, vcc, %3, %4\n v_add_co_u32 %5, vcc, %6, %7\n ")
KERNEL(k_addc_co, "v_addc_co_u32 %0, vcc, %1, %2, vcc\n v_addc_co_u32 %3, vcc, %4, %5, vcc\n v_addc_co_u32 %6, vcc, %7, %0, vcc\n v_addc_co_u32 %1, vcc, %2, %3, vcc\n v_addc_co_u32 %4, vcc, %5, %6, vcc\n v_addc_co_u32 %7, vcc, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %3, %4, vcc\n v_addc_co_u32 %5, vcc, %6, %7, vcc\n ")
KERNEL(k_cmp_lt_u32_e32, ICMP32("v_cmp_lt_u32"))
KERNEL(k_cmp_lt_u32_e64, ICMP64("v_cmp_lt_u32_e64"))
KERNEL(k_cmp_lt_f32_e32, ICMP32("v_cmp_lt_f32"))
KERNEL(k_cmp_cnd_vcc, "v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_lt_u32 vcc, %5, %6\n v_cndmask_b32 %7, %1, %2, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %5, %6, %0, vcc\n v_cmp_lt_u32 vcc, %7, %1\n v_cndmask_b32 %2, %3, %4, vcc\n ")
KERNEL(k_cmpf_cnd_vcc, "v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_lt_f32 vcc, %5, %6\n v_cndmask_b32 %7, %1, %2, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %5, %6, %0, vcc\n v_cmp_lt_f32 vcc, %7, %1\n v_cndmask_b32 %2, %3, %4, vcc\n ")
KERNEL(k_cnd_e64_sgpr, "v_cndmask_b32_e64 %0, %1, %2, s[40:41]\n v_cndmask_b32_e64 %3, %4, %5, s[42:43]\n v_cndmask_b32_e64 %6, %7, %0, s[40:41]\n v_cndmask_b32_e64 %1, %2, %3, s[42:43]\n v_cndmask_b32_e64 %4, %5, %6, s[40:41]\n v_cndmask_b32_e64 %7, %0, %1, s[42:43]\n v_cndmask_b32_e64 %2, %3, %4, s[40:41]\n v_cndmask_b32_e64 %5, %6, %7, s[42:43]\n ")
KERNEL(k_mov_dpp_shr, IMOVDPP(CTL_SHR))
KERNEL(k_max_i32_dpp, IDPP("v_max_i32_dpp", CTL_XOR1))
KERNEL(k_max_f32_dpp, IDPP("v_max_f32_dpp", CTL_XOR1))
KERNEL(k_add_u32_dpp, IDPP("v_add_u32_dpp", CTL_XOR1))
KERNEL(k_pk_max_u16, I3("v_pk_max_u16"))
KERNEL(k_pk_add_u16, I3("v_pk_add_u16"))
KERNEL(k_pk_min_u16, I3("v_pk_min_u16"))
KERNEL(k_pk_sub_u16, I3("v_pk_sub_u16"))
KERNEL(k_pk_max_f16, I3("v_pk_max_f16"))
KERNEL(k_pk_add_f16, I3("v_pk_add_f16"))
KERNEL(k_mix_add_max, "v_add_u32 %0, %1, %2\n v_max_u32 %3, %4, %5\n v_add_u32 %6, %7, %0\n v_max_u32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_max_u32 %7, %0, %1\n v_add_u32 %2, %3, %4\n v_max_u32 %5, %6, %7\n ")
KERNEL(k_mix_add_maxf, "v_add_u32 %0, %1, %2\n v_max_f32 %3, %4, %5\n v_add_u32 %6, %7, %0\n v_max_f32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_max_f32 %7, %0, %1\n v_add_u32 %2, %3, %4\n v_max_f32 %5, %6, %7\n ")
KERNEL(k_mix3_add_max, "v_add_u32 %0, %1, %2\n v_sub_u32 %3, %4, %5\n v_and_b32 %6, %7, %0\n v_max_u32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_xor_b32 %7, %0, %1\n v_or_b32 %2, %3, %4\n v_max_u32 %5, %6, %7\n ")

struct Ent { const char *name; void (*fn)(unsigned long long *, int); };
#define E(n) {#n, n}
static Ent ents[] = {E(k_add_u32), E(k_sub_u32), E(k_and), E(k_or), E(k_xor), E(k_mov), E(k_lshlrev), E(k_lshrrev), E(k_ashrrev),
    E(k_max_u32), E(k_min_u32), E(k_max_i32), E(k_max_f32), E(k_min_f32), E(k_add_f32), E(k_mul_f32), E(k_fma_f32), E(k_max3_f32), E(k_min3_f32),
    E(k_med3_f32), E(k_max3_u32), E(k_min3_u32), E(k_med3_i32), E(k_add3_u32), E(k_or3), E(k_and_or), E(k_lshl_or), E(k_lshl_add), E(k_xad),
    E(k_alignbit), E(k_perm), E(k_bfe), E(k_ffbl), E(k_ffbh), E(k_mad_u24), E(k_mul_u24), E(k_add_co), E(k_addc_co), E(k_cmp_lt_u32_e32),
    E(k_cmp_lt_u32_e64), E(k_cmp_lt_f32_e32), E(k_cmp_cnd_vcc), E(k_cmpf_cnd_vcc), E(k_cnd_e64_sgpr), E(k_mov_dpp_shr), E(k_max_i32_dpp),
    E(k_max_f32_dpp), E(k_add_u32_dpp), E(k_pk_max_u16), E(k_pk_add_u16), E(k_pk_min_u16), E(k_pk_sub_u16), E(k_pk_max_f16), E(k_pk_add_f16),
    E(k_mix_add_max), E(k_mix_add_maxf), E(k_mix3_add_max)};

int main(int argc, char **argv) {
    const int iters = 600;
    double    base[9] = {0};
    for (int w : {2, 4, 8}) {
        for (const Ent &e : ents) {
            if (argc > 1 && !strstr(e.name, argv[1])) continue;
            const int blocks = 256 * 4 * w;
            unsigned long long *d;
            CHK(hipMalloc(&d, blocks * 8));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d, iters);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(64), 0, 0, d, iters);
            CHK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(blocks);
            CHK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[blocks / 2], per = med / ((double)iters * 64);
            if (&e == &ents[0]) base[w] = per / w;
            printf("%-20s waves/SIMD %d  ticks/instr/wave %7.3f  SIMD ticks/instr %6.3f  vs v_add_u32 %5.2f\n", e.name + 2, w, per, per / w,
                   base[w] > 0 ? per / w / base[w] : 0.0);
            fflush(stdout);
            CHK(hipFree(d));
        }
    }
    return 0;
}
