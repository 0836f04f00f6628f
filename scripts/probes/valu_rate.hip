// Probe: VALU pipe cycles per wave64 instruction on gfx950 for the integer / DPP forms the alignment kernels use.
// 8 waves per SIMD, each issuing long runs of independent instructions; reports SIMD cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#define BODY(ins) asm volatile(REP16(ins) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"vcc", "s40", "s41", "s42", "s43")
        if (OP == 0) BODY("v_add_u32 %0, %1, %2\n v_add_u32 %3, %4, %5\n v_add_u32 %6, %7, %0\n v_add_u32 %1, %2, %3\n v_add_u32 %4, %5, %6\n v_add_u32 %7, %0, %1\n v_add_u32 %2, %3, %4\n v_add_u32 %5, %6, %7\n");
        if (OP == 1) BODY("v_max_u32 %0, %1, %2\n v_max_u32 %3, %4, %5\n v_max_u32 %6, %7, %0\n v_max_u32 %1, %2, %3\n v_max_u32 %4, %5, %6\n v_max_u32 %7, %0, %1\n v_max_u32 %2, %3, %4\n v_max_u32 %5, %6, %7\n");
        if (OP == 2) BODY("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %6, %7, %0, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %7, %0, %1, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n");
        if (OP == 3) BODY("v_cndmask_b32_e64 %0, %1, %2, s[40:41]\n v_cndmask_b32_e64 %3, %4, %5, s[42:43]\n v_cndmask_b32_e64 %6, %7, %0, s[40:41]\n v_cndmask_b32_e64 %1, %2, %3, s[42:43]\n v_cndmask_b32_e64 %4, %5, %6, s[40:41]\n v_cndmask_b32_e64 %7, %0, %1, s[42:43]\n v_cndmask_b32_e64 %2, %3, %4, s[40:41]\n v_cndmask_b32_e64 %5, %6, %7, s[42:43]\n");
        if (OP == 4) BODY("v_cmp_lt_u32_e64 s[40:41], %1, %2\n v_cmp_lt_u32_e64 s[42:43], %4, %5\n v_cmp_lt_u32_e64 s[40:41], %7, %0\n v_cmp_lt_u32_e64 s[42:43], %2, %3\n v_cmp_lt_u32_e64 s[40:41], %5, %6\n v_cmp_lt_u32_e64 s[42:43], %0, %1\n v_cmp_lt_u32_e64 s[40:41], %3, %4\n v_cmp_lt_u32_e64 s[42:43], %6, %7\n");
        if (OP == 5) BODY("v_min3_u32 %0, %1, %2, %3\n v_min3_u32 %3, %4, %5, %6\n v_min3_u32 %6, %7, %0, %1\n v_min3_u32 %1, %2, %3, %4\n v_min3_u32 %4, %5, %6, %7\n v_min3_u32 %7, %0, %1, %2\n v_min3_u32 %2, %3, %4, %5\n v_min3_u32 %5, %6, %7, %0\n");
        if (OP == 6) BODY("v_lshl_or_b32 %0, %1, 3, %2\n v_lshl_or_b32 %3, %4, 3, %5\n v_lshl_or_b32 %6, %7, 3, %0\n v_lshl_or_b32 %1, %2, 3, %3\n v_lshl_or_b32 %4, %5, 3, %6\n v_lshl_or_b32 %7, %0, 3, %1\n v_lshl_or_b32 %2, %3, 3, %4\n v_lshl_or_b32 %5, %6, 3, %7\n");
        if (OP == 7) BODY("v_alignbit_b32 %0, %1, %2, %3\n v_alignbit_b32 %3, %4, %5, %6\n v_alignbit_b32 %6, %7, %0, %1\n v_alignbit_b32 %1, %2, %3, %4\n v_alignbit_b32 %4, %5, %6, %7\n v_alignbit_b32 %7, %0, %1, %2\n v_alignbit_b32 %2, %3, %4, %5\n v_alignbit_b32 %5, %6, %7, %0\n");
        if (OP == 8) BODY("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n");
        if (OP == 9) BODY("v_max_i32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %4, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %6, %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n");
        if (OP == 10) BODY("v_addc_co_u32 %0, vcc, %1, %2, vcc\n v_addc_co_u32 %3, vcc, %4, %5, vcc\n v_addc_co_u32 %6, vcc, %7, %0, vcc\n v_addc_co_u32 %1, vcc, %2, %3, vcc\n v_addc_co_u32 %4, vcc, %5, %6, vcc\n v_addc_co_u32 %7, vcc, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %3, %4, vcc\n v_addc_co_u32 %5, vcc, %6, %7, vcc\n");
        if (OP == 11) BODY("v_ffbl_b32 %0, %1\n v_ffbl_b32 %2, %3\n v_ffbl_b32 %4, %5\n v_ffbl_b32 %6, %7\n v_ffbl_b32 %1, %0\n v_ffbl_b32 %3, %2\n v_ffbl_b32 %5, %4\n v_ffbl_b32 %7, %6\n");
        if (OP == 12) BODY("v_add3_u32 %0, %1, %2, %3\n v_add3_u32 %3, %4, %5, %6\n v_add3_u32 %6, %7, %0, %1\n v_add3_u32 %1, %2, %3, %4\n v_add3_u32 %4, %5, %6, %7\n v_add3_u32 %7, %0, %1, %2\n v_add3_u32 %2, %3, %4, %5\n v_add3_u32 %5, %6, %7, %0\n");
        if (OP == 13) BODY("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %3, %4, %5, %6\n v_fma_f32 %6, %7, %0, %1\n v_fma_f32 %1, %2, %3, %4\n v_fma_f32 %4, %5, %6, %7\n v_fma_f32 %7, %0, %1, %2\n v_fma_f32 %2, %3, %4, %5\n v_fma_f32 %5, %6, %7, %0\n");
        if (OP == 14) BODY("v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %4, 3, %5\n v_lshlrev_b32 %6, 3, %7\n v_lshlrev_b32 %1, 3, %0\n v_lshlrev_b32 %3, 3, %2\n v_lshlrev_b32 %5, 3, %4\n v_lshlrev_b32 %7, 3, %6\n");
        if (OP == 15) BODY("v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %7, %0\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %6, %7\n");
        if (OP == 16) BODY("v_add_f32 %0, %1, %2\n v_add_f32 %3, %4, %5\n v_add_f32 %6, %7, %0\n v_add_f32 %1, %2, %3\n v_add_f32 %4, %5, %6\n v_add_f32 %7, %0, %1\n v_add_f32 %2, %3, %4\n v_add_f32 %5, %6, %7\n");
        if (OP == 17) BODY("v_med3_i32 %0, %1, %2, %3\n v_med3_i32 %3, %4, %5, %6\n v_med3_i32 %6, %7, %0, %1\n v_med3_i32 %1, %2, %3, %4\n v_med3_i32 %4, %5, %6, %7\n v_med3_i32 %7, %0, %1, %2\n v_med3_i32 %2, %3, %4, %5\n v_med3_i32 %5, %6, %7, %0\n");
        if (OP == 18) BODY("v_sub_u32 %0, %1, %2\n v_xor_b32 %3, %4, %5\n v_and_b32 %6, %7, %0\n v_or_b32 %1, %2, %3\n v_min_u32 %4, %5, %6\n v_ashrrev_i32 %7, 4, %1\n v_lshrrev_b32 %2, 1, %4\n v_max_i32 %5, %6, %7\n");

        if (OP == 19) BODY("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_lt_u32 vcc, %5, %6\n v_cndmask_b32 %7, %1, %2, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %5, %6, %0, vcc\n v_cmp_lt_u32 vcc, %7, %1\n v_cndmask_b32 %2, %3, %4, vcc\n");
        if (OP == 20) BODY("v_cmp_lt_u32_e64 s[40:41], %1, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %3, %4, s[40:41]\n v_cmp_lt_u32_e64 s[42:43], %5, %6\n s_nop 1\n v_cndmask_b32_e64 %7, %1, %2, s[42:43]\n v_cmp_lt_u32_e64 s[40:41], %3, %4\n s_nop 1\n v_cndmask_b32_e64 %5, %6, %0, s[40:41]\n v_cmp_lt_u32_e64 s[42:43], %7, %1\n s_nop 1\n v_cndmask_b32_e64 %2, %3, %4, s[42:43]\n");
        if (OP == 21) BODY("v_cmp_lt_u32_e64 s[40:41], %1, %2\n v_cmp_lt_u32_e64 s[42:43], %5, %6\n s_nop 0\n v_cndmask_b32_e64 %0, %3, %4, s[40:41]\n v_cndmask_b32_e64 %7, %1, %2, s[42:43]\n v_cmp_lt_u32_e64 s[40:41], %3, %4\n v_cmp_lt_u32_e64 s[42:43], %7, %1\n s_nop 0\n v_cndmask_b32_e64 %5, %6, %0, s[40:41]\n v_cndmask_b32_e64 %2, %3, %4, s[42:43]\n");
        if (OP == 22) BODY("v_cndmask_b32_e64 %0, %1, %2, vcc\n v_cndmask_b32_e64 %3, %4, %5, vcc\n v_cndmask_b32_e64 %6, %7, %0, vcc\n v_cndmask_b32_e64 %1, %2, %3, vcc\n v_cndmask_b32_e64 %4, %5, %6, vcc\n v_cndmask_b32_e64 %7, %0, %1, vcc\n v_cndmask_b32_e64 %2, %3, %4, vcc\n v_cndmask_b32_e64 %5, %6, %7, vcc\n");
        if (OP == 23) BODY("v_sub_u32 %0, %1, %2\n v_sub_u32 %3, %4, %5\n v_sub_u32 %6, %7, %0\n v_sub_u32 %1, %2, %3\n v_sub_u32 %4, %5, %6\n v_sub_u32 %7, %0, %1\n v_sub_u32 %2, %3, %4\n v_sub_u32 %5, %6, %7\n");
        if (OP == 24) BODY("v_and_b32 %0, %1, %2\n v_and_b32 %3, %4, %5\n v_and_b32 %6, %7, %0\n v_and_b32 %1, %2, %3\n v_and_b32 %4, %5, %6\n v_and_b32 %7, %0, %1\n v_and_b32 %2, %3, %4\n v_and_b32 %5, %6, %7\n");
        if (OP == 25) BODY("v_mad_u32_u24 %0, %1, 1, %2\n v_mad_u32_u24 %3, %4, 1, %5\n v_mad_u32_u24 %6, %7, 1, %0\n v_mad_u32_u24 %1, %2, 1, %3\n v_mad_u32_u24 %4, %5, 1, %6\n v_mad_u32_u24 %7, %0, 1, %1\n v_mad_u32_u24 %2, %3, 1, %4\n v_mad_u32_u24 %5, %6, 1, %7\n");
        if (OP == 26) BODY("v_pk_max_u16 %0, %1, %2\n v_pk_max_u16 %3, %4, %5\n v_pk_max_u16 %6, %7, %0\n v_pk_max_u16 %1, %2, %3\n v_pk_max_u16 %4, %5, %6\n v_pk_max_u16 %7, %0, %1\n v_pk_max_u16 %2, %3, %4\n v_pk_max_u16 %5, %6, %7\n");
        if (OP == 27) BODY("v_pk_add_u16 %0, %1, %2\n v_pk_add_u16 %3, %4, %5\n v_pk_add_u16 %6, %7, %0\n v_pk_add_u16 %1, %2, %3\n v_pk_add_u16 %4, %5, %6\n v_pk_add_u16 %7, %0, %1\n v_pk_add_u16 %2, %3, %4\n v_pk_add_u16 %5, %6, %7\n");
        if (OP == 28) BODY("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n v_mov_b32 %1, %0\n v_mov_b32 %3, %2\n v_mov_b32 %5, %4\n v_mov_b32 %7, %6\n");
        if (OP == 29) BODY("v_bfe_u32 %0, %1, 3, 5\n v_bfe_u32 %2, %3, 3, 5\n v_bfe_u32 %4, %5, 3, 5\n v_bfe_u32 %6, %7, 3, 5\n v_bfe_u32 %1, %0, 3, 5\n v_bfe_u32 %3, %2, 3, 5\n v_bfe_u32 %5, %4, 3, 5\n v_bfe_u32 %7, %6, 3, 5\n");
        if (OP == 30) BODY("s_and_b64 s[40:41], s[40:41], s[42:43]\n s_or_b64 s[42:43], s[40:41], s[42:43]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n s_or_b64 s[42:43], s[40:41], s[42:43]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n s_or_b64 s[42:43], s[40:41], s[42:43]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n s_or_b64 s[42:43], s[40:41], s[42:43]\n");
        if (OP == 31) BODY("v_add_u32 %0, %1, %2\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_add_u32 %3, %4, %5\n s_or_b64 s[42:43], s[40:41], s[42:43]\n v_add_u32 %6, %7, %0\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_add_u32 %1, %2, %3\n s_or_b64 s[42:43], s[40:41], s[42:43]\n");
        if (OP == 32) BODY("v_max_u32 %0, %1, %2\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_max_u32 %3, %4, %5\n s_or_b64 s[42:43], s[40:41], s[42:43]\n v_max_u32 %6, %7, %0\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_max_u32 %1, %2, %3\n s_or_b64 s[42:43], s[40:41], s[42:43]\n");
        if (OP == 33) BODY("v_pk_min_u16 %0, %1, %2\n v_pk_sub_u16 %3, %4, %5\n v_pk_lshrrev_b16 %6, 15, %0\n v_pk_max_u16 %1, %2, %3\n v_pk_add_u16 %4, %5, %6\n v_pk_min_u16 %7, %0, %1\n v_pk_sub_u16 %2, %3, %4\n v_pk_lshrrev_b16 %5, 15, %7\n");
        if (OP == 34) BODY("v_max_u32 %0, %1, %2\n v_add_u32 %3, %4, %5\n v_max_u32 %6, %7, %0\n v_add_u32 %1, %2, %3\n v_max_u32 %4, %5, %6\n v_add_u32 %7, %0, %1\n v_max_u32 %2, %3, %4\n v_add_u32 %5, %6, %7\n");
        if (OP == 35) BODY("v_add_co_u32 %0, vcc, %1, %2\n v_add_co_u32 %3, vcc, %4, %5\n v_add_co_u32 %6, vcc, %7, %0\n v_add_co_u32 %1, vcc, %2, %3\n v_add_co_u32 %4, vcc, %5, %6\n v_add_co_u32 %7, vcc, %0, %1\n v_add_co_u32 %2, vcc, %3, %4\n v_add_co_u32 %5, vcc, %6, %7\n");
        if (OP == 36) BODY("v_max_i32 %0, %1, %2\n v_max_i32 %3, %4, %5\n v_min_i32 %6, %7, %0\n v_min_i32 %1, %2, %3\n v_max_i32 %4, %5, %6\n v_max_i32 %7, %0, %1\n v_min_i32 %2, %3, %4\n v_min_i32 %5, %6, %7\n");
        if (OP == 37) BODY("v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %3, %4, %5, %6\n v_perm_b32 %6, %7, %0, %1\n v_perm_b32 %1, %2, %3, %4\n v_perm_b32 %4, %5, %6, %7\n v_perm_b32 %7, %0, %1, %2\n v_perm_b32 %2, %3, %4, %5\n v_perm_b32 %5, %6, %7, %0\n");
        if (OP == 38) BODY("v_lshl_add_u32 %0, %1, 2, %2\n v_lshl_add_u32 %3, %4, 2, %5\n v_lshl_add_u32 %6, %7, 2, %0\n v_lshl_add_u32 %1, %2, 2, %3\n v_lshl_add_u32 %4, %5, 2, %6\n v_lshl_add_u32 %7, %0, 2, %1\n v_lshl_add_u32 %2, %3, 2, %4\n v_lshl_add_u32 %5, %6, 2, %7\n");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = 1;
}
template <int OP>
void run(const char *name, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * 4 * waves_per_simd;
    unsigned long long *d;
    hipMalloc(&d, blocks * 8);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[blocks / 2], n_ins = (double)iters * 128;
    // s_memtime ticks at 100 MHz (constant); convert with the nominal 2.4 GHz to shader cycles is unreliable, so print both
    printf("%-22s waves/SIMD %d  ticks/instr/wave %.4f  -> SIMD ticks per instr %.4f\n", name, waves_per_simd, med / n_ins,
           med / n_ins / waves_per_simd);
    hipFree(d);
}
#include <algorithm>
int main(int argc, char **argv) {
    int sel = argc > 1 ? atoi(argv[1]) : -1, w = argc > 2 ? atoi(argv[2]) : 8;
    if (sel == 0) run<0>("v_add_u32", w);
    if (sel == 23) run<23>("v_sub_u32", w);
    if (sel == 35) run<35>("v_add_co_u32", w);
    if (sel == 1) run<1>("v_max_u32", w);
    if (sel == 36) run<36>("v_max/min_i32", w);
    if (sel == 24) run<24>("v_and_b32", w);
    if (sel == 28) run<28>("v_mov_b32", w);
    if (sel == 34) run<34>("max+add alternating", w);
    if (sel == 25) run<25>("v_mad_u32_u24", w);
    if (sel == 38) run<38>("v_lshl_add_u32", w);
    if (sel == 29) run<29>("v_bfe_u32", w);
    if (sel == 37) run<37>("v_perm_b32", w);
    if (sel == 26) run<26>("v_pk_max_u16", w);
    if (sel == 27) run<27>("v_pk_add_u16", w);
    if (sel == 33) run<33>("v_pk mixed", w);
    if (sel == 2) run<2>("v_cndmask e32 vcc", w);
    if (sel == 22) run<22>("v_cndmask e64 vcc", w);
    if (sel == 3) run<3>("v_cndmask e64 sgpr", w);
    if (sel == 19) run<19>("cmp+cnd vcc pairs (2 ins)", w);
    if (sel == 20) run<20>("cmp,nop1,cnd sgpr (2 ins+nop)", w);
    if (sel == 21) run<21>("2cmp,nop0,2cnd sgpr", w);
    if (sel == 30) run<30>("s_and/or_b64", w);
    if (sel == 31) run<31>("v_add + s_and interleaved", w);
    if (sel == 32) run<32>("v_max + s_and interleaved", w);
    return 0;
}
