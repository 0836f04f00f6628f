// Probe of the DPP row operations the register-window kernel relies on (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned row_prev(unsigned cur, unsigned prev_tile) {
    int t = __builtin_amdgcn_update_dpp(0, (int)prev_tile, 0x121, 0xf, 0xf, false);  // row_ror:1
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)cur, 0x111, 0xf, 0xf, false);  // row_shr:1
}
__device__ __forceinline__ unsigned row_next(unsigned cur, unsigned next_tile) {
    int t = __builtin_amdgcn_update_dpp(0, (int)next_tile, 0x12F, 0xf, 0xf, false);  // row_ror:15
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)cur, 0x101, 0xf, 0xf, false);  // row_shl:1
}
__global__ void k(unsigned *out) {
    unsigned lane = threadIdx.x;
    unsigned cur = 1000 + lane, other = 2000 + lane;
    out[lane]       = row_prev(cur, other);
    out[64 + lane]  = row_next(cur, other);
    int v = (int)((lane * 7 + 3) % 23);
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    out[128 + lane] = (unsigned)v;
    unsigned long long b = __ballot((lane % 3) == 0);
    out[192 + lane] = (unsigned)((b >> (16 * (lane >> 4))) & 0xFFFFu);
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        unsigned ep = (l % 16 == 0) ? 2000 + l + 15 : 1000 + l - 1;
        unsigned en = (l % 16 == 15) ? 2000 + l - 15 : 1000 + l + 1;
        int mn = 99;
        for (int j = (l / 16) * 16; j < (l / 16) * 16 + 16; j++) mn = mn < (j * 7 + 3) % 23 ? mn : (j * 7 + 3) % 23;
        unsigned eb = 0;
        for (int j = 0; j < 16; j++) if (((l / 16) * 16 + j) % 3 == 0) eb |= 1u << j;
        if (h[l] != ep || h[64 + l] != en || h[128 + l] != (unsigned)mn || h[192 + l] != eb) {
            bad++;
            printf("lane %d prev %u (want %u) next %u (want %u) min %u (want %d) bits %x (want %x)\n", l, h[l], ep, h[64 + l], en, h[128 + l], mn, h[192+l], eb);
        }
    }
    printf(bad ? "DPP probe: %d lanes wrong\n" : "DPP probe ok\n", bad);
    return bad != 0;
}
