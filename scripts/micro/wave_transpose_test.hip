// unit check of wave_reduce_transpose / lane_xor on the device (run on a gfx950 box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../voldor_amd/csrc/vk_device.hpp"
using namespace vk;
__global__ void kt(float* out32, float* out8, float* x4) {
    const int lane = threadIdx.x;
    float v[32], w[8];
    for (int k = 0; k < 32; k++) v[k] = (float)((lane * 37 + k * 11) % 97) * 0.25f;
    for (int k = 0; k < 8; k++) w[k] = (float)((lane * 13 + k * 7) % 31);
    float r = wave_reduce_transpose<32>(v);
    out32[lane] = r; out32[64 + lane] = (float)wave_slot<32>(lane);
    float q = wave_reduce_transpose<8>(w);
    out8[lane] = q; out8[64 + lane] = (float)wave_slot<8>(lane);
    x4[lane] = lane_xor<4>((float)lane); x4[64 + lane] = lane_xor<8>((float)lane); x4[128 + lane] = lane_xor<16>((float)lane);
}
int main() {
    float *a, *b, *c; hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&c, 768);
    kt<<<1, 64>>>(a, b, c);
    float ha[128], hb[128], hc[192];
    hipMemcpy(ha, a, 512, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 512, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 768, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; L++) {
        if ((int)hc[L] != (L ^ 4) || (int)hc[64 + L] != (L ^ 8) || (int)hc[128 + L] != (L ^ 16)) { bad++; printf("xor lane %d: %g %g %g\n", L, hc[L], hc[64 + L], hc[128 + L]); }
        int k = (int)ha[64 + L]; double s = 0; for (int l = 0; l < 64; l++) s += (float)((l * 37 + k * 11) % 97) * 0.25f;
        if (fabs(s - ha[L]) > 1e-3) { bad++; printf("P32 lane %d slot %d got %g want %g\n", L, k, ha[L], s); }
        k = (int)hb[64 + L]; s = 0; for (int l = 0; l < 64; l++) s += (float)((l * 13 + k * 7) % 31);
        if (fabs(s - hb[L]) > 1e-3) { bad++; printf("P8 lane %d slot %d got %g want %g\n", L, k, hb[L], s); }
    }
    printf("wave_transpose_test: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
