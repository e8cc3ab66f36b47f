// micro-benchmark: does a wave64 with only its low 32 (16) lanes enabled issue a VALU instruction in fewer cycles on gfx950?
// One wave per workgroup, a dependent chain and 8 independent chains of v_fma_f32 / v_fma_f64, EXEC = low K lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, int active) {
    float a[8]; double b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i]; }
    const float m = 1.0001f, c = 0.5f;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = clock64();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);                       // 8 independent fp32 chains
                if (MODE == 1) a[0] = __builtin_fmaf(a[0], m, c);                       // one dependent fp32 chain
                if (MODE == 2) b[i] = __builtin_fma(b[i], (double)m, (double)c);        // 8 independent fp64 chains
                if (MODE == 3) b[0] = __builtin_fma(b[0], (double)m, (double)c);        // one dependent fp64 chain
            }
        }
        t1 = clock64();
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    const int iters = 4096;
    const char* names[4] = { "fp32 x8 independent", "fp32 dependent chain", "fp64 x8 independent", "fp64 dependent chain" };
    for (int active : {64, 32, 16, 1}) {
        long long h[4];
        k<0><<<1, 64>>>(out, cyc, iters, active); hipMemcpy(&h[0], cyc, 8, hipMemcpyDeviceToHost);
        k<1><<<1, 64>>>(out, cyc, iters, active); hipMemcpy(&h[1], cyc, 8, hipMemcpyDeviceToHost);
        k<2><<<1, 64>>>(out, cyc, iters, active); hipMemcpy(&h[2], cyc, 8, hipMemcpyDeviceToHost);
        k<3><<<1, 64>>>(out, cyc, iters, active); hipMemcpy(&h[3], cyc, 8, hipMemcpyDeviceToHost);
        printf("active lanes %2d:", active);
        for (int i = 0; i < 4; i++) printf("  %s %.2f", names[i], h[i] / (8.0 * iters));
        printf("  (clock64 ticks per instruction; clock64 runs at 100 MHz on gfx950: x24 for 2.4 GHz cycles)\n");
    }
    return 0;
}
