// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul/add on gfx950, 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i].x = a[i]; p[i].y = a[i] + 1; }
    const float m = 1.0001f, c = 0.5f; v2f mm; mm.x = m; mm.y = m; v2f cc; cc.x = c; cc.y = c;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], mm, cc);
            if (MODE == 2) { p[i] = p[i] * mm; }
            if (MODE == 3) { a[i] = a[i] * m; }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    const int iters = 4096;
    for (int threads : {64, 256, 512, 1024}) {
        long long h[4];
        k<0><<<1, threads>>>(out, cyc, iters); hipMemcpy(&h[0], cyc, 8, hipMemcpyDeviceToHost);
        k<1><<<1, threads>>>(out, cyc, iters); hipMemcpy(&h[1], cyc, 8, hipMemcpyDeviceToHost);
        k<2><<<1, threads>>>(out, cyc, iters); hipMemcpy(&h[2], cyc, 8, hipMemcpyDeviceToHost);
        k<3><<<1, threads>>>(out, cyc, iters); hipMemcpy(&h[3], cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %4d (waves/SIMD %.2f): cycles per instr (wave 0 view): v_fma_f32 %.2f  v_pk_fma_f32 %.2f  v_pk_mul_f32 %.2f  v_mul_f32 %.2f\n",
               threads, threads / 256.0, h[0] / (8.0 * iters), h[1] / (8.0 * iters), h[2] / (8.0 * iters), h[3] / (8.0 * iters));
    }
    return 0;
}
