// micro-benchmark: cost of a chain of dependent tiny kernels in one stream (launch floor of the per-window kernel chain)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1; }
__global__ void k_spin(int* p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1;
}
static void spin_test(hipStream_t s, int* d) {
    // kernels long enough (~8 us) for the host to run ahead: what is left is the GPU-side dispatch gap
    const long long cyc = 20000; const int n = 300;
    for (int rep = 0; rep < 2; rep++) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc);
        hipStreamSynchronize(s);
        auto t2 = std::chrono::high_resolution_clock::now();
        printf("stream: %d spin kernels of %lld cycles: %.2f us/kernel\n", n, cyc, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 4; i++) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t2 = std::chrono::high_resolution_clock::now();
    printf("graph : %d spin kernels of %lld cycles: %.2f us/kernel\n", n, cyc, std::chrono::duration<double, std::micro>(t2 - t0).count() / (4 * n));
}
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int blocks : {1, 1200}) {
        for (int rep = 0; rep < 2; rep++) {
            const int n = 3000;
            hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, d);
            auto t1 = std::chrono::high_resolution_clock::now();
            hipStreamSynchronize(s);
            auto t2 = std::chrono::high_resolution_clock::now();
            printf("blocks %4d: enqueue %.2f us/launch, total %.2f us/launch\n", blocks,
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
        }
    }
    spin_test(s, d);
    // same chain as a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 300; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, d);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 10; i++) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t2 = std::chrono::high_resolution_clock::now();
    printf("graph of 300 launches: %.2f us/kernel\n", std::chrono::duration<double, std::micro>(t2 - t0).count() / 3000);
    return 0;
}
