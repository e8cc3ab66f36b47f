// micro-benchmark (round 5): what a cross-stream dependency costs on the critical path of a stream of ~8 us kernels, and what a cooperative launch costs.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/event_cost.hip -o scripts/micro/event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <chrono>
__global__ void k_spin(int* p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1;
}
template <class F> static double timed(hipStream_t s, hipStream_t s2, int n, F f) {
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        hipStreamSynchronize(s); hipStreamSynchronize(s2);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < n; i++) f(i);
        hipStreamSynchronize(s); hipStreamSynchronize(s2);
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n);
    }
    return best;
}
int main() {
    int* d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const long long cyc = 800;  // clock64 ticks at 100 MHz: ~8 us
    const int n = 200;
    hipEvent_t ea[2], eb[2], ec[2];
    for (int i = 0; i < 2; i++) { hipEventCreateWithFlags(&ea[i], hipEventDisableTiming); hipEventCreateWithFlags(&eb[i], hipEventDisableTiming | hipEventReleaseToDevice); hipEventCreate(&ec[i]); }
    const double base = timed(s, s2, n, [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc); });
    printf("chain of spin kernels:                                  %.2f us / kernel\n", base);
    double t;
    t = timed(s, s2, n, [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc); hipEventRecord(ea[0], s); });
    printf("+ hipEventRecord (DisableTiming) after each:            %.2f (+%.2f)\n", t, t - base);
    t = timed(s, s2, n, [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc); hipEventRecord(eb[0], s); });
    printf("+ hipEventRecord (DisableTiming | ReleaseToDevice):     %.2f (+%.2f)\n", t, t - base);
    t = timed(s, s2, n, [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc); hipEventRecord(ec[0], s); });
    printf("+ hipEventRecord (default, timing):                     %.2f (+%.2f)\n", t, t - base);
    for (int flavour = 0; flavour < 2; flavour++) {
        hipEvent_t* ev = flavour ? eb : ea;
        // fork / join per step: main: K, record e0; side: wait e0, Kside (half as long), record e1; main: K, wait e1, K   -> 3 main kernels per step
        t = timed(s, s2, n / 2, [&](int) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc); hipEventRecord(ev[0], s);
            hipStreamWaitEvent(s2, ev[0], 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, d + 1, cyc / 2); hipEventRecord(ev[1], s2);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc);
            hipStreamWaitEvent(s, ev[1], 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc);
        });
        printf("fork/join per 3 main kernels (%s): %.2f us / step, overhead %.2f us / step\n", flavour ? "ReleaseToDevice" : "DisableTiming  ", t, t - 3 * base);
    }
    // cooperative launch in a chain of normal launches
    {
        void* args[] = { (void*)&d, (void*)&cyc };
        t = timed(s, s2, n, [&](int) { hipLaunchCooperativeKernel((const void*)k_spin, dim3(1), dim3(64), args, 0, s); });
        printf("chain of cooperative launches:                          %.2f (+%.2f)\n", t, t - base);
        t = timed(s, s2, n / 2, [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc); hipLaunchCooperativeKernel((const void*)k_spin, dim3(1), dim3(64), args, 0, s); });
        printf("normal + cooperative alternating (per pair):            %.2f (+%.2f)\n", t, t - 2 * base);
        long long c256 = cyc;
        t = timed(s, s2, n / 2, [&](int) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, s, d, c256); });
        printf("256 x 512-thread spin kernels, normal:                  %.2f\n", t);
        t = timed(s, s2, n / 2, [&](int) { hipLaunchCooperativeKernel((const void*)k_spin, dim3(256), dim3(512), args, 0, s); });
        printf("256 x 512-thread spin kernels, cooperative:             %.2f\n", t);
    }
    return 0;
}
