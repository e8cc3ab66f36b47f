// A C++ client of the B-inner boundary, built with plain g++ against include/gpu_kernels.h + include/py_export.h and linked
// with -lvoldor_hip, the way voldor/voldor.cpp and voldor/geometry.cpp link the reference's libgpu-kernels
// (slam_py/install/setup_linux_vo.py:16-27).  Exercises default arguments, bool parameters, host-pointer ownership and the
// int& out-parameter of py_voldor_wrapper.  Prints "CLIENT OK" on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpu_kernels.h"
#include "py_export.h"

int main() {
    // meanshift_gpu with its default arguments (gpu_kernels.h:11-15): a 6-D cluster around (1,2,3,4,5,6) + far outliers
    const int N = 4096, dims = 6;
    std::vector<float> space((size_t)N * dims);
    srand(7);
    for (int i = 0; i < N; i++)
        for (int d = 0; d < dims; d++) {
            const float u = rand() / (float)RAND_MAX - 0.5f;
            space[(size_t)i * dims + d] = (i % 4 == 0) ? 40.f * u : (float)(d + 1) + 0.2f * u;
        }
    float mean[6] = { 1.1f, 2.1f, 2.9f, 4.1f, 4.9f, 6.1f }, conf = 0.f;
    int iters = 0;
    if (meanshift_gpu(space.data(), 0.1f, mean, &conf, &iters, true, N, dims) != 0) { printf("meanshift_gpu failed\n"); return 1; }
    for (int d = 0; d < dims; d++)
        if (std::fabs(mean[d] - (float)(d + 1)) > 0.05f) { printf("mode[%d] = %f\n", d, mean[d]); return 2; }
    if (!(conf > 0.3f) || iters < 1) { printf("confidence %f iters %d\n", conf, iters); return 3; }

    // py_voldor_wrapper (py_export.h:3-11) on a pure forward translation over a fronto-parallel plane
    const int w = 96, h = 72, nf = 2;
    const float fx = 48.f, fy = 48.f, cx = 48.f, cy = 36.f, Z = 8.f, tz = 0.3f;
    std::vector<float> flows((size_t)nf * w * h * 2);
    for (int f = 0; f < nf; f++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const float Zf = Z - tz * f;  // camera moves towards the plane: X' = X, Z' = Z - tz
                const float X = (x - cx) / fx * Zf, Y = (y - cy) / fy * Zf;
                float* o = &flows[(((size_t)f * h + y) * w + x) * 2];
                o[0] = fx * X / (Zf - tz) + cx - x;
                o[1] = fy * Y / (Zf - tz) + cy - y;
            }
    int n_registered = -1;
    std::vector<float> poses(nf * 6), covar(nf * 36), depth((size_t)w * h), dconf((size_t)w * h);
    const int rc = py_voldor_wrapper(flows.data(), nullptr, nullptr, nullptr, nullptr, nullptr, fx, fy, cx, cy, 0.f, nf, 0, w, h,
                                     "--silent --max_iters 3", n_registered, poses.data(), covar.data(), depth.data(), dconf.data());
    if (rc != 0 || n_registered != nf) { printf("py_voldor_wrapper rc %d n_registered %d\n", rc, n_registered); return 4; }
    for (int f = 0; f < nf; f++) {  // monocular scale: mean |t| = 1; direction -z (points move towards the camera)
        const float* p = &poses[f * 6];
        const float tn = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
        if (std::fabs(p[0]) > 5e-3f || std::fabs(p[1]) > 5e-3f || std::fabs(p[2]) > 5e-3f || p[5] / tn > -0.99f) {
            printf("pose %d: %f %f %f | %f %f %f\n", f, p[0], p[1], p[2], p[3], p[4], p[5]);
            return 5;
        }
    }
    printf("CLIENT OK\n");
    return 0;
}
