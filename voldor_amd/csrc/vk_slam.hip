// voldor_amd/csrc/vk_slam.hip -- the step right after each VO call in the SLAM driver:
// eval_covisibility (slam_py/slam_utils.py:18-53, called once per registered camera at voldor_slam.py:496-504) on the
// device, so that the depth / confidence maps a window leaves in HBM are scored where they are (the reference downloads
// both maps and runs five numpy passes + a 2-D histogram per camera).
//
// score = 2 v c / max(v + c, 1) with  v = visible stride-sampled pixels / ((w/s)(h/s))   (strictly inside the image)
//                                     c = occupied cells of a (w/2s) x (h/2s) grid / cells  (np.histogram2d semantics:
//                                         edges linspace(0, w, nb+1), right-most edge inclusive, points outside dropped)
#include "vk_common.hpp"
#include "vk_internal.hpp"
#include "../../include/voldor_hip.h"

namespace vk {

struct CovisArgs { float Ki[9], K[9], R[9], t[3]; int w, h, stride, nx, ny, nbx, nby; };

// np.histogram2d cell along one axis: edges e_i = i * (len / nb) in double, [e_i, e_{i+1}) except the last cell which
// also takes v == len; -1 outside.
__device__ __forceinline__ int hist_cell(double v, double len, int nb) {
    if (!(v >= 0.0) || v > len) return -1;
    const double step = len / (double)nb;
    int i = (int)floor(v / step);
    i = max(0, min(i, nb - 1));
    while (i > 0 && v < (double)i * step) i--;
    while (i < nb - 1 && v >= (double)(i + 1) * step) i++;
    return i;
}

__global__ __launch_bounds__(256) static void k_covis(const float* __restrict__ depth, const unsigned char* __restrict__ mask, CovisArgs A,
                                                       unsigned* __restrict__ cells, int* __restrict__ n_vis) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool vis = false;
    if (i < A.nx * A.ny) {
        const int x = (i % A.nx) * A.stride, y = (i / A.nx) * A.stride;
        if (!mask || mask[y * A.w + x]) {
            const float d = depth[y * A.w + x], fx = (float)x, fy = (float)y;
            // (inv(K) [x y 1]^T) * depth ; R . + t ; K .   (slam_utils.py:33,37,40-43), float32 throughout
            const float c0 = (A.Ki[0] * fx + A.Ki[1] * fy + A.Ki[2]) * d, c1 = (A.Ki[3] * fx + A.Ki[4] * fy + A.Ki[5]) * d,
                        c2 = (A.Ki[6] * fx + A.Ki[7] * fy + A.Ki[8]) * d;
            const float p0 = A.R[0] * c0 + A.R[1] * c1 + A.R[2] * c2 + A.t[0], p1 = A.R[3] * c0 + A.R[4] * c1 + A.R[5] * c2 + A.t[1],
                        p2 = A.R[6] * c0 + A.R[7] * c1 + A.R[8] * c2 + A.t[2];
            const float q0 = A.K[0] * p0 + A.K[1] * p1 + A.K[2] * p2, q1 = A.K[3] * p0 + A.K[4] * p1 + A.K[5] * p2,
                        q2 = A.K[6] * p0 + A.K[7] * p1 + A.K[8] * p2;
            if (q2 > 0.f) {
                const float u = q0 / q2, v = q1 / q2;
                vis = u > 0.f && u < (float)A.w && v > 0.f && v < (float)A.h;
                const int bx = hist_cell((double)u, (double)A.w, A.nbx), by = hist_cell((double)v, (double)A.h, A.nby);
                if (bx >= 0 && by >= 0) cells[bx * A.nby + by] = 1u;  // same value from every writer
            }
        }
    }
    const unsigned long long b = __ballot(vis);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_vis, __popcll(b));  // integer: order-independent
}
__global__ __launch_bounds__(256) static void k_covis_count(const unsigned* __restrict__ cells, int n, int* __restrict__ n_cov) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(i < n && cells[i] != 0u);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_cov, __popcll(b));
}

static bool inv3x3(const float* K, float* Ki) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0.0) return false;
    const double m[9] = { e * i - f * h, c * h - b * i, b * f - c * e, f * g - d * i, a * i - c * g, c * d - a * f,
                          d * h - e * g, b * g - a * h, a * e - b * d };
    for (int k = 0; k < 9; k++) Ki[k] = (float)(m[k] / det);
    return true;
}

}  // namespace vk

using namespace vk;

extern "C" int vk_eval_covisibility(const float* depth, const unsigned char* mask, const float* T44, const float* K9, int w, int h, int stride,
                                    float* o_score, int* o_counts) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (!depth || !T44 || !K9 || !o_score || w <= 0 || h <= 0 || stride <= 0 || w / (2 * stride) <= 0 || h / (2 * stride) <= 0)
        return (int)hipErrorInvalidValue;
    CovisArgs A;
    if (!inv3x3(K9, A.Ki)) return (int)hipErrorInvalidValue;
    for (int k = 0; k < 9; k++) A.K[k] = K9[k];
    for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) A.R[r * 3 + q] = T44[r * 4 + q]; A.t[r] = T44[r * 4 + 3]; }
    A.w = w; A.h = h; A.stride = stride;
    A.nx = (w + stride - 1) / stride; A.ny = (h + stride - 1) / stride;  // np.mgrid[0:h:stride, 0:w:stride]
    A.nbx = w / (2 * stride); A.nby = h / (2 * stride);
    const size_t npx = (size_t)w * h;
    const int ncell = A.nbx * A.nby;
    // scratch: [cells | n_vis | n_cov] ints, then host images if the caller's pointers are host memory
    hipPointerAttribute_t at;
    const bool dev_depth = hipPointerGetAttributes(&at, depth) == hipSuccess && at.type == hipMemoryTypeDevice;
    const bool dev_mask = mask && hipPointerGetAttributes(&at, mask) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // host pointers make hipPointerGetAttributes report an error: not ours
    const size_t off_img = ((size_t)(ncell + 2) * sizeof(int) + 255) / 256 * 256;
    if (int e = c->tmp.reserve(off_img + (dev_depth ? 0 : npx * sizeof(float)) + ((mask && !dev_mask) ? npx : 0))) return e;
    char* base = c->tmp.as<char>();
    unsigned* cells = reinterpret_cast<unsigned*>(base);
    int* counters = reinterpret_cast<int*>(base) + ncell;
    VK_CHECK(hipMemsetAsync(base, 0, (size_t)(ncell + 2) * sizeof(int), c->stream));
    const float* d_depth = depth;
    const unsigned char* d_mask = mask;
    if (!dev_depth) {
        VK_CHECK(hipMemcpyAsync(base + off_img, depth, npx * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_depth = reinterpret_cast<const float*>(base + off_img);
    }
    if (mask && !dev_mask) {
        char* m = base + off_img + (dev_depth ? 0 : npx * sizeof(float));
        VK_CHECK(hipMemcpyAsync(m, mask, npx, hipMemcpyHostToDevice, c->stream));
        d_mask = reinterpret_cast<const unsigned char*>(m);
    }
    hipLaunchKernelGGL(k_covis, dim3((A.nx * A.ny + 255) / 256), dim3(256), 0, c->stream, d_depth, d_mask, A, cells, counters);
    hipLaunchKernelGGL(k_covis_count, dim3((ncell + 255) / 256), dim3(256), 0, c->stream, cells, ncell, counters + 1);
    VK_CHECK_LAST();
    int hc[2] = { 0, 0 };
    VK_CHECK(hipMemcpyAsync(hc, counters, sizeof hc, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    const double visibility = (double)hc[0] / (double)((w / stride) * (h / stride));
    const double coverage = (double)hc[1] / (double)ncell;
    const double den = visibility + coverage > 1.0 ? visibility + coverage : 1.0;
    *o_score = (float)(2.0 * (visibility * coverage) / den);
    if (o_counts) { o_counts[0] = hc[0]; o_counts[1] = hc[1]; }
    return 0;
}
