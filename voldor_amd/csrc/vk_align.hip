// voldor_amd/csrc/vk_align.hip -- frame-alignment residual / Jacobian maps of the mapping back-end on gfx950
// (SURVEY.md 8(f)-2): align_frame_init_gpu / align_frame_eval_gpu of gpu-kernels/gpu_kernels.h:60-74
// (gpu-kernels/align_frame.cu:414-554; kernels :153-412), the two remaining symbols of the reference's libgpu-kernels.
//
// The Ceres cost functor of frame-alignment/ calls eval once per solver iteration and edge: a [h][w] residual map and a
// [h][w*9] Jacobian map (6-DoF pose + depth scale + colour scale/offset of the reference frame) of a point-to-plane +
// optional photometric error between a reference and a target key-frame.
// Reference: zero-fill of both maps, a residual kernel, a loss kernel, 16x16 blocks, layered textures.
// Here: ONE kernel per eval (residual, chain-rule Jacobian and the weighted sqrt-Cauchy loss fused, each output written
// once, the per-pixel 9 Jacobian entries as whole 36-byte runs), 64x4 blocks (a wave reads whole rows), the XCD band order
// of the VO kernels, and the exact-weight bilinear fetch (deviation D2).  Quirks of the reference that change values
// are kept (see oracle/orc_align.c): theta^(3/2) in d/d(rvec), at_safe(-1) -> last index, raw residual below FLT_EPSILON.
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_internal.hpp"
#include "../../include/gpu_kernels.h"
#include "../../include/voldor_hip.h"
#include <cfloat>
#include <mutex>

namespace vk {

constexpr int AL_PARAMS = 9, AL_MAX_FRAMES = 64;  // align_frame.cu:7,9

struct AlignGeom { int N, w, h, photo; float fx, cx, fy, cy, fxi, cxi, fyi, cyi, vbf, crw; };
struct AlignState {  // per device; the reference keeps these as file statics (align_frame.cu:33-45)
    AlignGeom g{};
    DevBuf images, depths, weights, normals /*[N][h][w] float4*/, dimages /*[N][h][w] float2*/, residual, jacobian;
};
static std::map<int, AlignState*> g_align;

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return { x, y, z }; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

// Rodrigues rotation with optional Jacobians (align_frame.cu:47-137), T = theta^(3/2) as in the reference
template <bool WANT_JW, bool WANT_JP>
__device__ __forceinline__ F3 rot_rvec(F3 p, F3 w, float (&Jw)[9], float (&Jp)[9]) {
#pragma clang fp contract(off)
    const float th2 = dot(w, w);
    if (th2 > FLT_EPSILON) {
        const float th = sqrtf(th2), ith = 1.f / th;
        float s, c;
        sincosf(th, &s, &c);
        const F3 u = w * ith, uxp = cross(u, p);
        const float up = dot(u, p);
        if (WANT_JP) {
            const float uv[3] = { u.x, u.y, u.z }, ux[9] = { 0, -u.z, u.y, u.z, 0, -u.x, -u.y, u.x, 0 };
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) Jp[i * 3 + j] = (i == j ? c : 0.f) + s * ux[i * 3 + j] + (1.f - c) * uv[i] * uv[j];
        }
        if (WANT_JW) {
            const float T = sqrtf(th2 * th);
            const float pv[3] = { p.x, p.y, p.z }, wv[3] = { w.x, w.y, w.z }, uv[3] = { u.x, u.y, u.z }, uxpv[3] = { uxp.x, uxp.y, uxp.z };
            const F3 wxp = cross(w, p);
            const float wxpv[3] = { wxp.x, wxp.y, wxp.z }, wp = dot(w, p);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float e0 = j == 0, e1 = j == 1, e2 = j == 2;
                const float ejxp[3] = { e1 * pv[2] - e2 * pv[1], e2 * pv[0] - e0 * pv[2], e0 * pv[1] - e1 * pv[0] };
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const float dui = (i == j ? ith : 0.f) - wv[i] * wv[j] / T;
                    const float dup = pv[j] * ith - wv[j] * wp / T;
                    const float duxp = ejxp[i] * ith - wv[j] * wxpv[i] / T;
                    Jw[i * 3 + j] = -pv[i] * s * uv[j] + uxpv[i] * c * uv[j] + s * duxp + dui * up * (1.f - c) + uv[i] * dup * (1.f - c) +
                                    uv[i] * up * s * uv[j];
                }
            }
        }
        return p * c + uxp * s + u * (up * (1.0f - c));
    }
    if (WANT_JP) { const float m[9] = { 1, -w.z, w.y, w.z, 1, -w.x, -w.y, w.x, 1 }; for (int k = 0; k < 9; k++) Jp[k] = m[k]; }
    if (WANT_JW) { const float m[9] = { 0, p.z, -p.y, -p.z, 0, p.x, p.y, -p.x, 0 }; for (int k = 0; k < 9; k++) Jw[k] = m[k]; }
    return p + cross(w, p);
}

__device__ __forceinline__ int safe_idx(int i, int n) { return i < 0 ? n - 1 : min(i, n - 1); }  // size_t wrap of gmat.h:181-186
__device__ __forceinline__ float at_safe(const float* m, int w, int h, int x, int y) { return m[safe_idx(y, h) * w + safe_idx(x, w)]; }
__device__ __forceinline__ F3 backproj(const AlignGeom& g, float x, float y, float d) { return { (g.fxi * x + g.cxi) * d, (g.fyi * y + g.cyi) * d, d }; }
template <int NC>
__device__ __forceinline__ void bil_nc(const float* m, int w, int h, float x, float y, float (&out)[NC]) {
    const float fx = floorf(x), fy = floorf(y), a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy;
    const int x1 = min(max(x0 + 1, 0), w - 1), y1 = min(max(y0 + 1, 0), h - 1);
    x0 = min(max(x0, 0), w - 1); y0 = min(max(y0, 0), h - 1);
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
#pragma unroll
    for (int c = 0; c < NC; c++)
        out[c] = w00 * m[(y0 * w + x0) * NC + c] + w10 * m[(y0 * w + x1) * NC + c] + w01 * m[(y1 * w + x0) * NC + c] + w11 * m[(y1 * w + x1) * NC + c];
}

// normals from the depth neighbours + image gradients (align_frame.cu:153-205); the depth gradients the reference also
// stores (:176-185) are never read by the residual kernel and are not produced
__global__ __launch_bounds__(256) static void k_align_init(AlignGeom g, const float* __restrict__ depths, const float* __restrict__ images,
                                                            float4* __restrict__ normals, float2* __restrict__ dimages) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const size_t npx = (size_t)g.w * g.h;
    const float* D = depths + (size_t)f * npx;
    const F3 pt = backproj(g, (float)x, (float)(y - 1), at_safe(D, g.w, g.h, x, y - 1)), pb = backproj(g, (float)x, (float)(y + 1), at_safe(D, g.w, g.h, x, y + 1));
    const F3 pl = backproj(g, (float)(x - 1), (float)y, at_safe(D, g.w, g.h, x - 1, y)), pr = backproj(g, (float)(x + 1), (float)y, at_safe(D, g.w, g.h, x + 1, y));
    F3 n = cross(pt - pb, pl - pr);
    const float nn = sqrtf(dot(n, n));
    n = { n.x / nn, n.y / nn, n.z / nn };
    if (dot(backproj(g, (float)x, (float)y, 1.f), n) > 0) n = n * -1.f;
    normals[(size_t)f * npx + (size_t)y * g.w + x] = make_float4(n.x, n.y, n.z, 0.f);
    if (g.photo) {
        const float* I = images + (size_t)f * npx;
        const int w = g.w, h = g.h;
        const float gx = 0.3f * (at_safe(I, w, h, x + 1, y) - at_safe(I, w, h, x - 1, y)) + 0.1f * (at_safe(I, w, h, x + 1, y - 1) - at_safe(I, w, h, x - 1, y - 1)) +
                         0.1f * (at_safe(I, w, h, x + 1, y + 1) - at_safe(I, w, h, x - 1, y + 1));
        const float gy = 0.3f * (at_safe(I, w, h, x, y + 1) - at_safe(I, w, h, x, y - 1)) + 0.1f * (at_safe(I, w, h, x - 1, y + 1) - at_safe(I, w, h, x - 1, y - 1)) +
                         0.1f * (at_safe(I, w, h, x + 1, y + 1) - at_safe(I, w, h, x + 1, y - 1));
        dimages[(size_t)f * npx + (size_t)y * w + x] = make_float2(gx, gy);
    }
}

struct AlignParams { float ref[AL_PARAMS], tar[AL_PARAMS]; };

// compute_residual + apply_weighted_sqrt_cauchy_loss (align_frame.cu:207-412), one pixel per thread
template <bool WANT_J>
__global__ __launch_bounds__(256) static void k_align_eval(AlignGeom g, AlignParams P, int ref, int tar, const float* __restrict__ depths,
                                                            const float* __restrict__ images, const float* __restrict__ weights,
                                                            const float4* __restrict__ normals, const float2* __restrict__ dimages,
                                                            float* __restrict__ residual, float* __restrict__ jacobian, int apply_weights) {
#pragma clang fp contract(off)  // same un-fused operation order as the oracle
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x = (tile % gridDim.x) * 64 + (threadIdx.x & 63), y = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    if (x >= g.w || y >= g.h) return;
    const int w = g.w, h = g.h;
    const size_t npx = (size_t)w * h, pi = (size_t)y * w + x;
    float J[AL_PARAMS];
#pragma unroll
    for (int k = 0; k < AL_PARAMS; k++) J[k] = 0.f;
    float R = 0.f;
    bool valid = false;
    {
        const F3 rvec = f3(P.ref[0], P.ref[1], P.ref[2]), rvec0 = f3(-P.tar[0], -P.tar[1], -P.tar[2]);  // target pose inverted: world -> cam
        float dummy[9];
        F3 t0 = rot_rvec<false, false>(f3(P.tar[3], P.tar[4], P.tar[5]), rvec0, dummy, dummy);
        t0 = t0 * -1.f;
        const float d_ref = depths[(size_t)ref * npx + pi] * expf(P.ref[6]);
        const F3 p3r = backproj(g, (float)x, (float)y, d_ref);
        const float dp3r_dd[3] = { g.fxi * x + g.cxi, g.fyi * y + g.cyi, 1.f };
        float Jw_rvec[9], Jw_p3r[9], Jt_p3w[9];
        const F3 p3w = rot_rvec<WANT_J, WANT_J>(p3r, rvec, Jw_rvec, Jw_p3r) + f3(P.ref[3], P.ref[4], P.ref[5]);
        const F3 p3t = rot_rvec<false, WANT_J>(p3w, rvec0, dummy, Jt_p3w) + t0;
        const float u = (g.fx * p3t.x) / p3t.z + g.cx, v = (g.fy * p3t.y) / p3t.z + g.cy;
        if (!(u < 0 || u >= w || v < 0 || v >= h || p3t.z < 1.f)) {
            float dt[1], nv[4];
            bil_nc<1>(depths + (size_t)tar * npx, w, h, u, v, dt);
            bil_nc<4>(reinterpret_cast<const float*>(normals + (size_t)tar * npx), w, h, u, v, nv);
            const float d_tar = dt[0] * expf(P.tar[6]);
            const F3 n = f3(nv[0], nv[1], nv[2]);
            const F3 ray = p3t * (d_tar / p3t.z);
            const F3 diff = n * dot(n, ray - p3t);
            const F3 geo = p3t + diff;
            const float ug = (g.fx * geo.x) / geo.z + g.cx, vg = (g.fy * geo.y) / geo.z + g.cy;
            if (!(ug < 0 || ug >= w || vg < 0 || vg >= h)) {
                valid = true;
                const float res_d = 0.5f * dot(diff, diff);
                const float q = g.vbf / (fmaxf(geo.z, 1.0f) * fmaxf(p3t.z, 1.0f));
                const float drw = q * q;
                float c_ref = 0, c_tar = 0, res_c = 0;
                if (g.photo) {
                    float cb[1];
                    bil_nc<1>(images + (size_t)tar * npx, w, h, u, v, cb);
                    c_ref = images[(size_t)ref * npx + pi] + P.ref[8];
                    c_tar = (cb[0] + P.tar[8]) * (expf(P.ref[7]) / expf(P.tar[7]));
                    res_c = 0.5f * (c_ref - c_tar) * (c_ref - c_tar);
                }
                R = g.photo ? drw * res_d + g.crw * res_c : drw * res_d;
                if (WANT_J) {
                    float gp3t[3] = { -diff.x * drw, -diff.y * drw, -diff.z * drw };
                    float dc_scale = 0, dc_off = 0;
                    if (g.photo) {
                        float gI[2];
                        bil_nc<2>(reinterpret_cast<const float*>(dimages + (size_t)tar * npx), w, h, u, v, gI);
                        const float k = c_tar - c_ref;
                        const float du[3] = { g.fx / p3t.z, 0.f, -(g.fx * p3t.x) / (p3t.z * p3t.z) }, dv[3] = { 0.f, g.fy / p3t.z, -(g.fy * p3t.y) / (p3t.z * p3t.z) };
#pragma unroll
                        for (int i = 0; i < 3; i++) gp3t[i] += g.crw * ((gI[0] * k) * du[i] + (gI[1] * k) * dv[i]);
                        dc_scale = k * c_tar; dc_off = (c_ref - c_tar) * 1.f;
                    }
                    float gw[3], gr[3], gp[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) gw[j] = gp3t[0] * Jt_p3w[j] + gp3t[1] * Jt_p3w[3 + j] + gp3t[2] * Jt_p3w[6 + j];
#pragma unroll
                    for (int j = 0; j < 3; j++) gr[j] = gw[0] * Jw_rvec[j] + gw[1] * Jw_rvec[3 + j] + gw[2] * Jw_rvec[6 + j];
#pragma unroll
                    for (int j = 0; j < 3; j++) gp[j] = gw[0] * Jw_p3r[j] + gw[1] * Jw_p3r[3 + j] + gw[2] * Jw_p3r[6 + j];
                    J[0] = gr[0]; J[1] = gr[1]; J[2] = gr[2]; J[3] = gw[0]; J[4] = gw[1]; J[5] = gw[2];
                    J[6] = (gp[0] * dp3r_dd[0] + gp[1] * dp3r_dd[1] + gp[2] * dp3r_dd[2]) * d_ref;
                    J[7] = g.photo ? g.crw * dc_scale : 0.f;
                    J[8] = g.photo ? g.crw * dc_off : 0.f;
                }
            }
        }
    }
    if (!valid) R = __builtin_nanf("");
    const float wgt = apply_weights ? weights[(size_t)ref * npx + pi] : 1.f;
    const float r2 = wgt * R;
    if (r2 > FLT_EPSILON) {  // NaN and tiny residuals pass through untouched (:399)
        const float loss = logf(r2 + 1.f), sl = sqrtf(loss);
        R = sl;
        if (WANT_J) {
            const float k = (0.5f / sl) * (1.f / (r2 + 1.f)) * wgt;
#pragma unroll
            for (int j = 0; j < AL_PARAMS; j++) J[j] *= k;
        }
    }
    residual[pi] = R;
    if (WANT_J) {
#pragma unroll
        for (int j = 0; j < AL_PARAMS; j++) jacobian[pi * AL_PARAMS + j] = J[j];
    }
}

static AlignState* align_state(Context* c) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    AlignState*& s = g_align[c->device];
    if (!s) s = new AlignState();
    return s;
}

}  // namespace vk

using namespace vk;

// gpu_kernels.h:60-66
int align_frame_init_gpu(float* h_images[], float* h_depths[], float* h_weights[], float* h_K, float vbf, float crw, int N, int w, int h) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (!h_depths || !h_weights || !h_K || N <= 0 || N > AL_MAX_FRAMES || w <= 0 || h <= 0) return (int)hipErrorInvalidValue;
    AlignState* S = align_state(c);
    AlignGeom& g = S->g;
    g.N = N; g.w = w; g.h = h; g.vbf = vbf; g.crw = crw; g.photo = (h_images && crw > 0) ? 1 : 0;
    g.fx = h_K[0]; g.cx = h_K[2]; g.fy = h_K[4]; g.cy = h_K[5];
    g.fxi = 1.f / h_K[0]; g.cxi = -h_K[2] / h_K[0]; g.fyi = 1.f / h_K[4]; g.cyi = -h_K[5] / h_K[4];
    const size_t npx = (size_t)w * h, lb = npx * sizeof(float);
    if (int e = S->depths.reserve(lb * N)) return e;
    if (int e = S->weights.reserve(lb * N)) return e;
    if (int e = S->normals.reserve(lb * N * 4)) return e;
    if (int e = S->residual.reserve(lb)) return e;
    if (int e = S->jacobian.reserve(lb * AL_PARAMS)) return e;
    if (g.photo) {
        if (int e = S->images.reserve(lb * N)) return e;
        if (int e = S->dimages.reserve(lb * N * 2)) return e;
    }
    for (int i = 0; i < N; i++) {
        VK_CHECK(hipMemcpyAsync(S->depths.as<char>() + lb * i, h_depths[i], lb, hipMemcpyHostToDevice, c->stream));
        VK_CHECK(hipMemcpyAsync(S->weights.as<char>() + lb * i, h_weights[i], lb, hipMemcpyHostToDevice, c->stream));
        if (g.photo) VK_CHECK(hipMemcpyAsync(S->images.as<char>() + lb * i, h_images[i], lb, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_align_init, dim3((w + 63) / 64, (h + 3) / 4, N), dim3(256), 0, c->stream, g, S->depths.as<float>(), S->images.as<float>(),
                       S->normals.as<float4>(), S->dimages.as<float2>());
    VK_CHECK_LAST();
    VK_CHECK(hipStreamSynchronize(c->stream));  // the host arrays may be freed by the caller
    return 0;
}

// gpu_kernels.h:68-74.  NULL params keep the previous ones (align_frame.cu:425-428); NULL outputs are not downloaded and
// a NULL Jacobian also skips its computation (:434).
int align_frame_eval_gpu(int ref_fid, int tar_fid, const float* h_params_ref, const float* h_params_tar, float* h_o_residual, float* h_o_jacobian,
                         const bool apply_weights) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    AlignState* S = align_state(c);
    const AlignGeom& g = S->g;
    if (g.N <= 0 || ref_fid < 0 || ref_fid >= g.N || tar_fid < 0 || tar_fid >= g.N) return (int)hipErrorInvalidValue;
    static thread_local AlignParams P{};
    if (h_params_ref) memcpy(P.ref, h_params_ref, sizeof P.ref);
    if (h_params_tar) memcpy(P.tar, h_params_tar, sizeof P.tar);
    const dim3 grid((g.w + 63) / 64, (g.h + 3) / 4), block(256);
    const size_t npx = (size_t)g.w * g.h;
    if (h_o_jacobian)
        hipLaunchKernelGGL(k_align_eval<true>, grid, block, 0, c->stream, g, P, ref_fid, tar_fid, S->depths.as<float>(), S->images.as<float>(),
                           S->weights.as<float>(), S->normals.as<float4>(), S->dimages.as<float2>(), S->residual.as<float>(), S->jacobian.as<float>(),
                           apply_weights ? 1 : 0);
    else
        hipLaunchKernelGGL(k_align_eval<false>, grid, block, 0, c->stream, g, P, ref_fid, tar_fid, S->depths.as<float>(), S->images.as<float>(),
                           S->weights.as<float>(), S->normals.as<float4>(), S->dimages.as<float2>(), S->residual.as<float>(), (float*)nullptr,
                           apply_weights ? 1 : 0);
    VK_CHECK_LAST();
    if (h_o_residual) VK_CHECK(hipMemcpyAsync(h_o_residual, S->residual.p, npx * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (h_o_jacobian) VK_CHECK(hipMemcpyAsync(h_o_jacobian, S->jacobian.p, npx * sizeof(float) * AL_PARAMS, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" {
int vk_align_frame_init_gpu(float** h_images, float** h_depths, float** h_weights, float* h_K, float vbf, float crw, int N, int w, int h) {
    return align_frame_init_gpu(h_images, h_depths, h_weights, h_K, vbf, crw, N, w, h);
}
int vk_align_frame_eval_gpu(int ref_fid, int tar_fid, const float* h_params_ref, const float* h_params_tar, float* h_o_residual,
                            float* h_o_jacobian, int apply_weights) {
    return align_frame_eval_gpu(ref_fid, tar_fid, h_params_ref, h_params_tar, h_o_residual, h_o_jacobian, apply_weights != 0);
}
}
