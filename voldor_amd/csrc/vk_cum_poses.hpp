// voldor_amd/csrc/vk_cum_poses.hpp -- the rigid chain of a window folded into one projective map per frame (PoseBlock::cumM / cumT, fast path) and the
// world-scale factor of normalize_world_scale (voldor.cpp:309-317).  One workgroup's worth of work (12 active lanes, fp64), needed by the first cost
// kernel of the depth half: run by an extra workgroup of the first fb_smooth launch, or by its own launch k_cum_poses.
#pragma once
#include "vk_common.hpp"
#include "vk_device.hpp"

namespace vk {

// voldor.cpp:309-317: scale = n / sum ||t_i|| over the registered frames (frames dropped by this iteration's decision do not count)
__device__ __forceinline__ static float world_scale_factor(const PoseBlock* P, int n_launch, const float (*ts)[3]) {
    const int n = min(n_launch, P->n_active);
    if (n <= 0) return 1.f;  // window lost: nothing to normalise (deviation D6)
    float ws = 0.f;
    for (int i = 0; i < n; i++) {
        const float* t = ts[i];
        ws = (float)((double)ws + sqrt((double)t[0] * t[0] + (double)t[1] * t[1] + (double)t[2] * t[2]));  // float += double (cv::norm, voldor.cpp:312)
    }
    return (float)n / ws;
}
// The density reduction that closes an E-step (update_rigidnesses + host cv::sum, optimize_depth.cu:84-138, voldor.cpp:171) as ONE workgroup's work per
// frame: block f < n_launch sums the per-tile rigidness sums of frame f in a fixed order -> CamState::pose_rigidness_density; block n_launch (present
// when scale_out != NULL) is the pose half of normalize_world_scale (voldor.cpp:309-317).  Run by k_reduce_density, or -- window pipeline, fast mode --
// by extra workgroups of the NEXT launch of the stream, the correspondence trace of camera 0 (k_collect), which reads none of what is written here.
__device__ __forceinline__ static void reduce_density_block(int f, const ReduceArgs& a) {
    PoseBlock* P = a.P;
    if (f == a.n_launch) {
        if (threadIdx.x != 0) return;
        const int n = min(a.n_launch, P->n_active);  // frames dropped by this iteration's decision do not count
        // scale_ready: the factor was computed from these poses at the start of the call (cum_poses_block) and the depth map already
        // carries it (k_update_rigidness_lean); otherwise it is computed here and k_scale follows
        const float s = a.scale_ready ? *a.scale_out : world_scale_factor(P, a.n_launch, P->ts);
        for (int i = 0; i < n; i++)
            for (int d = 0; d < 3; d++) { P->ts[i][d] *= s; a.cams[i].t[d] = P->ts[i][d]; }
        *a.scale_out = s;
        return;
    }
    if (f >= P->n_active) return;
    __shared__ float s_red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < a.nblk; i += 256) acc += a.partial[(size_t)f * a.nblk + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.cams[f].pose_rigidness_density = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)a.npx;
}
// world_scale (may be NULL): the factor of normalize_world_scale (voldor.cpp:309-317), n / sum ||t_i|| over the registered frames, from
// the poses this optimize_depth call runs with; the E-step kernel stores the scaled depth, k_reduce_density then scales the poses.
__device__ __forceinline__ static void cum_poses_block(PoseBlock* P, int N, int N_dp, float* world_scale) {
    __shared__ double Rc[9], tc[3];
    __shared__ float sR[MAX_FRAMES][9], sT[MAX_FRAMES][3];  // one round trip to the pose block instead of one per frame of the chain
    const int l = threadIdx.x, nt = blockDim.x;  // any workgroup size (its own launch: 64; riding in the 512-thread mode kernel: every element once)
    for (int i = l; i < N * 9; i += nt) sR[i / 9][i % 9] = P->Rs[i / 9][i % 9];
    for (int i = l; i < N * 3; i += nt) sT[i / 3][i % 3] = P->ts[i / 3][i % 3];
    const double fx = P->K4[0], cx = P->K4[1], fy = P->K4[2], cy = P->K4[3];
    __syncthreads();
    auto emit = [&](const double* R, const double* t, float* M, float* T) {  // K R K^-1 and K t
        if (l < 9) {
            const int r = l / 3, c = l % 3;
            double kr[3];
            for (int j = 0; j < 3; j++) kr[j] = r == 0 ? fx * R[j] + cx * R[6 + j] : (r == 1 ? fy * R[3 + j] + cy * R[6 + j] : R[6 + j]);
            const float v = (float)(c == 0 ? kr[0] / fx : (c == 1 ? kr[1] / fy : kr[2] - kr[0] * cx / fx - kr[1] * cy / fy));
            M[l] = v;
        } else if (l < 12) {
            const int r = l - 9;
            const float v = (float)(r == 0 ? fx * t[0] + cx * t[2] : (r == 1 ? fy * t[1] + cy * t[2] : t[2]));
            T[r] = v;
        }
    };
    for (int f = 0; f < N; f++) {
        const float* R = sR[f]; const float* t = sT[f];
        double nv = 0.0;
        if (l < 9) {
            const int r = l / 3, c = l % 3;
            nv = f == 0 ? (double)R[l] : (double)R[r * 3] * Rc[c] + (double)R[r * 3 + 1] * Rc[3 + c] + (double)R[r * 3 + 2] * Rc[6 + c];
        } else if (l < 12) {
            const int r = l - 9;
            nv = f == 0 ? (double)t[r] : (double)R[r * 3] * tc[0] + (double)R[r * 3 + 1] * tc[1] + (double)R[r * 3 + 2] * tc[2] + (double)t[r];
        }
        __syncthreads();
        if (l < 9) Rc[l] = nv; else if (l < 12) tc[l - 9] = nv;
        __syncthreads();
        emit(Rc, tc, P->cumM[f], P->cumT[f]);
    }
    for (int f = 0; f < N_dp; f++) {
        double R[9], t[3];
        for (int k = 0; k < 9; k++) R[k] = P->dpRs[f][k];
        for (int k = 0; k < 3; k++) t[k] = P->dpts[f][k];
        emit(R, t, P->dpM[f], P->dpT[f]);
    }
    if (l == 0) {
        int ident = 0;
        for (int f = 0; f < N_dp; f++) {
            bool id = true;
            for (int k = 0; k < 9; k++) id = id && P->dpRs[f][k] == ((k % 4 == 0) ? 1.f : 0.f);
            for (int k = 0; k < 3; k++) id = id && P->dpts[f][k] == 0.f;
            ident |= id ? (1 << f) : 0;
        }
        P->dp_ident = ident;
    }
    if (world_scale && l == 0) *world_scale = world_scale_factor(P, N, sT);
}

}  // namespace vk
