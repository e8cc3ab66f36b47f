// oracle/_ref builder, part 2: the DEVICE code of the reference's CUDA files executed on the CPU.
// The Makefile cuts the device prefix of gpu-kernels/optimize_depth.cu (everything before the host entry point, lines 1-291)
// and the kernel of gpu-kernels/fb_smooth.h (lines 17-70) out of the files into temp files outside the repo and passes
// their paths as REF_OD_INC / REF_FB_INC; ref_stubs/emul/cuda_emul.h supplies threadIdx & co, a sequential launcher, a
// host-backed GMat and the two documented substitutions (D1 RNG, D2 bilinear).  The launch geometry and stage order
// below restate optimize_depth.cu:314-325,462-494 and fb_smooth.h:72-108 (host code that cannot be compiled: <<< >>>).
// TEST INFRASTRUCTURE ONLY: pins oracle/orc_model.c (tests/golden/ref_depth.npz, ref_fb.npz).
#if defined(REF_OD_INC) && defined(REF_FB_INC)
#include <vector>
#include "ref_stubs/emul/cuda_emul.h"
#include "gpu-kernels/utils.h"
#include "gpu-kernels/residual_model.h"

namespace ref_fb {
#define FB_MSG_L2R 0
#define FB_MSG_T2B 1
#define FB_MSG_R2L 2
#define FB_MSG_B2T 3
#define FB_POSTERIOR 4
#include REF_FB_INC
}  // namespace ref_fb

namespace ref_od {
#include REF_OD_INC
}  // namespace ref_od

extern "C" {

// fb_smooth_batch_inplace (fb_smooth.h:72-108): rows (L2R, R2L, posterior), then columns (T2B, B2T, posterior)
void ref_fb_smooth(float* maps, int N, int w, int h, float s0_ems_prob, float no_change_prob) {
    std::vector<float> fwd((size_t)N * w * h), bwd((size_t)N * w * h);
    GMatf m, f, b;
    m.bind(maps, w, h, N); f.bind(fwd.data(), w, h, N); b.bind(bwd.data(), w, h, N);
    const dim3 blk(16, 16, 1), grid(DIV_CEIL_EMUL(w, 16), DIV_CEIL_EMUL(h, 16), N);
    const dim3 blk_row(1, 128, 1), grid_row(1, DIV_CEIL_EMUL(h, 128), N), blk_col(128, 1, 1), grid_col(DIV_CEIL_EMUL(w, 128), 1, N);
    using namespace ref_fb;
    emul_launch(grid_row, blk_row, [&] { fb_smooth_inplace_kernel<FB_MSG_L2R>(m, f, b, s0_ems_prob, no_change_prob, N, w, h); });
    emul_launch(grid_row, blk_row, [&] { fb_smooth_inplace_kernel<FB_MSG_R2L>(m, f, b, s0_ems_prob, no_change_prob, N, w, h); });
    emul_launch(grid, blk, [&] { fb_smooth_inplace_kernel<FB_POSTERIOR>(m, f, b, s0_ems_prob, no_change_prob, N, w, h); });
    emul_launch(grid_col, blk_col, [&] { fb_smooth_inplace_kernel<FB_MSG_T2B>(m, f, b, s0_ems_prob, no_change_prob, N, w, h); });
    emul_launch(grid_col, blk_col, [&] { fb_smooth_inplace_kernel<FB_MSG_B2T>(m, f, b, s0_ems_prob, no_change_prob, N, w, h); });
    emul_launch(grid, blk, [&] { fb_smooth_inplace_kernel<FB_POSTERIOR>(m, f, b, s0_ems_prob, no_change_prob, N, w, h); });
}

// optimize_depth_gpu without the uploads (optimize_depth.cu:293-520): binds the "device" arrays to the caller's host arrays,
// fills the __constant__ block, then runs the kernels in the order of :462-494.  flows [N][h][w][2], rig [N][h][w],
// priors/pconfs/confs [N_dp][h][w], depth [h][w] (in/out), cost [h][w] (out), K4 = fx,cx,fy,cy, Rs [N][9], ts [N][3].
// rand_epoch: counter value the random-sample kernels start from (the reference keeps cuRAND states across calls).
// stage mask: 1 cost map, 2 random samples, 4 global propagation, 8 local propagation, 16 rigidness update, 32 fb_smooth.
void ref_optimize_depth(float* flows, float* rig, float* priors, float* pconfs, float* confs, float* depth, float* cost, const float* K4,
                        const float* Rs, const float* ts, const float* dpRs, const float* dpts, float abs_resize_factor, int N, int N_dp, int w,
                        int h, float basefocal, int n_rand_samples, int global_prop_step, int local_prop_width, float lambda, float omega,
                        float disp_delta, float delta, float s0_ems_prob, float no_change_prob, float range_factor, unsigned rand_epoch,
                        int stages) {
    using namespace ref_od;
    for (int k = 0; k < 4; k++) _K4[k] = K4[k];
    _K4_inv[0] = 1.f / K4[0]; _K4_inv[1] = -K4[1] / K4[0]; _K4_inv[2] = 1.f / K4[2]; _K4_inv[3] = -K4[3] / K4[2];  // :343-347
    if (N > 0) { memcpy(_Rs, Rs, sizeof(float) * 9 * N); memcpy(_ts, ts, sizeof(float) * 3 * N); }
    if (N_dp > 0) { memcpy(_dp_Rs, dpRs, sizeof(float) * 9 * N_dp); memcpy(_dp_ts, dpts, sizeof(float) * 3 * N_dp); }
    _N = N; _N_dp = N_dp; _w = w; _h = h; _abs_resize_factor = abs_resize_factor; _basefocal = basefocal;
    _lambda = lambda; _omega = omega; _delta = delta; _disp_delta = disp_delta; _range_factor = range_factor;
    std::vector<curandState> states((size_t)w * h);
    _d_rand_states.bind(states.data(), w, h, 1);
    _d_flows.bind(reinterpret_cast<float2*>(flows), w, h, N);
    _d_rigidnesses.bind(rig, w, h, N);
    _d_depth_priors.bind(priors, w, h, N_dp); _d_depth_prior_pconfs.bind(pconfs, w, h, N_dp); _d_depth_prior_confs.bind(confs, w, h, N_dp);
    _d_depth.bind(depth, w, h, 1); _d_cost_map.bind(cost, w, h, 1);
    const dim3 blk(16, 16), grid(DIV_CEIL_EMUL(w, 16), DIV_CEIL_EMUL(h, 16));                                     // :311-312
    const dim3 blk_rc(1, 64), grid_rc(1, DIV_CEIL_EMUL(h, 64)), blk_cc(64, 1), grid_cc(DIV_CEIL_EMUL(w, 64), 1);  // :314-318
    emul_launch(grid, blk, [&] { init_rand_states(); });
    for (auto& s : states) s.counter = rand_epoch;
    if (stages & 32) {
        if (N > 0) ref_fb_smooth(rig, N, w, h, s0_ems_prob, no_change_prob);
        if (N_dp > 0) ref_fb_smooth(confs, N_dp, w, h, s0_ems_prob, no_change_prob);
    }
    if (stages & 1) emul_launch(grid, blk, [&] { compute_cost_map(); });
    if (stages & 2) for (int it = 0; it < n_rand_samples; it++) emul_launch(grid, blk, [&] { optimize_depth_with_rand_inplace(); });
    if ((stages & 4) && global_prop_step > 0) {
        emul_launch(grid_rc, blk_rc, [&] { optimize_depth_with_global_propagation_inplace<PROPAGATE_L2R>(global_prop_step); });
        emul_launch(grid_cc, blk_cc, [&] { optimize_depth_with_global_propagation_inplace<PROPAGATE_B2T>(global_prop_step); });
        emul_launch(grid_rc, blk_rc, [&] { optimize_depth_with_global_propagation_inplace<PROPAGATE_R2L>(global_prop_step); });
        emul_launch(grid_cc, blk_cc, [&] { optimize_depth_with_global_propagation_inplace<PROPAGATE_T2B>(global_prop_step); });
    }
    if ((stages & 8) && local_prop_width > 0) {
        const dim3 grid_rs(DIV_CEIL_EMUL(w, 16 * local_prop_width), DIV_CEIL_EMUL(h, 16)), grid_cs(DIV_CEIL_EMUL(w, 16), DIV_CEIL_EMUL(h, 16 * local_prop_width));  // :321-325
        emul_launch(grid_rs, blk, [&] { optimize_depth_with_local_propagation_inplace<PROPAGATE_L2R>(local_prop_width); });
        emul_launch(grid_cs, blk, [&] { optimize_depth_with_local_propagation_inplace<PROPAGATE_B2T>(local_prop_width); });
        emul_launch(grid_rs, blk, [&] { optimize_depth_with_local_propagation_inplace<PROPAGATE_R2L>(local_prop_width); });
        emul_launch(grid_cs, blk, [&] { optimize_depth_with_local_propagation_inplace<PROPAGATE_T2B>(local_prop_width); });
    }
    if (stages & 16) emul_launch(grid, blk, [&] { update_rigidnesses(); });
}
}
#endif
