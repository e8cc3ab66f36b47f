/* oracle/orc_voldor.c -- ORACLE (test infrastructure only, see orc.h).
 * Whole-window EM schedule = py_voldor_wrapper: voldor/py_export.cpp:5-79,
 * voldor/voldor.cpp:4-317 (init/solve/bootstrap/optimize_cameras/optimize_depth/
 * normalize_world_scale), voldor/geometry.cpp:5-332, voldor/config.h:4-253. */
#include "orc.h"
#include <math.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include "../voldor_amd/csrc/vk_ref_cv.h"

/* ------------------------------------------------------------------ Config (config.h:4-82) */
typedef struct {
    float omega, disp_delta, delta, basefocal;
    int rg_refine, rg_refine_last_only; float rg_trunc_sigma, rg_covar_reg_lambda, rg_pose_scaling; int rg_max_iters; float rg_epsilon;
    float resize_factor, abs_resize_factor, fx, fy, cx, cy; int exclusive_gpu_context;
    int debug, silent, save_everything, viz_img_per_row; float viz_depth_scale;
    float lambda, meanshift_kernel_var, meanshift_rvec_scale; int norm_world_scale;
    int cpu_p3p, lambdatwist, n_poses_to_sample; float pose_sample_min_depth, pose_sample_max_depth;
    int max_trace_on_flow; float rigidness_threshold, rigidness_sum_threshold;
    float trunc_rigidness_density, trunc_sample_density, no_trunc_iters; int max_iters, min_iters_after_trunc;
    int fb_smooth; float fb_emm, fb_no_change_prob;
    int optimize_depth, depth_rand_samples, depth_global_prop_step, depth_local_prop_width; float depth_range_factor;
    int meanshift_max_iters, meanshift_max_init_trials; float meanshift_good_init_confidence, meanshift_epsilon;
    int kitti_estimate_ground, kitti_ground_holo_width; float kitti_ground_roi, kitti_ground_meanshift_kernel_var;
} orc_config;

static void cfg_defaults(orc_config* c) {
    memset(c, 0, sizeof *c);
    c->omega = 0.15f; c->disp_delta = 1.f; c->delta = 0.5f; c->basefocal = 0;
    c->rg_refine = 1; c->rg_refine_last_only = 1; c->rg_trunc_sigma = 3.f; c->rg_covar_reg_lambda = 0.001f;
    c->rg_pose_scaling = 100.f; c->rg_max_iters = 100; c->rg_epsilon = 1e-5f;
    c->resize_factor = 1.f; c->abs_resize_factor = 1.f; c->exclusive_gpu_context = 1;
    c->viz_img_per_row = 2; c->viz_depth_scale = 5;
    c->lambda = 0.15f; c->meanshift_kernel_var = 0.1f; c->meanshift_rvec_scale = 25.f; c->norm_world_scale = 1;
    c->cpu_p3p = 0; c->lambdatwist = 1; c->n_poses_to_sample = 8192; c->pose_sample_min_depth = 0.1f;
    c->pose_sample_max_depth = 1000.f; c->max_trace_on_flow = 3; c->rigidness_threshold = 0.5f; c->rigidness_sum_threshold = 1.f;
    c->trunc_rigidness_density = 0.05f; c->trunc_sample_density = 0.001f; c->no_trunc_iters = 2; c->max_iters = 5; c->min_iters_after_trunc = 3;
    c->fb_smooth = 1; c->fb_emm = 0.5f; c->fb_no_change_prob = 0.9f;
    c->optimize_depth = 1; c->depth_rand_samples = 10; c->depth_global_prop_step = 8; c->depth_local_prop_width = 32; c->depth_range_factor = 1.f;
    c->meanshift_max_iters = 100; c->meanshift_max_init_trials = 20; c->meanshift_good_init_confidence = 0.5f; c->meanshift_epsilon = 1e-5f;
    c->kitti_estimate_ground = 0; c->kitti_ground_holo_width = 5; c->kitti_ground_roi = 0.4f; c->kitti_ground_meanshift_kernel_var = 0.01f;
}

/* config.h:110-253.  str_to_arg's switch has no breaks (config.h:85-99): every numeric field
 * ends up as the value of stod(str) converted to the field type. Unknown key / missing value
 * -> the reference prints and exit(1)s; the oracle returns nonzero instead. */
typedef struct { const char* key; int is_int; size_t off; } cfg_key;
#define KF(name) { "--" #name, 0, offsetof(orc_config, name) }
#define KI(name) { "--" #name, 1, offsetof(orc_config, name) }
static const cfg_key CFG_KEYS[] = {
    KF(basefocal), KF(omega), KF(disp_delta), KF(delta),
    KI(rg_refine), KI(rg_refine_last_only), KF(rg_trunc_sigma), KF(rg_covar_reg_lambda), KF(rg_epsilon), KI(rg_max_iters), KF(rg_pose_scaling),
    KF(resize_factor), KF(abs_resize_factor), KF(fx), KF(fy), KF(cx), KF(cy),
    KI(viz_img_per_row), KF(viz_depth_scale), KI(exclusive_gpu_context),
    KF(lambda), KF(meanshift_kernel_var), KF(meanshift_rvec_scale), KI(norm_world_scale),
    KI(cpu_p3p), KI(lambdatwist), KI(max_trace_on_flow), KI(n_poses_to_sample), KF(pose_sample_min_depth), KF(pose_sample_max_depth),
    KF(rigidness_threshold), KF(rigidness_sum_threshold),
    KF(trunc_rigidness_density), KF(trunc_sample_density), KI(max_iters), KF(no_trunc_iters), KI(min_iters_after_trunc),
    KI(fb_smooth), KF(fb_emm), KF(fb_no_change_prob),
    KI(optimize_depth), KI(depth_rand_samples), KI(depth_global_prop_step), KI(depth_local_prop_width), KF(depth_range_factor),
    KI(meanshift_max_iters), KI(meanshift_max_init_trials), KF(meanshift_good_init_confidence), KF(meanshift_epsilon),
    KI(kitti_estimate_ground), KI(kitti_ground_holo_width), KF(kitti_ground_roi), KF(kitti_ground_meanshift_kernel_var),
};
static int cfg_parse(orc_config* c, const char* s) {
    char* buf = strdup(s ? s : "");
    char* save = NULL;
    int rc = 0;
    for (char* tok = strtok_r(buf, " \t\r\n", &save); tok; tok = strtok_r(NULL, " \t\r\n", &save)) {
        if (!strcmp(tok, "--debug")) { c->debug = 1; continue; }
        if (!strcmp(tok, "--silent")) { c->silent = 1; continue; }
        if (!strcmp(tok, "--save_everything")) { c->save_everything = 1; continue; }
        const cfg_key* k = NULL;
        for (size_t i = 0; i < sizeof CFG_KEYS / sizeof CFG_KEYS[0]; i++)
            if (!strcmp(tok, CFG_KEYS[i].key)) { k = &CFG_KEYS[i]; break; }
        if (!k) { rc = 1; break; }
        char* val = strtok_r(NULL, " \t\r\n", &save);
        if (!val) { rc = 2; break; }
        char* end = NULL;
        double v = strtod(val, &end);
        if (end == val) { rc = 3; break; }
        if (k->is_int) *(int*)((char*)c + k->off) = (int)v;
        else *(float*)((char*)c + k->off) = (float)v;
    }
    free(buf);
    return rc;
}

/* ------------------------------------------------------------------ Camera (utils.h:31-77) */
typedef struct {
    float R[9], t[3], rvec[3], pose_covar[36];
    float pose_density; int pose_sample_count; float pose_rigidness_density;
    int last_used_ms_iters, last_used_gu_iters;
} orc_cam;
static void cam_init(orc_cam* c) {
    memset(c, 0, sizeof *c);
    c->R[0] = c->R[4] = c->R[8] = 1.f;
}

typedef struct {
    orc_config cfg;
    int n_flows, n_flows_init, n_dp, w, h, iters_cur, iters_remain, has_disparity;
    float* depth; float* cost;
    float* priors; float* pconfs; float* confs; orc_cam dp_poses[ORC_MAX_FRAMES];
    const float* flows; float* rig; orc_cam cams[ORC_MAX_FRAMES];
    uint32_t rand_epoch;
    float* od_depth; /* ORC_EMULATE_B1 only: the depth map as optimize_depth.cu's device copy holds it */
} orc_voldor_t;

enum { OD_DEFAULT = 0, OD_ONLY_USE_DEPTH_PRIOR = 1, OD_UPDATE_RIGIDNESS_ONLY = 2 };

/* voldor.cpp:203-307 */
static void v_optimize_depth(orc_voldor_t* v, int flag) {
    if (v->n_flows == 0 && v->n_dp == 0) return;
    orc_od_params p;
    memset(&p, 0, sizeof p);
    const orc_config* c = &v->cfg;
    p.N = (flag == OD_ONLY_USE_DEPTH_PRIOR) ? 0 : v->n_flows;
    p.N_dp = v->n_dp; p.w = v->w; p.h = v->h;
    float K[9] = { c->fx, 0, c->cx, 0, c->fy, c->cy, 0, 0, 1 };
    memcpy(p.K, K, sizeof K);
    for (int i = 0; i < v->n_flows; i++) { memcpy(p.Rs[i], v->cams[i].R, 36); memcpy(p.ts[i], v->cams[i].t, 12); }
    for (int i = 0; i < v->n_dp; i++) { memcpy(p.dp_Rs[i], v->dp_poses[i].R, 36); memcpy(p.dp_ts[i], v->dp_poses[i].t, 12); }
    p.abs_resize_factor = c->abs_resize_factor; p.basefocal = c->basefocal;
    p.n_rand_samples = c->depth_rand_samples; p.global_prop_step = c->depth_global_prop_step; p.local_prop_width = c->depth_local_prop_width;
    p.lambda = c->lambda; p.omega = c->omega; p.disp_delta = v->has_disparity ? c->disp_delta : -1; p.delta = c->delta;
    p.fb_smooth = c->fb_smooth; p.s0_ems_prob = c->fb_emm; p.no_change_prob = c->fb_no_change_prob;
    p.range_factor = c->depth_range_factor; p.update_rigidness_only = (flag == OD_UPDATE_RIGIDNESS_ONLY);
    /* D4 vs SURVEY Appendix B-1: with exclusive_gpu_context the reference stops uploading the depth map after iteration 1
     * (voldor.cpp:275-291 passes NULL), so optimize_depth.cu keeps searching from ITS copy, which never saw
     * normalize_world_scale().  The oracle (and the product) use the one normalised map.  ORC_EMULATE_B1=1 reproduces the
     * reference's behaviour instead, for the whole-window comparison with the reference pipeline run on the CPU. */
    const char* b1 = getenv("ORC_EMULATE_B1");
    const size_t bytes = sizeof(float) * (size_t)v->w * v->h;
    if (b1 && b1[0] == '1') {
        const int stale = c->exclusive_gpu_context && v->iters_cur >= 2 && v->od_depth != NULL;
        if (!v->od_depth) v->od_depth = malloc(bytes);
        if (!stale) memcpy(v->od_depth, v->depth, bytes);
        orc_optimize_depth(&p, v->flows, v->rig, v->priors, v->pconfs, v->confs, v->od_depth, v->cost, &v->rand_epoch);
        memcpy(v->depth, v->od_depth, bytes);
        return;
    }
    orc_optimize_depth(&p, v->flows, v->rig, v->priors, v->pconfs, v->confs, v->depth, v->cost, &v->rand_epoch);
}

/* cv::Mat /= double is a.convertTo(a, -1, 1./b) (OpenCV core/mat.inl.hpp, CV_MAT_AUG_OPERATOR): float data are MULTIPLIED
 * by (float)(1./b), not divided -- one ulp apart for most b.  The element-wise at<float>() /= of :233-236 is a true division. */
static inline float cv_div_scale(float b) { return (float)(1.0 / (double)b); }

/* geometry.cpp:5-265 */
static int v_optimize_camera_pose(orc_voldor_t* v, int active_idx, int successive_pose, int rg_refine) {
    const orc_config* c = &v->cfg;
    const int w = v->w, h = v->h, npx = w * h;
    orc_cam* cam = &v->cams[active_idx];
    float K[9] = { c->fx, 0, c->cx, 0, c->fy, c->cy, 0, 0, 1 };
    float (*Rs)[9] = malloc(sizeof(float) * 9 * ORC_MAX_FRAMES);
    float (*ts)[3] = malloc(sizeof(float) * 3 * ORC_MAX_FRAMES);
    for (int i = 0; i < v->n_flows; i++) { memcpy(Rs[i], v->cams[i].R, 36); memcpy(ts[i], v->cams[i].t, 12); }
    float* p2m = malloc(sizeof(float) * npx * 2); float* p3m = malloc(sizeof(float) * npx * 3);
    orc_collect_p3p(v->flows, v->rig, v->depth, K, (const float (*)[9])Rs, (const float (*)[3])ts, p2m, p3m,
                    v->n_flows, w, h, active_idx, c->rigidness_threshold, c->rigidness_sum_threshold,
                    c->pose_sample_min_depth, c->pose_sample_max_depth, c->max_trace_on_flow);
    int n_points = 0; /* geometry.cpp:68-80: number of finite correspondences */
    for (int i = 0; i < npx; i++) if (isfinite(p2m[i * 2])) n_points++;
    free(Rs); free(ts);
    if (getenv("ORC_PRINT_POINTS")) fprintf(stderr, "orc: iter %d camera %d: %d valid correspondences of %d pixels\n", v->iters_cur, active_idx, n_points, npx);
    if (n_points < 4) { free(p2m); free(p3m); return 0; } /* :84 */

    const int np = c->n_poses_to_sample;
    float* rv = malloc(sizeof(float) * np * 3); float* tv = malloc(sizeof(float) * np * 3);
    float* pool = malloc(sizeof(float) * np * 6);
    /* :99-170. cpu_p3p=1 is the reference's CPU path: lambdatwist_p4p<double,...> (:112).  Draw: the reference's own (index into the
     * row-major compacted list, geometry.cpp:68-80 + solve_batch_lambdatwist.cu:16-19, clamp D3) -- the default since round 3: over the
     * 24-window ensemble the rejection draw D3b, being an INDEPENDENT sample of the hypotheses, puts the depth maps 1.4x further from the
     * reference's than the reference's own 1-ulp self-distance (tests/test_gpu_ensemble.py), the reference's draw does not.
     * ORC_REFERENCE_DRAW=0 (product: --reference_draw 0) selects D3b: rejection over the map, which keeps the reference's draw as the
     * fallback below 5 % valid pixels, where 256 probes per point start to lose hypotheses (product: DRAW_RANK_INV_DENSITY). */
    const char* ref_draw = getenv("ORC_REFERENCE_DRAW");
    if (!(ref_draw && ref_draw[0] == '0') || (long long)n_points * 20 < (long long)npx) {
        float* c2 = malloc(sizeof(float) * npx * 2); float* c3 = malloc(sizeof(float) * npx * 3);
        const int nc = orc_compact_p3p(p2m, p3m, npx, c2, c3);
        orc_solve_batch_p3p(c3, c2, rv, tv, K, nc, np, !c->lambdatwist, c->cpu_p3p ? 1 : 0);
        free(c2); free(c3);
    } else
        orc_solve_batch_p3p_maps(p2m, p3m, npx, n_points, rv, tv, K, np, !c->lambdatwist, c->cpu_p3p ? 1 : 0);
    free(p2m); free(p3m);
    int used = 0;
    for (int i = 0; i < np; i++) {
        float s = rv[i * 3] + rv[i * 3 + 1] + rv[i * 3 + 2] + tv[i * 3] + tv[i * 3 + 1] + tv[i * 3 + 2];
        if (isfinite(s)) {
            memcpy(pool + used * 6, rv + i * 3, 12); memcpy(pool + used * 6 + 3, tv + i * 3, 12);
            used++;
        }
    }
    free(rv); free(tv);
    if (used == 0) { free(pool); return 0; }
    cam->pose_sample_count = used;

    float pose_opm[6] = { cam->rvec[0], cam->rvec[1], cam->rvec[2], cam->t[0], cam->t[1], cam->t[2] };
    for (int i = 0; i < used; i++) for (int d = 0; d < 3; d++) pool[i * 6 + d] *= c->meanshift_rvec_scale; /* :191 */
    for (int d = 0; d < 3; d++) pose_opm[d] *= c->meanshift_rvec_scale;
    { /* test aid: ORC_DUMP_POOL="<iter>,<camera>,<path>" writes the scaled pool and the start of the mean shift of that call ([used][6] floats,
       * then 6 floats), for stage-level replays against the reference's meanshift_gpu */
        const char* dp = getenv("ORC_DUMP_POOL");
        int di, dc; char path[512];
        if (dp && sscanf(dp, "%d,%d,%511s", &di, &dc, path) == 3 && di == v->iters_cur && dc == active_idx) {
            FILE* f = fopen(path, "wb");
            if (f) { fwrite(&used, sizeof used, 1, f); fwrite(pool, sizeof(float), (size_t)used * 6, f); fwrite(pose_opm, sizeof(float), 6, f); fclose(f); }
        }
    }
    orc_meanshift(pool, c->meanshift_kernel_var, pose_opm, &cam->pose_density, &cam->last_used_ms_iters,
                  successive_pose, used, 6, c->meanshift_epsilon, c->meanshift_max_iters,
                  c->meanshift_max_init_trials, c->meanshift_good_init_confidence);
    if (rg_refine) { /* :201-246 */
        memset(cam->pose_covar, 0, sizeof cam->pose_covar);
        for (int d = 0; d < 6; d++) cam->pose_covar[d * 6 + d] = c->meanshift_kernel_var;
        const float sc = c->rg_pose_scaling;
        for (int d = 0; d < 36; d++) cam->pose_covar[d] *= (sc * sc);
        for (int d = 0; d < 6; d++) pose_opm[d] *= sc;
        for (int i = 0; i < used * 6; i++) pool[i] *= sc;
        int ret = orc_fit_robust_gaussian(pool, pose_opm, cam->pose_covar, c->rg_trunc_sigma, c->rg_covar_reg_lambda,
                                          &cam->pose_density, &cam->last_used_gu_iters, used, 6, c->rg_epsilon, c->rg_max_iters);
        if (ret == 0) {
            { const float isc2 = cv_div_scale(sc * sc); for (int d = 0; d < 36; d++) cam->pose_covar[d] *= isc2; }
            for (int i1 = 0; i1 < 6; i1++)
                for (int i2 = 0; i2 < 6; i2++) {
                    if (i1 < 3 || i2 < 3) cam->pose_covar[i1 * 6 + i2] /= c->meanshift_rvec_scale;
                    if (i1 < 3 && i2 < 3) cam->pose_covar[i1 * 6 + i2] /= c->meanshift_rvec_scale;
                }
        } else memset(cam->pose_covar, 0, sizeof cam->pose_covar);
        { const float isc = cv_div_scale(sc); for (int d = 0; d < 6; d++) pose_opm[d] *= isc; }
    }
    { const float irs = cv_div_scale(c->meanshift_rvec_scale); for (int d = 0; d < 3; d++) pose_opm[d] *= irs; }
    free(pool);
    int ok = 1; /* checkRange :256 */
    for (int d = 0; d < 6; d++) if (!isfinite(pose_opm[d])) ok = 0;
    if (!ok) return 0;
    orc_rvec_to_rotmat(pose_opm, cam->R);            /* Rodrigues(pose_opm -> cams[i].R), :258 */
    vrcv_rvec_of_R32(cam->R, cam->rvec, orc_get_strict_math()); /* the reference keeps the float MATRIX only: Camera::rvec() (utils.h:49-53) is what the
                                                      * next mean shift starts from (:184) and what pose6() returns -- a round trip through float R */
    memcpy(cam->t, pose_opm + 3, 12);
    return 1;
}

/* voldor.cpp:164-201 */
static void v_optimize_cameras(orc_voldor_t* v) {
    const orc_config* c = &v->cfg;
    int allow_trunc = v->iters_cur > c->no_trunc_iters;
    const int npx = v->w * v->h;
    for (int i = 0; i < v->n_flows; i++) {
        double s = 0; /* cv::sum accumulates in double */
        for (int k = 0; k < npx; k++) s += v->rig[(size_t)i * npx + k];
        v->cams[i].pose_rigidness_density = (float)s / (float)npx;
        int ok = 0;
        if (!allow_trunc || v->cams[i].pose_rigidness_density > c->trunc_rigidness_density)
            ok = v_optimize_camera_pose(v, i, v->cams[i].pose_sample_count == 0 ? 0 : 1,
                                        c->rg_refine && (!c->rg_refine_last_only || v->iters_remain == 0));
        if (getenv("ORC_TRACE")) { /* the lines of Camera::print_info (utils.h:66-76), for side-by-side traces with a non-silent reference run */
            const orc_cam* q = &v->cams[i];
            fprintf(stderr, "orc: iter %d cam %d pool %d rigidness density %.6g pose density %.6g ms iters %d gu iters %d trans mag %.6g rot mag %.6g\n",
                    v->iters_cur, i, q->pose_sample_count, q->pose_rigidness_density, q->pose_density, q->last_used_ms_iters, q->last_used_gu_iters,
                    sqrt((double)q->t[0] * q->t[0] + (double)q->t[1] * q->t[1] + (double)q->t[2] * q->t[2]),
                    sqrt((double)q->rvec[0] * q->rvec[0] + (double)q->rvec[1] * q->rvec[1] + (double)q->rvec[2] * q->rvec[2]) * 180 / 3.14159);
        }
        if (!ok || (allow_trunc && v->cams[i].pose_density < c->trunc_sample_density)) {
            v->iters_remain = v->iters_remain > c->min_iters_after_trunc ? v->iters_remain : c->min_iters_after_trunc;
            v->n_flows = i;
            break;
        }
    }
}

/* voldor.cpp:309-317 */
static void v_normalize_world_scale(orc_voldor_t* v) {
    float world_scale = 0;
    for (int i = 0; i < v->n_flows; i++) {
        const float* t = v->cams[i].t;
        world_scale = (float)((double)world_scale + sqrt((double)t[0] * t[0] + (double)t[1] * t[1] + (double)t[2] * t[2])); /* float += double (cv::norm) */
    }
    const float s = v->n_flows / world_scale;
    for (int i = 0; i < v->n_flows; i++) for (int d = 0; d < 3; d++) v->cams[i].t[d] *= s;
    const int npx = v->w * v->h;
    for (int k = 0; k < npx; k++) v->depth[k] *= s;
}

/* ------------------------------------------------------------------ bootstrap
 * geometry.cpp:267-285 closed-form depth */
void orc_estimate_depth_closed_form(const float* flow, float* depth, const float* K,
                                    const float* R, const float* t, int w, int h,
                                    float min_depth, float max_depth) {
    /* b = K t ; KRKinv = K R K^-1 (float, cv::Mat products) */
    /* K.inv() (voldor.cpp:101): cv::invert of a 3x3 CV_32F goes through the adjugate with the determinant in double */
    float Kinv[9];
    {
        double m[9]; for (int i = 0; i < 9; i++) m[i] = K[i];
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        const double id = 1. / det;
        const double a[9] = { (m[4] * m[8] - m[5] * m[7]) * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id,
                              (m[5] * m[6] - m[3] * m[8]) * id, (m[0] * m[8] - m[2] * m[6]) * id, (m[2] * m[3] - m[0] * m[5]) * id,
                              (m[3] * m[7] - m[4] * m[6]) * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id };
        for (int i = 0; i < 9; i++) Kinv[i] = (float)a[i];
    }
    float KR[9], M[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += K[i * 3 + k] * R[k * 3 + j]; KR[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += KR[i * 3 + k] * Kinv[k * 3 + j]; M[i * 3 + j] = s; }
    float b1 = K[0] * t[0] + K[1] * t[1] + K[2] * t[2];
    float b2 = K[3] * t[0] + K[4] * t[1] + K[5] * t[2];
    float b3 = K[6] * t[0] + K[7] * t[1] + K[8] * t[2];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float dx = flow[(y * w + x) * 2], dy = flow[(y * w + x) * 2 + 1];
            float w1 = M[0] * x + M[1] * y + M[2], w2 = M[3] * x + M[4] * y + M[5], w3 = M[6] * x + M[7] * y + M[8];
            float a1 = x + dx, a2 = y + dy;
            float zn = (a1 * b3 - b1) * (w1 - a1 * w3) + (a2 * b3 - b2) * (w2 - a2 * w3);
            float zd = (w1 - a1 * w3) * (w1 - a1 * w3) + (w2 - a2 * w3) * (w2 - a2 * w3);
            depth[y * w + x] = fminf(fmaxf(zn / zd, min_depth), max_depth);
        }
}

/* Symmetric Jacobi eigen-decomposition (double, n<=9): A = V diag(e) V^T, ascending e. */
static void jacobi_eig(double* A, int n, double* V, double* e) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-30) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) e[i] = A[i * n + i];
    for (int i = 0; i < n; i++) { /* selection sort ascending */
        int m = i;
        for (int j = i + 1; j < n; j++) if (e[j] < e[m]) m = j;
        if (m != i) {
            double t = e[i]; e[i] = e[m]; e[m] = t;
            for (int k = 0; k < n; k++) { t = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = t; }
        }
    }
}
static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
/* SVD of a 3x3 (rank>=2): E = U diag(s) V^T, s descending, det(U)=det(V)=+1 */
static void svd3(const double* E, double* U, double* s, double* V) {
    double EtE[9], ev[3], Vv[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += E[k * 3 + i] * E[k * 3 + j]; EtE[i * 3 + j] = a; }
    jacobi_eig(EtE, 3, Vv, ev);
    for (int c = 0; c < 3; c++) { /* descending */
        for (int r = 0; r < 3; r++) V[r * 3 + c] = Vv[r * 3 + (2 - c)];
        s[c] = sqrt(ev[2 - c] > 0 ? ev[2 - c] : 0);
    }
    if (det3(V) < 0) for (int r = 0; r < 3; r++) V[r * 3 + 2] = -V[r * 3 + 2];
    double u[3][3];
    for (int c = 0; c < 2; c++) {
        double n = 0;
        for (int r = 0; r < 3; r++) { u[c][r] = E[r * 3 + 0] * V[0 * 3 + c] + E[r * 3 + 1] * V[1 * 3 + c] + E[r * 3 + 2] * V[2 * 3 + c]; n += u[c][r] * u[c][r]; }
        n = sqrt(n); if (n < 1e-300) n = 1;
        for (int r = 0; r < 3; r++) u[c][r] /= n;
    }
    { /* re-orthogonalise u1 against u0, u2 = u0 x u1 */
        double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2], n = 0;
        for (int r = 0; r < 3; r++) { u[1][r] -= d * u[0][r]; n += u[1][r] * u[1][r]; }
        n = sqrt(n); if (n < 1e-300) n = 1;
        for (int r = 0; r < 3; r++) u[1][r] /= n;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) U[r * 3 + c] = u[c][r];
}

/* Null vector of the 8x9 epipolar constraint matrix (8 correspondences -> the 9 entries of E up to scale) by
 * Gaussian elimination with complete pivoting in double: after 8 pivots the remaining column is the free one, it is
 * set to 1 and the others follow by back substitution.  A is destroyed.  (A minimal sample has an exact null space,
 * so this is the same vector as the smallest eigenvector of A^T A at a fraction of the work; a rank-deficient
 * sample gives a finite but meaningless vector that the LMedS score discards.) */
static void null9(double* A, double* f) {
    int perm[9];
    for (int c = 0; c < 9; c++) perm[c] = c;
    for (int k = 0; k < 8; k++) {
        int pr = k, pc = k; double best = -1.0;
        for (int r = k; r < 8; r++) for (int c = k; c < 9; c++) { double v = fabs(A[r * 9 + c]); if (v > best) { best = v; pr = r; pc = c; } }
        if (pr != k) for (int c = 0; c < 9; c++) { double t = A[k * 9 + c]; A[k * 9 + c] = A[pr * 9 + c]; A[pr * 9 + c] = t; }
        if (pc != k) { for (int r = 0; r < 8; r++) { double t = A[r * 9 + k]; A[r * 9 + k] = A[r * 9 + pc]; A[r * 9 + pc] = t; } int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
        double piv = A[k * 9 + k];
        if (!(fabs(piv) > 1e-300)) piv = 1e-300; /* rank deficient sample */
        for (int r = k + 1; r < 8; r++) {
            const double m = A[r * 9 + k] / piv;
            for (int c = k + 1; c < 9; c++) A[r * 9 + c] -= m * A[k * 9 + c];
        }
        A[k * 9 + k] = piv;
    }
    double x[9]; x[8] = 1.0;
    for (int k = 7; k >= 0; k--) {
        double sacc = 0;
        for (int c = k + 1; c < 9; c++) sacc += A[k * 9 + c] * x[c];
        x[k] = -sacc / A[k * 9 + k];
    }
    double nrm = 0;
    for (int c = 0; c < 9; c++) nrm += x[c] * x[c];
    nrm = sqrt(nrm);
    for (int c = 0; c < 9; c++) f[perm[c]] = x[c] / nrm;
}

static int cmp_double(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }

/* geometry.cpp:288-332 estimate_camera_pose_epipolar.  The reference calls OpenCV 3.4
 * findEssentialMat(LMEDS, 0.999, 1.0) [Nister 5-point inside an LMedS loop] + recoverPose, which
 * are not under /root/reference.  Deviation D5: the same LMedS principle is
 * restated with a normalised 8-point minimal solver, a fixed budget of 256 hypotheses, the
 * median of the squared Sampson distance over a <=2048-point scoring subset, then the
 * cheirality vote of recoverPose, then cam.t = R*t (:330).  Returns 1 on success. */
#define BOOT_HYPS 256
#define BOOT_SCORE_MAX 2048
static float g_last_two_view_t[3]; /* unit translation of the last call BEFORE cam.t = R*t, read by orc_last_two_view_translation() */
void orc_last_two_view_translation(float* t3) { memcpy(t3, g_last_two_view_t, sizeof g_last_two_view_t); }
int orc_estimate_pose_epipolar(const float* flow, const float* K, int w, int h, int step, float* R9, float* t3) {
    const int nx = (w + step - 1) / step, ny = (h + step - 1) / step, n = nx * ny;
    double* q1 = malloc(sizeof(double) * n * 2); double* q2 = malloc(sizeof(double) * n * 2);
    const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    int m = 0;
    for (int y = 0; y < h; y += step)
        for (int x = 0; x < w; x += step) {
            float x2 = x + flow[(y * w + x) * 2], y2 = y + flow[(y * w + x) * 2 + 1];
            q1[m * 2] = (x - cx) / fx; q1[m * 2 + 1] = (y - cy) / fy;
            q2[m * 2] = (x2 - cx) / fx; q2[m * 2 + 1] = (y2 - cy) / fy;
            m++;
        }
    if (n < 8) { free(q1); free(q2); return 0; }
    const int sstride = (n + BOOT_SCORE_MAX - 1) / BOOT_SCORE_MAX;
    const int ns = (n + sstride - 1) / sstride;
    double best_med = INFINITY, bestE[9] = { 0 };
    double* errs = malloc(sizeof(double) * ns);
    for (int hy = 0; hy < BOOT_HYPS; hy++) {
        double A[72];
        for (int k = 0; k < 8; k++) {
            int i = (int)(orc_rng(233u, (uint32_t)hy, 0x100u + (uint32_t)k) % (uint32_t)n);
            double a[9] = { q2[i * 2] * q1[i * 2], q2[i * 2] * q1[i * 2 + 1], q2[i * 2],
                            q2[i * 2 + 1] * q1[i * 2], q2[i * 2 + 1] * q1[i * 2 + 1], q2[i * 2 + 1],
                            q1[i * 2], q1[i * 2 + 1], 1.0 };
            for (int c = 0; c < 9; c++) A[k * 9 + c] = a[c];
        }
        double E0[9];
        null9(A, E0);
        double U[9], s[3], Vt[9], E[9];
        svd3(E0, U, s, Vt);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) /* E = U diag(1,1,0) V^T */
            E[r * 3 + c] = U[r * 3 + 0] * Vt[c * 3 + 0] + U[r * 3 + 1] * Vt[c * 3 + 1];
        for (int j = 0; j < ns; j++) {
            int i = j * sstride;
            double x1 = q1[i * 2], y1 = q1[i * 2 + 1], x2 = q2[i * 2], y2 = q2[i * 2 + 1];
            double Ex0 = E[0] * x1 + E[1] * y1 + E[2], Ex1 = E[3] * x1 + E[4] * y1 + E[5], Ex2 = E[6] * x1 + E[7] * y1 + E[8];
            double Et0 = E[0] * x2 + E[3] * y2 + E[6], Et1 = E[1] * x2 + E[4] * y2 + E[7];
            double num = x2 * Ex0 + y2 * Ex1 + Ex2;
            errs[j] = num * num / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1);
            if (!(errs[j] == errs[j])) errs[j] = INFINITY; /* a degenerate hypothesis ranks last */
        }
        qsort(errs, ns, sizeof(double), cmp_double);
        double med = errs[ns / 2];
        if (med < best_med) { best_med = med; memcpy(bestE, E, sizeof E); }
    }
    free(errs);
    /* recoverPose: decompose, cheirality vote */
    double U[9], s[3], V[9];
    svd3(bestE, U, s, V);
    const double W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    double Rc[2][9];
    for (int k = 0; k < 2; k++) { /* R = U W V^T , U W^T V^T */
        double UW[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int j = 0; j < 3; j++) a += U[r * 3 + j] * (k == 0 ? W[j * 3 + c] : W[c * 3 + j]); UW[r * 3 + c] = a; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int j = 0; j < 3; j++) a += UW[r * 3 + j] * V[c * 3 + j]; Rc[k][r * 3 + c] = a; }
    }
    double tc[3] = { U[2], U[5], U[8] };
    int best = 0, best_cnt = -1;
    for (int cand = 0; cand < 4; cand++) {
        const double* R = Rc[cand >> 1];
        double sg = (cand & 1) ? -1.0 : 1.0;
        double t[3] = { sg * tc[0], sg * tc[1], sg * tc[2] };
        int cnt = 0;
        for (int j = 0; j < ns; j++) {
            int i = j * sstride;
            double a[3] = { q1[i * 2], q1[i * 2 + 1], 1 }, b[3] = { q2[i * 2], q2[i * 2 + 1], 1 };
            double Ra[3] = { R[0] * a[0] + R[1] * a[1] + R[2] * a[2], R[3] * a[0] + R[4] * a[1] + R[5] * a[2], R[6] * a[0] + R[7] * a[1] + R[8] * a[2] };
            /* z1*Ra + t = z2*b  -> least squares for (z1,z2) */
            double A11 = Ra[0] * Ra[0] + Ra[1] * Ra[1] + Ra[2] * Ra[2], A12 = -(Ra[0] * b[0] + Ra[1] * b[1] + Ra[2] * b[2]);
            double A22 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
            double r1 = -(Ra[0] * t[0] + Ra[1] * t[1] + Ra[2] * t[2]), r2 = b[0] * t[0] + b[1] * t[1] + b[2] * t[2];
            double det = A11 * A22 - A12 * A12;
            if (fabs(det) < 1e-12) continue;
            double z1 = (r1 * A22 - A12 * r2) / det, z2 = (A11 * r2 - A12 * r1) / det;
            if (z1 > 0 && z2 > 0) cnt++;
        }
        if (cnt > best_cnt) { best_cnt = cnt; best = cand; }
    }
    const double* R = Rc[best >> 1];
    double sg = (best & 1) ? -1.0 : 1.0;
    float Rf[9], tf[3] = { (float)(sg * tc[0]), (float)(sg * tc[1]), (float)(sg * tc[2]) };
    for (int i = 0; i < 9; i++) Rf[i] = (float)R[i];
    memcpy(R9, Rf, sizeof Rf);
    memcpy(g_last_two_view_t, tf, sizeof tf);
    for (int r = 0; r < 3; r++) t3[r] = Rf[r * 3] * tf[0] + Rf[r * 3 + 1] * tf[1] + Rf[r * 3 + 2] * tf[2]; /* cam.t = R*t :330 */
    free(q1); free(q2);
    return 1;
}

/* ------------------------------------------------------------------ py_voldor_wrapper */
int orc_voldor(const float* flows, const float* disparity, const float* disparity_pconf,
               const float* depth_priors, const float* depth_prior_poses,
               const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
               float basefocal, int N, int N_dp_in, int w, int h, const char* config,
               int* n_registered, float* poses, float* poses_covar, float* depth_out,
               float* depth_conf) {
    orc_voldor_t* v = calloc(1, sizeof *v);
    cfg_defaults(&v->cfg);
    v->cfg.fx = fx; v->cfg.cx = cx; v->cfg.fy = fy; v->cfg.cy = cy; v->cfg.basefocal = basefocal;
    int rc = cfg_parse(&v->cfg, config);
    if (rc) { free(v); return 100 + rc; }
    if (v->cfg.resize_factor != 1.f) { free(v); return 110; } /* deprecated path (voldor.cpp:24-27) not restated */
    if (N > ORC_MAX_FRAMES || N_dp_in + (disparity ? 1 : 0) > ORC_MAX_FRAMES || N < 1) { free(v); return 111; }
    const int npx = w * h;
    const orc_config* c = &v->cfg;
    /* ---- init: voldor.cpp:4-128 ---- */
    v->w = w; v->h = h; v->n_flows = v->n_flows_init = N; v->flows = flows;
    v->iters_cur = 0; v->iters_remain = c->max_iters;
    v->n_dp = N_dp_in + (disparity ? 1 : 0);
    v->has_disparity = disparity != NULL;
    v->depth = malloc(sizeof(float) * npx); v->cost = malloc(sizeof(float) * npx);
    v->rig = malloc(sizeof(float) * (size_t)npx * N);
    for (size_t k = 0; k < (size_t)npx * N; k++) v->rig[k] = 1.f;
    if (v->n_dp > 0) {
        v->priors = malloc(sizeof(float) * (size_t)npx * v->n_dp);
        v->pconfs = malloc(sizeof(float) * (size_t)npx * v->n_dp);
        v->confs = malloc(sizeof(float) * (size_t)npx * v->n_dp);
        for (size_t k = 0; k < (size_t)npx * v->n_dp; k++) v->confs[k] = 1.f;
    }
    int o = 0;
    if (disparity) {
        for (int k = 0; k < npx; k++) v->priors[k] = c->basefocal / disparity[k]; /* :33 */
        for (int k = 0; k < npx; k++) v->pconfs[k] = disparity_pconf ? disparity_pconf[k] : 1.f;
        cam_init(&v->dp_poses[0]);
        o = 1;
    }
    for (int i = 0; i < N_dp_in; i++) {
        memcpy(v->priors + (size_t)(o + i) * npx, depth_priors + (size_t)i * npx, sizeof(float) * npx);
        if (depth_prior_pconfs) memcpy(v->pconfs + (size_t)(o + i) * npx, depth_prior_pconfs + (size_t)i * npx, sizeof(float) * npx);
        else for (int k = 0; k < npx; k++) v->pconfs[(size_t)(o + i) * npx + k] = 1.f;
        cam_init(&v->dp_poses[o + i]);
        memcpy(v->dp_poses[o + i].rvec, depth_prior_poses + i * 6, 12);
        orc_rvec_to_rotmat(v->dp_poses[o + i].rvec, v->dp_poses[o + i].R);
        memcpy(v->dp_poses[o + i].t, depth_prior_poses + i * 6 + 3, 12);
    }
    for (int i = 0; i < N; i++) cam_init(&v->cams[i]);
    if (v->n_dp > 0) { /* :106-117 */
        memcpy(v->depth, v->priors, sizeof(float) * npx);
        if (!disparity) v_optimize_depth(v, OD_ONLY_USE_DEPTH_PRIOR);
    } else
        for (int k = 0; k < npx; k++) v->depth[k] = 1.f;

    /* ---- solve: voldor.cpp:130-149 ---- */
    if (v->n_dp == 0) { /* bootstrap :151-162 */
        float K[9] = { c->fx, 0, c->cx, 0, c->fy, c->cy, 0, 0, 1 };
        orc_cam* cam = &v->cams[0];
        if (orc_estimate_pose_epipolar(flows, K, w, h, 4, cam->R, cam->t)) vrcv_rvec_of_R32(cam->R, cam->rvec, orc_get_strict_math());
        orc_estimate_depth_closed_form(flows, v->depth, K, cam->R, cam->t, w, h, 1e-2f, 1e10f);
    }
    while (v->iters_remain > 0 && v->n_flows > 0) {
        v->iters_cur++; v->iters_remain--;
        v_optimize_cameras(v);
        v_optimize_depth(v, c->optimize_depth ? OD_DEFAULT : OD_UPDATE_RIGIDNESS_ONLY);
        /* the reference also runs this with n_flows==0 (0/0 -> NaN depth on a lost window): guarded, D6 */
        if (c->norm_world_scale && v->n_dp == 0 && v->n_flows > 0) v_normalize_world_scale(v);
    }
    /* ---- outputs: py_export.cpp:56-76 ---- */
    *n_registered = v->n_flows;
    for (int i = 0; i < v->n_flows; i++) {
        if (poses) { memcpy(poses + i * 6, v->cams[i].rvec, 12); memcpy(poses + i * 6 + 3, v->cams[i].t, 12); }
        if (poses_covar) memcpy(poses_covar + i * 36, v->cams[i].pose_covar, sizeof(float) * 36);
    }
    if (depth_out) memcpy(depth_out, v->depth, sizeof(float) * npx);
    if (depth_conf) {
        for (int k = 0; k < npx; k++) {
            float s = 0;
            for (int i = 0; i < v->n_flows; i++) s += v->rig[(size_t)i * npx + k];
            for (int i = 0; i < v->n_dp; i++) s += v->confs[(size_t)i * npx + k];
            depth_conf[k] = s * cv_div_scale((float)(v->n_flows + v->n_dp)); /* Mat /= : py_export.cpp:74 */
        }
    }
    free(v->od_depth); free(v->depth); free(v->cost); free(v->rig); free(v->priors); free(v->pconfs); free(v->confs); free(v);
    return 0;
}
