/* oracle/orc_model.c -- ORACLE (test infrastructure only, see orc.h).
 * Depth half of the EM loop: residual model, pixel cost, PatchMatch-style depth search,
 * rigidness E-step, forward-backward smoothing, Gaussian blur.
 * Restated from gpu-kernels/{residual_model.h,optimize_depth.cu,fb_smooth.h,gblur.cu};
 * citations inline (file:line relative to /root/reference). */
#include "orc.h"
#include "orc_math.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ZDE FLT_EPSILON /* gpu-kernels/utils.h:19 */

int orc_strict_math_flag = 0;
void orc_set_strict_math(int on) { orc_strict_math_flag = on ? 1 : 0; }
int orc_get_strict_math(void) { return orc_strict_math_flag; }

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ RNG (deviation D1)
 * The reference uses per-pixel cuRAND XORWOW states (optimize_depth.cu:273,290; seed 233,
 * subsequence = pixel index).  cuRAND is not reproducible here, so product and oracle share
 * this SPEC instead: a counter-based generator keyed by (seed, stream, counter). */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
uint32_t orc_rng(uint32_t seed, uint32_t stream, uint32_t counter) {
    uint32_t h = fmix32(seed ^ 0x9E3779B9u);
    h = fmix32(h ^ stream);
    h = fmix32(h + counter * 0x9E3779B1u + 0x7F4A7C15u);
    return h;
}
float orc_u01(uint32_t r) { return (float)((r >> 8) + 1u) * (1.0f / 16777216.0f); }

/* ---- reference mode, round 4: D1 and D2 switched back to CUDA's published behaviour (voldor_amd/csrc/vk_ref_cuda.h, shared with the
 * product and with the launch-emulation layer of the reference build).  orc_set_reference_rng(1): cuRAND XORWOW streams as the
 * reference seeds and advances them; orc_set_reference_tex(1): 8-bit-fraction linear filtering over the STACKED layers. */
#include "../voldor_amd/csrc/vk_ref_cuda.h"
static int g_ref_rng = 0, g_ref_tex = 0;
void orc_set_reference_rng(int on) { g_ref_rng = on ? 1 : 0; }
void orc_set_reference_tex(int on) { g_ref_tex = on ? 1 : 0; }
int orc_get_reference_rng(void) { return g_ref_rng; }
int orc_get_reference_tex(void) { return g_ref_tex; }
static uint32_t* g_xw_J = NULL;  /* [32][VRC_XW_MAT] sequence jumps, built once */
const uint32_t* orc_xorwow_jumps(void) {
    if (!g_xw_J) {
#pragma omp critical(orc_xw_jumps)
        if (!g_xw_J) {
            uint32_t* J = (uint32_t*)malloc(sizeof(uint32_t) * 32 * VRC_XW_MAT);
            vrc_build_sequence_jumps(J);
            g_xw_J = J;
        }
    }
    return g_xw_J;
}
/* test access: state after curand_init(seed, sub, 0) and the first n outputs of curand_uniform / curand */
void orc_xorwow_stream(unsigned long long seed, uint32_t sub, int n, uint32_t* raw, float* uni, uint32_t* state6) {
    vrc_xorwow s;
    vrc_xorwow_init(orc_xorwow_jumps(), seed, sub, &s);
    if (state6) { for (int k = 0; k < 5; k++) state6[k] = s.v[k]; state6[5] = s.d; }
    for (int i = 0; i < n; i++) { const uint32_t x = vrc_xorwow_next(&s); if (raw) raw[i] = x; if (uni) uni[i] = vrc_uniform(x); }
}
/* per-pixel states of the depth sampling (optimize_depth.cu:286-291: initialised when the size changes, then one draw per pixel per
 * sample launch, persisting): a cache that stands `epoch` draws after curand_init; rebuilt when the caller's epoch does not continue it */
static vrc_xorwow* g_px_states = NULL;
static int g_px_n = 0;
static uint32_t g_px_epoch = 0;
static vrc_xorwow* pixel_states(int npx, uint32_t epoch) {
    if (g_px_n != npx || g_px_epoch != epoch || !g_px_states) {
        const uint32_t* J = orc_xorwow_jumps();
        free(g_px_states);
        g_px_states = (vrc_xorwow*)malloc(sizeof(vrc_xorwow) * (size_t)npx);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < npx; i++) {
            vrc_xorwow_init(J, 233ull, (uint32_t)i, &g_px_states[i]);
            for (uint32_t e = 0; e < epoch; e++) (void)vrc_xorwow_next(&g_px_states[i]);
        }
        g_px_n = npx; g_px_epoch = epoch;
    }
    return g_px_states;
}

/* ------------------------------------------------------------------ residual model
 * residual_model.h:6-13 */
#define EST_RF 0.5
#define FISK_A1 0.01f
#define FISK_A2 0.09f
#define FISK_B1 1.0f
#define FISK_B2 -0.0022f
#define MIN_OBS_FMAG 2.f
#define MAX_OBS_FMAG 100.f

/* residual_model.h:15-19.  NB `fmag*EST_RF` is float*double -> double, then fmaxf truncates
 * to float: restated literally. */
float orc_fun_fmag_c(float fmag) {
    fmag = fminf(fmaxf((float)(fmag * EST_RF), MIN_OBS_FMAG), MAX_OBS_FMAG);
    return FISK_B1 + FISK_B2 * fmag;
}
/* residual_model.h:21-25 */
float orc_fun_fmag_scale(float fmag) {
    fmag = fminf(fmaxf((float)(fmag * EST_RF), MIN_OBS_FMAG), MAX_OBS_FMAG);
    return FISK_A1 * om_expf(FISK_A2 * fmag);
}
/* residual_model.h:28-31 */
float orc_fisk_dist_pdf(float x, float c, float scale) {
    x = fmaxf((float)(x * EST_RF), ZDE);
    return (c * om_powf((x * x) / scale, -c - 1.f) * om_powf(1 + om_powf((x * x) / scale, -c), -2.f)) / scale;
}
static inline float l2norm(float x, float y) { return sqrtf(x * x + y * y); }
/* residual_model.h:34-42 */
float orc_fun_rigidness(float dx1, float dy1, float dx2, float dy2, float lambda, float abs_rf) {
    const float obs_fmag = l2norm(dx2, dy2) / abs_rf;
    const float diff_fmag = l2norm(dx1 - dx2, dy1 - dy2) / abs_rf;
    const float c = orc_fun_fmag_c(obs_fmag);
    const float s = orc_fun_fmag_scale(obs_fmag);
    const float fisk_prob = orc_fisk_dist_pdf(diff_fmag, c, s);
    const float mu = orc_fisk_dist_pdf(lambda * obs_fmag, c, s);
    return fisk_prob / (fisk_prob + mu);
}
/* residual_model.h:45-49 */
static inline void fun_cost(float dx1, float dy1, float dx2, float dy2, float weight,
                            float* io_cost, float* io_wsum, float lambda, float abs_rf) {
    *io_cost -= weight * om_logf(orc_fun_rigidness(dx1, dy1, dx2, dy2, lambda, abs_rf));
    *io_wsum += weight;
}
/* residual_model.h:51-61 */
float orc_fun_depth_rigidness(float d1, float d2, float basefocal, float omega, float abs_rf) {
    const float disp1 = (basefocal / d1) / abs_rf;
    const float disp2 = (basefocal / d2) / abs_rf;
    const float obs_disp = disp2;
    const float diff_disp = fabsf(disp1 - disp2);
    const float c = orc_fun_fmag_c(obs_disp);
    const float s = orc_fun_fmag_scale(obs_disp);
    const float fisk_prob = orc_fisk_dist_pdf(diff_disp, c, s);
    const float mu = orc_fisk_dist_pdf(omega * obs_disp, c, s);
    return fisk_prob / (fisk_prob + mu);
}
/* residual_model.h:64-68 */
static inline void fun_depth_cost(float d1, float d2, float basefocal, float weight,
                                  float* io_cost, float* io_wsum, float omega, float abs_rf) {
    *io_cost -= weight * om_logf(orc_fun_depth_rigidness(d1, d2, basefocal, omega, abs_rf));
    *io_wsum += weight;
}

/* ------------------------------------------------------------------ geometry helpers
 * optimize_depth.cu:54-81; K4 = fx,cx,fy,cy ; K4inv = 1/fx,-cx/fx,1/fy,-cy/fy (:346-347) */
typedef struct { float K4[4], K4i[4]; } k4_t;
static k4_t make_k4(const float* K) {
    k4_t k;
    k.K4[0] = K[0]; k.K4[1] = K[2]; k.K4[2] = K[4]; k.K4[3] = K[5];
    k.K4i[0] = 1.f / K[0]; k.K4i[1] = -K[2] / K[0]; k.K4i[2] = 1.f / K[4]; k.K4i[3] = -K[5] / K[4];
    return k;
}
static inline void p2_to_p3(const k4_t* k, float px, float py, float d, float* o) {
    o[0] = (k->K4i[0] * px + k->K4i[1]) * d;
    o[1] = (k->K4i[2] * py + k->K4i[3]) * d;
    o[2] = d;
}
static inline void p3_to_p2(const k4_t* k, const float* o, float* px, float* py) {
    *px = (k->K4[0] * o[0] + k->K4[1] * o[2]) / o[2];
    *py = (k->K4[2] * o[1] + k->K4[3] * o[2]) / o[2];
}
static inline void trans_p3(float* o, const float* R, const float* t) {
    float x = o[0] * R[0] + o[1] * R[1] + o[2] * R[2];
    float y = o[0] * R[3] + o[1] * R[4] + o[2] * R[5];
    float z = o[0] * R[6] + o[1] * R[7] + o[2] * R[8];
    o[0] = x + t[0]; o[1] = y + t[1]; o[2] = z + t[2];
}

/* Bilinear fetch of one layer (deviation D2): CUDA formula
 * (1-a)(1-b)T00 + a(1-b)T10 + (1-a)bT01 + abT11 with exact fp32 fractions and per-layer
 * clamping.  gmat.h:175-179 samples at (x+.5, d*h+y+.5) of the stacked texture. */
static inline void bil_idx(float x, float y, int w, int h, int* x0, int* x1, int* y0, int* y1,
                           float* a, float* b) {
    float fx = floorf(x), fy = floorf(y);
    *a = x - fx; *b = y - fy;
    int ix = (int)fx, iy = (int)fy;
    int ix1 = ix + 1, iy1 = iy + 1;
    if (ix < 0) ix = 0; if (ix > w - 1) ix = w - 1;
    if (ix1 < 0) ix1 = 0; if (ix1 > w - 1) ix1 = w - 1;
    if (iy < 0) iy = 0; if (iy > h - 1) iy = h - 1;
    if (iy1 < 0) iy1 = 0; if (iy1 > h - 1) iy1 = h - 1;
    *x0 = ix; *x1 = ix1; *y0 = iy; *y1 = iy1;
}
float orc_bilinear1(const float* img, int w, int h, float x, float y) {
    int x0, x1, y0, y1; float a, b;
    bil_idx(x, y, w, h, &x0, &x1, &y0, &y1, &a, &b);
    float t00 = img[y0 * w + x0], t10 = img[y0 * w + x1], t01 = img[y1 * w + x0], t11 = img[y1 * w + x1];
    return (1.f - a) * (1.f - b) * t00 + a * (1.f - b) * t10 + (1.f - a) * b * t01 + a * b * t11;
}
void orc_bilinear2(const float* img, int w, int h, float x, float y, float* ox, float* oy) {
    int x0, x1, y0, y1; float a, b;
    bil_idx(x, y, w, h, &x0, &x1, &y0, &y1, &a, &b);
    const float* p00 = img + 2 * (y0 * w + x0); const float* p10 = img + 2 * (y0 * w + x1);
    const float* p01 = img + 2 * (y1 * w + x0); const float* p11 = img + 2 * (y1 * w + x1);
    float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    *ox = w00 * p00[0] + w10 * p10[0] + w01 * p01[0] + w11 * p11[0];
    *oy = w00 * p00[1] + w10 * p10[1] + w01 * p01[1] + w11 * p11[1];
}

/* at_tex of layer f of a stack of n_layers (gmat.h:175-179): D2's exact per-layer bilinear, or CUDA's filter over the stack */
float orc_fetch1(const float* stack, int f, int n_layers, int w, int h, float x, float y) {
    if (g_ref_tex) return vrc_tex_fetch1(stack, x, y, f, w, h, n_layers);
    return orc_bilinear1(stack + (size_t)f * w * h, w, h, x, y);
}
void orc_fetch2(const float* stack, int f, int n_layers, int w, int h, float x, float y, float* ox, float* oy) {
    if (g_ref_tex) { vrc_tex_fetch2(stack, x, y, f, w, h, n_layers, ox, oy); return; }
    orc_bilinear2(stack + (size_t)f * w * h * 2, w, h, x, y, ox, oy);
}

/* ------------------------------------------------------------------ pixel cost
 * optimize_depth.cu:140-198 */
typedef struct {
    const orc_od_params* p; k4_t k;
    const float* flows; const float* rig; const float* priors; const float* pconfs; const float* confs;
} cost_ctx;

static float pixel_cost(const cost_ctx* c, int px, int py, float depth) {
    const orc_od_params* p = c->p;
    const int w = p->w, h = p->h, npx = w * h;
    float cost_sum = 0, wsum = 0;
    float o[3], px1, py1, px2, py2;
    p2_to_p3(&c->k, (float)px, (float)py, depth, o);
    px1 = (float)px; py1 = (float)py;
    for (int f = 0; f < p->N; f++) {
        trans_p3(o, p->Rs[f], p->ts[f]);
        p3_to_p2(&c->k, o, &px2, &py2);
        if (o[2] > 0 && px1 >= 0 && px1 < w && py1 >= 0 && py1 < h) {
            float d2x, d2y;
            orc_fetch2(c->flows, f, p->N, w, h, px1, py1, &d2x, &d2y);
            float dx1 = px2 - px1, dy1 = py2 - py1;
            px1 = px2; py1 = py2;
            fun_cost(dx1, dy1, d2x, d2y, c->rig[(size_t)f * npx + py * w + px], &cost_sum, &wsum,
                     p->lambda, p->abs_resize_factor);
        }
    }
    for (int f = 0; f < p->N_dp; f++) {
        p2_to_p3(&c->k, (float)px, (float)py, depth, o);
        trans_p3(o, p->dp_Rs[f], p->dp_ts[f]);
        p3_to_p2(&c->k, o, &px1, &py1);
        if (o[2] > 0 && px1 >= 0 && px1 < w && py1 >= 0 && py1 < h) {
            float td = orc_fetch1(c->priors, f, p->N_dp, w, h, px1, py1);
            float tpc = orc_fetch1(c->pconfs, f, p->N_dp, w, h, px1, py1);
            float tc = orc_fetch1(c->confs, f, p->N_dp, w, h, px1, py1);
            if (td > 0) {
                if (p->disp_delta > 0 && f == 0)
                    fun_depth_cost(o[2], td, p->basefocal, tpc * tc * p->disp_delta, &cost_sum, &wsum,
                                   p->omega, p->abs_resize_factor);
                else
                    fun_depth_cost(o[2], td, p->basefocal, tpc * tc * p->delta, &cost_sum, &wsum,
                                   p->omega, p->abs_resize_factor);
            }
        }
    }
    if (wsum == 0) return INFINITY;
    return cost_sum / fmaxf(wsum, ZDE);
}

/* optimize_depth.cu:201-207 */
static inline void replace_if_better(const cost_ctx* c, int px, int py, float depth_new,
                                     float* o_depth, float* io_cost) {
    float cost = pixel_cost(c, px, py, depth_new);
    if (cost < *io_cost) { *o_depth = depth_new; *io_cost = cost; }
}

void orc_compute_cost_map(const orc_od_params* p, const float* flows, const float* rig,
                          const float* priors, const float* pconfs, const float* confs,
                          const float* depth, float* cost) {
    cost_ctx c = { p, make_k4(p->K), flows, rig, priors, pconfs, confs };
    const int w = p->w, h = p->h;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            cost[y * w + x] = pixel_cost(&c, x, y, depth[y * w + x]); /* :279-284 */
}

/* ------------------------------------------------------------------ E-step
 * optimize_depth.cu:84-138 */
void orc_update_rigidnesses(const orc_od_params* p, const float* flows, float* rig,
                            const float* priors, float* confs, const float* depth) {
    const k4_t k = make_k4(p->K);
    const int w = p->w, h = p->h, npx = w * h;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float o[3], px1, py1, px2, py2;
            p2_to_p3(&k, (float)x, (float)y, depth[y * w + x], o);
            px1 = (float)x; py1 = (float)y;
            for (int f = 0; f < p->N; f++) {
                trans_p3(o, p->Rs[f], p->ts[f]);
                p3_to_p2(&k, o, &px2, &py2);
                if (o[2] > 0 && px1 >= 0 && px1 < w && py1 >= 0 && py1 < h) {
                    float d2x, d2y;
                    orc_fetch2(flows, f, p->N, w, h, px1, py1, &d2x, &d2y);
                    float dx1 = px2 - px1, dy1 = py2 - py1;
                    px1 = px2; py1 = py2;
                    rig[(size_t)f * npx + y * w + x] =
                        orc_fun_rigidness(dx1, dy1, d2x, d2y, p->lambda, p->abs_resize_factor);
                } else {
                    rig[(size_t)f * npx + y * w + x] = 0; /* px1 NOT advanced (SURVEY B-10) */
                }
            }
            for (int f = 0; f < p->N_dp; f++) {
                p2_to_p3(&k, (float)x, (float)y, depth[y * w + x], o);
                trans_p3(o, p->dp_Rs[f], p->dp_ts[f]);
                p3_to_p2(&k, o, &px1, &py1);
                if (o[2] > 0 && px1 >= 0 && px1 < w && py1 >= 0 && py1 < h) {
                    float td = orc_fetch1(priors, f, p->N_dp, w, h, px1, py1);
                    if (td > 0) /* else: conf left untouched (:129) */
                        confs[(size_t)f * npx + y * w + x] =
                            orc_fun_depth_rigidness(o[2], td, p->basefocal, p->omega, p->abs_resize_factor);
                } else {
                    confs[(size_t)f * npx + y * w + x] = 0;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ forward-backward smoothing
 * fb_smooth.h:26-70 (kernels), :89-106 (rows, posterior, columns, posterior) */
static void fb_line(float* e1, int n, int stride, float e0, float p, float* F, float* B) {
    float prev = e1[0], s0, s1;
    for (int i = 0; i < n; i++) { /* FB_MSG_L2R / T2B :27-36, :47-55 */
        s0 = (prev * (1.f - p) + (1.f - prev) * p) * e0;
        s1 = (prev * p + (1.f - prev) * (1 - p)) * e1[i * stride];
        prev = s1 / (s0 + s1);
        F[i] = prev;
    }
    prev = e1[(n - 1) * stride];
    for (int i = n - 1; i >= 0; i--) { /* FB_MSG_R2L / B2T :37-46, :56-64 */
        s0 = prev * e1[i * stride] * (1.f - p) + (1.f - prev) * p * e0;
        s1 = prev * e1[i * stride] * p + (1.f - prev) * (1.f - p) * e0;
        prev = s1 / (s0 + s1);
        B[i] = prev;
    }
    for (int i = 0; i < n; i++) { /* FB_POSTERIOR :65-69 */
        s0 = (1.f - F[i]) * (1.f - B[i]);
        s1 = F[i] * B[i];
        e1[i * stride] = s1 / (s0 + s1);
    }
}
void orc_fb_smooth(float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob) {
    const int n = w > h ? w : h;
#pragma omp parallel
    {
        float* F = (float*)malloc(sizeof(float) * n * 2);
        float* B = F + n;
#pragma omp for schedule(static)
        for (int i = 0; i < n_maps * h; i++) { /* rows */
            int d = i / h, y = i % h;
            fb_line(maps + (size_t)d * w * h + (size_t)y * w, w, 1, s0_ems_prob, no_change_prob, F, B);
        }
#pragma omp for schedule(static)
        for (int i = 0; i < n_maps * w; i++) { /* columns, on the row-smoothed result */
            int d = i / w, x = i % w;
            fb_line(maps + (size_t)d * w * h + x, h, w, s0_ems_prob, no_change_prob, F, B);
        }
        free(F);
    }
}

/* ------------------------------------------------------------------ depth search passes */
#define MAXIMUM_DEPTH 1e5f /* optimize_depth.cu:15 */

/* optimize_depth.cu:269-277 with deviation D1 for the uniform */
static void pass_rand(const cost_ctx* c, float* depth, float* cost, uint32_t epoch) {
    const int w = c->p->w, h = c->p->h;
    vrc_xorwow* states = g_ref_rng ? pixel_states(w * h, epoch) : NULL;  /* curand_uniform(&_d_rand_states.at(x, y)) */
    if (states) g_px_epoch = epoch + 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float u = states ? vrc_uniform(vrc_xorwow_next(&states[y * w + x])) : orc_u01(orc_rng(233u, (uint32_t)(y * w + x), epoch));
            float depth_rnd = 1.0f / (c->p->range_factor * u + (1.0f / MAXIMUM_DEPTH));
            replace_if_better(c, x, y, depth_rnd, &depth[y * w + x], &cost[y * w + x]);
        }
}
/* optimize_depth.cu:209-235; dir 0 L2R, 1 T2B, 2 R2L, 3 B2T (:10-13) */
static void pass_global(const cost_ctx* c, float* depth, float* cost, int dir, int step) {
    const int w = c->p->w, h = c->p->h;
    if (dir == 0 || dir == 2) {
#pragma omp parallel for schedule(static)
        for (int ty = 0; ty < h; ty++) {
            if (dir == 0)
                for (int x = 1; x < w; x += step)
                    replace_if_better(c, x, ty, depth[ty * w + x - 1], &depth[ty * w + x], &cost[ty * w + x]);
            else
                for (int x = w - 2; x >= 0; x -= step)
                    replace_if_better(c, x, ty, depth[ty * w + x + 1], &depth[ty * w + x], &cost[ty * w + x]);
        }
    } else {
#pragma omp parallel for schedule(static)
        for (int tx = 0; tx < w; tx++) {
            if (dir == 1)
                for (int y = 1; y < h; y += step)
                    replace_if_better(c, tx, y, depth[(y - 1) * w + tx], &depth[y * w + tx], &cost[y * w + tx]);
            else
                for (int y = h - 2; y >= 0; y -= step)
                    replace_if_better(c, tx, y, depth[(y + 1) * w + tx], &depth[y * w + tx], &cost[y * w + tx]);
        }
    }
}
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
/* optimize_depth.cu:237-267 */
static void pass_local(const cost_ctx* c, float* depth, float* cost, int dir, int width) {
    const int w = c->p->w, h = c->p->h;
    if (dir == 0 || dir == 2) {
        const int nseg = (w + width - 1) / width;
#pragma omp parallel for schedule(static)
        for (int ty = 0; ty < h; ty++)
            for (int s = 0; s < nseg; s++) {
                int px = s * width;
                if (dir == 0)
                    for (int x = imax(1, px + 1); x < imin(w, px + width); x++)
                        replace_if_better(c, x, ty, depth[ty * w + x - 1], &depth[ty * w + x], &cost[ty * w + x]);
                else
                    for (int x = imin(w - 2, px + width - 2); x >= imax(0, px); x--)
                        replace_if_better(c, x, ty, depth[ty * w + x + 1], &depth[ty * w + x], &cost[ty * w + x]);
            }
    } else {
        const int nseg = (h + width - 1) / width;
#pragma omp parallel for schedule(static)
        for (int tx = 0; tx < w; tx++)
            for (int s = 0; s < nseg; s++) {
                int py = s * width;
                if (dir == 1)
                    for (int y = imax(1, py + 1); y < imin(h, py + width); y++)
                        replace_if_better(c, tx, y, depth[(y - 1) * w + tx], &depth[y * w + tx], &cost[y * w + tx]);
                else
                    for (int y = imin(h - 2, py + width - 2); y >= imax(0, py); y--)
                        replace_if_better(c, tx, y, depth[(y + 1) * w + tx], &depth[y * w + tx], &cost[y * w + tx]);
            }
    }
}

/* optimize_depth.cu:462-494 stage order */
void orc_optimize_depth(const orc_od_params* p, const float* flows, float* rig,
                        const float* priors, const float* pconfs, float* confs,
                        float* depth, float* cost, uint32_t* rand_epoch) {
    if (!p->update_rigidness_only) {
        if (p->fb_smooth) {
            if (p->N > 0) orc_fb_smooth(rig, p->N, p->w, p->h, p->s0_ems_prob, p->no_change_prob);
            if (p->N_dp > 0) orc_fb_smooth(confs, p->N_dp, p->w, p->h, p->s0_ems_prob, p->no_change_prob);
        }
        cost_ctx c = { p, make_k4(p->K), flows, rig, priors, pconfs, confs };
        orc_compute_cost_map(p, flows, rig, priors, pconfs, confs, depth, cost);
        for (int it = 0; it < p->n_rand_samples; it++) pass_rand(&c, depth, cost, (*rand_epoch)++);
        if (p->global_prop_step > 0) { /* :480-485 L2R,B2T,R2L,T2B */
            pass_global(&c, depth, cost, 0, p->global_prop_step);
            pass_global(&c, depth, cost, 3, p->global_prop_step);
            pass_global(&c, depth, cost, 2, p->global_prop_step);
            pass_global(&c, depth, cost, 1, p->global_prop_step);
        }
        if (p->local_prop_width > 0) { /* :486-491 */
            pass_local(&c, depth, cost, 0, p->local_prop_width);
            pass_local(&c, depth, cost, 3, p->local_prop_width);
            pass_local(&c, depth, cost, 2, p->local_prop_width);
            pass_local(&c, depth, cost, 1, p->local_prop_width);
        }
    }
    orc_update_rigidnesses(p, flows, rig, priors, confs, depth); /* :494 */
}

/* ------------------------------------------------------------------ gblur
 * gblur.cu:12-72: vertical then horizontal, half kernel exp(-i^2/2s^2), border renormalised */
int orc_gblur(const float* src, float* dst, int w, int h, int d, float sigma, int ksize) {
    if (ksize == 0) { ksize = (int)ceilf(6 * sigma); if (ksize < 3) ksize = 3; }
    int half = ksize / 2 + 1;
    if (half > 128) return 1;
    float g[128];
    for (int i = 0; i < half; i++) g[i] = expf(-(float)(i * i) / (float)(2 * sigma * sigma));
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h * d);
    for (int pass = 0; pass < 2; pass++) {
        const float* s = pass == 0 ? src : tmp;
        float* o = pass == 0 ? tmp : dst;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < d * h; i++) {
            int z = i / h, y = i % h;
            const float* sl = s + (size_t)z * w * h;
            for (int x = 0; x < w; x++) {
                float sum = g[0] * sl[y * w + x], sw = g[0];
                for (int k = 1; k < half; k++) {
                    if (pass == 1) { /* horizontal */
                        if (x + k < w) { sum += g[k] * sl[y * w + x + k]; sw += g[k]; }
                        if (x - k >= 0) { sum += g[k] * sl[y * w + x - k]; sw += g[k]; }
                    } else {
                        if (y + k < h) { sum += g[k] * sl[(y + k) * w + x]; sw += g[k]; }
                        if (y - k >= 0) { sum += g[k] * sl[(y - k) * w + x]; sw += g[k]; }
                    }
                }
                o[(size_t)z * w * h + y * w + x] = sum / sw;
            }
        }
    }
    free(tmp);
    return 0;
}
