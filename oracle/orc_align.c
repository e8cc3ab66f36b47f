/* oracle/orc_align.c -- CPU restatement of the frame-alignment residual / Jacobian maps of the reference's mapping
 * back-end: gpu-kernels/align_frame.cu:47-137 (rot_with_rvec), :140-162 (pin-hole helpers),
 * :153-205 (normals and image gradients), :207-388 (compute_residual), :390-417 (weighted sqrt-Cauchy loss),
 * :414-554 (host entry points), with the GMat access rules of gpu-kernels/gmat.h:171-186.
 * TEST INFRASTRUCTURE ONLY (tests/ link it through oracle/orc.py).  PINNED against the reference's own code: rot_with_rvec
 * against the function compiled in place (oracle/_ref, tests/golden/ref_rot.npz), and the whole pair of entry points against
 * align_frame.cu itself compiled for the CPU on the launch-emulation layer (oracle/ref_wrap_kernels.cpp,
 * tests/golden/ref_kernels.npz "align/..."): residual maps bit-exact incl. the NaN mask, Jacobian maps to 1e-6 of each
 * parameter's scale (tests/test_oracle_vs_ref_kernels.py::test_align_frame_matches_reference).
 *
 * Same deviation as the VO path: D2, bilinear fetches use exact fp32 weights per layer with clamp-to-edge instead of the
 * 8-bit CUDA texture filter over vertically stacked layers.
 * Replicated quirks: (1) the d/d(rvec) Jacobian divides by theta^(3/2) where theta^3 is meant (align_frame.cu:70,
 * `sqrt(theta2*theta)`); (2) at_safe takes size_t, so index -1 wraps to the LAST row/column, not 0 (gmat.h:181-186);
 * (3) residuals with weight*r <= FLT_EPSILON skip the loss and keep the raw, unweighted value (:399).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

typedef struct { float x, y, z; } v3;
static v3 v3add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static v3 v3sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static v3 v3mul(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 v3cross(v3 a, v3 b) { v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; return r; }

/* q = Rot(w) p (Rodrigues), optional d q / d w and d q / d p, row-major 3x3.  align_frame.cu:47-137.
 * With th = |w|, u = w/th:  q = p cos + (u x p) sin + u (u.p)(1 - cos); the derivative is assembled from
 * d cos = -sin u_j, d sin = cos u_j, d u_k / d w_j = delta_kj / th - w_k w_j / T  with T = th^(3/2) (sic, quirk 1). */
void orc_rot_with_rvec(const float* p3, const float* rvec, float* out3, float* Jw, float* Jp) {
    const v3 p = { p3[0], p3[1], p3[2] }, w = { rvec[0], rvec[1], rvec[2] };
    const float th2 = v3dot(w, w);
    const float pv[3] = { p.x, p.y, p.z }, wv[3] = { w.x, w.y, w.z };
    if (th2 > FLT_EPSILON) {
        const float th = sqrtf(th2), c = cosf(th), s = sinf(th), ith = 1.f / th;
        const v3 u = v3mul(w, ith), uxp = v3cross(u, p);
        const float up = v3dot(u, p);
        const v3 q = v3add(v3add(v3mul(p, c), v3mul(uxp, s)), v3mul(u, up * (1.0f - c)));
        out3[0] = q.x; out3[1] = q.y; out3[2] = q.z;
        if (Jp) { /* the rotation matrix itself: cos I + sin [u]x + (1 - cos) u u^T */
            const float uv[3] = { u.x, u.y, u.z };
            const float ux[9] = { 0, -u.z, u.y, u.z, 0, -u.x, -u.y, u.x, 0 };
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) Jp[i * 3 + j] = (i == j ? c : 0.f) + s * ux[i * 3 + j] + (1.f - c) * uv[i] * uv[j];
        }
        if (Jw) {
            const float T = sqrtf(th2 * th); /* quirk 1 */
            const float uv[3] = { u.x, u.y, u.z }, uxpv[3] = { uxp.x, uxp.y, uxp.z };
            const v3 wxp = v3cross(w, p);
            const float wxpv[3] = { wxp.x, wxp.y, wxp.z };
            const float wp = v3dot(w, p);
            for (int j = 0; j < 3; j++) {
                /* e_j x p */
                const float e[3] = { j == 0, j == 1, j == 2 };
                const float ejxp[3] = { e[1] * pv[2] - e[2] * pv[1], e[2] * pv[0] - e[0] * pv[2], e[0] * pv[1] - e[1] * pv[0] };
                for (int i = 0; i < 3; i++) {
                    const float dui = (i == j ? ith : 0.f) - wv[i] * wv[j] / T;          /* d u_i / d w_j */
                    const float dup = pv[j] * ith - wv[j] * wp / T;                       /* d (u.p) / d w_j */
                    const float duxp = ejxp[i] * ith - wv[j] * wxpv[i] / T;               /* d (u x p)_i / d w_j */
                    Jw[i * 3 + j] = -pv[i] * s * uv[j] + uxpv[i] * c * uv[j] + s * duxp + dui * up * (1.f - c) + uv[i] * dup * (1.f - c) +
                                    uv[i] * up * s * uv[j];
                }
            }
        }
    } else { /* first order: q = p + w x p */
        const v3 q = v3add(p, v3cross(w, p));
        out3[0] = q.x; out3[1] = q.y; out3[2] = q.z;
        if (Jp) { const float m[9] = { 1, -w.z, w.y, w.z, 1, -w.x, -w.y, w.x, 1 }; memcpy(Jp, m, sizeof m); }
        if (Jw) { const float m[9] = { 0, p.z, -p.y, -p.z, 0, p.x, p.y, -p.x, 0 }; memcpy(Jw, m, sizeof m); }
    }
}

struct orc_align {
    int N, w, h, photo;
    float fx, cx, fy, cy, fxi, cxi, fyi, cyi, vbf, crw;
    float *images, *depths, *weights, *normals /*[N][h][w][4]*/, *dimages /*[N][h][w][2]*/;
};

static int safe_idx(int i, int n) { return i < 0 ? n - 1 : (i > n - 1 ? n - 1 : i); } /* quirk 2 */
static float at_safe(const float* m, int w, int h, int x, int y) { return m[safe_idx(y, h) * w + safe_idx(x, w)]; }
static v3 backproj(const orc_align* A, float x, float y, float d) { v3 r = { (A->fxi * x + A->cxi) * d, (A->fyi * y + A->cyi) * d, d }; return r; }
/* clamp-to-edge bilinear of channel ch of an interleaved [h][w][nc] layer (deviation D2) */
static float bil(const float* m, int w, int h, int nc, int ch, float x, float y) {
    const float fx = floorf(x), fy = floorf(y), a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    return w00 * m[(y0 * w + x0) * nc + ch] + w10 * m[(y0 * w + x1) * nc + ch] + w01 * m[(y1 * w + x0) * nc + ch] + w11 * m[(y1 * w + x1) * nc + ch];
}
static void grad33(const float* m, int w, int h, int x, int y, float* gx, float* gy) { /* :176-185, :196-204 */
    *gx = 0.3f * (at_safe(m, w, h, x + 1, y) - at_safe(m, w, h, x - 1, y)) + 0.1f * (at_safe(m, w, h, x + 1, y - 1) - at_safe(m, w, h, x - 1, y - 1)) +
          0.1f * (at_safe(m, w, h, x + 1, y + 1) - at_safe(m, w, h, x - 1, y + 1));
    *gy = 0.3f * (at_safe(m, w, h, x, y + 1) - at_safe(m, w, h, x, y - 1)) + 0.1f * (at_safe(m, w, h, x - 1, y + 1) - at_safe(m, w, h, x - 1, y - 1)) +
          0.1f * (at_safe(m, w, h, x + 1, y + 1) - at_safe(m, w, h, x + 1, y - 1));
}

void orc_align_free(orc_align* A) {
    if (!A) return;
    free(A->images); free(A->depths); free(A->weights); free(A->normals); free(A->dimages); free(A);
}
/* align_frame_init_gpu, :443-554.  images may be NULL (or crw <= 0): geometry only. */
orc_align* orc_align_init(const float* images, const float* depths, const float* weights, const float* K9, float vbf, float crw, int N, int w, int h) {
    orc_align* A = calloc(1, sizeof *A);
    const size_t npx = (size_t)w * h;
    A->N = N; A->w = w; A->h = h; A->vbf = vbf; A->crw = crw; A->photo = images && crw > 0;
    A->fx = K9[0]; A->cx = K9[2]; A->fy = K9[4]; A->cy = K9[5];
    A->fxi = 1.f / K9[0]; A->cxi = -K9[2] / K9[0]; A->fyi = 1.f / K9[4]; A->cyi = -K9[5] / K9[4];
    A->depths = malloc(sizeof(float) * npx * N); memcpy(A->depths, depths, sizeof(float) * npx * N);
    A->weights = malloc(sizeof(float) * npx * N); memcpy(A->weights, weights, sizeof(float) * npx * N);
    A->normals = malloc(sizeof(float) * npx * N * 4);
    if (A->photo) {
        A->images = malloc(sizeof(float) * npx * N); memcpy(A->images, images, sizeof(float) * npx * N);
        A->dimages = malloc(sizeof(float) * npx * N * 2);
    }
    for (int f = 0; f < N; f++) {
        const float* D = A->depths + (size_t)f * npx;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const v3 pt = backproj(A, (float)x, (float)(y - 1), at_safe(D, w, h, x, y - 1)), pb = backproj(A, (float)x, (float)(y + 1), at_safe(D, w, h, x, y + 1));
                const v3 pl = backproj(A, (float)(x - 1), (float)y, at_safe(D, w, h, x - 1, y)), pr = backproj(A, (float)(x + 1), (float)y, at_safe(D, w, h, x + 1, y));
                v3 n = v3cross(v3sub(pt, pb), v3sub(pl, pr));
                const float nn = sqrtf(v3dot(n, n));
                n.x /= nn; n.y /= nn; n.z /= nn;
                if (v3dot(backproj(A, (float)x, (float)y, 1.f), n) > 0) n = v3mul(n, -1.f); /* towards the view point */
                float* o = A->normals + ((size_t)f * npx + (size_t)y * w + x) * 4;
                o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = 0.f;
                if (A->photo) grad33(A->images + (size_t)f * npx, w, h, x, y, A->dimages + ((size_t)f * npx + (size_t)y * w + x) * 2, A->dimages + ((size_t)f * npx + (size_t)y * w + x) * 2 + 1);
            }
    }
    return A;
}

/* align_frame_eval_gpu (:414-441) = compute_residual (:207-388) + apply_weighted_sqrt_cauchy_loss (:390-412).
 * residual [h][w]; jacobian [h][w*9] or NULL. */
void orc_align_eval(const orc_align* A, int ref, int tar, const float* pr, const float* pt, float* residual, float* jacobian, int apply_weights) {
    const int w = A->w, h = A->h; const size_t npx = (size_t)w * h;
    memset(residual, 0, sizeof(float) * npx);
    if (jacobian) memset(jacobian, 0, sizeof(float) * npx * 9);
    const float* Dr = A->depths + (size_t)ref * npx; const float* Dt = A->depths + (size_t)tar * npx;
    const float* Nt = A->normals + (size_t)tar * npx * 4;
    const float rvec[3] = { pr[0], pr[1], pr[2] }, rvec0[3] = { -pt[0], -pt[1], -pt[2] }; /* target pose inverted: world -> cam */
    float t0[3];
    { const float tt[3] = { pt[3], pt[4], pt[5] }; orc_rot_with_rvec(tt, rvec0, t0, NULL, NULL); t0[0] = -t0[0]; t0[1] = -t0[1]; t0[2] = -t0[2]; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float* R = residual + (size_t)y * w + x;
            float* J = jacobian ? jacobian + ((size_t)y * w + x) * 9 : NULL;
            const float d_ref = Dr[(size_t)y * w + x] * expf(pr[6]);
            const v3 p3r = backproj(A, (float)x, (float)y, d_ref);
            const float dp3r_dd[3] = { A->fxi * x + A->cxi, A->fyi * y + A->cyi, 1.f };
            float p3w[3], Jw_rvec[9], Jw_p3r[9];
            { const float pp[3] = { p3r.x, p3r.y, p3r.z }; orc_rot_with_rvec(pp, rvec, p3w, J ? Jw_rvec : NULL, J ? Jw_p3r : NULL); }
            p3w[0] += pr[3]; p3w[1] += pr[4]; p3w[2] += pr[5];
            float p3tv[3], Jt_p3w[9];
            orc_rot_with_rvec(p3w, rvec0, p3tv, NULL, J ? Jt_p3w : NULL);
            const v3 p3t = { p3tv[0] + t0[0], p3tv[1] + t0[1], p3tv[2] + t0[2] };
            const float u = (A->fx * p3t.x) / p3t.z + A->cx, v = (A->fy * p3t.y) / p3t.z + A->cy;
            if (u < 0 || u >= w || v < 0 || v >= h || p3t.z < 1.f) { *R = NAN; continue; }
            const float d_tar = bil(Dt, w, h, 1, 0, u, v) * expf(pt[6]);
            const v3 n = { bil(Nt, w, h, 4, 0, u, v), bil(Nt, w, h, 4, 1, u, v), bil(Nt, w, h, 4, 2, u, v) };
            const v3 ray = v3mul(p3t, d_tar / p3t.z);                 /* the measured surface point on this ray */
            const v3 diff = v3mul(n, v3dot(n, v3sub(ray, p3t)));    /* point-to-plane offset, target -> reference */
            const v3 geo = v3add(p3t, diff);
            const float ug = (A->fx * geo.x) / geo.z + A->cx, vg = (A->fy * geo.y) / geo.z + A->cy;
            if (ug < 0 || ug >= w || vg < 0 || vg >= h) { *R = NAN; continue; }
            const float res_d = 0.5f * v3dot(diff, diff);
            const float q = A->vbf / (fmaxf(geo.z, 1.0f) * fmaxf(p3t.z, 1.0f));
            const float drw = q * q;
            float c_ref = 0, c_tar = 0, res_c = 0;
            if (A->photo) {
                c_ref = A->images[(size_t)ref * npx + (size_t)y * w + x] + pr[8];
                const float c_bs = bil(A->images + (size_t)tar * npx, w, h, 1, 0, u, v) + pt[8];
                c_tar = c_bs * (expf(pr[7]) / expf(pt[7]));
                res_c = 0.5f * (c_ref - c_tar) * (c_ref - c_tar);
            }
            *R = A->photo ? drw * res_d + A->crw * res_c : drw * res_d;
            if (J) {
                /* d res / d p3t : geometric part -diff (normal, weight and measured depth held fixed), photometric part through
                 * the image gradient at the target pixel and the pin-hole Jacobian */
                float g[3] = { -diff.x * drw, -diff.y * drw, -diff.z * drw };
                float dc_scale = 0, dc_off = 0;
                if (A->photo) {
                    const float gI[2] = { bil(A->dimages + (size_t)tar * npx * 2, w, h, 2, 0, u, v), bil(A->dimages + (size_t)tar * npx * 2, w, h, 2, 1, u, v) };
                    const float k = c_tar - c_ref;
                    const float du[3] = { A->fx / p3t.z, 0.f, -(A->fx * p3t.x) / (p3t.z * p3t.z) }, dv[3] = { 0.f, A->fy / p3t.z, -(A->fy * p3t.y) / (p3t.z * p3t.z) };
                    for (int i = 0; i < 3; i++) g[i] += A->crw * ((gI[0] * k) * du[i] + (gI[1] * k) * dv[i]);
                    dc_scale = k * c_tar; dc_off = (c_ref - c_tar) * 1.f;
                }
                float gw[3], gr[3], gp[3];
                for (int j = 0; j < 3; j++) gw[j] = g[0] * Jt_p3w[0 * 3 + j] + g[1] * Jt_p3w[1 * 3 + j] + g[2] * Jt_p3w[2 * 3 + j];
                for (int j = 0; j < 3; j++) gr[j] = gw[0] * Jw_rvec[0 * 3 + j] + gw[1] * Jw_rvec[1 * 3 + j] + gw[2] * Jw_rvec[2 * 3 + j];
                for (int j = 0; j < 3; j++) gp[j] = gw[0] * Jw_p3r[0 * 3 + j] + gw[1] * Jw_p3r[1 * 3 + j] + gw[2] * Jw_p3r[2 * 3 + j];
                J[0] = gr[0]; J[1] = gr[1]; J[2] = gr[2]; J[3] = gw[0]; J[4] = gw[1]; J[5] = gw[2];
                J[6] = (gp[0] * dp3r_dd[0] + gp[1] * dp3r_dd[1] + gp[2] * dp3r_dd[2]) * d_ref;
                J[7] = A->photo ? A->crw * dc_scale : 0.f;
                J[8] = A->photo ? A->crw * dc_off : 0.f;
            }
        }
    /* weighted sqrt-Cauchy loss (:390-412) */
    for (size_t i = 0; i < npx; i++) {
        const float wgt = apply_weights ? A->weights[(size_t)ref * npx + i] : 1.f;
        const float r2 = wgt * residual[i];
        if (r2 > FLT_EPSILON) { /* quirk 3: otherwise (incl. NaN) untouched */
            const float loss = logf(r2 + 1.f), sl = sqrtf(loss);
            residual[i] = sl;
            if (jacobian) { const float k = (0.5f / sl) * (1.f / (r2 + 1.f)) * wgt; for (int j = 0; j < 9; j++) jacobian[i * 9 + j] *= k; }
        }
    }
}
