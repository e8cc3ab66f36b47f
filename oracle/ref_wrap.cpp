// oracle/_ref builder: compiles the REFERENCE's own header-only math, in place from
// /root/reference (never copied), into oracle/_ref/libvoldor_ref.so.
// TEST INFRASTRUCTURE ONLY: used to pin oracle/ (the C restatement) and to generate
// tests/golden/*.npz.  /root/reference does not exist on the GPU box; the built .so
// travels with the snapshot, the golden vectors are committed.
//
// What can be compiled from the reference without nvcc/OpenCV:
//   lambdatwist/lambdatwist_p4p.h     -> ref_lambdatwist_p4p_{f,d}
//   gpu-kernels/residual_model.h      -> ref_fun_rigidness, ref_fun_depth_rigidness, ref_fisk_pdf, ...
//   gpu-kernels/rodrigues.h + svd3_cuda.h -> ref_rodrigues (SVD-orthonormalise + angle-axis)
// Everything else in gpu-kernels/*.cu needs the CUDA launch syntax / textures / cuRAND and
// voldor/*.cpp needs OpenCV: unbuildable here.
#include <utility>
#include <algorithm>
#include <cmath>
#include <cstring>
#include "ref_stubs/emul/cuda_emul.h"   // (includes cuda_stub_common.h) the libm switch ref_set_math_mode also governs these headers
#include "lambdatwist/lambdatwist_p4p.h"   // -I/root/reference
#include "gpu-kernels/residual_model.h"
#include "gpu-kernels/rodrigues.h"
#ifdef REF_ROT_INC   // rot_with_rvec of gpu-kernels/align_frame.cu:47-137, cut out of the .cu at build time into a temp file
#include "gpu-kernels/vops.h"
#include REF_ROT_INC
#endif

extern "C" {

// lambdatwist_p4p<_T=float> is the reference GPU path (solve_batch_lambdatwist.cu:22),
// <_T=double> the reference CPU path (voldor/geometry.cpp:112).
int ref_lambdatwist_p4p_f(const float* y, const float* x, float fx, float fy, float cx, float cy,
                          float* R9, float* t3) {
    float yy[8], xx[12]; memcpy(yy, y, sizeof yy); memcpy(xx, x, sizeof xx);
    float R[3][3], t[3];
    bool ok = lambdatwist_p4p<float, float, 5>(yy, yy + 2, yy + 4, yy + 6, xx, xx + 3, xx + 6, xx + 9,
                                               fx, fy, cx, cy, R, t);
    if (ok) { memcpy(R9, R, sizeof R); memcpy(t3, t, sizeof t); }
    return ok ? 1 : 0;
}
int ref_lambdatwist_p4p_d(const float* y, const float* x, float fx, float fy, float cx, float cy,
                          float* R9, float* t3) {
    float yy[8], xx[12]; memcpy(yy, y, sizeof yy); memcpy(xx, x, sizeof xx);
    float R[3][3], t[3];
    bool ok = lambdatwist_p4p<double, float, 5>(yy, yy + 2, yy + 4, yy + 6, xx, xx + 3, xx + 6, xx + 9,
                                                fx, fy, cx, cy, R, t);
    if (ok) { memcpy(R9, R, sizeof R); memcpy(t3, t, sizeof t); }
    return ok ? 1 : 0;
}

float ref_fun_fmag_c(float m) { return fun_fmag_c(m); }
float ref_fun_fmag_scale(float m) { return fun_fmag_scale(m); }
float ref_fisk_dist_pdf(float x, float c, float s) { return fisk_dist_pdf(x, c, s); }
float ref_fun_rigidness(float dx1, float dy1, float dx2, float dy2, float lambda, float arf) {
    return fun_rigidness(dx1, dy1, dx2, dy2, lambda, arf);
}
float ref_fun_depth_rigidness(float d1, float d2, float bf, float omega, float arf) {
    return fun_depth_rigidness(d1, d2, bf, omega, arf);
}
void ref_fun_cost(float dx1, float dy1, float dx2, float dy2, float w, float lambda, float arf,
                  float* io_cost, float* io_wsum) {
    fun_cost(dx1, dy1, dx2, dy2, w, *io_cost, *io_wsum, lambda, arf);
}
void ref_fun_depth_cost(float d1, float d2, float bf, float w, float omega, float arf,
                        float* io_cost, float* io_wsum) {
    fun_depth_cost(d1, d2, bf, w, *io_cost, *io_wsum, omega, arf);
}
// rodrigues(R, rvec): R is modified in place (projected to SO(3)) exactly as the reference does.
void ref_rodrigues(const float* R9, float* rvec3, float* Rproj9) {
    float R[3][3]; memcpy(R, R9, sizeof R);
    rodrigues(R, rvec3);
    if (Rproj9) memcpy(Rproj9, R, sizeof R);
}
void ref_rotmat_to_angle_axis(const float* R9, float* rvec3) {
    float R[3][3]; memcpy(R, R9, sizeof R);
    RotationMatrixToAngleAxis(R, rvec3);
}
#ifdef REF_ROT_INC
// rot_with_rvec (align_frame.cu:47-137): rotated point, d/d rvec and d/d point (row-major 3x3 each)
void ref_rot_with_rvec(const float* p3, const float* rvec, float* out3, float* J_rvec9, float* J_p39) {
    float Jr[3][3], Jp[3][3];
    float3 q = rot_with_rvec(make_float3(p3[0], p3[1], p3[2]), make_float3(rvec[0], rvec[1], rvec[2]), Jr, Jp);
    out3[0] = q.x; out3[1] = q.y; out3[2] = q.z;
    memcpy(J_rvec9, Jr, sizeof Jr); memcpy(J_p39, Jp, sizeof Jp);
}
#endif
}
