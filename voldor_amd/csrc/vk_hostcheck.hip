// voldor_amd/csrc/vk_hostcheck.hip -- host instantiations of the __host__ __device__ solver math
// (vk_p3p.hpp), exported so that the CPU-only test tier can compare the exact arithmetic the GPU
// lanes run against the oracle.  Not used by the product path.
#include "vk_p3p.hpp"
#include "vk_device.hpp"
#include "../../include/voldor_hip.h"

extern "C" {
int vk_host_lambdatwist_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy, int use_double,
                            float* R9, float* t3) {
    float yu[4], yv[4], xp[4][3];
    for (int k = 0; k < 4; k++) { yu[k] = y8[k * 2]; yv[k] = y8[k * 2 + 1]; for (int d = 0; d < 3; d++) xp[k][d] = x12[k * 3 + d]; }
    bool ok = use_double ? vk::lambdatwist_p4p<double>(yu, yv, xp, fx, fy, cx, cy, R9, t3)
                         : vk::lambdatwist_p4p<float>(yu, yv, xp, fx, fy, cx, cy, R9, t3);
    return ok ? 1 : 0;
}
int vk_host_ap3p_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy, float* R9, float* t3) {
    float yu[4], yv[4], xp[4][3];
    for (int k = 0; k < 4; k++) { yu[k] = y8[k * 2]; yv[k] = y8[k * 2 + 1]; for (int d = 0; d < 3; d++) xp[k][d] = x12[k * 3 + d]; }
    return vk::ap3p_p4p(yu, yv, xp, fx, fy, cx, cy, R9, t3) ? 1 : 0;
}
void vk_host_rodrigues(const float* R9, float* rvec3) {
    float R[9];
    for (int i = 0; i < 9; i++) R[i] = R9[i];
    vk::nearest_rotation(R);
    vk::rotmat_to_angle_axis(R, rvec3);
}
void vk_host_rvec_to_rotmat(const float* rvec3, float* R9) { vk::angle_axis_to_rotmat(rvec3, R9); }
unsigned vk_host_rng(unsigned seed, unsigned stream, unsigned counter) { return vk::rng3(seed, stream, counter); }
float vk_host_u01(unsigned r) { return vk::u01(r); }
}
