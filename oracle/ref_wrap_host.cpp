// oracle/_ref builder, part 3: the reference's HOST pipeline.  voldor/voldor.cpp, geometry.cpp, utils.cpp and py_export.cpp
// are compiled in place from /root/reference (no rewriting at all) against oracle/ref_stubs/minicv (a stand-in for the slice
// of OpenCV they use) and linked to the CPU-emulated kernels of ref_wrap_kernels.cpp: what runs behind ref_py_voldor_wrapper
// is the reference's own py_voldor_wrapper (voldor/py_export.cpp:5-79) -> VOLDOR::init / solve -> optimize_camera_pose ->
// the reference's kernels.  TEST INFRASTRUCTURE ONLY: pins oracle/orc_voldor.c (tests/golden/ref_window.npz).
//
// Second build of the same file (-DREF_HOST_ON_HIP, oracle/_ref/libvoldor_refhost_hip.so): the SAME reference host objects
// linked against voldor_amd/lib/libvoldor_hip.so instead of the emulated kernels -- the reference's voldor.cpp / geometry.cpp
// then call optimize_depth_gpu, collect_p3p_instances, solve_batch_p3p_*_gpu, meanshift_gpu and fit_robust_gaussian of the
// HIP library through the mangled gpu_kernels.h symbols, i.e. the drop-in a maintainer gets by swapping -lgpu-kernels for
// -lvoldor_hip.  tests/test_gpu_reference_host_on_hip.py runs it on the GPU.
#if defined(REF_PREP_DIR) || defined(REF_HOST_ON_HIP)
#include "ref_stubs/minicv/minicv.hpp"
#include "voldor/py_export.h"

double minicv_two_view_R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, minicv_two_view_t[3] = { 0, 0, 1 };
#ifdef REF_HOST_ON_HIP
extern "C" { int ref_math_mode = 0; }  // minicv's Rodrigues switch (the kernels come from libvoldor_hip.so here, which has its own strict mode)
extern "C" int vk_set_rand_epoch(unsigned epoch);  // include/voldor_hip.h:108
static void ref_reset_window_state(unsigned rand_epoch) { vk_set_rand_epoch(rand_epoch); }
#else
extern "C" void ref_reset_window_state(unsigned rand_epoch);
#endif

cv::Mat load_flow(const char* file_path);  // voldor/utils.cpp:23-41 (declared in voldor/utils.h, which py_export.h does not pull in)

extern "C" {
// the reference's Middlebury .flo reader on a file written by voldor_amd/formats.py / vk_write_flo
int ref_load_flow(const char* path, int* w, int* h, float* out, size_t cap_floats) {
    cv::Mat m = load_flow(path);
    *w = m.cols; *h = m.rows;
    const size_t n = (size_t)m.cols * m.rows * 2;
    if (n > cap_floats) return 1;
    memcpy(out, m.data, n * sizeof(float));
    return 0;
}
// what cv::recoverPose will hand to estimate_camera_pose_epipolar (geometry.cpp:288-332); deviation D5
void ref_set_two_view_pose(const double* R9, const double* t3) {
    memcpy(minicv_two_view_R, R9, sizeof minicv_two_view_R);
    memcpy(minicv_two_view_t, t3, sizeof minicv_two_view_t);
}
int ref_py_voldor_wrapper(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                          const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy, float basefocal,
                          int N, int N_dp, int w, int h, const char* config, unsigned rand_epoch, int* n_registered, float* poses,
                          float* poses_covar, float* depth, float* depth_conf) {
    ref_reset_window_state(rand_epoch);
    int n = 0;
    const int rc = py_voldor_wrapper(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy, cx, cy, basefocal,
                                     N, N_dp, w, h, config, n, poses, poses_covar, depth, depth_conf);
    *n_registered = n;
    return rc;
}
}
#endif
