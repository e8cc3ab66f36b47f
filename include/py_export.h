// include/py_export.h -- B-outer drop-in boundary: the whole VO window in one call.
// Same declaration as /root/reference/voldor/py_export.h:3-11 (C++ linkage, `int&` output),
// which is what slam_py/install/pyvoldor_vo.pyx:5-12 binds; implemented by
// voldor_amd/csrc/vk_voldor.hip with a device-resident EM loop instead of voldor/voldor.cpp.
#pragma once

#pragma GCC visibility push(default)
extern int py_voldor_wrapper(
	// inputs
	const float* flows, const float* disparity, const float* disparity_pconf,
	const float* depth_priors, const float* depth_prior_poses, const float* depth_prior_pconfs,
	const float fx, const float fy, const float cx, const float cy, const float basefocal,
	const int N, const int N_dp, const int w, const int h,
	const char* config,
	// outputs
	int& n_registered, float* poses, float* poses_covar, float* depth, float* depth_conf);

#pragma GCC visibility pop
