/* stands in for gpu-kernels/aux_funs.h: the reference implements these three on cv::Matx66d (aux_funs.cpp:97-141, OpenCV,
 * absent here) -- oracle/ref_wrap_kernels.cpp defines them with a double-precision partial-pivot LU, the algorithm
 * cv::Matx::inv / cv::determinant use for 6x6.  TEST INFRASTRUCTURE ONLY. */
#pragma once
double inverse(double* mat, double* mat_inv, int N);
double determinant(double* mat, int N);
double regularize_covar_LW_given_lambda(double* mat, double* mat_ret, double lambda, int dims);
