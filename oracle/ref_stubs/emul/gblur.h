/* stands in for gpu-kernels/gblur.h (which would pull in the real gmat.h): align_frame.cu includes it but never calls it */
#pragma once
#include "gmat.h"
int gblur_gpu(GMatf src, GMatf& dst, float sigma, int ksize = 0);
