/* stands in for gpu-kernels/gmat.h when the reference's device code is compiled for the CPU (cuda_emul.h) */
#pragma once
#include "cuda_emul.h"
