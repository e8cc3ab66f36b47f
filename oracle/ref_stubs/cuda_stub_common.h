/* Host-compile stubs so that the reference's pure device-function headers
 * (gpu-kernels/residual_model.h, rodrigues.h, svd3_cuda.h) can be compiled by
 * g++ IN PLACE from /root/reference.  Test infrastructure only. Nothing here is
 * reference code: these are empty stand-ins for the CUDA toolkit headers that
 * gpu-kernels/utils.h includes. */
#pragma once
#include <math.h>
#include <float.h>
#include <algorithm>
#include <utility>
#define __device__
#define __host__
#define __global__
#define __inline__ inline
#define __constant__
#define CUDART_NAN_F (__builtin_nanf(""))
typedef int cudaError_t;
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
/* CUDA __frsqrt_rn is IEEE round-to-nearest 1/sqrt(x). */
static inline float __frsqrt_rn(float x) { return (float)(1.0 / sqrt((double)x)); }
using std::max;
using std::min;

/* CUDA vector types used by gpu-kernels/vops.h and the host/device helpers of align_frame.cu */
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r = { x, y }; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r = { x, y, z }; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
