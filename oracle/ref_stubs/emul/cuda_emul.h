/* oracle/ref_stubs/emul/cuda_emul.h -- a few lines of CUDA execution model on the CPU, so that the reference's .cu files
 * (kernels, device functions AND their host entry points; rewritten at build time by oracle/ref_prep.pl into temp files,
 * never copied into the repo) can be compiled by g++ and run thread by thread.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
 * Nothing here is reference code.
 *
 * What is emulated: threadIdx / blockIdx / blockDim / gridDim, a launcher that walks the grid sequentially (the only
 * source rewrite is `kernel<<<grid, block>>>(args);` -> `emul_launch(emul_cfg(grid, block), [&]{ kernel(args); });`),
 * cudaMalloc / cudaMemcpy / cudaMemcpyToSymbol on host memory, GMat (the reference's device array, gmat.h) backed by
 * host memory, cuRAND state.  Two deliberate substitutions, the same ones the oracle documents: D1 curand_uniform draws
 * from the counter-based generator of oracle/orc_model.c instead of XORWOW; D2 at_tex is an exact-weight clamp-to-edge
 * bilinear fetch per layer instead of the 8-bit texture filter.  Kernels that need __syncthreads / warp lock-step
 * (reduce_vector_sum.h only) are replaced by a stand-in header that restates their summation order. */
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "../cuda_stub_common.h"

/* ---- the C library's transcendentals, switchable at run time (ref_set_math_mode, defined in ref_wrap_kernels.cpp):
 *  0  libm, as the reference calls it;
 *  1  voldor_amd/csrc/vk_strict_math.h -- what the oracle (orc_set_strict_math) and the product (--strict_math 1) use in
 *     strict mode, so the reference's own code can be run in strict mode too (tests/golden/ref_window_strict.npz);
 *  2  libm with every expf / powf / logf result moved by -1 / 0 / +1 ulp (a hash of the argument bits decides): a stand-in
 *     for "some other correctly-behaving libm / GPU math library", used to MEASURE how far two runs of the reference's own
 *     pipeline drift apart when only the last bit of the transcendentals differs (tests/golden/ref_selfnoise.npz). */
#include "../../../voldor_amd/csrc/vk_strict_math.h"
extern "C" int ref_math_mode;
extern "C" unsigned int ref_jitter_salt; /* ref_set_jitter_salt: independent jitter patterns (0 = the pattern of ref_selfnoise.npz) */
static inline float ref_ulp_jitter(float r, float x, float y) {
    uint32_t a, b; memcpy(&a, &x, 4); memcpy(&b, &y, 4);
    uint32_t h = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u) ^ (ref_jitter_salt * 0xC2B2AE3Du); h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    const int k = (int)(h % 3u) - 1;
    if (k == 0 || !(r == r) || std::isinf(r)) return r;
    return std::nextafterf(r, k > 0 ? INFINITY : -INFINITY);
}
static inline float ref_expf(float x) { return ref_math_mode == 1 ? vsm_expf(x) : (ref_math_mode == 2 ? ref_ulp_jitter(::expf(x), x, 1.f) : ::expf(x)); }
static inline float ref_logf(float x) { return ref_math_mode == 1 ? vsm_logf(x) : (ref_math_mode == 2 ? ref_ulp_jitter(::logf(x), x, 2.f) : ::logf(x)); }
static inline float ref_powf(float x, float y) { return ref_math_mode == 1 ? vsm_powf(x, y) : (ref_math_mode == 2 ? ref_ulp_jitter(::powf(x, y), x, y) : ::powf(x, y)); }
static inline float ref_atan2f(float y, float x) { return ref_math_mode == 1 ? vsm_atan2f(y, x) : ::atan2f(y, x); }
static inline float ref_sinf(float x) { return ref_math_mode == 1 ? vsm_sinf(x) : ::sinf(x); }
static inline float ref_cosf(float x) { return ref_math_mode == 1 ? vsm_cosf(x) : ::cosf(x); }
static inline float ref_cbrtf(float x) { return ref_math_mode == 1 ? vsm_cbrtf(x) : ::cbrtf(x); }
#define expf ref_expf
#define logf ref_logf
#define powf ref_powf
#define atan2f ref_atan2f
#define sinf ref_sinf
#define cosf ref_cosf
#define cbrtf ref_cbrtf

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct emul_idx3 { unsigned x, y, z; };
static emul_idx3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;
struct emul_cfg { dim3 grid, block; emul_cfg(dim3 g, dim3 b, size_t = 0) : grid(g), block(b) {} };
template <class F> static void emul_launch(dim3 grid, dim3 block, F kernel) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++)
        for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned tx = 0; tx < block.x; tx++) {
            blockIdx = { bx, by, bz }; threadIdx = { tx, ty, tz };
            kernel();
        }
}
template <class F> static void emul_launch(const emul_cfg& c, F kernel) { emul_launch(c.grid, c.block, kernel); }
#define DIV_CEIL_EMUL(a, b) (((a) + (b) - 1) / (b))

/* ---- runtime API on host memory.  Allocations are padded: the reference's `int i = curand_uniform() * N` indexes one
 * past the end when the draw is exactly 1.0 (solve_batch_lambdatwist.cu:16-19). */
enum { cudaSuccess = 0, cudaErrorInvalidFilterSetting = 26 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)calloc(n + 64, 1); return cudaSuccess; }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
template <class S> static inline cudaError_t cudaMemcpyToSymbol(S& sym, const void* s, size_t n) { memcpy((void*)&sym, s, n); return cudaSuccess; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "emulated"; }
struct cudaPos { size_t x, y, z; };
static inline cudaPos make_cudaPos(size_t x, size_t y, size_t z) { cudaPos p = { x, y, z }; return p; }

/* host rand() of meanshift.cu:76 (ref_wrap_kernels.cpp maps rand to emul_host_rand around that file): the wrapper sets
 * the modulus so that `rand() % N` equals the oracle's draw */
static uint32_t emul_rand_trial = 0, emul_rand_mod = 1;

/* D1: counter-based generator (orc_model.c orc_rng / orc_u01): murmur3 finaliser over (seed, stream, counter) */
static inline uint32_t emul_fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
static inline uint32_t emul_rng3(uint32_t seed, uint32_t stream, uint32_t counter) {
    uint32_t h = emul_fmix32(seed ^ 0x9E3779B9u);
    h = emul_fmix32(h ^ stream);
    h = emul_fmix32(h + counter * 0x9E3779B1u + 0x7F4A7C15u);
    return h;
}
static inline int emul_host_rand() { return (int)(emul_rng3(233u, emul_rand_trial++, 0x4D53u) % emul_rand_mod); }
/* ref_set_reference_rng(1) / ref_set_reference_tex(1) (round 4): D1 / D2 OFF -- cuRAND's XORWOW and CUDA's linear texture filter restated
 * from their published definitions (voldor_amd/csrc/vk_ref_cuda.h, the header the oracle and the HIP kernels use for the same switches) */
#include "../../../voldor_amd/csrc/vk_ref_cuda.h"
extern "C" int ref_reference_rng, ref_reference_tex;
static uint32_t* emul_xw_J = nullptr;
static inline const uint32_t* emul_xw_jumps() {
    if (!emul_xw_J) { emul_xw_J = (uint32_t*)malloc(sizeof(uint32_t) * 32 * VRC_XW_MAT); vrc_build_sequence_jumps(emul_xw_J); }
    return emul_xw_J;
}
struct curandState { uint32_t seed, stream, counter; vrc_xorwow xw; };
static uint32_t emul_curand_epoch = 0;  /* the wrapper's stand-in for "states persist across calls": added to the offset */
extern "C" unsigned int ref_rand_salt; /* ref_set_rand_salt: "the reference run again with another seed of its random streams" (0 = RAND_SEED as is) */
static inline void curand_init(unsigned long long seed, unsigned long long sequence, unsigned long long offset, curandState* s) {
    s->seed = (uint32_t)seed ^ (ref_rand_salt * 0x9E3779B1u); s->stream = (uint32_t)sequence; s->counter = (uint32_t)offset + emul_curand_epoch;
    if (ref_reference_rng) {  /* curand_init(seed, subsequence, offset): seed scramble, subsequence * 2^67 outputs skipped, then `offset` outputs */
        vrc_xorwow_init(emul_xw_jumps(), seed ^ (unsigned long long)(ref_rand_salt * 0x9E3779B1u), (uint32_t)sequence, &s->xw);
        for (uint32_t e = 0; e < s->counter; e++) (void)vrc_xorwow_next(&s->xw);
    }
}
static inline float curand_uniform(curandState* s) {  /* (0, 1] like cuRAND */
    if (ref_reference_rng) return vrc_uniform(vrc_xorwow_next(&s->xw));
    const uint32_t r = emul_rng3(s->seed, s->stream, s->counter++);
    return (float)((r >> 8) + 1u) * (1.0f / 16777216.0f);
}

/* the reference's GMat<T> (gmat.h:4-204): 3-D array, x fastest; here dense host memory, no pitch.  create() keeps the
 * reference's reuse rule (gmat.h:19-32: reallocate unless the size matches, or only the depth shrank with lazy_depth). */
template <typename T> struct GMat {
    T* ptr = nullptr; int _width = 0, _height = 0, _depth = 0; bool owned = false;
    void bind(T* p, int w, int h, int d) { ptr = p; _width = w; _height = h; _depth = d; owned = false; }
    int create(size_t w, size_t h, size_t d, bool lazy_depth = false) {
        if (((int)w == _width && (int)h == _height && (int)d == _depth) || (lazy_depth && (int)w == _width && (int)h == _height && (int)d <= _depth)) return 0;
        if (owned) ::free(ptr);
        ptr = (T*)calloc(w * h * (d ? d : 1) + 16, sizeof(T)); owned = true;
        _width = (int)w; _height = (int)h; _depth = (int)d;
        return 1;
    }
    size_t width() const { return (size_t)_width; }
    size_t height() const { return (size_t)_height; }
    size_t depth() const { return (size_t)_depth; }
    int free() { if (owned) ::free(ptr); ptr = nullptr; owned = false; _width = _height = _depth = 0; return cudaSuccess; }
    int bind_tex() { return 1; }
    int zeros() { memset(ptr, 0, (size_t)_width * _height * (_depth ? _depth : 1) * sizeof(T)); return cudaSuccess; }
    int copy_from_host(const T* src, cudaPos pos, size_t w, size_t h, size_t d) {
        for (size_t z = 0; z < d; z++) for (size_t y = 0; y < h; y++) memcpy(&at(pos.x, pos.y + y, pos.z + z), src + (z * h + y) * w, w * sizeof(T));
        return cudaSuccess;
    }
    int copy_to_host(const T* dst, cudaPos pos, size_t w, size_t h, size_t d) {
        for (size_t z = 0; z < d; z++) for (size_t y = 0; y < h; y++) memcpy(const_cast<T*>(dst) + (z * h + y) * w, &at(pos.x, pos.y + y, pos.z + z), w * sizeof(T));
        return cudaSuccess;
    }
    T& at(const size_t x, const size_t y, const size_t d = 0) { return ptr[(d * (size_t)_height + y) * (size_t)_width + x]; }
    T at_tex(const float x, const float y, const int d = 0) const;  /* D2 */
    /* gmat.h:181-186: the indices are size_t, so -1 wraps and min() sends it to the LAST row / column / layer */
    T& at_safe(const size_t x, const size_t y, const size_t d = 0) {
        return at(std::min(x, (size_t)(_width - 1)), std::min(y, (size_t)(_height - 1)), std::min(d, (size_t)(_depth - 1)));
    }
    T at_tex_safe(const float x, const float y, const int d = 0) const {  /* gmat.h:188-195 */
        return at_tex(std::max(std::min(x, (float)(_width - 1)), 0.f), std::max(std::min(y, (float)(_height - 1)), 0.f), std::max(std::min(d, _depth - 1), 0));
    }
};
static inline void emul_bil_idx(float x, float y, int w, int h, int& x0, int& x1, int& y0, int& y1, float& a, float& b) {
    const float fx = floorf(x), fy = floorf(y);
    a = x - fx; b = y - fy;
    int ix = (int)fx, iy = (int)fy, ix1 = ix + 1, iy1 = iy + 1;
    ix = ix < 0 ? 0 : (ix > w - 1 ? w - 1 : ix); ix1 = ix1 < 0 ? 0 : (ix1 > w - 1 ? w - 1 : ix1);
    iy = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy); iy1 = iy1 < 0 ? 0 : (iy1 > h - 1 ? h - 1 : iy1);
    x0 = ix; x1 = ix1; y0 = iy; y1 = iy1;
}
template <> inline float GMat<float>::at_tex(const float x, const float y, const int d) const {
    if (ref_reference_tex) return vrc_tex_fetch1(ptr, x, y, d, _width, _height, _depth);  /* gmat.h:49-62, :175-179: one texture over the stacked layers */
    int x0, x1, y0, y1; float a, b;
    emul_bil_idx(x, y, _width, _height, x0, x1, y0, y1, a, b);
    const float* m = ptr + (size_t)d * _height * _width;
    return (1.f - a) * (1.f - b) * m[y0 * _width + x0] + a * (1.f - b) * m[y0 * _width + x1] + (1.f - a) * b * m[y1 * _width + x0] + a * b * m[y1 * _width + x1];
}
template <> inline float2 GMat<float2>::at_tex(const float x, const float y, const int d) const {
    if (ref_reference_tex) { float2 r; vrc_tex_fetch2(&ptr->x, x, y, d, _width, _height, _depth, &r.x, &r.y); return r; }
    int x0, x1, y0, y1; float a, b;
    emul_bil_idx(x, y, _width, _height, x0, x1, y0, y1, a, b);
    const float2* m = ptr + (size_t)d * _height * _width;
    const float2 t00 = m[y0 * _width + x0], t10 = m[y0 * _width + x1], t01 = m[y1 * _width + x0], t11 = m[y1 * _width + x1];
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    float2 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    return r;
}
template <> inline float4 GMat<float4>::at_tex(const float x, const float y, const int d) const {
    int x0, x1, y0, y1; float a, b;
    emul_bil_idx(x, y, _width, _height, x0, x1, y0, y1, a, b);
    const float4* m = ptr + (size_t)d * _height * _width;
    const float4 t00 = m[y0 * _width + x0], t10 = m[y0 * _width + x1], t01 = m[y1 * _width + x0], t11 = m[y1 * _width + x1];
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    float4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}
typedef GMat<float> GMatf;
typedef GMat<float2> GMatf2;
typedef GMat<float3> GMatf3;
typedef GMat<float4> GMatf4;
typedef GMat<curandState> GMatRnd;
