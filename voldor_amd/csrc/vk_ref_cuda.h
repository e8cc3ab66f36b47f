/* voldor_amd/csrc/vk_ref_cuda.h -- the two pieces of CUDA behaviour the reference's results depend on and that rounds 1-3 replaced
 * (deviations D1, D2), restated from their PUBLISHED definitions so that reference mode can switch them back on
 * (config keys --reference_rng 1 / --reference_tex 1; oracle: orc_set_reference_rng / _tex; emulated reference: ref_set_reference_rng / _tex).
 * One header for the product (hipcc), the oracle (gcc -std=gnu11) and the launch-emulation layer of the reference build (g++): plain C,
 * integer arithmetic and IEEE + - * only.
 *
 * D1  cuRAND's XORWOW generator as the reference drives it:
 *       curand_init(RAND_SEED, pixel, 0, &state) once per image size, then ONE curand_uniform per pixel per sample launch, the state
 *       persisting across launches, iterations and windows (gpu-kernels/optimize_depth.cu:269-277, :286-291);
 *       curand_init(RAND_SEED, idx, 0, &state) per solver call and four curand_uniform per hypothesis, index = (int)(u * N_pts)
 *       (solve_batch_lambdatwist.cu:16-19, :44-48; solve_batch_ap3p.cu:336-339, :380-384).
 *     Algorithm (curand_kernel.h, "XORWOW"; Marsaglia, Xorshift RNGs, JSS 8(14), 2003, with the Weyl sequence d += 362437):
 *       state = five 32-bit words v[0..4] and d;  t = v0 ^ (v0 >> 2); v0..v3 = v1..v4; v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1)); d += 362437;
 *       output v4 + d.  curand_init scrambles the seed into the state (constants below), then skips subsequence * 2^67 outputs -- a power of
 *       the GF(2)-linear transition of v (d is untouched: 362437 * 2^67 = 0 mod 2^32) -- then `offset` outputs.
 *       curand_uniform(x) = x * 2^-32 + 2^-33 in float: (0, 1].
 *     The skip-ahead matrices are COMPUTED here from the transition (vrc_build_sequence_jumps); tests hold them against rocRAND's
 *     published table of the same recurrence (/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h, A^(2^67)) and the stream against an
 *     independent big-integer implementation (tests/golden/gen_golden_xorwow.py).  NOT verifiable in this container: the four seed-scramble
 *     constants of curand_init, which are taken from the public header curand_kernel.h as remembered (rocRAND deliberately uses others).
 *
 * D2  CUDA linear texture filtering as the reference uses it (gpu-kernels/gmat.h:49-62: ONE pitched 2-D texture of height depth * height
 *     over all layers, clamp addressing, linear filter, unnormalised coordinates; :175-179: tex2D(x + 0.5f, d * _height + y + 0.5f)):
 *     CUDA C Programming Guide, "Texture Fetching / Linear Filtering":  xB = x - 0.5, i = floor(xB), alpha = frac(xB) "stored in 9-bit fixed
 *     point format with 8 bits of fractional value", tex = (1-a)(1-b) T[i,j] + a (1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1], indices
 *     clamped to the texture -- the STACKED image, so the row below the last row of layer d is row 0 of layer d + 1 (only the ends of
 *     the whole stack clamp).  The guide does not say how the fraction is rounded to 8 bits nor in which order the unit adds the four
 *     terms: taken here as round-to-nearest (floor(a * 256 + 0.5)) and left-to-right fp32 -- "parity unpinned" for those two choices. */
#ifndef VK_REF_CUDA_H
#define VK_REF_CUDA_H
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VRC_FN __host__ __device__ static inline
#else
#define VRC_FN static inline
#endif
#if defined(__clang__)
#define VRC_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define VRC_NO_CONTRACT /* gcc: the oracle and the reference build are compiled with -ffp-contract=off */
#endif

/* ---- D1: XORWOW ---------------------------------------------------------------------------------------------------------------- */
typedef struct { uint32_t v[5]; uint32_t d; } vrc_xorwow;
#define VRC_XW_BITS 160
#define VRC_XW_MAT (VRC_XW_BITS * 5) /* a transition matrix: for every state bit b = 32 i + j (bit j of v[i]) the five words of its image */

/* curand_init(seed, 0, 0): the seed scrambled into the state (curand_kernel.h, _curand_init_scratch) */
VRC_FN void vrc_xorwow_seed(unsigned long long seed, vrc_xorwow* s) {
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0;
    s->v[1] = 362436069u ^ t0;
    s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;
    s->v[4] = 5783321u + t0;
}
/* curand(&state) */
VRC_FN uint32_t vrc_xorwow_next(vrc_xorwow* s) {
    const uint32_t t = s->v[0] ^ (s->v[0] >> 2);
    s->v[0] = s->v[1]; s->v[1] = s->v[2]; s->v[2] = s->v[3]; s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437u;
    return s->v[4] + s->d;
}
/* curand_uniform: _curand_uniform(x) = x * CURAND_2POW32_INV + CURAND_2POW32_INV / 2 (float; the product with 2^-32 is exact, one rounding) */
VRC_FN float vrc_uniform(uint32_t x) {
    VRC_NO_CONTRACT
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}
/* v <- M v over GF(2) */
VRC_FN void vrc_matvec(const uint32_t* M, uint32_t* v) {
    uint32_t r[5] = { 0u, 0u, 0u, 0u, 0u };
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++) {
            const uint32_t b = (v[i] >> j) & 1u ? 0xffffffffu : 0u;
            const uint32_t* row = M + (size_t)(32 * i + j) * 5;
            for (int k = 0; k < 5; k++) r[k] ^= b & row[k];
        }
    for (int k = 0; k < 5; k++) v[k] = r[k];
}
/* the state of subsequence `sub` from the state of subsequence 0: J[k] = T^(2^67 * 2^k) (vrc_build_sequence_jumps), k < 32 */
VRC_FN void vrc_xorwow_skip_subsequences(const uint32_t* J, uint32_t sub, vrc_xorwow* s) {
    for (int k = 0; k < 32; k++)
        if ((sub >> k) & 1u) vrc_matvec(J + (size_t)k * VRC_XW_MAT, s->v);
}
/* curand_init(seed, sub, 0, &state) */
VRC_FN void vrc_xorwow_init(const uint32_t* J, unsigned long long seed, uint32_t sub, vrc_xorwow* s) {
    vrc_xorwow_seed(seed, s);
    vrc_xorwow_skip_subsequences(J, sub, s);
}

/* host: the transition of one output, C = A o B, and the 32 sequence jumps */
static inline void vrc_mat_step(uint32_t* T) {
    for (int b = 0; b < VRC_XW_BITS; b++) {
        vrc_xorwow s;
        for (int k = 0; k < 5; k++) s.v[k] = 0u;
        s.d = 0u;
        s.v[b / 32] = 1u << (b % 32);
        (void)vrc_xorwow_next(&s);
        for (int k = 0; k < 5; k++) T[(size_t)b * 5 + k] = s.v[k];
    }
}
static inline void vrc_mat_mul(const uint32_t* A, const uint32_t* B, uint32_t* C) { /* C = A after B; C may not alias A or B */
    for (int b = 0; b < VRC_XW_BITS; b++) {
        uint32_t col[5];
        for (int k = 0; k < 5; k++) col[k] = B[(size_t)b * 5 + k];
        vrc_matvec(A, col);
        for (int k = 0; k < 5; k++) C[(size_t)b * 5 + k] = col[k];
    }
}
static inline void vrc_build_sequence_jumps(uint32_t* J /* [32][VRC_XW_MAT] */) {
    uint32_t a[VRC_XW_MAT], b[VRC_XW_MAT];
    vrc_mat_step(a);
    for (int i = 0; i < 67; i++) { vrc_mat_mul(a, a, b); for (int k = 0; k < VRC_XW_MAT; k++) a[k] = b[k]; }  /* T^(2^67) */
    for (int k = 0; k < 32; k++) {
        for (int q = 0; q < VRC_XW_MAT; q++) J[(size_t)k * VRC_XW_MAT + q] = a[q];
        vrc_mat_mul(a, a, b);
        for (int q = 0; q < VRC_XW_MAT; q++) a[q] = b[q];
    }
}

/* ---- D2: linear filtering of the stacked layers ----------------------------------------------------------------------------------- */
typedef struct { int i00, i10, i01, i11; float w00, w10, w01, w11; } vrc_tex;  /* element offsets into the layer STACK [n_layers * h][w] and their weights */
VRC_FN float vrc_tex_frac8(float f) {  /* 9-bit fixed point, 8 fractional bits (1.0 representable); rounding: nearest (see the header note) */
    VRC_NO_CONTRACT
    const float q = f * 256.0f + 0.5f;
    return (float)(int)q * 0.00390625f;  /* q >= 0: the conversion truncates = floor */
}
VRC_FN vrc_tex vrc_tex_setup(float x, float y, int layer, int w, int h, int n_layers) {
    VRC_NO_CONTRACT
    /* gmat.h:178: tex2D(_tex_obj, x + 0.5f, d * _height + y + 0.5f) -- the caller's float sums, then the unit's xB = x - 0.5 */
    const float X = x + 0.5f, Y = ((float)(layer * h) + y) + 0.5f;
    const float xB = X - 0.5f, yB = Y - 0.5f;
    float fi = (float)(int)xB, fj = (float)(int)yB;
    if (fi > xB) fi -= 1.0f;  /* floor for negative positions (clamped below anyway) */
    if (fj > yB) fj -= 1.0f;
    const float a = vrc_tex_frac8(xB - fi), b = vrc_tex_frac8(yB - fj);
    const int H = (n_layers > 0 ? n_layers : 1) * h;
    int i0 = (int)fi, j0 = (int)fj, i1 = i0 + 1, j1 = j0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > w - 1 ? w - 1 : i0); i1 = i1 < 0 ? 0 : (i1 > w - 1 ? w - 1 : i1);
    j0 = j0 < 0 ? 0 : (j0 > H - 1 ? H - 1 : j0); j1 = j1 < 0 ? 0 : (j1 > H - 1 ? H - 1 : j1);  /* clamp at the ends of the STACK only */
    vrc_tex t;
    t.i00 = j0 * w + i0; t.i10 = j0 * w + i1; t.i01 = j1 * w + i0; t.i11 = j1 * w + i1;
    t.w00 = (1.0f - a) * (1.0f - b); t.w10 = a * (1.0f - b); t.w01 = (1.0f - a) * b; t.w11 = a * b;
    return t;
}
VRC_FN float vrc_tex_fetch1(const float* stack, float x, float y, int layer, int w, int h, int n_layers) {
    VRC_NO_CONTRACT
    const vrc_tex t = vrc_tex_setup(x, y, layer, w, h, n_layers);
    return t.w00 * stack[t.i00] + t.w10 * stack[t.i10] + t.w01 * stack[t.i01] + t.w11 * stack[t.i11];
}
VRC_FN void vrc_tex_fetch2(const float* stack /* interleaved (x, y) texels */, float x, float y, int layer, int w, int h, int n_layers, float* ox, float* oy) {
    VRC_NO_CONTRACT
    const vrc_tex t = vrc_tex_setup(x, y, layer, w, h, n_layers);
    *ox = t.w00 * stack[2 * t.i00] + t.w10 * stack[2 * t.i10] + t.w01 * stack[2 * t.i01] + t.w11 * stack[2 * t.i11];
    *oy = t.w00 * stack[2 * t.i00 + 1] + t.w10 * stack[2 * t.i10 + 1] + t.w01 * stack[2 * t.i01 + 1] + t.w11 * stack[2 * t.i11 + 1];
}

#endif /* VK_REF_CUDA_H */
