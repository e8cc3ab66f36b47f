/* stands in for gpu-kernels/reduce_vector_sum.h when the reference's .cu files are compiled for the CPU (cuda_emul.h).
 * The reference kernel needs __syncthreads and warp lock-step, which a sequential launcher cannot give, so its
 * SUMMATION ORDER is restated here (reduce_vector_sum.h:12-42,45-61): per level, blocks of 2*block_size elements; thread
 * t starts from x[t] + x[t + block_size]; then a binary tree over strides block_size/2 ... 1; the block sums are appended
 * behind the level's data and become the next level.  Same buffer contract as the reference (data_ext has room for the
 * partial sums of every level).  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include "cuda_emul.h"
template <int block_size> static int reduce_vector_sum(float* d_data_ext, float* h_o_data, int N, int dims) {
    int N_remain = N;
    float* lvl = d_data_ext;
    float s[block_size];
    while (N_remain > 1) {
        const int n_blocks = (N_remain + 2 * block_size - 1) / (2 * block_size);
        for (int b = 0; b < n_blocks; b++) for (int d = 0; d < dims; d++) {
            for (int t = 0; t < block_size; t++) {
                const int idx = b * 2 * block_size + t;
                s[t] = 0;
                if (idx < N_remain) { s[t] = lvl[idx * dims + d]; if (idx + block_size < N_remain) s[t] += lvl[(idx + block_size) * dims + d]; }
            }
            for (int stride = block_size / 2; stride >= 1; stride >>= 1) for (int t = 0; t < stride; t++) s[t] += s[t + stride];
            lvl[(N_remain + b) * dims + d] = s[0];
        }
        lvl += N_remain * dims;
        N_remain = n_blocks;
    }
    memcpy(h_o_data, lvl, dims * sizeof(float));
    return 0;
}
