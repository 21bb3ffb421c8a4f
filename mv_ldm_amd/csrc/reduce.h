// Second stage of the deterministic two-stage column reductions of the training kernels (bias gradients, norm parameter
// gradients): out[row][col] = / += sum over k < P of part[(row * P + k) * n + col].
// The first stages write P = several hundred to two thousand partial rows so that they fill the chip; folding them with one
// thread per column (two workgroups on the whole GPU walking P dependent loads) cost 25-100 us per launch -- more than the
// first stage.  Here a workgroup owns four columns of one output row, its 256 threads stride over the P partial rows with
// independent 16-byte loads, and a fixed shuffle / LDS tree finishes: same result for the same P on every run.
#pragma once
#include "common.h"

namespace mvldm {

// MODE 0: out0[row * ld_out + col], col < n_valid.   MODE 1: the columns are (dbeta, dgamma) pairs of channel col / 2:
// out0[ch] (+)= even columns, out1[ch] (+)= odd columns.  `accumulate` 0: the outputs are written (a window's first write), 1: += .
template <int MODE>
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ part, int P, int n, int n_valid, float* __restrict__ out0,
                                                            float* __restrict__ out1, int ld_out, int accumulate) {
    __shared__ float s_w[4][4];
    const int g4 = blockIdx.x, row = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = part + (size_t)row * P * n + (size_t)g4 * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < P; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)k * n);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = wave_sum(a[e]);
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s_w[wave][e] = a[e];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float t = (s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + (s_w[2][threadIdx.x] + s_w[3][threadIdx.x]);
        const int col = g4 * 4 + threadIdx.x;
        if (col < n_valid) {
            if constexpr (MODE == 0) {
                float* o = out0 + (size_t)row * ld_out + col;
                *o = accumulate ? *o + t : t;
            } else {
                float* o = ((col & 1) ? out1 : out0) + (col >> 1);
                *o = accumulate ? *o + t : t;
            }
        }
    }
}

}  // namespace mvldm
