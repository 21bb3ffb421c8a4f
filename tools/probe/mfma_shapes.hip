// Probe: sustained bf16 MFMA rate under the power limit, by instruction shape -- v_mfma_f32_32x32x16_bf16 (this library's kernels) against
// v_mfma_f32_16x16x32_bf16 (the vendor GEMM's, MI16x16x1) -- with no memory traffic: 4 x 4 operand fragments of random N(0,1)-like
// values, 16 accumulator tiles per wave, one wave per SIMD or two.  Per 32768 flop the 32x32 shape moves 8 operand + 32 accumulator
// registers through the register file, the 16x16 shape 16 operand + 16 accumulator registers.
// hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip ; ./mfma_shapes [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float rnd(unsigned& s) {      // sum of 4 uniforms, roughly N(0, 1)
    float a = 0.f;
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    return a * 1.7320508f;
}

template <bool BIG> __global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, int zero) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    bf16x8 a[4], b[4];
    for (int f = 0; f < 4; ++f)
        for (int i = 0; i < 8; ++i) { a[f][i] = (__bf16)(zero ? 0.f : rnd(s)); b[f][i] = (__bf16)(zero ? 0.f : rnd(s)); }
    float sum = 0.f;
    if constexpr (BIG) {
        f32x16 acc[16];
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    } else {
        f32x4 acc[16];
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) sum += acc[j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}


// the same loop on f16 operands (the reference's own 16-mixed type): 10 mantissa bits in the multipliers instead of 7
__global__ __launch_bounds__(256) void mfma_loop_f16(float* out, int iters, int zero) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    f16x8 a[4], b[4];
    for (int f = 0; f < 4; ++f)
        for (int i = 0; i < 8; ++i) { a[f][i] = (_Float16)(zero ? 0.f : rnd(s)); b[f][i] = (_Float16)(zero ? 0.f : rnd(s)); }
    f32x16 acc[16];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
    }
    float sum = 0.f;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    float* out; hipMalloc(&out, 1024 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero)
        for (int wps = 1; wps <= 2; ++wps)
            for (int big = 1; big >= 0; --big) {
                const int blocks = 256 * wps;
                const double fl_it = big ? 16 * 32768.0 : 16 * 16384.0;
                int iters = big ? 20000 : 40000;
                double best = 0, last = 0, t_tot = 0;
                int reps = 0;
                while (t_tot < secs * 1e3) {       // back-to-back launches for `secs`: the figure of the LAST launch is the settled one
                    hipEventRecord(e0);
                    if (big) mfma_loop<true><<<blocks, 256>>>(out, iters, zero); else mfma_loop<false><<<blocks, 256>>>(out, iters, zero);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    last = (double)blocks * 4 * iters * fl_it / ms / 1e9;
                    best = last > best ? last : best;
                    t_tot += ms; ++reps;
                }
                printf("%-7s %s  %d wave(s)/SIMD: first-launch-best %.0f  settled %.0f TFLOP/s  (%d launches)\n", zero ? "zeros" : "random",
                       big ? "32x32x16" : "16x16x32", wps, best, last, reps);
            }
    for (int zero = 0; zero < 2; ++zero) {
        const int blocks = 512, iters = 20000;
        double last = 0, t_tot = 0;
        while (t_tot < secs * 1e3) {
            hipEventRecord(e0);
            mfma_loop_f16<<<blocks, 256>>>(out, iters, zero);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            last = (double)blocks * 4 * iters * 16 * 32768.0 / ms / 1e9;
            t_tot += ms;
        }
        printf("%-7s 32x32x16 f16, 2 wave(s)/SIMD: settled %.0f TFLOP/s\n", zero ? "zeros" : "random", last);
    }
    return 0;
}
