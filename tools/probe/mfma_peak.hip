// Probe: sustained v_mfma_f32_32x32x16_bf16 rate with no memory traffic (how much of the 2.5 PFLOP/s
// datasheet figure a pure MFMA loop reaches on this box).  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC> __global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x % 7 + i)); b[i] = (__bf16)(seed * (threadIdx.x % 5 + 2 * i)); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int blocks = 256 * 2, iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        const float seed = zero ? 0.f : 0.37f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            mfma_loop<4><<<blocks, 256>>>(out, iters, seed);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*NACC*/ * 32768.0;
            printf("data=%s  %d blocks x 4 waves, 4 acc: %.2f ms  %.1f TFLOP/s\n", zero ? "zeros" : "random", blocks, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
