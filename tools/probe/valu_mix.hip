// Probe: do plain VALU instructions issue in the shadow of v_exp_f32 (8 cycles per wave64)?  Per iteration 16 independent
// v_exp_f32 plus N x 16 independent v_fma_f32 on other registers, interleaved one exp : N fma.  If the time per iteration stays
// at the exp-only time, the transcendental pipe runs beside the main VALU; if it grows by N x 16 x 2 cycles, they share the port.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NF> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float e[16], f[16];
    for (int i = 0; i < 16; ++i) { e[i] = seed + threadIdx.x * 1e-3f + i; f[i] = seed * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
#pragma unroll
            for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(i + 5 * j) & 15]) : "v"(seed));
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += e[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NF> void run() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256, 1024}) {
        const int iters = 20000;
        k<NF><<<blocks, 256>>>(out, 10, 1.f);
        (void)hipEventRecord(e0); k<NF><<<blocks, 256>>>(out, iters, 1.f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double waves_per_simd = blocks * 4.0 / 1024.0;
        printf("1 exp : %d fma, %4d blocks: %.3f ms -> %.2f ns per (exp + %d fma) group per wave per SIMD\n", NF, blocks, ms, ms * 1e6 / ((double)iters * 16 * waves_per_simd), NF);
    }
}
int main() { run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); return 0; }
