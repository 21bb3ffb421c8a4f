// Probe: issue cost of v_exp_f32 / v_rcp_f32 / v_fma_f32 / v_pk_fma_f32 / v_max3_f32 per wave64 instruction (cycles per SIMD),
// from a dependent-free unrolled loop at 1 wave per SIMD and at 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float v[16]; f32x2 w[8];
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) w[i] = f32x2{v[2 * i], v[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
            if (OP == 1) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            if (OP == 2) v[i] = __builtin_amdgcn_rcpf(v[i]);
            if (OP == 4) v[i] = __builtin_fmaxf(__builtin_fmaxf(v[i], v[(i + 1) & 15]), v[(i + 2) & 15]);
            if (OP == 5) { _Float16 h = (_Float16)v[i]; asm volatile("v_exp_f16 %0, %1" : "=v"(h) : "v"(h)); v[i] = (float)h; }   // + 2 conversions
            if (OP == 6) { _Float16 h = (_Float16)v[i]; asm volatile("v_mov_b32 %0, %1" : "=v"(h) : "v"(h)); v[i] = (float)h; }   // the conversions alone
        }
        if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = __builtin_elementwise_fma(w[i], f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += v[i]; for (int i = 0; i < 8; ++i) s += w[i][0] + w[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int per_iter) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 1024}) {
        const int iters = 20000;
        k<OP><<<blocks, 256>>>(out, 10, 1.f);
        hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves_per_simd = blocks * 4.0 / 1024.0;
        const double inst = (double)iters * per_iter * waves_per_simd;      // wave-instructions per SIMD
        printf("%-14s %4d blocks: %.3f ms -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, blocks, ms, ms * 1e6 / inst);
    }
}
int main() { run<1>("v_fma_f32", 16); run<0>("v_exp_f32", 16); run<2>("v_rcp_f32", 16); run<3>("v_pk_fma_f32", 8); run<4>("v_max3_f32", 16);
    run<5>("v_exp_f16+2cvt", 16); run<6>("v_mov+2cvt", 16); return 0; }
