// Probe: how should the matrix work and the softmax VALU work of one attention tile share a SIMD?  Register-only model of the
// d = 40 tile of attention.hip per wave (32 queries x 64 keys): 6 QK^T MFMAs (32x32x16 bf16), 8 PV MFMAs, and the softmax between
// them: 18 max, 33 fma + 33 v_exp_f32, 16 v_cvt_pk.  No memory, no LDS: only the issue structure differs between the modes.
//   MODE 0  serial in one wave (QK, softmax, PV), W independent waves per SIMD              = attention_kernel
//   MODE 1  software-pipelined in one wave: QK(t+1) and PV(t-1) MFMAs interleaved with softmax(t) (sched_group_barrier 1 : 8)
//   MODE 2  ping-pong: waves w / w+4 of an 8-wave workgroup a barrier apart (matrix segment | VALU segment)  = attention_pp_kernel
//   MODE 3  ping-pong with 16 waves per workgroup (two waves of a SIMD in each role)
// Output: cycles per wave-tile per SIMD (clock from the wall-clock of a calibrated s_memtime loop is not needed: we print ns and
// the v_fma reference of the same launch geometry).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct St {
    f32x16 s0, s1, o0, o1;
    bf16x8 kf[3], qf[3], vf[4], pf[4];     // (fragments re-used by both chains: the instruction mix is what matters)
    float m_run, c;
};

__device__ __forceinline__ void qk(St& t, f32x16& s0, f32x16& s1) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.kf[0], t.qf[0], z, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.kf[1], t.qf[0], z, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.kf[1], t.qf[1], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.kf[2], t.qf[1], s1, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.kf[2], t.qf[2], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.kf[0], t.qf[2], s1, 0, 0, 0);
}
#ifndef O_AGPR
#define O_AGPR 0
#endif
__device__ __forceinline__ void pv(St& t) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#if O_AGPR
        // the O accumulators live in the accumulation registers: the MFMA's C / D traffic stays off the VGPR ports
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(t.o0) : "v"(t.vf[ks]), "v"(t.pf[ks]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(t.o1) : "v"(t.vf[(ks + 1) & 3]), "v"(t.pf[ks]));
#else
        t.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.vf[ks], t.pf[ks], t.o0, 0, 0, 0);
        t.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.vf[(ks + 1) & 3], t.pf[ks], t.o1, 0, 0, 0);
#endif
    }
}
__device__ __forceinline__ void softmax(St& t, f32x16& s0, f32x16& s1) {
    float mx = s0[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    const float m_new = fmaxf(t.m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((t.m_run - m_new) * t.c);
    t.m_run = m_new;
    const float mc = -m_new * t.c;
    if (!__all(alpha == 1.0f)) {
#if O_AGPR
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(t.o0), "+a"(t.o1));     // MFMA D -> non-MFMA reader: wait states
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) { t.o0[r] *= alpha; t.o1[r] *= alpha; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], t.c, mc));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], t.c, mc));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.pf[ks][j] = (__bf16)((ks >> 1) ? s1[(ks & 1) * 8 + j] : s0[(ks & 1) * 8 + j]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(t.pf[ks]));
}
__device__ __forceinline__ void init(St& t, float seed) {
    const float x = seed + (threadIdx.x & 63) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.kf[i][j] = (__bf16)(x * (i + 1) + j * 0.01f);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.qf[i][j] = (__bf16)(0.3f - x * (i + 1) + j * 0.02f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.vf[i][j] = (__bf16)(0.1f * x + i * 0.01f + j * 0.03f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.pf[i][j] = (__bf16)0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { t.o0[r] = 0.f; t.o1[r] = 0.f; t.s0[r] = 0.f; t.s1[r] = 0.f; }
    t.m_run = -1e30f;
    t.c = 0.2f;
}
__device__ __forceinline__ void touch(St& t) {       // new operands every tile: nothing is loop-invariant
    asm volatile("" : "+v"(t.qf[0]), "+v"(t.kf[0]), "+v"(t.vf[0]));
}
__device__ __forceinline__ float fold(const St& t) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += t.o0[r] + t.o1[r];
    return a;
}

template <int MODE, int NT> __global__ __launch_bounds__(NT) void k(float* out, int iters, float seed) {
    extern __shared__ char pad[];
    St t;
    init(t, seed);
    if constexpr (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            touch(t);
            qk(t, t.s0, t.s1);
            softmax(t, t.s0, t.s1);
            pv(t);
        }
    } else if constexpr (MODE == 1) {
        f32x16 n0, n1;
        qk(t, t.s0, t.s1);
        for (int it = 0; it < iters; ++it) {
            touch(t);
            pv(t);                     // P of the previous tile
            qk(t, n0, n1);             // S of the next tile
            softmax(t, t.s0, t.s1);    // this tile
#pragma unroll
            for (int g = 0; g < 14; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            t.s0 = n0; t.s1 = n1;
        }
    } else {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int grp = wave >= (NT / 128);
        if (grp) __builtin_amdgcn_s_barrier();
        qk(t, t.s0, t.s1);
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        softmax(t, t.s0, t.s1);
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        for (int it = 1; it < iters; ++it) {
            touch(t);
            pv(t);
            qk(t, t.s0, t.s1);
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
            softmax(t, t.s0, t.s1);
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        }
        pv(t);
        if (!grp) __builtin_amdgcn_s_barrier();
    }
#if O_AGPR
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(t.o0), "+a"(t.o1));
#endif
    out[(size_t)blockIdx.x * NT + threadIdx.x] = fold(t) + t.m_run;
}

template <int MODE, int NT> void run(const char* name, int wg_per_cu) {
    float* out;
    (void)hipMalloc(&out, (size_t)256 * 8 * 1024 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * wg_per_cu;
    const int lds = 160 * 1024 / wg_per_cu - 1024;           // caps residency at wg_per_cu workgroups per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<MODE, NT><<<blocks, NT, lds>>>(out, 10, 1.f);
    (void)hipEventRecord(e0);
    k<MODE, NT><<<blocks, NT, lds>>>(out, iters, 1.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = (double)wg_per_cu * NT / 256.0;
    printf("%-52s %2.0f waves/SIMD: %8.3f ms -> %7.1f ns per wave-tile per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / (iters * waves_per_simd));
    (void)hipFree(out);
}

int main() {
    run<0, 256>("serial, 4-wave workgroups", 1);
    run<0, 256>("serial, 4-wave workgroups", 2);
    run<0, 256>("serial, 4-wave workgroups", 3);
    run<0, 256>("serial, 4-wave workgroups", 4);
    run<1, 256>("software-pipelined 1 MFMA : 8 VALU", 1);
    run<1, 256>("software-pipelined 1 MFMA : 8 VALU", 2);
    run<1, 256>("software-pipelined 1 MFMA : 8 VALU", 3);
    run<1, 256>("software-pipelined 1 MFMA : 8 VALU", 4);
    run<2, 512>("ping-pong, 8-wave workgroup", 1);
    run<2, 512>("ping-pong, 8-wave workgroup", 2);
    run<3, 1024>("ping-pong, 16-wave workgroup", 1);
    return 0;
}
