// Shared device helpers for libmvldm_hip.so (gfx950 / CDNA4 only: wave64, MFMA 32x32, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <type_traits>

#include "../../include/mvldm.h"

namespace mvldm {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

int set_error(int code, const char* fmt, ...);
#define MVLDM_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ::mvldm::set_error(MVLDM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define MVLDM_REQUIRE(cond, ...)                                        \
    do {                                                                \
        if (!(cond)) return ::mvldm::set_error(MVLDM_ERR_ARG, __VA_ARGS__); \
    } while (0)
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: raise it once per (kernel, device).
// `mask` is the kernel's own bitmap of devices already done (one `static std::atomic<uint64_t>` per launch site); a race
// between two threads only repeats the idempotent call.
inline int ensure_dyn_smem(const void* kern, int smem, std::atomic<uint64_t>& mask) {
    if (smem <= 48 * 1024) return MVLDM_OK;
    int dev = 0;
    MVLDM_CHECK_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return MVLDM_OK;
    MVLDM_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    mask.fetch_or(bit, std::memory_order_release);
    return MVLDM_OK;
}
inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(MVLDM_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return MVLDM_OK;
}

// ---- element traits ------------------------------------------------------------------------------
template <typename T> struct Elt;
template <> struct Elt<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int DT = MVLDM_F32;
};
template <> struct Elt<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int DT = MVLDM_BF16;
};
template <> struct Elt<f16_t> {
    static constexpr int EPC = 8;
    static constexpr int DT = MVLDM_F16;
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }  // RNE for bf16/f16

// A 16-byte chunk viewed as EPC elements of T.
template <typename T> struct Chunk {
    static constexpr int N = Elt<T>::EPC;
    union {
        u32x4 raw;
        T e[N];
    };
    __device__ __forceinline__ Chunk() {}
    __device__ __forceinline__ void zero() { raw = u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ float get(int i) const { return to_f32<T>(e[i]); }
    __device__ __forceinline__ void set(int i, float v) { e[i] = from_f32<T>(v); }
};

template <typename T> __device__ __forceinline__ Chunk<T> load_chunk(const T* p) {
    Chunk<T> c;
    c.raw = *reinterpret_cast<const u32x4*>(p);
    return c;
}
template <typename T> __device__ __forceinline__ void store_chunk(T* p, const Chunk<T>& c) {
    *reinterpret_cast<u32x4*>(p) = c.raw;
}
// host: should a kernel that writes `bytes` of output use streaming stores?  Outputs of half the 256 MB Infinity Cache or more;
// MVLDM_STREAM_STORES=0 / 1 forces it (A/B knob).
// A/B and tuning knobs of decisions already made exist in EXPERIMENT builds only (-DMVLDM_EXPERIMENTS: tools/*_probe.sh,
// MVLDM_EXPERIMENTS=1 python -m mv_ldm_amd._build): the product library's kernels and dispatch read no environment variable.
static inline int knob_int(const char* name, int dflt) {
#ifdef MVLDM_EXPERIMENTS
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
static inline int stream_stores(size_t bytes) {
    static const int force = knob_int("MVLDM_STREAM_STORES", -1);
    return force >= 0 ? force : (bytes >= ((size_t)128 << 20));
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// below fp32 round-off of the product for the tolerances in use): one rcp + one exp + a degree-5 Horner
// chain instead of libm's erff -- the GEGLU epilogue evaluates it for every element of the 8C-wide FF.
__device__ __forceinline__ float erf_as_f(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float y = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }

// The same exact-GELU through the same A-S 7.1.26 polynomial, arranged for the GEGLU epilogue (which evaluates
// it for every element of the 8C-wide FF and was VALU-bound: PMC showed 61 VALU instructions per output).
// gelu(x) = x * Phi(x);  with z = |x|/sqrt(2), t = 1/(1 + p z):  q = 0.5 * erfc(z) = 0.5 * poly(t) * t * exp(-z^2);
// Phi = x >= 0 ? 1 - q : q.  Raw v_rcp_f32 / v_exp_f32 (1 ulp each, no denormal fix-up sequences: t is in
// (0,1], and an underflowing exp is the correct 0), the 0.5 folded into the coefficients, exp(-z^2) as
// exp2(x^2 * -0.5*log2(e)): 17 VALU instructions.  |error| <= 1.5e-7 * |x| as before.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
    const float q = (poly * t) * e;
    const float phi = x >= 0.f ? 1.0f - q : q;
    return x * phi;
}

// The 16-bit kernels' GELU (round 6): the same construction on Abramowitz-Stegun 7.1.25 (three coefficients, |error| <= 2.5e-5 in erf,
// i.e. <= 1.25e-5 in Phi) -- 20x below the rounding of an f16 output (2^-11), 80x below bf16's: two FMAs fewer of the 17 VALU
// instructions the GEGLU epilogues spend per element (the level-0 GEGLU projection evaluates 755 M of them per launch, ~ 25 % of its
// time).  The f32 kernels keep gelu_erf_fast / gelu_erf_f (their parity bound against the oracle is 1e-6-class).
__device__ __forceinline__ float gelu_erf_16(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f * 0.70710678118654752440f, ax, 1.0f));
    float poly = fmaf(0.5f * 0.7478556f, t, 0.5f * -0.0958798f);
    poly = fmaf(poly, t, 0.5f * 0.3480242f);
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
    const float q = (poly * t) * e;
    const float phi = x >= 0.f ? 1.0f - q : q;
    return x * phi;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware remap of a linear workgroup id: hardware places workgroup b on XCD b % 8, so give each
// XCD one contiguous range of logical ids (bijective for any n).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    constexpr int X = 8;
    int q = n / X, r = n % X;
    int xcd = b % X, i = b / X;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

template <typename F> inline int dispatch_dtype(int dtype, F&& f) {
    switch (dtype) {
        case MVLDM_F32: return f(float{});
        case MVLDM_BF16: return f(bf16_t{});
        case MVLDM_F16: return f(f16_t{});
        default: return set_error(MVLDM_ERR_ARG, "unknown dtype %d", dtype);
    }
}
inline size_t dtype_size(int dtype) { return dtype == MVLDM_F32 ? 4 : 2; }

}  // namespace mvldm
