// Backward of GroupNorm(+SiLU) and LayerNorm for NHWC / token-major activations (include/mvldm.h, "Training").
//
// GroupNorm: y = act(z * gamma + beta), z = (x - mean_g) * rstd_g over (pixels, channels of the group) of one image,
// act = SiLU or identity.  With dg = dy * act'(z gamma + beta), dz = dg * gamma:
//     dgamma[c] = sum dg * z      dbeta[c] = sum dg      dx = rstd * (dz - mean_g(dz) - z * mean_g(dz * z))
// Four launches: (1) per (image, row slab): per-channel partial sums of dg and dg*z (a thread owns one 16-byte channel
// column, fixed order, no atomics to HBM); (2) per (image, group): fold the image's partials into the two group means (once --
// the first version re-folded them in every one of the 512-1024 slab workgroups of step 3); (3) per (image, row slab): write
// dx (x and dy are read a second time: L2/MALL-resident at these sizes); (4) fold the partials over images and slabs into
// dgamma / dbeta (reduce.h; accumulating: parameters may collect several micro-batches).
// (mean, rstd) come from the forward pass (`stats_out` of mvldm_groupnorm_fwd).  The input may be the channel concat of
// two tensors (the up-path skip concat): dx is then written as two tensors.
//
// LayerNorm: one wave per row, the row in registers, mean / rstd recomputed (cheaper than saving them); per-block
// partials of dgamma / dbeta folded by a second launch.
#include <algorithm>

#include "common.h"
#include "reduce.h"

namespace mvldm {

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0, const T* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ stats, float* __restrict__ part, int hw, int c, int groups,
                                                             int rows_per_chunk, int nchunk, int silu) {
    constexpr int EPC = Elt<T>::EPC;
    extern __shared__ float s_acc[];    // [RB][span][EPC][2]
    const int img = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int ncc = c / EPC, cpg = c / groups;
    const int r0 = chunk * rows_per_chunk, r1 = min(hw, r0 + rows_per_chunk);
    float* out = part + ((size_t)img * nchunk + chunk) * c * 2;
    for (int cc0 = 0; cc0 < ncc; cc0 += 256) {
        const int span = min(ncc - cc0, 256);
        const int RB = max(1, 256 / span);
        const int cc = cc0 + threadIdx.x % span, rl = threadIdx.x / span;
        if (rl < RB) {
            const int ch0 = cc * EPC;
            const bool first = ch0 < c0;
            const int cs = first ? c0 : c - c0;
            const T* xs = (first ? x0 + ch0 : x1 + (ch0 - c0)) + (size_t)img * hw * cs;
            const T* ds = dy + (size_t)img * hw * c + ch0;
            float mu[EPC], rs[EPC], ga[EPC], be[EPC], a1[EPC], a2[EPC];
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const int g = (ch0 + i) / cpg;
                mu[i] = stats[((size_t)img * groups + g) * 2];
                rs[i] = stats[((size_t)img * groups + g) * 2 + 1];
                ga[i] = gamma[ch0 + i];
                be[i] = beta[ch0 + i];
                a1[i] = a2[i] = 0.f;
            }
            for (int r = r0 + rl; r < r1; r += RB) {
                const Chunk<T> xv = load_chunk<T>(xs + (size_t)r * cs), dv = load_chunk<T>(ds + (size_t)r * c);
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float z = (xv.get(i) - mu[i]) * rs[i];
                    float dg = dv.get(i);
                    if (silu) {
                        const float g = z * ga[i] + be[i], sg = sigm(g);
                        dg *= sg * (1.0f + g * (1.0f - sg));
                    }
                    a1[i] += dg;
                    a2[i] += dg * z;
                }
            }
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                s_acc[((rl * span + (cc - cc0)) * EPC + i) * 2] = a1[i];
                s_acc[((rl * span + (cc - cc0)) * EPC + i) * 2 + 1] = a2[i];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < span * EPC * 2; e += 256) {
            float t = 0.f;
            for (int r = 0; r < RB; ++r) t += s_acc[r * span * EPC * 2 + e];
            out[cc0 * EPC * 2 + e] = t;
        }
        __syncthreads();
    }
}

// means[img][g] = (mean_g(dz), mean_g(dz * z)), dz = dg * gamma: one wave per (image, group), fixed order
__global__ __launch_bounds__(256) void gn_bwd_means_kernel(const float* __restrict__ part, const float* __restrict__ gamma, float* __restrict__ means,
                                                           int n_img, int hw, int c, int groups, int nchunk) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n_img * groups) return;
    const int img = idx / groups, g = idx - img * groups, cpg = c / groups;
    float t1 = 0.f, t2 = 0.f;
    for (int e = lane; e < cpg * nchunk; e += 64) {
        const int k = e / cpg, ch = g * cpg + (e - k * cpg);
        const float* pp = part + (((size_t)img * nchunk + k) * c + ch) * 2;
        t1 += pp[0] * gamma[ch];
        t2 += pp[1] * gamma[ch];
    }
    t1 = wave_sum(t1);
    t2 = wave_sum(t2);
    if (lane == 0) {
        const float inv = 1.0f / ((float)hw * (float)cpg);
        means[(size_t)idx * 2] = t1 * inv;
        means[(size_t)idx * 2 + 1] = t2 * inv;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0, const T* __restrict__ dy,
                                                           T* __restrict__ dx0, T* __restrict__ dx1, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats,
                                                           const float* __restrict__ means, int hw, int c, int groups, int rows_per_chunk,
                                                           int nchunk, int silu) {
    constexpr int EPC = Elt<T>::EPC;
    const int img = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int ncc = c / EPC, cpg = c / groups;
    const float* mimg = means + (size_t)img * groups * 2;
    const int r0 = chunk * rows_per_chunk, r1 = min(hw, r0 + rows_per_chunk);
    for (int cc0 = 0; cc0 < ncc; cc0 += 256) {
        const int span = min(ncc - cc0, 256);
        const int RB = max(1, 256 / span);
        const int cc = cc0 + threadIdx.x % span, rl = threadIdx.x / span;
        if (rl >= RB) continue;
        const int ch0 = cc * EPC;
        const bool first = ch0 < c0;
        const int cs = first ? c0 : c - c0;
        const T* xs = (first ? x0 + ch0 : x1 + (ch0 - c0)) + (size_t)img * hw * cs;
        T* dst = (first ? dx0 + ch0 : dx1 + (ch0 - c0)) + (size_t)img * hw * cs;
        const T* ds = dy + (size_t)img * hw * c + ch0;
        float mu[EPC], rs[EPC], ga[EPC], be[EPC], m1[EPC], m2[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            const int g = (ch0 + i) / cpg;
            mu[i] = stats[((size_t)img * groups + g) * 2];
            rs[i] = stats[((size_t)img * groups + g) * 2 + 1];
            ga[i] = gamma[ch0 + i];
            be[i] = beta[ch0 + i];
            m1[i] = mimg[2 * g];
            m2[i] = mimg[2 * g + 1];
        }
        for (int r = r0 + rl; r < r1; r += RB) {
            const Chunk<T> xv = load_chunk<T>(xs + (size_t)r * cs), dv = load_chunk<T>(ds + (size_t)r * c);
            Chunk<T> o;
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const float z = (xv.get(i) - mu[i]) * rs[i];
                float dg = dv.get(i);
                if (silu) {
                    const float g = z * ga[i] + be[i], sg = sigm(g);
                    dg *= sg * (1.0f + g * (1.0f - sg));
                }
                o.set(i, rs[i] * (dg * ga[i] - m1[i] - z * m2[i]));
            }
            store_chunk<T>(dst + (size_t)r * cs, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------- LN
template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                     const float* __restrict__ gamma, float* __restrict__ part, int rows, int c, float eps) {
    constexpr int EPC = Elt<T>::EPC;
    extern __shared__ float s_p[];      // [4 waves][c][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncc = c / EPC;
    float a1[MAXCH][EPC], a2[MAXCH][EPC];
#pragma unroll
    for (int k = 0; k < MAXCH; ++k)
#pragma unroll
        for (int i = 0; i < EPC; ++i) a1[k][i] = a2[k][i] = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const T* xr = x + (size_t)row * c;
        const T* dr = dy + (size_t)row * c;
        Chunk<T> v[MAXCH], d[MAXCH];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int cc = lane + k * 64;
            if (cc < ncc) {
                v[k] = load_chunk<T>(xr + cc * EPC);
                d[k] = load_chunk<T>(dr + cc * EPC);
#pragma unroll
                for (int i = 0; i < EPC; ++i) s += v[k].get(i);
            }
        }
        const float mean = wave_sum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int cc = lane + k * 64;
            if (cc < ncc) {
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float t = v[k].get(i) - mean;
                    q += t * t;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int cc = lane + k * 64;
            if (cc < ncc) {
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float z = (v[k].get(i) - mean) * rstd, dg = d[k].get(i);
                    const float dz = dg * gamma[cc * EPC + i];
                    s1 += dz;
                    s2 += dz * z;
                    a1[k][i] += dg;
                    a2[k][i] += dg * z;
                }
            }
        }
        s1 = wave_sum(s1) / (float)c;
        s2 = wave_sum(s2) / (float)c;
        T* xo = dx + (size_t)row * c;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int cc = lane + k * 64;
            if (cc < ncc) {
                Chunk<T> o;
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float z = (v[k].get(i) - mean) * rstd;
                    o.set(i, rstd * (d[k].get(i) * gamma[cc * EPC + i] - s1 - z * s2));
                }
                store_chunk<T>(xo + cc * EPC, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int cc = lane + k * 64;
        if (cc < ncc) {
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                s_p[((size_t)wave * c + cc * EPC + i) * 2] = a1[k][i];
                s_p[((size_t)wave * c + cc * EPC + i) * 2 + 1] = a2[k][i];
            }
        }
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * c * 2;
    for (int e = threadIdx.x; e < c * 2; e += 256) out[e] = (s_p[e] + s_p[(size_t)c * 2 + e]) + (s_p[(size_t)2 * c * 2 + e] + s_p[(size_t)3 * c * 2 + e]);
}

// ---------------------------------------------------------------------------------------------- host
static void gn_chunks(int n_img, int hw, int& nchunk, int& rpc) {
    nchunk = std::min(MVLDM_GN_MAX_CHUNKS, std::max(1, std::min(hw / 8, (1024 + n_img - 1) / n_img)));
    rpc = (hw + nchunk - 1) / nchunk;
    nchunk = (hw + rpc - 1) / rpc;
}

int groupnorm_bwd_run(const void* x0, const void* x1, const void* dy, void* dx0, void* dx1, const float* gamma, const float* beta,
                      const float* stats, float* dgamma, float* dbeta, int n_img, int hw, int c0, int c1, int groups, int silu, int dtype,
                      float* ws, size_t ws_bytes, hipStream_t s) {
    if (n_img == 0 || hw == 0) return MVLDM_OK;
    const int accumulate = (silu & MVLDM_NORM_BWD_STORE) ? 0 : 1;      // dgamma / dbeta: += (default) or = (the window's first write)
    silu &= ~MVLDM_NORM_BWD_STORE;
    const int c = c0 + c1, epc = dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(x0 && dy && dx0 && gamma && beta && stats && dgamma && dbeta && ws, "groupnorm_bwd: null pointer");
    MVLDM_REQUIRE((c1 == 0) == (x1 == nullptr) && (c1 == 0) == (dx1 == nullptr), "groupnorm_bwd: x1/dx1/c1 mismatch");
    MVLDM_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0 && c0 % epc == 0 && c1 % epc == 0, "groupnorm_bwd: c=(%d,%d) groups=%d", c0, c1, groups);
    int nchunk, rpc;
    gn_chunks(n_img, hw, nchunk, rpc);
    const size_t n_part = (size_t)n_img * nchunk * c * 2;       // floats; the group means follow them
    MVLDM_REQUIRE((n_part + (size_t)n_img * groups * 2) * sizeof(float) <= ws_bytes && ((uintptr_t)ws % 16) == 0,
                  "groupnorm_bwd: workspace too small or not 16-byte aligned (need %zu bytes)", (n_part + (size_t)n_img * groups * 2) * sizeof(float));
    float* means = ws + n_part;
    const int span = std::min(c / epc, 256), rb = std::max(1, 256 / span);
    const size_t smem = (size_t)rb * span * epc * 2 * sizeof(float);
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(gn_bwd_partial_kernel<T>, dim3(n_img * nchunk), dim3(256), smem, s, reinterpret_cast<const T*>(x0),
                           reinterpret_cast<const T*>(x1), c0, reinterpret_cast<const T*>(dy), gamma, beta, stats, ws, hw, c, groups, rpc, nchunk, silu);
        int rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(gn_bwd_means_kernel, dim3((n_img * groups + 3) / 4), dim3(256), 0, s, ws, gamma, means, n_img, hw, c, groups, nchunk);
        rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3(n_img * nchunk), dim3(256), 0, s, reinterpret_cast<const T*>(x0), reinterpret_cast<const T*>(x1),
                           c0, reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(dx0), reinterpret_cast<T*>(dx1), gamma, beta, stats, means, hw, c,
                           groups, rpc, nchunk, silu);
        rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(fold_partials_kernel<1>, dim3(c * 2 / 4, 1), dim3(256), 0, s, ws, n_img * nchunk, c * 2, c * 2, dbeta, dgamma, 0, accumulate);
        return check_launch();
    });
}

int layernorm_bwd_run(const void* x, const void* dy, void* dx, const float* gamma, float* dgamma, float* dbeta, int rows, int c, float eps, int dtype,
                      float* ws, size_t ws_bytes, hipStream_t s) {
    if (rows == 0) return MVLDM_OK;
    const int accumulate = (dtype & MVLDM_NORM_BWD_STORE) ? 0 : 1;
    dtype &= ~MVLDM_NORM_BWD_STORE;
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    MVLDM_REQUIRE(x && dy && dx && gamma && dgamma && dbeta && ws, "layernorm_bwd: null pointer");
    MVLDM_REQUIRE(c % epc == 0 && c / epc <= 64 * 8, "layernorm_bwd: c=%d", c);
    const int blocks = std::max(1, std::min((rows + 3) / 4, 512));
    MVLDM_REQUIRE((size_t)blocks * c * 2 * sizeof(float) <= ws_bytes && ((uintptr_t)ws % 16) == 0,
                  "layernorm_bwd: workspace too small or not 16-byte aligned (need %zu bytes)", (size_t)blocks * c * 2 * sizeof(float));
    const size_t smem = (size_t)4 * c * 2 * sizeof(float);
    const int ncc = c / epc;
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
#define MVLDM_LN_BWD(K_)                                                                                                                  \
    {                                                                                                                                     \
        auto kern = ln_bwd_kernel<T, K_>;                                                                                                 \
        if (smem > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, s, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy),         \
                           reinterpret_cast<T*>(dx), gamma, ws, rows, c, eps);                                                           \
    }
        if (ncc <= 64) MVLDM_LN_BWD(1)
        else if (ncc <= 128) MVLDM_LN_BWD(2)
        else if (ncc <= 256) MVLDM_LN_BWD(4)
        else MVLDM_LN_BWD(8)
#undef MVLDM_LN_BWD
        int rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(fold_partials_kernel<1>, dim3(c * 2 / 4, 1), dim3(256), 0, s, ws, blocks, c * 2, c * 2, dbeta, dgamma, 0, accumulate);
        return check_launch();
    });
}

}  // namespace mvldm

using namespace mvldm;
extern "C" int mvldm_groupnorm_bwd(const void* x0, const void* x1, const void* dy, void* dx0, void* dx1, const float* gamma, const float* beta,
                                   const float* stats, float* dgamma, float* dbeta, int n_img, int hw, int c0, int c1, int groups, int silu,
                                   int dtype, float* workspace, size_t workspace_bytes, mvldm_stream_t stream) {
    return groupnorm_bwd_run(x0, x1, dy, dx0, dx1, gamma, beta, stats, dgamma, dbeta, n_img, hw, c0, c1, groups, silu, dtype, workspace,
                             workspace_bytes, (hipStream_t)stream);
}
extern "C" int mvldm_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, float* dgamma, float* dbeta, int rows, int c,
                                   float eps, int dtype, float* workspace, size_t workspace_bytes, mvldm_stream_t stream) {
    return layernorm_bwd_run(x, dy, dx, gamma, dgamma, dbeta, rows, c, eps, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}
