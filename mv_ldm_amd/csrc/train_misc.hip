// Training-step kernels that are neither GEMMs nor norms (include/mvldm.h, "Training"): column sums (bias /
// time-embedding-row gradients), SiLU / GEGLU forward+backward, gradient accumulation, 2x2 sum pooling (backward of the
// nearest-2x upsample), noise injection, the MSE loss and its gradient, global gradient norm + clip coefficient, fused AdamW.
// All HBM-bound streaming kernels: 16-byte accesses, one pass over their operands.
#include <algorithm>

#include "common.h"
#include "reduce.h"

namespace mvldm {

static inline int grid_for(size_t n, int per = 256) { return (int)std::min<size_t>((n + per - 1) / per, 8192); }

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x), Phi through the same A-S 7.1.26 erf as the forward (common.h)
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float phi = 0.5f * (1.0f + erf_as_f(x * 0.70710678118654752440f));
    return phi + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---- column sums ------------------------------------------------------------------------------------
// x [n_seg * rows_per_seg][ld] (activation dtype), columns [0, n): part[seg][chunk][n] = sum over the chunk's rows.
// A thread owns one 16-byte chunk column; the block sweeps RB rows at a time; deterministic (fixed order, no atomics to HBM).
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int rows_per_seg,
                                                             int n, int ld, int rows_per_chunk, int nchunk) {
    constexpr int EPC = Elt<T>::EPC;
    extern __shared__ float s_acc[];          // [RB][span*EPC]
    const int seg = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int ncc = n / EPC;
    const int r0 = chunk * rows_per_chunk, r1 = min(rows_per_seg, r0 + rows_per_chunk);
    const T* base = x + (size_t)seg * rows_per_seg * ld;
    float* out = part + ((size_t)seg * nchunk + chunk) * n;
    for (int cc0 = 0; cc0 < ncc; cc0 += 256) {
        const int span = min(ncc - cc0, 256);
        const int RB = max(1, 256 / span);
        const int cc = cc0 + threadIdx.x % span, rl = threadIdx.x / span;
        float a[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) a[i] = 0.f;
        if (rl < RB) {
            for (int r = r0 + rl; r < r1; r += RB) {
                const Chunk<T> v = load_chunk<T>(base + (size_t)r * ld + cc * EPC);
#pragma unroll
                for (int i = 0; i < EPC; ++i) a[i] += v.get(i);
            }
#pragma unroll
            for (int i = 0; i < EPC; ++i) s_acc[(rl * span + (cc - cc0)) * EPC + i] = a[i];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < span * EPC; e += 256) {
            float t = 0.f;
            for (int r = 0; r < RB; ++r) t += s_acc[r * span * EPC + e];
            out[cc0 * EPC + e] = t;
        }
        __syncthreads();
    }
}
// ---- elementwise forward/backward --------------------------------------------------------------------
enum { TE_SILU_BWD = 0, TE_ADD = 1, TE_GEGLU_FWD = 2, TE_GEGLU_BWD = 3, TE_GELU_BWD = 4 };

// dx = dy * silu'(x)
template <typename TX, typename T>
__global__ __launch_bounds__(256) void silu_bwd_kernel(const TX* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = to_f32<TX>(x[i]), s = sigmoid_f(v);
        dx[i] = from_f32<T>(to_f32<T>(dy[i]) * s * (1.0f + v * (1.0f - s)));
    }
}
// dx = dy * gelu'(x)   (exact GELU of the ViT feed-forward)
template <typename TX, typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const TX* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dx[i] = from_f32<T>(to_f32<T>(dy[i]) * gelu_grad_f(to_f32<TX>(x[i])));
}
// a += b   (gradient accumulation where two consumers feed one tensor)
template <typename T> __global__ __launch_bounds__(256) void add_kernel(T* __restrict__ a, const T* __restrict__ b, size_t nchunks) {
    constexpr int EPC = Elt<T>::EPC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
        Chunk<T> u = load_chunk<T>(a + i * EPC);
        const Chunk<T> v = load_chunk<T>(b + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) u.set(e, u.get(e) + v.get(e));
        store_chunk<T>(a + i * EPC, u);
    }
}
// GEGLU (diffusers GEGLU / mvdream attention.py:60-73): ag [rows][2D] = [value | gate];  h = value * gelu(gate)
template <typename T> __global__ __launch_bounds__(256) void geglu_fwd_kernel(const T* __restrict__ ag, T* __restrict__ h, size_t rows, int D) {
    constexpr int EPC = Elt<T>::EPC;
    const int dc = D / EPC;
    const size_t total = rows * dc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / dc;
        const int c = (int)(i - r * dc) * EPC;
        const Chunk<T> a = load_chunk<T>(ag + r * 2 * D + c), g = load_chunk<T>(ag + r * 2 * D + D + c);
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, a.get(e) * gelu_erf_f(g.get(e)));
        store_chunk<T>(h + r * D + c, o);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ ag, const T* __restrict__ dh, T* __restrict__ dag, size_t rows, int D) {
    constexpr int EPC = Elt<T>::EPC;
    const int dc = D / EPC;
    const size_t total = rows * dc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / dc;
        const int c = (int)(i - r * dc) * EPC;
        const Chunk<T> a = load_chunk<T>(ag + r * 2 * D + c), g = load_chunk<T>(ag + r * 2 * D + D + c), d = load_chunk<T>(dh + r * D + c);
        Chunk<T> da, dg;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            da.set(e, d.get(e) * gelu_erf_f(g.get(e)));
            dg.set(e, d.get(e) * a.get(e) * gelu_grad_f(g.get(e)));
        }
        store_chunk<T>(dag + r * 2 * D + c, da);
        store_chunk<T>(dag + r * 2 * D + D + c, dg);
    }
}
// backward of nearest-2x upsampling: dx[n][i][j][c] = sum of the 2x2 block of du[n][2i..][2j..][c]
template <typename T> __global__ __launch_bounds__(256) void pool2x2_kernel(const T* __restrict__ du, T* __restrict__ dx, int n_img, int h, int w, int c) {
    constexpr int EPC = Elt<T>::EPC;
    const int cc = c / EPC;
    const size_t total = (size_t)n_img * h * w * cc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i % cc) * EPC;
        const size_t p = i / cc;
        const int x = (int)(p % w), y = (int)((p / w) % h), img = (int)(p / ((size_t)w * h));
        const T* s = du + (((size_t)img * 2 * h + 2 * y) * (2 * w) + 2 * x) * c + ch;
        const Chunk<T> a = load_chunk<T>(s), b = load_chunk<T>(s + c), d = load_chunk<T>(s + (size_t)2 * w * c), e = load_chunk<T>(s + (size_t)2 * w * c + c);
        Chunk<T> o;
#pragma unroll
        for (int k = 0; k < EPC; ++k) o.set(k, (a.get(k) + b.get(k)) + (d.get(k) + e.get(k)));
        store_chunk<T>(dx + p * c + ch, o);
    }
}

// backward of a stride-2 subsampling: out[n][2i][2j][c] = x[n][i][j][c], zero at the odd positions
template <typename T> __global__ __launch_bounds__(256) void zero_insert_kernel(const T* __restrict__ x, T* __restrict__ out, int n_img, int h, int w, int c) {
    constexpr int EPC = Elt<T>::EPC;
    const int cc = c / EPC;
    const size_t total = (size_t)n_img * 2 * h * 2 * w * cc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i % cc) * EPC;
        const size_t pp = i / cc;
        const int X = (int)(pp % (2 * w)), Y = (int)((pp / (2 * w)) % (2 * h)), img = (int)(pp / ((size_t)4 * w * h));
        Chunk<T> v;
        if ((X | Y) & 1) v.zero();
        else v = load_chunk<T>(x + (((size_t)img * h + (Y >> 1)) * w + (X >> 1)) * c + ch);
        store_chunk<T>(out + pp * c + ch, v);
    }
}

// ---- noise injection, loss ---------------------------------------------------------------------------------
// DDIMScheduler.add_noise (diffusion_wrapper.py:370): noisy = sqrt(a_t) * x0 + sqrt(1 - a_t) * noise, per image coefficients
// coef [n][2]; x0 / noise fp32 NCHW [n][c][hw]; written into channels [c_off, c_off + c) of NHWC image img_map[i]
// (separately rounded fp32 mul/mul/add like the torch expression)
template <typename T>
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const float* __restrict__ coef,
                                                        T* __restrict__ dst, int n, int c, int hw, int dst_c, int dst_c_off, const int32_t* __restrict__ img_map) {
    const size_t total = (size_t)n * hw * c;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ch = (int)(idx % c);
        const size_t pi = idx / c;
        const int pix = (int)(pi % hw), img = (int)(pi / hw);
        const size_t s = ((size_t)img * c + ch) * hw + pix;
        const float v = __fadd_rn(__fmul_rn(coef[2 * img], x0[s]), __fmul_rn(coef[2 * img + 1], noise[s]));
        const int dimg = img_map ? img_map[img] : img;
        dst[((size_t)dimg * hw + pix) * dst_c + dst_c_off + ch] = from_f32<T>(v);
    }
}

// F.mse_loss(pred[:, v_c:], noise, reduction="mean") and its gradient (diffusion_wrapper.py:405-411):
// pred fp32 NHWC [n_img][hw][c]; target t: image tgt_img[t] of pred, noise fp32 NCHW [n_tgt][c][hw].
// partial[block] = sum (pred - noise)^2;  dpred [n_img][hw][dc] (activation dtype, zero-filled by the caller for the
// non-target images and the padding channels) = grad_scale * 2 (pred - noise) / N
template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ noise, const int32_t* __restrict__ tgt_img,
                                                  int n_tgt, int hw, int c, double* __restrict__ partial, T* __restrict__ dpred, int dc, float gscale) {
    __shared__ double s_red[256];
    const size_t per = (size_t)hw * c, total = (size_t)n_tgt * per;
    double acc = 0.0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int t = (int)(idx / per);
        const size_t rem = idx - (size_t)t * per;
        const int pix = (int)(rem / c), ch = (int)(rem - (size_t)pix * c);
        const int img = tgt_img[t];
        const float d = pred[((size_t)img * hw + pix) * c + ch] - noise[((size_t)t * c + ch) * hw + pix];
        acc += (double)d * (double)d;
        if (dpred) dpred[((size_t)img * hw + pix) * dc + ch] = from_f32<T>(gscale * d);
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}
// out[0] (+)= scale * sum(partial[0..n))
__global__ __launch_bounds__(256) void sum_finish_kernel(const double* __restrict__ partial, int n, double scale, float* __restrict__ out, int accumulate) {
    __shared__ double s_red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(scale * s_red[0]);
}

// ---- optimizer ----------------------------------------------------------------------------------------------
// sum of squares of a flat fp32 buffer -> partial[block] (fp64)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
    __shared__ double s_red[256];
    double acc = 0.0;
    const size_t n4 = n / 4, step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // four 16-byte loads in flight per lane (one per iteration left the kernel at 2.5 TB/s: rocprofv3, 100 us per 247 MB bucket); the
    // products are summed in the order of the one-load loop, so the result is bit-identical to it
    for (; i + 3 * step < n4; i += 4 * step) {
        const f32x4 v0 = reinterpret_cast<const f32x4*>(g)[i], v1 = reinterpret_cast<const f32x4*>(g)[i + step];
        const f32x4 v2 = reinterpret_cast<const f32x4*>(g)[i + 2 * step], v3 = reinterpret_cast<const f32x4*>(g)[i + 3 * step];
        acc += (double)v0[0] * v0[0] + (double)v0[1] * v0[1] + (double)v0[2] * v0[2] + (double)v0[3] * v0[3];
        acc += (double)v1[0] * v1[0] + (double)v1[1] * v1[1] + (double)v1[2] * v1[2] + (double)v1[3] * v1[3];
        acc += (double)v2[0] * v2[0] + (double)v2[1] * v2[1] + (double)v2[2] * v2[2] + (double)v2[3] * v2[3];
        acc += (double)v3[0] * v3[0] + (double)v3[1] * v3[1] + (double)v3[2] * v3[2] + (double)v3[3] * v3[3];
    }
    for (; i < n4; i += step) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        acc += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float v = g[n4 * 4 + threadIdx.x]; acc += (double)v * v; }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}
// torch.nn.utils.clip_grad_norm_: total = sqrt(extra_sumsq + sum partial) [extra: other ranks' shards, already all-reduced];
// norm_out[0] = total, norm_out[1] = clip coefficient min(1, max_norm / (total + 1e-6))  (max_norm <= 0: no clipping)
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int n, const float* __restrict__ sumsq_in, float max_norm,
                                                        float* __restrict__ norm_out) {
    __shared__ double s_red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tot2 = sumsq_in ? (double)sumsq_in[0] : s_red[0];
        const float total = (float)sqrt(tot2);
        norm_out[0] = total;
        norm_out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (total + 1e-6f)) : 1.0f;
        norm_out[2] = (float)s_red[0];      // this buffer's own sum of squares (what a sharded optimizer all-reduces)
    }
}
// torch.optim.AdamW (decoupled weight decay, no amsgrad), one launch over the flat fp32 master parameters:
//   g *= gscale * clip;  p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)          (bc = 1 - beta^step, computed on the host in fp64)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    size_t n, float lr, float b1, float b2, float eps, float wd, float step_size, float inv_sqrt_bc2,
                                                    float gscale, const float* __restrict__ clip) {
    const float gs = gscale * (clip ? clip[1] : 1.0f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
        p[i] = pi;
    }
}

// torch.optim.swa_utils.get_ema_multi_avg_fn(decay): avg.lerp_(p, 1 - decay) = avg + w (p - avg) for w < 0.5 (the form torch's lerp takes
// there; w >= 0.5: p - (p - avg)(1 - w), torch's other branch)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ avg, const float* __restrict__ p, size_t n, float w) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float a = avg[i], q = p[i], d = __fsub_rn(q, a);
        avg[i] = w < 0.5f ? fmaf(w, d, a) : fmaf(-d, __fsub_rn(1.0f, w), q);      // ATen's lerp: fmadd(weight, diff, start) in its vectorised form
    }
}

// 16-byte form of the same update: four parameters per thread and memory instruction (7 streams x 3.7 GB per step: the scalar
// kernel moved them at 4.2 TB/s)
template <bool NT>
__global__ __launch_bounds__(256) void adamw_kernel4(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v,
                                                     size_t n4, float lr, float b1, float b2, float eps, float wd, float step_size, float inv_sqrt_bc2,
                                                     float gscale, const float* __restrict__ clip) {
    const float gs = gscale * (clip ? clip[1] : 1.0f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        // streamed once per step, 26 GB in all: nothing here is worth a cache line (MVLDM_ADAMW_NT=0: plain loads / stores, A/B)
        const f32x4 g4 = NT ? __builtin_nontemporal_load(g + i) : g[i], m4 = NT ? __builtin_nontemporal_load(m + i) : m[i],
                    v4 = NT ? __builtin_nontemporal_load(v + i) : v[i];
        f32x4 p4 = NT ? __builtin_nontemporal_load(p + i) : p[i], mo, vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = g4[e] * gs;
            float pi = p4[e] * (1.0f - lr * wd);
            const float mi = b1 * m4[e] + (1.0f - b1) * gi;
            const float vi = b2 * v4[e] + (1.0f - b2) * gi * gi;
            mo[e] = mi;
            vo[e] = vi;
            pi -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
            p4[e] = pi;
        }
        if (NT) {
            __builtin_nontemporal_store(mo, m + i);
            __builtin_nontemporal_store(vo, v + i);
            __builtin_nontemporal_store(p4, p + i);
        } else {
            m[i] = mo;
            v[i] = vo;
            p[i] = p4;
        }
    }
}

// ---- host entry points ---------------------------------------------------------------------------------------
int colsum_run(const void* x, float* dst, float* ws, size_t ws_bytes, int n_seg, int rows_per_seg, int n, int ld, int ld_dst, int per_seg,
               int accumulate, int dtype, hipStream_t s) {
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    if (n_seg == 0 || rows_per_seg == 0 || n == 0) return MVLDM_OK;
    MVLDM_REQUIRE(x && dst && ws, "colsum: null pointer");
    // a column count that is not a whole number of 16-byte chunks (conv_out: 4) is summed over the padded width -- the
    // buffer must then hold that padding (ld >= roundup(n)) -- and only the first n sums are written
    const int n_valid = n;
    n = (n + epc - 1) / epc * epc;
    MVLDM_REQUIRE(ld % epc == 0 && ld >= n, "colsum: n=%d ld=%d (ld must be a multiple of %d and cover the padded width)", n_valid, ld, epc);
    // about 1024 first-stage workgroups in total (the second stage folds any number of partial rows in parallel: reduce.h)
    int nchunk = std::max(1, std::min(std::min(rows_per_seg / 8, 512), (1024 + n_seg - 1) / n_seg));
    const int rpc = (rows_per_seg + nchunk - 1) / nchunk;
    nchunk = (rows_per_seg + rpc - 1) / rpc;
    MVLDM_REQUIRE((size_t)n_seg * nchunk * n * sizeof(float) <= ws_bytes && ((uintptr_t)ws % 16) == 0,
                  "colsum: workspace of %zu bytes too small (need %zu) or not 16-byte aligned", ws_bytes, (size_t)n_seg * nchunk * n * sizeof(float));
    const int span = std::min(n / epc, 256), rb = std::max(1, 256 / span);
    const size_t smem = (size_t)rb * span * epc * sizeof(float);
    int rc = dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3(n_seg * nchunk), dim3(256), smem, s, reinterpret_cast<const T*>(x), ws, rows_per_seg, n, ld, rpc, nchunk);
        return check_launch();
    });
    if (rc) return rc;
    // dst[seg][n] (ld_dst) = / += sum over the segment's chunks (per_seg), or dst[n] = / += sum over all segments and chunks
    if (per_seg) hipLaunchKernelGGL(fold_partials_kernel<0>, dim3(n / 4, n_seg), dim3(256), 0, s, ws, nchunk, n, n_valid, dst, nullptr, ld_dst, accumulate);
    else hipLaunchKernelGGL(fold_partials_kernel<0>, dim3(n / 4, 1), dim3(256), 0, s, ws, n_seg * nchunk, n, n_valid, dst, nullptr, 0, accumulate);
    return check_launch();
}

int train_eltwise_run(int op, const void* a, const void* b, void* out, size_t rows, int d, int a_dtype, int dtype, hipStream_t s) {
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    if (rows == 0 || d == 0) return MVLDM_OK;
    MVLDM_REQUIRE(a && out, "train_eltwise: null pointer");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        const size_t n = rows * (size_t)d;
        switch (op) {
            case TE_SILU_BWD:
                MVLDM_REQUIRE(b, "silu_bwd: null dy");
                return dispatch_dtype(a_dtype, [&](auto tx) {
                    using TX = decltype(tx);
                    hipLaunchKernelGGL((silu_bwd_kernel<TX, T>), dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const TX*>(a),
                                       reinterpret_cast<const T*>(b), reinterpret_cast<T*>(out), n);
                    return check_launch();
                });
            case TE_GELU_BWD:
                MVLDM_REQUIRE(b, "gelu_bwd: null dy");
                return dispatch_dtype(a_dtype, [&](auto tx) {
                    using TX = decltype(tx);
                    hipLaunchKernelGGL((gelu_bwd_kernel<TX, T>), dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const TX*>(a),
                                       reinterpret_cast<const T*>(b), reinterpret_cast<T*>(out), n);
                    return check_launch();
                });
            case TE_ADD:
                MVLDM_REQUIRE(n % epc == 0, "add: element count must be a multiple of %d", epc);
                hipLaunchKernelGGL(add_kernel<T>, dim3(grid_for(n / epc)), dim3(256), 0, s, reinterpret_cast<T*>(out), reinterpret_cast<const T*>(a), n / epc);
                return check_launch();
            case TE_GEGLU_FWD:
                MVLDM_REQUIRE(d % epc == 0, "geglu: D=%d", d);
                hipLaunchKernelGGL(geglu_fwd_kernel<T>, dim3(grid_for(n / epc)), dim3(256), 0, s, reinterpret_cast<const T*>(a), reinterpret_cast<T*>(out), rows, d);
                return check_launch();
            case TE_GEGLU_BWD:
                MVLDM_REQUIRE(d % epc == 0 && b, "geglu_bwd: D=%d", d);
                hipLaunchKernelGGL(geglu_bwd_kernel<T>, dim3(grid_for(n / epc)), dim3(256), 0, s, reinterpret_cast<const T*>(a), reinterpret_cast<const T*>(b),
                                   reinterpret_cast<T*>(out), rows, d);
                return check_launch();
            default: return set_error(MVLDM_ERR_ARG, "train_eltwise: op %d", op);
        }
    });
}

int pool2x2_run(const void* du, void* dx, int n_img, int h, int w, int c, int dtype, hipStream_t s) {
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    if (n_img == 0 || h == 0 || w == 0) return MVLDM_OK;
    MVLDM_REQUIRE(du && dx && c % epc == 0, "pool2x2: bad arguments");
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(pool2x2_kernel<T>, dim3(grid_for((size_t)n_img * h * w * (c / epc))), dim3(256), 0, s, reinterpret_cast<const T*>(du),
                           reinterpret_cast<T*>(dx), n_img, h, w, c);
        return check_launch();
    });
}

int zero_insert_run(const void* x, void* out, int n_img, int h, int w, int c, int dtype, hipStream_t s) {
    const int epc = dtype == MVLDM_F32 ? 4 : 8;
    if (n_img == 0 || h == 0 || w == 0) return MVLDM_OK;
    MVLDM_REQUIRE(x && out && c % epc == 0, "zero_insert2x: bad arguments");
    return dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(zero_insert_kernel<T>, dim3(grid_for((size_t)n_img * 4 * h * w * (c / epc))), dim3(256), 0, s, reinterpret_cast<const T*>(x),
                           reinterpret_cast<T*>(out), n_img, h, w, c);
        return check_launch();
    });
}

int add_noise_run(const float* x0, const float* noise, const float* coef, void* dst, int n, int c, int hw, int dst_c, int dst_c_off, int dst_dtype,
                  const int32_t* img_map, hipStream_t s) {
    if (n == 0 || hw == 0) return MVLDM_OK;
    MVLDM_REQUIRE(x0 && noise && coef && dst && dst_c_off + c <= dst_c, "add_noise: bad arguments");
    return dispatch_dtype(dst_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(add_noise_kernel<T>, dim3(grid_for((size_t)n * c * hw)), dim3(256), 0, s, x0, noise, coef, reinterpret_cast<T*>(dst), n, c, hw,
                           dst_c, dst_c_off, img_map);
        return check_launch();
    });
}

constexpr int kLossBlocks = 256;
int mse_run(const float* pred, const float* noise, const int32_t* tgt_img, int n_tgt, int hw, int c, float* loss, int accumulate, float loss_scale,
            void* dpred, int dc, int dtype, float grad_scale, double* ws, hipStream_t s) {
    MVLDM_REQUIRE(pred && noise && tgt_img && loss && ws && n_tgt > 0 && hw > 0, "mse_loss: bad arguments");
    MVLDM_REQUIRE(!dpred || dc >= c, "mse_loss: dpred has %d channels, need %d", dc, c);
    const double N = (double)n_tgt * hw * c;
    int rc = dispatch_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(mse_kernel<T>, dim3(kLossBlocks), dim3(256), 0, s, pred, noise, tgt_img, n_tgt, hw, c, ws, reinterpret_cast<T*>(dpred), dc,
                           (float)(2.0 * grad_scale / N));
        return check_launch();
    });
    if (rc) return rc;
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, s, ws, kLossBlocks, (double)loss_scale / N, loss, accumulate);
    return check_launch();
}

constexpr int kNormBlocks = 1024;
int grad_norm_run(const float* g, size_t n, const float* sumsq_in, float max_norm, float* norm_out, double* ws, hipStream_t s) {
    MVLDM_REQUIRE(g && norm_out && ws, "grad_norm: null pointer");
    hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, s, g, n, ws);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, ws, kNormBlocks, sumsq_in, max_norm, norm_out);
    return check_launch();
}

int adamw_run(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
              float grad_scale, const float* clip, hipStream_t s) {
    if (n == 0) return MVLDM_OK;
    MVLDM_REQUIRE(p && g && m && v && step >= 1, "adamw: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const bool vec = n % 4 == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    static const bool nt = knob_int("MVLDM_ADAMW_NT", 1) != 0;
    if (vec && nt)
        hipLaunchKernelGGL(adamw_kernel4<true>, dim3(grid_for(n / 4, 2048)), dim3(256), 0, s, reinterpret_cast<f32x4*>(p), reinterpret_cast<const f32x4*>(g),
                           reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v), n / 4, lr, beta1, beta2, eps, weight_decay, (float)(lr / bc1),
                           (float)(1.0 / sqrt(bc2)), grad_scale, clip);
    else if (vec)
        hipLaunchKernelGGL(adamw_kernel4<false>, dim3(grid_for(n / 4, 2048)), dim3(256), 0, s, reinterpret_cast<f32x4*>(p), reinterpret_cast<const f32x4*>(g),
                           reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v), n / 4, lr, beta1, beta2, eps, weight_decay, (float)(lr / bc1),
                           (float)(1.0 / sqrt(bc2)), grad_scale, clip);
    else
        hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)(lr / bc1),
                           (float)(1.0 / sqrt(bc2)), grad_scale, clip);
    return check_launch();
}

int ema_run(float* avg, const float* p, size_t n, float weight, hipStream_t s) {
    if (n == 0) return MVLDM_OK;
    MVLDM_REQUIRE(avg && p && weight >= 0.f && weight <= 1.f, "ema_update: bad arguments");
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, avg, p, n, weight);
    return check_launch();
}

}  // namespace mvldm

using namespace mvldm;
extern "C" int mvldm_colsum(const void* x, float* dst, float* workspace, size_t workspace_bytes, int n_seg, int rows_per_seg, int n, int ld, int ld_dst,
                            int per_seg, int accumulate, int dtype, mvldm_stream_t stream) {
    return colsum_run(x, dst, workspace, workspace_bytes, n_seg, rows_per_seg, n, ld, ld_dst, per_seg, accumulate, dtype, (hipStream_t)stream);
}
extern "C" int mvldm_train_eltwise(int op, const void* a, const void* b, void* out, size_t rows, int d, int a_dtype, int dtype, mvldm_stream_t stream) {
    return train_eltwise_run(op, a, b, out, rows, d, a_dtype, dtype, (hipStream_t)stream);
}
extern "C" int mvldm_pool2x2_sum(const void* du, void* dx, int n_img, int h, int w, int c, int dtype, mvldm_stream_t stream) {
    return pool2x2_run(du, dx, n_img, h, w, c, dtype, (hipStream_t)stream);
}
extern "C" int mvldm_zero_insert2x(const void* x, void* out, int n_img, int h, int w, int c, int dtype, mvldm_stream_t stream) {
    return zero_insert_run(x, out, n_img, h, w, c, dtype, (hipStream_t)stream);
}
extern "C" int mvldm_add_noise(const float* x0, const float* noise, const float* coef, void* dst, int n, int c, int hw, int dst_c, int dst_c_off,
                               int dst_dtype, const int32_t* img_map, mvldm_stream_t stream) {
    return add_noise_run(x0, noise, coef, dst, n, c, hw, dst_c, dst_c_off, dst_dtype, img_map, (hipStream_t)stream);
}
extern "C" int mvldm_mse_loss(const float* pred, const float* noise, const int32_t* tgt_img, int n_tgt, int hw, int c, float* loss, int accumulate,
                              float loss_scale, void* dpred, int dpred_c, int dpred_dtype, float grad_scale, double* workspace, mvldm_stream_t stream) {
    return mse_run(pred, noise, tgt_img, n_tgt, hw, c, loss, accumulate, loss_scale, dpred, dpred_c, dpred_dtype, grad_scale, workspace, (hipStream_t)stream);
}
extern "C" int mvldm_grad_norm(const float* g, size_t n, const float* sumsq_in, float max_norm, float* norm_out, double* workspace, mvldm_stream_t stream) {
    return grad_norm_run(g, n, sumsq_in, max_norm, norm_out, workspace, (hipStream_t)stream);
}
extern "C" int mvldm_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int step, float grad_scale, const float* clip, mvldm_stream_t stream) {
    return adamw_run(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, clip, (hipStream_t)stream);
}
extern "C" int mvldm_ema_update(float* avg, const float* p, size_t n, float weight, mvldm_stream_t stream) {
    return ema_run(avg, p, n, weight, (hipStream_t)stream);
}
