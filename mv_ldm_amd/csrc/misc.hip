// Small HBM-bound kernels: sinusoidal timestep projection, elementwise SiLU / dtype conversion,
// fused CFG + DDIM update, DDIM step bookkeeping, NCHW<->NHWC boundary plumbing (include/mvldm.h).
#include <algorithm>

#include "common.h"

namespace mvldm {

template <typename T>
__global__ __launch_bounds__(256) void temb_kernel(const int64_t* __restrict__ ts, const float* __restrict__ freqs,
                                                   T* __restrict__ out, int n, int dim, int flip) {
    const int half = dim / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * dim) return;
    const int i = idx / dim, j = idx - i * dim;
    const int fj = j < half ? j : j - half;
    const float arg = (float)ts[i] * freqs[fj];  // timesteps[:, None].float() * emb[None, :]
    const bool use_cos = flip ? (j < half) : (j >= half);
    out[idx] = from_f32<T>(use_cos ? cosf(arg) : sinf(arg));
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void eltwise_kernel(const TS* __restrict__ x, TD* __restrict__ y, size_t n, int op) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = to_f32<TS>(x[i]);
        if (op == MVLDM_ELT_SILU) v = silu_f(v);
        else if (op == MVLDM_ELT_GELU) v = gelu_erf_f(v);
        y[i] = from_f32<TD>(v);
    }
}

// fp32, separately rounded operations in the order of diffusers' DDIMScheduler.step (eta = 0):
// the mul/sub/div/add sequence must not be contracted into FMAs to stay bit-identical to torch.
template <typename TU>
__global__ __launch_bounds__(256) void ddim_kernel(const float* __restrict__ eps, const float* __restrict__ x_t,
                                                   float* __restrict__ x_next, const int32_t* __restrict__ cond_img,
                                                   const int32_t* __restrict__ uncond_img, int n_tgt, int hw, int c,
                                                   float cfg_scale, const float* __restrict__ coef,
                                                   const int32_t* __restrict__ step_ptr, TU* __restrict__ unet_in,
                                                   int unet_in_c, int n_steps, float clip_range) {
    // a replay past the end of the schedule re-applies the last step's coefficients instead of reading out of bounds
    const int step = min(max(*step_ptr, 0), n_steps - 1);
    const float sb = coef[step * 4 + 0], sa = coef[step * 4 + 1], sp = coef[step * 4 + 2], sd = coef[step * 4 + 3];
    const size_t per = (size_t)hw * c, total = (size_t)n_tgt * per;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int t = (int)(idx / per);
        const size_t rem = idx - (size_t)t * per;
        const int ci = cond_img[t];
        const float ec = eps[(size_t)ci * per + rem];
        float e = ec;
        int ui = -1;
        if (uncond_img) {
            ui = uncond_img[t];
            const float eu = eps[(size_t)ui * per + rem];
            e = __fadd_rn(eu, __fmul_rn(cfg_scale, __fsub_rn(ec, eu)));
        }
        const float x = x_t[idx];
        float x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(sb, e)), sa);
        if (clip_range > 0.f) x0 = fminf(fmaxf(x0, -clip_range), clip_range);   // diffusers `clip_sample` (x0 only; eps is kept)
        const float xn = __fadd_rn(__fmul_rn(sp, x0), __fmul_rn(sd, e));
        x_next[idx] = xn;
        if (unet_in) {
            const int pix = (int)(rem / c), ch = (int)(rem - (size_t)pix * c);
            unet_in[((size_t)ci * hw + pix) * unet_in_c + ch] = from_f32<TU>(xn);
            if (ui >= 0) unet_in[((size_t)ui * hw + pix) * unet_in_c + ch] = from_f32<TU>(xn);
        }
    }
}

// diffusers' DDPMScheduler.step (epsilon prediction, variance_type "fixed_small"), optionally behind the CFG compose, in the
// order of its fp32 operations:  x0 = (x - sqrt(1-a_t) e) / sqrt(a_t)  [clamp]  ;  prev = c0 x0 + c1 x  ;  out = prev + sigma z
// coef = {sqrt(1-a_t), sqrt(a_t), c0, c1, sigma}; sigma = 0 at t = 0 (no noise is added there; `noise` may then be null)
__global__ __launch_bounds__(256) void ddpm_kernel(const float* __restrict__ eps_c, const float* __restrict__ eps_u,
                                                   const float* __restrict__ x_t, const float* __restrict__ noise,
                                                   float* __restrict__ x_next, size_t n, float cfg_scale,
                                                   const float* __restrict__ coef, float clip_range) {
    const float sb = coef[0], sa = coef[1], c0 = coef[2], c1 = coef[3], sigma = coef[4];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float e = eps_c[i];
        if (eps_u) {
            const float eu = eps_u[i];
            e = __fadd_rn(eu, __fmul_rn(cfg_scale, __fsub_rn(e, eu)));
        }
        const float x = x_t[i];
        float x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(sb, e)), sa);
        if (clip_range > 0.f) x0 = fminf(fmaxf(x0, -clip_range), clip_range);
        float xn = __fadd_rn(__fmul_rn(c0, x0), __fmul_rn(c1, x));
        xn = __fadd_rn(xn, noise ? __fmul_rn(sigma, noise[i]) : 0.0f);
        x_next[i] = xn;
    }
}

// dst row d(k) = src row s(k) with optional index vectors (16-byte chunks; one row = one image's NHWC feature map)
__global__ __launch_bounds__(256) void gather_rows_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, const int32_t* __restrict__ src_index,
                                                          const int32_t* __restrict__ dst_index, int n_rows, size_t chunks_per_row) {
    const size_t total = (size_t)n_rows * chunks_per_row;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / chunks_per_row, c = i - row * chunks_per_row;
        const size_t sr = src_index ? (size_t)src_index[row] : row, dr = dst_index ? (size_t)dst_index[row] : row;
        dst[dr * chunks_per_row + c] = src[sr * chunks_per_row + c];
    }
}

__global__ void ddim_advance_kernel(int32_t* step_ptr, const int64_t* t_table, int n_steps, int64_t* timesteps,
                                    const int32_t* tgt_rows, int n_rows) {
    const int next = *step_ptr + 1;
    __syncthreads();
    const int64_t t = t_table[next < n_steps ? next : n_steps - 1];
    for (int i = threadIdx.x; i < n_rows; i += blockDim.x) timesteps[tgt_rows[i]] = t;
    if (threadIdx.x == 0) *step_ptr = next;
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int n_img,
                                                           int c, int hw, int dst_c, int dst_c_off, float scale, float shift,
                                                           const int32_t* __restrict__ img_map) {
    const size_t total = (size_t)n_img * hw * c;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        // idx enumerates the destination (pixel-major, channel fastest)
        const int ch = (int)(idx % c);
        const size_t pi = idx / c;
        const int pix = (int)(pi % hw), img = (int)(pi / hw);
        const int dimg = img_map ? img_map[img] : img;     // source image i lands in destination image img_map[i]
        dst[((size_t)dimg * hw + pix) * dst_c + dst_c_off + ch] = from_f32<T>(src[((size_t)img * c + ch) * hw + pix] * scale + shift);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int n_img,
                                                           int c, int hw, int src_c, int src_c_off, float scale,
                                                           float shift, int clamp01) {
    const size_t total = (size_t)n_img * hw * c;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        // idx enumerates the destination (channel-major, pixel fastest)
        const int pix = (int)(idx % hw);
        const size_t ic = idx / hw;
        const int ch = (int)(ic % c), img = (int)(ic / c);
        float v = to_f32<T>(src[((size_t)img * hw + pix) * src_c + src_c_off + ch]) * scale + shift;
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        dst[idx] = v;
    }
}


// Per-pixel camera rays of the latent grid (DiffusionWrapper.ray_encode, diffusion_wrapper.py:301-322 ->
// generate_image_rays :169-190 -> sample_image_grid / unproject / get_world_rays, projection.py:74-138):
//   xy = ((j + .5)/w, (i + .5)/h);  d = normalize(K^-1 [x, y, 1]);  d_world = R d;  origin = c2w translation.
// One thread per (camera, pixel); K^-1 by the adjugate (fp32).  Output channels: origin xyz, direction xyz.
// Either / both of: fp32 NCHW [n_cam][6][hw] (the reference's `ray_encodings` tensor) and a slice
// [c_off, c_off+6) of an NHWC activation buffer (the UNet input), image i -> row block img_map[i].
__host__ __device__ inline int ray_channels(int mode, int no, int nd) {
    if (mode == MVLDM_RAYS_POSITIONAL) return (no > 0 ? 6 * no : 3) + (nd > 0 ? 6 * nd : 3);
    if (mode == MVLDM_RAYS_SRT) return 6 * no + 6 * nd;
    return 6;
}

template <typename T>
__global__ __launch_bounds__(256) void ray_kernel(const float* __restrict__ extr, const float* __restrict__ intr, int n_cam,
                                                  int h, int w, float* __restrict__ out_nchw, T* __restrict__ out_nhwc,
                                                  int nhwc_c, int nhwc_c_off, const int32_t* __restrict__ img_map, int mode,
                                                  int no, int nd, int plucker) {
    const int hw = h * w;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_cam * hw) return;
    const int cam = idx / hw, pix = idx - cam * hw;
    const int i = pix / w, j = pix - i * w;
    const float* K = intr + cam * 9;
    const float* E = extr + cam * 16;
    const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], hh = K[7], k = K[8];
    const float A = e * k - f * hh, B = -(d * k - f * g), C = d * hh - e * g;
    const float det = a * A + b * B + c * C;
    const float id = 1.0f / det;
    const float inv[9] = {A * id, -(b * k - c * hh) * id, (b * f - c * e) * id,
                          B * id, (a * k - c * g) * id, -(a * f - c * d) * id,
                          C * id, -(a * hh - b * g) * id, (a * e - b * d) * id};
    const float x = ((float)j + 0.5f) / (float)w, y = ((float)i + 0.5f) / (float)h;
    float dx = inv[0] * x + inv[1] * y + inv[2];
    float dy = inv[3] * x + inv[4] * y + inv[5];
    float dz = inv[6] * x + inv[7] * y + inv[8];
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= n; dy /= n; dz /= n;
    float v[6];
    v[0] = E[3]; v[1] = E[7]; v[2] = E[11];
    v[3] = E[0] * dx + E[1] * dy + E[2] * dz;
    v[4] = E[4] * dx + E[5] * dy + E[6] * dz;
    v[5] = E[8] * dx + E[9] * dy + E[10] * dz;
    if (plucker) {      // origins = cross(origins, directions)   (diffusion_wrapper.py:309-310)
        const float ox = v[1] * v[5] - v[2] * v[4], oy = v[2] * v[3] - v[0] * v[5], oz = v[0] * v[4] - v[1] * v[3];
        v[0] = ox; v[1] = oy; v[2] = oz;
    }
    const int nch = ray_channels(mode, no, nd);
    const int dimg = img_map ? img_map[cam] : cam;
    T* onh = out_nhwc ? out_nhwc + ((size_t)dimg * hw + pix) * nhwc_c + nhwc_c_off : nullptr;
    auto put = [&](int ch, float val) {
        if (out_nchw) out_nchw[((size_t)cam * nch + ch) * hw + pix] = val;
        if (onh) onh[ch] = from_f32<T>(val);
    };
    if (mode == MVLDM_RAYS_RAW) {
#pragma unroll
        for (int q = 0; q < 6; ++q) put(q, v[q]);
    } else if (mode == MVLDM_RAYS_POSITIONAL) {
        // positional_encoding.py:30-33: sin(x * (2 pi 2^f) + phase), phase in {0, pi/2}; channel order (coordinate, octave, phase)
        int ch = 0;
        for (int part = 0; part < 2; ++part) {
            const int oct = part == 0 ? no : nd;
            for (int q = 0; q < 3; ++q) {
                const float val = v[part * 3 + q];
                if (oct == 0) { put(ch++, val); continue; }
                for (int fo = 0; fo < oct; ++fo) {
                    const float fr = __fmul_rn(6.283185307179586f, exp2f((float)fo));       // 2 * torch.pi * 2**octave in fp32
                    const float arg = __fmul_rn(val, fr);
                    put(ch++, sinf(arg));
                    put(ch++, sinf(__fadd_rn(arg, 1.5707963267948966f)));
                }
            }
        }
    } else {
        // srt/layers.py:17-33: multipliers 2^f pi; [sines (coordinate, octave) | cosines (coordinate, octave)] for o, then for d
        int base = 0;
        for (int part = 0; part < 2; ++part) {
            const int oct = part == 0 ? no : nd;
            for (int q = 0; q < 3; ++q)
                for (int fo = 0; fo < oct; ++fo) {
                    const float arg = __fmul_rn(v[part * 3 + q], __fmul_rn(exp2f((float)fo), 3.141592653589793f));
                    put(base + q * oct + fo, sinf(arg));
                    put(base + 3 * oct + q * oct + fo, cosf(arg));
                }
            base += 6 * oct;
        }
    }
}

// DiagonalGaussianDistribution.sample() of diffusers' AutoencoderKL.encode (diffusion_wrapper.py:283):
// moments fp32 NCHW [n][2c][hw] = [mean | logvar]; z = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale
__global__ __launch_bounds__(256) void posterior_sample_kernel(const float* __restrict__ moments, const float* __restrict__ noise,
                                                               float* __restrict__ out, int n, int c, int hw, float scale) {
    const size_t total = (size_t)n * c * hw;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t per = (size_t)c * hw;
        const int img = (int)(idx / per);
        const size_t rem = idx - (size_t)img * per;
        const float mean = moments[(size_t)img * 2 * per + rem];
        const float logvar = fminf(fmaxf(moments[(size_t)img * 2 * per + per + rem], -30.f), 20.f);
        const float stdv = expf(0.5f * logvar);
        out[idx] = __fmul_rn(__fadd_rn(mean, __fmul_rn(stdv, noise[idx])), scale);
    }
}

static inline int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 8192); }

int ray_run(const float* extr, const float* intr, int n_cam, int h, int w, float* out_nchw, void* out_nhwc, int nhwc_c,
            int nhwc_c_off, int nhwc_dtype, const int32_t* img_map, int mode, int no, int nd, int plucker, hipStream_t s) {
    MVLDM_REQUIRE(extr && intr && (out_nchw || out_nhwc), "ray_encode: null pointer");
    MVLDM_REQUIRE(mode >= MVLDM_RAYS_RAW && mode <= MVLDM_RAYS_SRT && no >= 0 && nd >= 0 && no <= 32 && nd <= 32, "ray_encode: mode %d octaves (%d, %d)", mode, no, nd);
    const int nch = ray_channels(mode, no, nd);
    MVLDM_REQUIRE(!out_nhwc || nhwc_c_off + nch <= nhwc_c, "ray_encode: channel slice [%d, %d) outside %d", nhwc_c_off, nhwc_c_off + nch, nhwc_c);
    const size_t total = (size_t)n_cam * h * w;
    if (total == 0) return MVLDM_OK;
    return dispatch_dtype(out_nhwc ? nhwc_dtype : MVLDM_F32, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(ray_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, extr, intr, n_cam, h, w, out_nchw,
                           reinterpret_cast<T*>(out_nhwc), nhwc_c, nhwc_c_off, img_map, mode, no, nd, plucker);
        return check_launch();
    });
}

int posterior_run(const float* moments, const float* noise, float* out, int n, int c, int hw, float scale, hipStream_t s) {
    MVLDM_REQUIRE(moments && noise && out, "posterior_sample: null pointer");
    const size_t total = (size_t)n * c * hw;
    if (total == 0) return MVLDM_OK;
    hipLaunchKernelGGL(posterior_sample_kernel, dim3(grid_for(total)), dim3(256), 0, s, moments, noise, out, n, c, hw, scale);
    return check_launch();
}

int temb_run(const int64_t* ts, const float* freqs, void* out, int n, int dim, int flip, int dst_dtype, hipStream_t s) {
    MVLDM_REQUIRE(ts && freqs && out && dim % 2 == 0, "timestep_embed: bad arguments");
    if (n == 0) return MVLDM_OK;
    return dispatch_dtype(dst_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(temb_kernel<T>, dim3((n * dim + 255) / 256), dim3(256), 0, s, ts, freqs, reinterpret_cast<T*>(out), n, dim, flip);
        return check_launch();
    });
}

int eltwise_run(const void* x, void* y, size_t n, int op, int src_dtype, int dst_dtype, hipStream_t s) {
    MVLDM_REQUIRE(x && y, "eltwise: null pointer");
    if (n == 0) return MVLDM_OK;
    return dispatch_dtype(src_dtype, [&](auto ts_) {
        using TS = decltype(ts_);
        return dispatch_dtype(dst_dtype, [&](auto td_) {
            using TD = decltype(td_);
            hipLaunchKernelGGL((eltwise_kernel<TS, TD>), dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const TS*>(x),
                               reinterpret_cast<TD*>(y), n, op);
            return check_launch();
        });
    });
}

int ddim_run(const float* eps, const float* x_t, float* x_next, const int32_t* cond_img, const int32_t* uncond_img,
             int n_tgt, int hw, int c, float cfg_scale, const float* coef, const int32_t* step_ptr, void* unet_in,
             int unet_in_c, int unet_in_dtype, int n_steps, float clip_range, hipStream_t s) {
    MVLDM_REQUIRE(eps && x_t && x_next && cond_img && coef && step_ptr && n_steps >= 1, "ddim: null pointer / n_steps");
    const size_t total = (size_t)n_tgt * hw * c;
    if (total == 0) return MVLDM_OK;
    return dispatch_dtype(unet_in_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(ddim_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, eps, x_t, x_next, cond_img, uncond_img,
                           n_tgt, hw, c, cfg_scale, coef, step_ptr, reinterpret_cast<T*>(unet_in), unet_in_c, n_steps, clip_range);
        return check_launch();
    });
}

int ddpm_run(const float* eps_c, const float* eps_u, const float* x_t, const float* noise, float* x_next, size_t n, float cfg_scale,
             const float* coef, float clip_range, hipStream_t s) {
    if (n == 0) return MVLDM_OK;
    MVLDM_REQUIRE(eps_c && x_t && x_next && coef, "ddpm_step: null pointer");
    hipLaunchKernelGGL(ddpm_kernel, dim3(grid_for(n)), dim3(256), 0, s, eps_c, eps_u, x_t, noise, x_next, n, cfg_scale, coef, clip_range);
    return check_launch();
}

int gather_rows_run(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int n_rows, size_t row_bytes, hipStream_t s) {
    if (n_rows == 0 || row_bytes == 0) return MVLDM_OK;
    MVLDM_REQUIRE(src && dst && row_bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "gather_rows: bad arguments");
    const size_t cpr = row_bytes / 16;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)n_rows * cpr)), dim3(256), 0, s, reinterpret_cast<const u32x4*>(src),
                       reinterpret_cast<u32x4*>(dst), src_index, dst_index, n_rows, cpr);
    return check_launch();
}

int advance_run(int32_t* step_ptr, const int64_t* t_table, int n_steps, int64_t* timesteps, const int32_t* tgt_rows,
                int n_rows, hipStream_t s) {
    MVLDM_REQUIRE(step_ptr && t_table && n_steps > 0, "ddim_advance: bad arguments");
    MVLDM_REQUIRE(n_rows == 0 || (timesteps && tgt_rows), "ddim_advance: null rows");
    hipLaunchKernelGGL(ddim_advance_kernel, dim3(1), dim3(64), 0, s, step_ptr, t_table, n_steps, timesteps, tgt_rows, n_rows);
    return check_launch();
}

int to_nhwc_run(const float* src, void* dst, int n_img, int c, int hw, int dst_c, int dst_c_off, int dst_dtype, float scale,
                float shift, const int32_t* img_map, hipStream_t s) {
    MVLDM_REQUIRE(src && dst && dst_c_off + c <= dst_c, "nchw_to_nhwc: bad arguments");
    const size_t total = (size_t)n_img * hw * c;
    if (total == 0) return MVLDM_OK;
    return dispatch_dtype(dst_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, src, reinterpret_cast<T*>(dst), n_img, c, hw, dst_c, dst_c_off, scale, shift, img_map);
        return check_launch();
    });
}

int to_nchw_run(const void* src, float* dst, int n_img, int c, int hw, int src_c, int src_c_off, int src_dtype, float scale,
                float shift, int clamp01, hipStream_t s) {
    MVLDM_REQUIRE(src && dst && src_c_off + c <= src_c, "nhwc_to_nchw: bad arguments");
    const size_t total = (size_t)n_img * hw * c;
    if (total == 0) return MVLDM_OK;
    return dispatch_dtype(src_dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<const T*>(src), dst, n_img, c, hw, src_c, src_c_off, scale, shift, clamp01);
        return check_launch();
    });
}

}  // namespace mvldm

using namespace mvldm;
extern "C" int mvldm_timestep_embed_fwd(const int64_t* timesteps, const float* freqs, void* out, int n, int dim,
                                        int flip_sin_to_cos, int dst_dtype, mvldm_stream_t stream) {
    return temb_run(timesteps, freqs, out, n, dim, flip_sin_to_cos, dst_dtype, (hipStream_t)stream);
}
extern "C" int mvldm_eltwise_fwd(const void* x, void* y, size_t n, int op, int src_dtype, int dst_dtype, mvldm_stream_t stream) {
    return eltwise_run(x, y, n, op, src_dtype, dst_dtype, (hipStream_t)stream);
}
extern "C" int mvldm_ddpm_cfg_step(const float* eps_c, const float* eps_u, const float* x_t, const float* noise, float* x_next, size_t n,
                                   float cfg_scale, const float* coef, float clip_range, mvldm_stream_t stream) {
    return ddpm_run(eps_c, eps_u, x_t, noise, x_next, n, cfg_scale, coef, clip_range, (hipStream_t)stream);
}
extern "C" int mvldm_gather_rows(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int n_rows, size_t row_bytes,
                                 mvldm_stream_t stream) {
    return gather_rows_run(src, dst, src_index, dst_index, n_rows, row_bytes, (hipStream_t)stream);
}
extern "C" int mvldm_ddim_cfg_step(const float* eps, const float* x_t, float* x_next, const int32_t* cond_img,
                                   const int32_t* uncond_img, int n_tgt, int hw, int c, float cfg_scale, const float* coef,
                                   const int32_t* step_ptr, void* unet_in, int unet_in_c, int unet_in_dtype,
                                   int n_steps, float clip_range, mvldm_stream_t stream) {
    return ddim_run(eps, x_t, x_next, cond_img, uncond_img, n_tgt, hw, c, cfg_scale, coef, step_ptr, unet_in, unet_in_c,
                    unet_in_dtype, n_steps, clip_range, (hipStream_t)stream);
}
extern "C" int mvldm_ddim_advance(int32_t* step_ptr, const int64_t* t_table, int n_steps, int64_t* timesteps,
                                  const int32_t* tgt_rows, int n_rows, mvldm_stream_t stream) {
    return advance_run(step_ptr, t_table, n_steps, timesteps, tgt_rows, n_rows, (hipStream_t)stream);
}
extern "C" int mvldm_nchw_to_nhwc(const float* src, void* dst, int n_img, int c, int hw, int dst_c, int dst_c_off,
                                  int dst_dtype, float scale, float shift, const int32_t* img_map, mvldm_stream_t stream) {
    return to_nhwc_run(src, dst, n_img, c, hw, dst_c, dst_c_off, dst_dtype, scale, shift, img_map, (hipStream_t)stream);
}
extern "C" int mvldm_ray_channels(int mode, int n_origin_octaves, int n_dir_octaves) { return ray_channels(mode, n_origin_octaves, n_dir_octaves); }
extern "C" int mvldm_ray_encode(const float* extrinsics, const float* intrinsics, int n_cam, int h, int w, float* out_nchw,
                                void* out_nhwc, int nhwc_c, int nhwc_c_off, int nhwc_dtype, const int32_t* img_map, int mode,
                                int n_origin_octaves, int n_dir_octaves, int plucker, mvldm_stream_t stream) {
    return ray_run(extrinsics, intrinsics, n_cam, h, w, out_nchw, out_nhwc, nhwc_c, nhwc_c_off, nhwc_dtype, img_map, mode, n_origin_octaves,
                   n_dir_octaves, plucker, (hipStream_t)stream);
}
extern "C" int mvldm_posterior_sample(const float* moments, const float* noise, float* out, int n, int c, int hw, float scale,
                                      mvldm_stream_t stream) {
    return posterior_run(moments, noise, out, n, c, hw, scale, (hipStream_t)stream);
}
extern "C" int mvldm_nhwc_to_nchw(const void* src, float* dst, int n_img, int c, int hw, int src_c, int src_c_off,
                                  int src_dtype, float scale, float shift, int clamp01, mvldm_stream_t stream) {
    return to_nchw_run(src, dst, n_img, c, hw, src_c, src_c_off, src_dtype, scale, shift, clamp01, (hipStream_t)stream);
}
