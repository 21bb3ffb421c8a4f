/*
 * mvldm.h -- C ABI of libmvldm_hip.so: the MI355X (gfx950) kernels behind the multi-view
 * latent-diffusion denoising path of mohammadasim98/mv-ldm.
 *
 * The reference has no FFI of its own (it is pure Python on top of torch + diffusers); the
 * "interface each entry point replaces" is therefore the Python call it stands in for, cited per
 * function as /root/reference path:line (or the diffusers==0.27.2 class the reference instantiates).
 * Plain pointers and sizes only: no torch types.  All pointers are DEVICE pointers unless marked
 * "host".  All entry points return 0 on success or a negative MVLDM_ERR_*; text via
 * mvldm_last_error().  Kernels are enqueued on the given hipStream_t and never synchronise,
 * allocate or free (graph-capture safe); scratch comes from caller-provided workspaces.
 *
 * Data layout: activations are NHWC ("token-major": [image][pixel][channel]) in the activation
 * dtype (bf16 / f16 / f32); weights are pre-packed [n_pad][k_pad] K-major in the activation dtype by
 * mvldm_pack_weight(); biases, norm affine parameters, statistics and the DDIM state are fp32.
 */
#ifndef MVLDM_H
#define MVLDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVLDM_ABI_VERSION 2

typedef void* mvldm_stream_t; /* hipStream_t */

enum { MVLDM_F32 = 0, MVLDM_BF16 = 1, MVLDM_F16 = 2 };
enum { MVLDM_OK = 0, MVLDM_ERR_ARG = -1, MVLDM_ERR_HIP = -2, MVLDM_ERR_UNSUPPORTED = -3 };

/* epilogue selector of the implicit GEMM */
enum { MVLDM_EPI_NONE = 0, MVLDM_EPI_SILU = 1, MVLDM_EPI_GEGLU = 2 };
/* elementwise op selector */
enum { MVLDM_ELT_COPY = 0, MVLDM_ELT_SILU = 1 };

int mvldm_abi_version(void);
const char* mvldm_last_error(void);
/* cu_count / hbm_bytes of the current device; arch receives e.g. "gfx950" */
int mvldm_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Implicit GEMM: 3x3 / 1x1 convolution and Linear, one kernel family.
 *   replaces  torch.nn.Conv2d / torch.nn.Linear as called from diffusers ResnetBlock2D
 *             (conv1/conv2/conv_shortcut/time_emb_proj; mvunet.py:121,150,159,177), Downsample2D /
 *             Upsample2D (mvunet.py:145-148,198-200), conv_in/conv_out (mvunet.py:113,205),
 *             Transformer2DModel proj_in/proj_out + Attention to_q/k/v/to_out + GEGLU FF
 *             (mvunet.py:131-134,158), and SpatialTransformer3D proj_in/proj_out, CrossAttention
 *             projections, FeedForward (src/model/denoiser/mvdream/attention.py:60-87,174-205,416-439).
 *   out[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] + row_bias[img(m)][n] ) * out_scale + residual[m][n]
 *   A is gathered on the fly from up to two NHWC sources concatenated along C (the UNet skip concat,
 *   mvunet.py:176, never materialised); k = (tap, channel); optional nearest x2 upsampling of the
 *   input (Upsample2D) and stride 2 (Downsample2D; pad=0 gives the VAE's asymmetric (0,1,0,1) pad).
 *   GEGLU: W rows are packed in alternating 32-row blocks [value | gate]; out has n_out/2 columns.
 */
typedef struct mvldm_igemm_desc {
    const void* src0;      /* [n_img][h_in][w_in][c0] */
    const void* src1;      /* [n_img][h_in][w_in][c1] or NULL */
    const void* weight;    /* packed [n_pad][k_pad], k_pad = roundup(ksize^2*(c0+c1), 128 bytes) */
    const float* bias;     /* [n_out] or NULL */
    const float* row_bias; /* [n_img][row_bias_ld] (time-embedding projection) or NULL */
    const void* residual;  /* [m][n_dst] activation dtype or NULL */
    void* dst;             /* [m][n_dst], n_dst = n_out (GEGLU: n_out/2) */
    float* workspace;      /* split-K scratch, >= splitk*m*n_pad floats when splitk > 1 */
    int32_t c0, c1;
    int32_t n_img, h_in, w_in, h_out, w_out;
    int32_t ksize, stride, pad, upsample;   /* ksize 1 | 3 (2 with upsample >= 2).  upsample: 0 none; 1 nearest-2x source
                                               indexing in front of a 3x3 conv (diffusers Upsample2D); 2 + 2*py + px = sub-pixel
                                               phase (py, px) of the same conv decomposed into four 2x2 convs on the LOW-resolution
                                               image (h_out = h_in, w_out = w_in; rows are scattered to (2i+py, 2j+px) of dst) */
    int32_t n_out, n_pad, k_pad;
    int32_t row_bias_ld;
    int32_t epilogue;   /* MVLDM_EPI_* */
    int32_t act_dtype;  /* dtype of src/weight/residual */
    int32_t dst_dtype;  /* act_dtype or MVLDM_F32 */
    int32_t splitk;     /* >= 1; 0 = let the library choose (needs workspace) */
    int32_t tile;       /* 0 = auto; else force a tile config (tests / tuning) */
    int32_t k_order;    /* K order of the packed weight: 0 = (tap, channel); 1 = (64-channel block, tap, channel) */
    int32_t dst_ld;     /* row stride of dst in elements; 0 = n_dst (dense).  > n_dst writes into a wider buffer */
    float out_scale;
    size_t workspace_bytes;
} mvldm_igemm_desc;
int mvldm_igemm_fwd(const mvldm_igemm_desc* d, mvldm_stream_t stream);
/* bytes of split-K workspace the auto heuristic may use for this problem */
size_t mvldm_igemm_workspace_bytes(const mvldm_igemm_desc* d);

/* Pack a PyTorch-layout fp32 weight ([n_out][c_in][k][k] conv or [n_out][c_in] linear) into the
 * kernel layout: dst[n'][ (ky*k+kx)*c_pad + c ], zero padded to [n_pad][k_pad]; `geglu` != 0
 * interleaves rows n and n + n_out/2 in blocks of 32; `k_order` 1 stores k as (channel block of 64 [32 for
 * f32], tap, channel in block) so that the 9 taps of one channel block are consecutive K-tiles (their
 * activation reads hit L2 instead of crossing the fabric 9 times); needs c_pad % 64 == 0.  replaces: nothing in the reference (weights
 * there stay in torch layout); this is the one-time load-time transform. */
int mvldm_pack_weight(const float* src, void* dst, int n_out, int c_in, int ksize, int c_pad, int n_pad,
                      int k_pad, int geglu, int k_order, int dst_dtype, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), NHWC.   replaces torch.nn.GroupNorm + SiLU in ResnetBlock2D
 * (norm1/norm2 + nonlinearity), conv_norm_out + conv_act (mvunet.py:203-204), Transformer2DModel.norm
 * and SpatialTransformer3D.norm (mvdream/attention.py:96-97,423).  Statistics in fp64.
 * The input may be the channel concatenation of two tensors x0 [.., c0] | x1 [.., c1] (x1 NULL, c1 0
 * otherwise): the up-path skip concat (mvunet.py:176) feeds norm1 directly and is never materialised;
 * groups are taken over the concatenated c0+c1 channels and y is [n_img][hw][c0+c1].
 * stats_ws: >= n_img * MVLDM_GN_MAX_CHUNKS * groups * 2 doubles.
 */
#define MVLDM_GN_MAX_CHUNKS 32
int mvldm_groupnorm_fwd(const void* x0, const void* x1, void* y, const float* gamma, const float* beta, int n_img,
                        int hw, int c0, int c1, int groups, float eps, int silu, int dtype, void* stats_ws,
                        mvldm_stream_t stream);

/* LayerNorm over the last dim of [rows][c].  replaces torch.nn.LayerNorm in BasicTransformerBlock
 * (diffusers) and BasicTransformerBlock3D norm1-3 (mvdream/attention.py:286-288,363-367). */
int mvldm_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, int rows, int c, float eps,
                        int dtype, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention: out = softmax(q k^T * scale) v per (segment, head), flash-style (scores never leave
 * the chip), fp32 online softmax, QK^T accumulated in fp32 (ATTN_PRECISION=fp32,
 * mvdream/attention.py:20,185-188).   replaces CrossAttention.forward core
 * (mvdream/attention.py:180-203: the 3-D attention over all views' tokens and the per-view
 * attention) and diffusers Attention's F.scaled_dot_product_attention.
 * q/k/v/out are row-major token matrices; head h occupies columns [h*head_dim, (h+1)*head_dim);
 * ld_* are row strides in elements, so q/k/v may alias one fused [tokens][3C] projection.
 * seg: device int32 [n_seg][4] = {q_row0, q_len, kv_row0, kv_len}.
 */
int mvldm_attention_fwd(const void* q, const void* k, const void* v, void* out, int ld_q, int ld_k, int ld_v,
                        int ld_o, int heads, int head_dim, const int32_t* seg, int n_seg, int max_q_len,
                        float scale, int dtype, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sinusoidal timestep projection.   replaces diffusers Timesteps(320, flip_sin_to_cos=True, shift 0)
 * (mvunet.py:107).  out[i] = [cos(t_i * f) | sin(t_i * f)] (flip) or [sin | cos]; freqs: fp32 [dim/2]
 * table built on the host exactly as the reference does (exp(-ln(10000) * j / half)).
 */
int mvldm_timestep_embed_fwd(const int64_t* timesteps, const float* freqs, void* out, int n, int dim,
                             int flip_sin_to_cos, int dst_dtype, mvldm_stream_t stream);

/* elementwise y = f(x) with dtype conversion; n elements.  SiLU on the time embedding
 * (ResnetBlock2D: time_emb_proj(nonlinearity(temb))). */
int mvldm_eltwise_fwd(const void* x, void* y, size_t n, int op, int src_dtype, int dst_dtype, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused classifier-free-guidance compose + DDIM update (eta = 0, epsilon prediction).
 *   replaces  DiffusionWrapper.step's tail (src/model/diffusion_wrapper.py:444,451-453) and
 *             diffusers DDIMScheduler.step.
 *   eps  = eps_u + cfg_scale * (eps_c - eps_u)            (use_cfg) | eps_c
 *   x0   = (x - sqrt(1-a_t) * eps) / sqrt(a_t);  x' = sqrt(a_prev) * x0 + sqrt(1-a_prev) * eps
 * evaluated in fp32 with separately rounded mul/add/div in exactly that order (bit-identical to
 * the torch CPU fp32 expression given the same eps).
 * eps: fp32 NHWC [n_img_total][hw][c]; x_t / x_next: fp32 NHWC [n_tgt][hw][c];
 * cond_img / uncond_img: device int32 [n_tgt] image index of target view t in the conditional /
 * unconditional pass (uncond_img NULL => no CFG).
 * coef: device fp32 [n_steps][4] = {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)}; step_ptr:
 * device int32 current step index (read only here; advanced by mvldm_ddim_advance) so that one
 * captured graph serves every step.
 * If unet_in != NULL the new latents are also scattered (activation dtype) into channels [0,c) of
 * the UNet input rows cond_img[t] and uncond_img[t] ([n_img_total][hw][unet_in_c]).
 * n_steps: rows of coef (the step index is clamped to [0, n_steps-1], so a replay past the end of the
 * schedule never reads out of bounds).  clip_range > 0: diffusers `clip_sample=True` --
 * x0 = clamp(x0, -clip_range, clip_range) before the update (`clip_sample_range`, default 1.0); 0 = off
 * (the released config, config/model/scheduler/ddim.yaml:9).
 */
int mvldm_ddim_cfg_step(const float* eps, const float* x_t, float* x_next, const int32_t* cond_img,
                        const int32_t* uncond_img, int n_tgt, int hw, int c, float cfg_scale, const float* coef,
                        const int32_t* step_ptr, void* unet_in, int unet_in_c, int unet_in_dtype, int n_steps,
                        float clip_range, mvldm_stream_t stream);
/* step_ptr += 1; timesteps[tgt_rows[i]] = t_table[min(step, n_steps-1)] for i < n_rows (the
 * per-image timestep vector the UNet reads: context views stay at 0, diffusion_wrapper.py:419-428) */
int mvldm_ddim_advance(int32_t* step_ptr, const int64_t* t_table, int n_steps, int64_t* timesteps,
                       const int32_t* tgt_rows, int n_rows, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layout plumbing at the boundary: NCHW fp32 <-> NHWC activation dtype with channel offset/padding.
 *   replaces the torch.concat input assembly of DiffusionWrapper.step / .sample
 *   (diffusion_wrapper.py:429-432,476-481), the `inputs * 2.0 - 1.0` / `(1 / 0.18215) * latents` /
 *   `(image / 2 + 0.5).clamp(0, 1)` arithmetic around the VAE (:281,293,298).
 * nchw_to_nhwc: dst[img_map ? img_map[i] : i][pix][dst_c_off + ch] = src[i][ch][pix] * scale + shift
 *   (img_map: device int32 [n_img] or NULL).
 */
int mvldm_nchw_to_nhwc(const float* src, void* dst, int n_img, int c, int hw, int dst_c, int dst_c_off,
                       int dst_dtype, float scale, float shift, const int32_t* img_map, mvldm_stream_t stream);
int mvldm_nhwc_to_nchw(const void* src, float* dst, int n_img, int c, int hw, int src_c, int src_c_off,
                       int src_dtype, float scale, float shift, int clamp01, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Camera ray grid of the latent image.   replaces DiffusionWrapper.ray_encode with the raw [origin | direction]
 * encoding (diffusion_wrapper.py:301-322, generate_image_rays :169-190; src/geometry/projection.py:74-138:
 * sample_image_grid, unproject, get_world_rays).  extrinsics: fp32 [n_cam][4][4] camera-to-world;
 * intrinsics: fp32 [n_cam][3][3] normalised.  Per latent pixel (i, j): xy = ((j+.5)/w, (i+.5)/h),
 * d = normalize(K^-1 [x y 1]), direction = R d, origin = translation.  Outputs (either may be NULL):
 * out_nchw fp32 [n_cam][6][h*w]; out_nhwc: channels [nhwc_c_off, +6) of an NHWC buffer [..][h*w][nhwc_c]
 * in nhwc_dtype, camera i -> image img_map[i] (NULL: i).
 */
int mvldm_ray_encode(const float* extrinsics, const float* intrinsics, int n_cam, int h, int w, float* out_nchw,
                     void* out_nhwc, int nhwc_c, int nhwc_c_off, int nhwc_dtype, const int32_t* img_map,
                     mvldm_stream_t stream);

/* AutoencoderKL.encode(x).latent_dist.sample() * scale (diffusion_wrapper.py:283; diffusers
 * DiagonalGaussianDistribution): moments fp32 NCHW [n][2c][hw] = [mean | logvar], noise / out fp32 [n][c][hw];
 * out = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale. */
int mvldm_posterior_sample(const float* moments, const float* noise, float* out, int n, int c, int hw, float scale,
                           mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Plans: a whole forward (UNet walk, VAE decoder, DDIM step) as a flat list of the ops above with
 * all pointers resolved -- built once by the host (mv_ldm_amd/plan.py), executed here without
 * touching Python, optionally captured into a hipGraph.   replaces MultiViewUNet.forward's module
 * walk (mvunet.py:90-208) and the per-step Python of DiffusionWrapper.step/sample.
 */
enum {
    MVLDM_OP_IGEMM = 1, MVLDM_OP_GROUPNORM, MVLDM_OP_LAYERNORM, MVLDM_OP_ATTENTION, MVLDM_OP_TIMESTEP_EMBED,
    MVLDM_OP_ELTWISE, MVLDM_OP_DDIM_STEP, MVLDM_OP_DDIM_ADVANCE, MVLDM_OP_NCHW_TO_NHWC, MVLDM_OP_NHWC_TO_NCHW,
    MVLDM_OP_MEMCPY, MVLDM_OP_RAY_ENCODE, MVLDM_OP_POSTERIOR_SAMPLE
};

typedef struct mvldm_op {
    int32_t kind;
    int32_t tag; /* caller-defined label (layer id) echoed by the profiler */
    union {
        mvldm_igemm_desc igemm;
        struct { const void* x; const void* x1; void* y; const float* gamma; const float* beta; void* stats_ws;
                 int32_t n_img, hw, c0, c1, groups, silu, dtype; float eps; } groupnorm;
        struct { const void* x; void* y; const float* gamma; const float* beta;
                 int32_t rows, c, dtype; float eps; } layernorm;
        struct { const void* q; const void* k; const void* v; void* out; const int32_t* seg;
                 int32_t ld_q, ld_k, ld_v, ld_o, heads, head_dim, n_seg, max_q_len, dtype; float scale; } attention;
        struct { const int64_t* timesteps; const float* freqs; void* out;
                 int32_t n, dim, flip, dst_dtype; } temb;
        struct { const void* x; void* y; size_t n; int32_t op, src_dtype, dst_dtype; } eltwise;
        struct { const float* eps; const float* x_t; float* x_next; const int32_t* cond_img; const int32_t* uncond_img;
                 const float* coef; const int32_t* step_ptr; void* unet_in;
                 int32_t n_tgt, hw, c, unet_in_c, unet_in_dtype; float cfg_scale; int32_t n_steps; float clip_range; } ddim;
        struct { int32_t* step_ptr; const int64_t* t_table; int64_t* timesteps; const int32_t* tgt_rows;
                 int32_t n_steps, n_rows; } advance;
        struct { const void* src; void* dst; int32_t n_img, c, hw, other_c, other_c_off, dtype, clamp01;
                 float scale, shift; const int32_t* img_map; } layout;
        struct { const float* extrinsics; const float* intrinsics; float* out_nchw; void* out_nhwc; const int32_t* img_map;
                 int32_t n_cam, h, w, nhwc_c, nhwc_c_off, nhwc_dtype; } rays;
        struct { const float* moments; const float* noise; float* out; int32_t n, c, hw; float scale; } posterior;
        struct { const void* src; void* dst; size_t bytes; } memcpy_;
    } u;
} mvldm_op;

typedef struct mvldm_plan mvldm_plan;
int mvldm_op_run(const mvldm_op* op, mvldm_stream_t stream);                   /* one op, eagerly */
int mvldm_plan_create(const mvldm_op* ops, int n_ops, mvldm_plan** out);      /* ops: host array, copied */
int mvldm_plan_num_ops(const mvldm_plan* p);
int mvldm_plan_run(mvldm_plan* p, mvldm_stream_t stream);                      /* eager launches */
int mvldm_plan_run_range(mvldm_plan* p, int first, int last, mvldm_stream_t stream);
int mvldm_plan_capture(mvldm_plan* p, mvldm_stream_t stream);                  /* record into a hipGraph */
int mvldm_plan_replay(mvldm_plan* p, mvldm_stream_t stream);                   /* launch the captured graph */
/* hipEvent-bracket every op on `stream` (eager), `iters` passes; per_op_ms: host float[n_ops] averages */
int mvldm_plan_profile(mvldm_plan* p, mvldm_stream_t stream, int iters, float* per_op_ms);
void mvldm_plan_destroy(mvldm_plan* p);

#ifdef __cplusplus
}
#endif
#endif /* MVLDM_H */
