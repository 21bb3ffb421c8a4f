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

/* 3 (round 3): + mvldm_gather_rows, mvldm_ddpm_cfg_step, mvldm_ema_update; plan ops MVLDM_OP_PAR_BEGIN / _NEXT / _END and
 * MVLDM_OP_GATHER_ROWS / MVLDM_OP_ATTN_MERGE; bits 8-9 of mvldm_wgrad_desc.accumulate select the weight-gradient kernel form.  Everything of version 2 is
 * unchanged (additive).
 * 4 (round 5): + mvldm_pack_skinny and tile 15 / k_order 2 of mvldm_igemm_fwd (the skinny-M weight-streaming GEMM); additive over 3.
 * 5 (round 6): + tile 19 of mvldm_igemm_fwd (register-staged Linear), tile 13's bits 13 / 14, mvldm_build_flags; additive over 4. */
#define MVLDM_ABI_VERSION 5

typedef void* mvldm_stream_t; /* hipStream_t */

enum { MVLDM_F32 = 0, MVLDM_BF16 = 1, MVLDM_F16 = 2 };
enum { MVLDM_OK = 0, MVLDM_ERR_ARG = -1, MVLDM_ERR_HIP = -2, MVLDM_ERR_UNSUPPORTED = -3 };

/* epilogue selector of the implicit GEMM */
enum { MVLDM_EPI_NONE = 0, MVLDM_EPI_SILU = 1, MVLDM_EPI_GEGLU = 2, MVLDM_EPI_GELU = 3 /* exact (erf) GELU: the ViT feed-forward of the
       "standard" multi-view block, src/model/transformer/feed_forward.py:31-40 */ };
/* elementwise op selector */
enum { MVLDM_ELT_COPY = 0, MVLDM_ELT_SILU = 1, MVLDM_ELT_GELU = 2 };

int mvldm_abi_version(void);
/* bit 0: the library was compiled with -DMVLDM_EXPERIMENTS (its kernels read the A/B environment knobs of tools/; the product build reads none) */
int mvldm_build_flags(void);
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
    int32_t tile;       /* 0 = auto; else force a tile config (tests / tuning / plan-time selection): 1..11 igemm.hip tiles,
                           12 = persistent Linear with the epilogue pipelined under the next tile (linear_pp.hip: 1x1, one or two
                           sources, K >= 320 a multiple of 64, 16-bit in and out; any other problem is an error, not a fallback),
                           13 = persistent wide Linear (linear_pw.hip), 14 = weight-stationary Linear for K = 320 (linear_ws.hip),
                           17 = 256 x 320 tile with the pixel halo resident in LDS (igemm.hip, round 5: one-source 3x3 / stride 1 / pad 1 convs
                                on maps up to 24 pixels wide; anything else runs as tile 7, same values),
                           18 = 192 x 128 tile with a 4-slot ring (igemm.hip; 1x1 / 3x3, no upsampling forms, no GEGLU: refused),
                           19 = register-staged persistent Linear (linear_rs.hip, round 6: 256 x 256 tiles, 4 waves of 128 x 128, the K-steps in
                                flight held in registers; 1x1, one or two sources, K >= 256 a multiple of 128, 16-bit in and out, bias /
                                residual / GEGLU; any other problem is an error, not a fallback),
                           15 = skinny-M weight-streaming GEMM (skinny.hip, round 5: launches of a few hundred rows -- one scene at the
                                8x8 / 4x4 levels, mvunet.py:150-200 -- whose cost is the weight stream): `weight` is the FRAGMENT-ORDER
                                pack of mvldm_pack_skinny (k_order must be 2), whole K per workgroup, no split-K slab and no reduce launch;
                                1x1 / 3x3 (stride 1 or 2) / 2x2 phase convs, one or two sources, channels in multiples of 64, every
                                epilogue; bits 8-13 = its configuration (0 = rule; a configuration that does not fit is an error).
                           Bits 0-5 = the tile id; bits 8-11 / 12 = tuning overrides (XCD grid, register-prefetch loop); tile 13 only:
                           bit 13 = L2 prefetch of the activation rows, bit 14 = issue a K-step's LDS-DMA pieces in one burst behind its
                           barrier (the form before round 6; the default spreads them over three sub-steps -- same values, A/B) */
    int32_t k_order;    /* K order of the packed weight: 0 = (tap, channel); 1 = (64-channel block, tap, channel); 2 = the k_order-1
                           sequence in MFMA-fragment order (mvldm_pack_skinny; tile 15 only) */
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
 * there stay in torch layout); this is the one-time load-time transform.
 * `transpose` != 0 packs the weight of the DATA-GRADIENT convolution instead (backward of Conv2d / Linear w.r.t. its input):
 * rows are `n_rows` INPUT channels starting at `c_off` (a skip-concat conv yields two gradients: two packs), K runs over
 * (tap, output channel) with the taps flipped: dst[r][(tap', n)] = src[n][c_off + r][taps-1-tap']; c_pad then pads n_out,
 * n_pad pads n_rows. */
int mvldm_pack_weight(const float* src, void* dst, int n_out, int c_in, int ksize, int c_pad, int n_pad,
                      int k_pad, int geglu, int k_order, int dst_dtype, int transpose, int c_off, int n_rows,
                      mvldm_stream_t stream);

/* Re-order a K-major 16-bit pack made with k_order 1 (`packed`: [n_pad][k_pad], n_pad % 64 == 0, k_pad % 64 == 0) into the
 * fragment order tile 15 of mvldm_igemm_fwd streams: 16-byte unit ((nt * (k_pad/32) + kg) * 64 + lane) of dst =
 * packed[row(nt, lane & 15)][32 kg + 8 (lane >> 4) .. + 7], i.e. one v_mfma_f32_16x16x32 A-operand fragment (16 output columns x 32 k)
 * is 1 KB of consecutive bytes and a 16-column tile is one contiguous stream over K.  row(nt, r) = 16 nt + r, except for a GEGLU
 * pack (`geglu` != 0: rows alternate [32 value | 32 gate]) where tiles (2q, 2q+1) are the value / gate rows of output columns
 * 16 q .. 16 q + 15.  Same byte count as `packed`.  replaces: nothing in the reference (a load-time transform like mvldm_pack_weight). */
int mvldm_pack_skinny(const void* packed, void* dst, int n_pad, int k_pad, int geglu, int dtype, mvldm_stream_t stream);
/* Host-side query (no GPU work): the tile-15 configuration the library's rule picks for the problem `d` describes (pointers may be the
 * K-major ones: only shapes, dtypes, alignment and the epilogue are looked at), or 0 when tile 15 cannot compute it (an image larger than
 * a workgroup's row tiles, f32, channels that are not multiples of 64 ...).  Plan builders ask before they commit an op to tile 15. */
int mvldm_igemm_skinny_config(const mvldm_igemm_desc* d);

/* The same transform for MANY weights in one launch (round 3): after an optimizer step the training path re-packs every
 * trained weight (forward + data-gradient packs, ~380 of them) -- as one launch per pack these are latency-bound (7 ms for
 * 7.4 GB).  A job is the argument list of mvldm_pack_weight(); mvldm_pack_job_prepare() validates it and fills `kind`
 * and `blocks` (host side, no GPU work).  The caller sets `block0` = exclusive prefix sum of `blocks` over the job list,
 * copies the list to the device and passes it with `total_blocks` = the sum.  Output bytes are identical to n_jobs calls
 * of mvldm_pack_weight().  replaces: nothing in the reference (torch keeps one weight layout; its optimizer step is the
 * last touch). */
typedef struct mvldm_pack_job {
    const float* src; void* dst;
    int n_out, c_in, ksize, c_pad, n_pad, k_pad, geglu, k_order, transpose, c_off, n_rows;
    int kind;        /* filled by mvldm_pack_job_prepare: which packer body */
    int blocks;      /* filled by mvldm_pack_job_prepare: workgroups of this job */
    int block0;      /* caller: first workgroup of this job inside the batched launch */
} mvldm_pack_job;
int mvldm_pack_job_prepare(mvldm_pack_job* job /* host */, int dst_dtype);
/* `block_job` (device, total_blocks int32, may be NULL): the job index of every workgroup -- one load instead of a binary
 * search over the job list per workgroup (a workgroup moves only 4-18 KB). */
int mvldm_pack_weight_batch(const mvldm_pack_job* jobs /* device */, int n_jobs, const int32_t* block_job, int total_blocks,
                            int dst_dtype, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), NHWC.   replaces torch.nn.GroupNorm + SiLU in ResnetBlock2D
 * (norm1/norm2 + nonlinearity), conv_norm_out + conv_act (mvunet.py:203-204), Transformer2DModel.norm
 * and SpatialTransformer3D.norm (mvdream/attention.py:96-97,423).  Statistics in fp64.
 * The input may be the channel concatenation of two tensors x0 [.., c0] | x1 [.., c1] (x1 NULL, c1 0
 * otherwise): the up-path skip concat (mvunet.py:176) feeds norm1 directly and is never materialised;
 * groups are taken over the concatenated c0+c1 channels and y is [n_img][hw][c0+c1].
 * stats_ws: >= n_img * MVLDM_GN_MAX_CHUNKS * groups * 2 doubles.
 * stats_out (optional): fp32 [n_img][groups][2] = (mean, rstd), saved for mvldm_groupnorm_bwd.
 */
#define MVLDM_GN_MAX_CHUNKS 32
int mvldm_groupnorm_fwd(const void* x0, const void* x1, void* y, const float* gamma, const float* beta, int n_img,
                        int hw, int c0, int c1, int groups, float eps, int silu, int dtype, void* stats_ws,
                        float* stats_out, mvldm_stream_t stream);
/* how many times that call moves the tensor over HBM: 2 (one launch, the (image, channel-span) slab stays in registers:
 * 1 read + 1 write) or 3 (statistics launch + apply launch: 2 reads + 1 write).  Host-side query (no GPU work) for the
 * byte accounting of plans and benches. */
int mvldm_groupnorm_passes(int n_img, int hw, int c, int groups, int dtype);

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
 * lse (optional): fp32 [heads][lse_ld], entry [h][q row] = log2-domain log-sum-exp of the scaled scores of query row
 * `q row` (P = exp2(s * scale * log2(e) - lse)): saved for mvldm_attention_bwd.
 */
int mvldm_attention_fwd(const void* q, const void* k, const void* v, void* out, int ld_q, int ld_k, int ld_v,
                        int ld_o, int heads, int head_dim, const int32_t* seg, int n_seg, int max_q_len,
                        float scale, int dtype, float* lse, int lse_ld, mvldm_stream_t stream);

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
/* SCHEDULER["ddpm"].step (src/model/scheduler/__init__.py:19-22; called at diffusion_wrapper.py:451 when the config names the DDPM
 * scheduler): diffusers' DDPMScheduler.step for epsilon prediction with variance_type "fixed_small", optionally behind the CFG compose
 * of diffusion_wrapper.py:444 (eps_u NULL => no CFG), on flat fp32 arrays of n elements:
 *   e = eps_u + cfg_scale (eps_c - eps_u);  x0 = (x_t - coef[0] e) / coef[1];  [x0 = clamp(x0, +-clip_range) if clip_range > 0];
 *   x_next = coef[2] x0 + coef[3] x_t + coef[4] noise
 * coef (device fp32 [5]) = {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev) b_t / (1-a_t), sqrt(alpha_t) (1-a_prev) / (1-a_t), sqrt(variance)};
 * the host passes coef[4] = 0 at t = 0, where diffusers adds no noise (noise may then be NULL).  Separately rounded fp32 operations in
 * diffusers' order: bit-identical to the torch CPU expression given the same inputs. */
int mvldm_ddpm_cfg_step(const float* eps_c, const float* eps_u, const float* x_t, const float* noise, float* x_next, size_t n,
                        float cfg_scale, const float* coef, float clip_range, mvldm_stream_t stream);
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
 * out_nchw fp32 [n_cam][C][h*w]; out_nhwc: channels [nhwc_c_off, +C) of an NHWC buffer [..][h*w][nhwc_c]
 * in nhwc_dtype, camera i -> image img_map[i] (NULL: i).
 * Encodings of the (origin, direction) pair (diffusion_wrapper.py:98-127,301-322):
 *   MVLDM_RAYS_RAW        [o | d]: 6 channels (the released config: use_ray_encoding = srt_ray_encoding = false)
 *   MVLDM_RAYS_POSITIONAL src/model/encodings/positional_encoding.py:8-36 on o (n_origin_octaves) then d (n_dir_octaves):
 *                         per coordinate, per octave f: sin(x * 2 pi 2^f), sin(x * 2 pi 2^f + pi/2); 6 * octaves channels each
 *                         (0 octaves: that part is passed through raw, `nn.Identity`)
 *   MVLDM_RAYS_SRT        src/model/srt/layers.py:11-58 (RayEncoder): [sin(o 2^f pi) | cos(o 2^f pi) | sin(d 2^f pi) | cos(...)]
 * plucker != 0: o is replaced by o x d first (diffusion_wrapper.py:309-310).  The channel count is mvldm_ray_channels().
 */
enum { MVLDM_RAYS_RAW = 0, MVLDM_RAYS_POSITIONAL = 1, MVLDM_RAYS_SRT = 2 };
int mvldm_ray_channels(int mode, int n_origin_octaves, int n_dir_octaves);
int mvldm_ray_encode(const float* extrinsics, const float* intrinsics, int n_cam, int h, int w, float* out_nchw,
                     void* out_nhwc, int nhwc_c, int nhwc_c_off, int nhwc_dtype, const int32_t* img_map, int mode,
                     int n_origin_octaves, int n_dir_octaves, int plucker, mvldm_stream_t stream);

/* AutoencoderKL.encode(x).latent_dist.sample() * scale (diffusion_wrapper.py:283; diffusers
 * DiagonalGaussianDistribution): moments fp32 NCHW [n][2c][hw] = [mean | logvar], noise / out fp32 [n][c][hw];
 * out = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale. */
int mvldm_posterior_sample(const float* moments, const float* noise, float* out, int n, int c, int hw, float scale,
                           mvldm_stream_t stream);

/* ================================================================================================
 * Training: the backward kernels of the same path.   replaces torch.autograd through
 * MultiViewUNet.forward inside DiffusionWrapper.training_step (src/model/diffusion_wrapper.py:324-411:
 * add_noise :370, denoiser.forward :401, F.mse_loss :405-411) and the optimizer step Lightning drives for it
 * (configure_optimizers :1111-1122: AdamW lr 2e-5 + LinearLR warm-up, config/experiment/baseline.yaml:62-73;
 * Trainer(gradient_clip_val=0.1, accumulate_grad_batches=2), src/main.py:119-136, config/main.yaml:82-84).
 * Data gradients of convolutions / Linears are mvldm_igemm_fwd on weights packed with `transpose` (above).
 * ================================================================================================ */

/* Weight gradient of a convolution / Linear: grad[n][c][ky][kx] (PyTorch fp32 layout; [n][c] for a Linear)
 *   = / += sum over output pixels m of dy[m][n] * A[m][(tap, c)], A = the forward pass's on-the-fly gather
 * (two sources concatenated along C, stride, zero padding, nearest-2x upsampling: same fields as mvldm_igemm_desc).
 * dy: [m][dy_ld] activation dtype, columns [0, n_out) used -- a column slice of a wider matrix (fused QKV / all
 * time_emb_proj at once) is addressed by offsetting the pointer.  c_in <= c0 + c1: padding channels of the gather
 * (conv_in: 11 -> 16) are dropped from grad.  workspace: fp32 split-K slabs, >= n_out * ksize^2 * (c0+c1) * 4 bytes
 * (more lets small layers split the pixel range over more workgroups); reduced in a fixed order: deterministic. */
typedef struct mvldm_wgrad_desc {
    const void* src0; const void* src1; const void* dy;
    float* grad;
    float* workspace; size_t workspace_bytes;
    int32_t c0, c1, c_in;
    int32_t n_img, h_in, w_in, h_out, w_out;
    int32_t ksize, stride, pad, upsample;      /* upsample: 0 | 1 (nearest-2x gather in front of a 3x3 conv) */
    int32_t n_out, dy_ld;
    int32_t act_dtype;
    int32_t accumulate;                        /* bit 0: 0 grad = ..., 1 grad += ... (accumulation over micro-batches);
                                                * bits 8-9: kernel form, 0 = the library's rule, 1 = register-staged [128 n x 64 c] tile,
                                                * 2 = wide LDS-DMA [320 n x 128 c] tile (refused where it does not apply);
                                                * bits 10-12: workgroup target of the pixel split, 0 = the library's, else 64 << code */
} mvldm_wgrad_desc;
int mvldm_igemm_wgrad(const mvldm_wgrad_desc* d, mvldm_stream_t stream);

/* Column sums of a [n_seg * rows_per_seg][ld] activation matrix, columns [0, n): bias gradients (per_seg = 0:
 * dst[n] (+)= sum over all rows) and the gradient of the per-image time-embedding row added by ResnetBlock2D
 * (per_seg = 1: dst[seg][n] (ld_dst) (+)= sum over the rows of image seg).  workspace: fp32, 16-byte aligned,
 * >= (1024 + n_seg) * roundup(n, 8) floats. */
int mvldm_colsum(const void* x, float* dst, float* workspace, size_t workspace_bytes, int n_seg, int rows_per_seg, int n,
                 int ld, int ld_dst, int per_seg, int accumulate, int dtype, mvldm_stream_t stream);

/* GroupNorm(+SiLU) backward (NHWC, optional two-source concat input as in the forward): dx0 / dx1 written,
 * dgamma / dbeta ACCUMULATED (+=) -- or WRITTEN (=) when `silu` (here) / `dtype` (mvldm_layernorm_bwd) carries
 * MVLDM_NORM_BWD_STORE: the first write of an accumulation window needs no zeroed gradient.  stats: the forward's stats_out.  workspace: fp32,
 * >= n_img * (MVLDM_GN_MAX_CHUNKS * (c0+c1) + groups) * 2 floats, 16-byte aligned. */
#define MVLDM_NORM_BWD_STORE 0x100
int mvldm_groupnorm_bwd(const void* x0, const void* x1, const void* dy, void* dx0, void* dx1, const float* gamma,
                        const float* beta, const float* stats, float* dgamma, float* dbeta, int n_img, int hw, int c0,
                        int c1, int groups, int silu, int dtype, float* workspace, size_t workspace_bytes,
                        mvldm_stream_t stream);
/* LayerNorm backward over [rows][c]: dx written, dgamma / dbeta accumulated.  workspace: fp32, >= 512 * c * 2 floats, 16-byte aligned. */
int mvldm_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, float* dgamma, float* dbeta,
                        int rows, int c, float eps, int dtype, float* workspace, size_t workspace_bytes,
                        mvldm_stream_t stream);

/* Flash-attention backward (probabilities recomputed from `lse`; see mvldm_attention_fwd for the layout conventions).
 * dq/dk/dv may be column slices of one [tokens][3C] gradient of a fused QKV projection.  delta: fp32 scratch
 * [heads][stat_ld] (written here: sum_d dout * out per query row); lse: [heads][stat_ld] from the forward.
 * total_q_rows: number of query rows covered by the segments (rows 0 .. total_q_rows-1 of q/out/dout). */
typedef struct mvldm_attn_bwd_desc {
    const void* q; const void* k; const void* v; const void* out; const void* dout;
    void* dq; void* dk; void* dv;
    const float* lse; float* delta;
    const int32_t* seg;
    int32_t ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
    int32_t heads, head_dim, n_seg, max_q_len, max_kv_len, total_q_rows, stat_ld, dtype;
    float scale;
} mvldm_attn_bwd_desc;
int mvldm_attention_bwd(const mvldm_attn_bwd_desc* d, mvldm_stream_t stream);

/* Elementwise training ops on [rows][d] activation matrices:
 *   MVLDM_TE_SILU_BWD   out = b * silu'(a)                      (a: pre-activation, dtype a_dtype; b: upstream gradient)
 *   MVLDM_TE_ADD        out += a                                 (gradient accumulation; b unused)
 *   MVLDM_TE_GEGLU_FWD  out[rows][d] = a[:, :d] * gelu(a[:, d:])            (a: [rows][2d]; diffusers GEGLU / mvdream attention.py:60-73)
 *   MVLDM_TE_GEGLU_BWD  out[rows][2d] = d/da of the above times b[rows][d]
 *   MVLDM_TE_GELU_BWD   out = b * gelu'(a)                      (a: pre-activation) */
enum { MVLDM_TE_SILU_BWD = 0, MVLDM_TE_ADD = 1, MVLDM_TE_GEGLU_FWD = 2, MVLDM_TE_GEGLU_BWD = 3, MVLDM_TE_GELU_BWD = 4 };
int mvldm_train_eltwise(int op, const void* a, const void* b, void* out, size_t rows, int d, int a_dtype, int dtype,
                        mvldm_stream_t stream);
/* backward of nearest-2x upsampling: dx[n][i][j][c] = sum of the 2x2 block of du [n][2h][2w][c] */
int mvldm_pool2x2_sum(const void* du, void* dx, int n_img, int h, int w, int c, int dtype, mvldm_stream_t stream);
/* backward of a stride-2 subsampling: out [n][2h][2w][c] = x [n][h][w][c] at the even positions, zero elsewhere */
int mvldm_zero_insert2x(const void* x, void* out, int n_img, int h, int w, int c, int dtype, mvldm_stream_t stream);

/* DDIMScheduler.add_noise (diffusion_wrapper.py:370) fused with the UNet-input assembly: channels [dst_c_off, +c) of NHWC
 * image img_map[i] = sqrt(a_t) * x0[i] + sqrt(1 - a_t) * noise[i]; x0 / noise fp32 NCHW [n][c][hw], coef fp32 [n][2]. */
int mvldm_add_noise(const float* x0, const float* noise, const float* coef, void* dst, int n, int c, int hw, int dst_c,
                    int dst_c_off, int dst_dtype, const int32_t* img_map, mvldm_stream_t stream);
/* F.mse_loss(pred[:, v_c:], noise, reduction="mean") (diffusion_wrapper.py:405-411) and its gradient: pred fp32 NHWC
 * [n_img][hw][c]; target t is image tgt_img[t]; noise fp32 NCHW [n_tgt][c][hw].  loss[0] (+)= loss_scale * mean;
 * dpred (optional, [n_img][hw][dpred_c], zero-filled by the caller) = grad_scale * 2 (pred - noise) / N at the target
 * images.  workspace: 256 doubles. */
int mvldm_mse_loss(const float* pred, const float* noise, const int32_t* tgt_img, int n_tgt, int hw, int c, float* loss,
                   int accumulate, float loss_scale, void* dpred, int dpred_c, int dpred_dtype, float grad_scale,
                   double* workspace, mvldm_stream_t stream);

/* torch.nn.utils.clip_grad_norm_ over a flat fp32 gradient buffer (Lightning gradient_clip_val, src/main.py:131):
 * norm_out[0] = total norm, norm_out[1] = min(1, max_norm / (total + 1e-6)) (max_norm <= 0: 1), norm_out[2] = this
 * buffer's sum of squares.  sumsq_in (optional, device): use this total sum of squares instead (a sharded optimizer
 * all-reduces the per-rank norm_out[2] first).  workspace: 1024 doubles. */
int mvldm_grad_norm(const float* g, size_t n, const float* sumsq_in, float max_norm, float* norm_out, double* workspace,
                    mvldm_stream_t stream);
/* torch.optim.AdamW step (decoupled weight decay, no amsgrad) on flat fp32 master parameters:
 * g' = g * grad_scale * clip[1] (clip optional: norm_out of mvldm_grad_norm); step >= 1 is the 1-based step count. */
int mvldm_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, const float* clip, mvldm_stream_t stream);

/* DiffusionWrapper.on_before_zero_grad -> self.ema.update_parameters(self.denoiser) (diffusion_wrapper.py:138-142,152-154):
 * torch.optim.swa_utils.AveragedModel with get_ema_multi_avg_fn(0.995), i.e. avg.lerp_(p, weight) with weight = 1 - decay, on the flat
 * fp32 parameter buffer (the first update is a plain copy, done by the caller).  avg += weight (p - avg), torch's lerp arithmetic. */
int mvldm_ema_update(float* avg, const float* p, size_t n, float weight, mvldm_stream_t stream);

/* dst row (dst_index ? dst_index[k] : k) = src row (src_index ? src_index[k] : k), k < n_rows; rows of row_bytes bytes (a multiple of
 * 16); src and dst may be the same buffer as long as no destination row is also a source row.  Used by the fused CFG forward: the
 * unconditional pass of DiffusionWrapper.step (diffusion_wrapper.py:437-441) feeds the target views the SAME latents, mask, rays and
 * timestep as the conditional pass, the context views never change during sampling, and every layer before the first multi-view
 * attention block works per image -- those layers run once per step on the target views (and once per sample on the context views)
 * and the feature maps are copied to the rows that need them (mv_ldm_amd/mvunet.py, `dup`). */
int mvldm_gather_rows(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int n_rows, size_t row_bytes,
                      mvldm_stream_t stream);

/* Merge two softmax-attention results of the SAME queries over DISJOINT key sets (the flash-attention combine): with the
 * log-sum-exp outputs of mvldm_attention_fwd (log2 domain, [heads][lse_ld]) lse = log2(2^lse_a + 2^lse_b) and
 * out = oa 2^(lse_a - lse) + ob 2^(lse_b - lse), per head.  Rows are addressed per image of `tokens` rows: image k of the merge reads
 * rows a_img[k] * tokens + t of oa / lse_a, b_img[k] * tokens + t of ob / lse_b and writes row out_img[k] * tokens + t of out (out may
 * be oa or ob when the image maps keep reads and writes of different k apart).  Used in the first multi-view block of the fused CFG
 * forward: the target views' queries and keys are identical in the conditional and the unconditional pass there, so their scores over
 * the target keys are computed once (the unconditional result) and the conditional result adds the context views' keys
 * (mvdream/attention.py:174-205 evaluated once for both passes of diffusion_wrapper.py:435-441). */
int mvldm_attention_merge(const void* oa, const float* lse_a, const void* ob, const float* lse_b, void* out, const int32_t* a_img,
                          const int32_t* b_img, const int32_t* out_img, int n_img, int tokens, int heads, int head_dim, int ld_a, int ld_b,
                          int ld_o, int lse_ld_a, int lse_ld_b, int dtype, mvldm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Plans: a whole forward (UNet walk, VAE decoder, DDIM step) as a flat list of the ops above with
 * all pointers resolved -- built once by the host (mv_ldm_amd/plan.py), executed here without
 * touching Python, optionally captured into a hipGraph.   replaces MultiViewUNet.forward's module
 * walk (mvunet.py:90-208) and the per-step Python of DiffusionWrapper.step/sample.
 */
enum {
    MVLDM_OP_IGEMM = 1, MVLDM_OP_GROUPNORM, MVLDM_OP_LAYERNORM, MVLDM_OP_ATTENTION, MVLDM_OP_TIMESTEP_EMBED,
    MVLDM_OP_ELTWISE, MVLDM_OP_DDIM_STEP, MVLDM_OP_DDIM_ADVANCE, MVLDM_OP_NCHW_TO_NHWC, MVLDM_OP_NHWC_TO_NCHW,
    MVLDM_OP_MEMCPY, MVLDM_OP_RAY_ENCODE, MVLDM_OP_POSTERIOR_SAMPLE,
    MVLDM_OP_WGRAD, MVLDM_OP_ATTENTION_BWD, MVLDM_OP_GROUPNORM_BWD, MVLDM_OP_LAYERNORM_BWD, MVLDM_OP_COLSUM,
    MVLDM_OP_TRAIN_ELTWISE, MVLDM_OP_POOL2X2, MVLDM_OP_ZERO_INSERT, MVLDM_OP_ADD_NOISE, MVLDM_OP_MSE_LOSS, MVLDM_OP_FILL_ZERO,
    /* markers (no payload): the ops between PAR_BEGIN and PAR_END form lanes separated by PAR_NEXT; the caller declares the lanes
     * mutually independent (disjoint outputs and workspaces).  mvldm_plan_run / _capture put lane 0 on the caller's stream and the
     * others on the plan's own side streams, forked and joined with events -- under capture these become parallel branches of the
     * hipGraph (the four sub-pixel phase convs of an upsampler, a resnet's 1x1 shortcut beside its main chain: what fills the chip at
     * one or two scenes).  mvldm_op_run and mvldm_plan_profile treat them as no-ops (serial order is always valid). */
    MVLDM_OP_PAR_BEGIN, MVLDM_OP_PAR_NEXT, MVLDM_OP_PAR_END,
    MVLDM_OP_GATHER_ROWS, MVLDM_OP_ATTN_MERGE
};

typedef struct mvldm_op {
    int32_t kind;
    int32_t tag; /* caller-defined label (layer id) echoed by the profiler */
    union {
        mvldm_igemm_desc igemm;
        struct { const void* x; const void* x1; void* y; const float* gamma; const float* beta; void* stats_ws;
                 int32_t n_img, hw, c0, c1, groups, silu, dtype; float eps; float* stats_out; } groupnorm;
        struct { const void* x; void* y; const float* gamma; const float* beta;
                 int32_t rows, c, dtype; float eps; } layernorm;
        struct { const void* q; const void* k; const void* v; void* out; const int32_t* seg;
                 int32_t ld_q, ld_k, ld_v, ld_o, heads, head_dim, n_seg, max_q_len, dtype; float scale;
                 float* lse; int32_t lse_ld; } attention;
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
                 int32_t n_cam, h, w, nhwc_c, nhwc_c_off, nhwc_dtype, mode, n_origin_octaves, n_dir_octaves, plucker; } rays;
        struct { const float* moments; const float* noise; float* out; int32_t n, c, hw; float scale; } posterior;
        mvldm_wgrad_desc wgrad;
        mvldm_attn_bwd_desc attention_bwd;
        struct { const void* x0; const void* x1; const void* dy; void* dx0; void* dx1; const float* gamma; const float* beta;
                 const float* stats; float* dgamma; float* dbeta; float* workspace; size_t workspace_bytes;
                 int32_t n_img, hw, c0, c1, groups, silu, dtype; } groupnorm_bwd;
        struct { const void* x; const void* dy; void* dx; const float* gamma; float* dgamma; float* dbeta; float* workspace;
                 size_t workspace_bytes; int32_t rows, c, dtype; float eps; } layernorm_bwd;
        struct { const void* x; float* dst; float* workspace; size_t workspace_bytes;
                 int32_t n_seg, rows_per_seg, n, ld, ld_dst, per_seg, accumulate, dtype; } colsum;
        struct { const void* a; const void* b; void* out; size_t rows; int32_t op, d, a_dtype, dtype; } train_eltwise;
        struct { const void* src; void* dst; int32_t n_img, h, w, c, dtype; } resample;      /* pool2x2 / zero_insert2x */
        struct { const float* x0; const float* noise; const float* coef; void* dst; const int32_t* img_map;
                 int32_t n, c, hw, dst_c, dst_c_off, dst_dtype; } add_noise;
        struct { const float* pred; const float* noise; const int32_t* tgt_img; float* loss; void* dpred; double* workspace;
                 int32_t n_tgt, hw, c, accumulate, dpred_c, dpred_dtype; float loss_scale, grad_scale; } mse;
        struct { void* dst; size_t bytes; } fill;
        struct { const void* src; void* dst; size_t bytes; } memcpy_;
        struct { const void* src; void* dst; const int32_t* src_index; const int32_t* dst_index; size_t row_bytes; int32_t n_rows; } gather;
        struct { const void* oa; const void* ob; void* out; const float* lse_a; const float* lse_b; const int32_t* a_img; const int32_t* b_img;
                 const int32_t* out_img; int32_t n_img, tokens, heads, head_dim, ld_a, ld_b, ld_o, lse_ld_a, lse_ld_b, dtype; } attn_merge;
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
