"""ctypes binding of libmvldm_hip.so (the C ABI declared in include/mvldm.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load, importing
anything that computes raises.  (`load(required=False)` exists only so that CPU-only tooling --
symbol-export tests, plan building on meta tensors -- can introspect without a GPU.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libmvldm_hip.so"

ABI_VERSION = 5
F32, BF16, F16 = 0, 1, 2
EPI_NONE, EPI_SILU, EPI_GEGLU, EPI_GELU = 0, 1, 2, 3
RAYS_RAW, RAYS_POSITIONAL, RAYS_SRT = 0, 1, 2
ELT_COPY, ELT_SILU, ELT_GELU = 0, 1, 2
NORM_BWD_STORE = 0x100          # include/mvldm.h MVLDM_NORM_BWD_STORE: norm backward writes (=) dgamma / dbeta instead of +=
GN_MAX_CHUNKS = 32

(OP_IGEMM, OP_GROUPNORM, OP_LAYERNORM, OP_ATTENTION, OP_TIMESTEP_EMBED, OP_ELTWISE, OP_DDIM_STEP, OP_DDIM_ADVANCE,
 OP_NCHW_TO_NHWC, OP_NHWC_TO_NCHW, OP_MEMCPY, OP_RAY_ENCODE, OP_POSTERIOR_SAMPLE,
 OP_WGRAD, OP_ATTENTION_BWD, OP_GROUPNORM_BWD, OP_LAYERNORM_BWD, OP_COLSUM, OP_TRAIN_ELTWISE, OP_POOL2X2, OP_ZERO_INSERT,
 OP_ADD_NOISE, OP_MSE_LOSS, OP_FILL_ZERO, OP_PAR_BEGIN, OP_PAR_NEXT, OP_PAR_END, OP_GATHER_ROWS, OP_ATTN_MERGE) = range(1, 30)
TE_SILU_BWD, TE_ADD, TE_GEGLU_FWD, TE_GEGLU_BWD, TE_GELU_BWD = 0, 1, 2, 3, 4

vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t


class IgemmDesc(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("weight", vp), ("bias", vp), ("row_bias", vp), ("residual", vp),
                ("dst", vp), ("workspace", vp),
                ("c0", i32), ("c1", i32),
                ("n_img", i32), ("h_in", i32), ("w_in", i32), ("h_out", i32), ("w_out", i32),
                ("ksize", i32), ("stride", i32), ("pad", i32), ("upsample", i32),
                ("n_out", i32), ("n_pad", i32), ("k_pad", i32),
                ("row_bias_ld", i32), ("epilogue", i32), ("act_dtype", i32), ("dst_dtype", i32),
                ("splitk", i32), ("tile", i32), ("k_order", i32), ("dst_ld", i32), ("out_scale", f32), ("workspace_bytes", sz)]


class _GroupNorm(C.Structure):
    _fields_ = [("x", vp), ("x1", vp), ("y", vp), ("gamma", vp), ("beta", vp), ("stats_ws", vp),
                ("n_img", i32), ("hw", i32), ("c0", i32), ("c1", i32), ("groups", i32), ("silu", i32), ("dtype", i32),
                ("eps", f32), ("stats_out", vp)]


class _LayerNorm(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("gamma", vp), ("beta", vp), ("rows", i32), ("c", i32), ("dtype", i32), ("eps", f32)]


class _Attention(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp), ("seg", vp),
                ("ld_q", i32), ("ld_k", i32), ("ld_v", i32), ("ld_o", i32), ("heads", i32), ("head_dim", i32),
                ("n_seg", i32), ("max_q_len", i32), ("dtype", i32), ("scale", f32), ("lse", vp), ("lse_ld", i32)]


class _Temb(C.Structure):
    _fields_ = [("timesteps", vp), ("freqs", vp), ("out", vp), ("n", i32), ("dim", i32), ("flip", i32), ("dst_dtype", i32)]


class _Eltwise(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("n", sz), ("op", i32), ("src_dtype", i32), ("dst_dtype", i32)]


class _Ddim(C.Structure):
    _fields_ = [("eps", vp), ("x_t", vp), ("x_next", vp), ("cond_img", vp), ("uncond_img", vp), ("coef", vp),
                ("step_ptr", vp), ("unet_in", vp),
                ("n_tgt", i32), ("hw", i32), ("c", i32), ("unet_in_c", i32), ("unet_in_dtype", i32), ("cfg_scale", f32),
                ("n_steps", i32), ("clip_range", f32)]


class _Advance(C.Structure):
    _fields_ = [("step_ptr", vp), ("t_table", vp), ("timesteps", vp), ("tgt_rows", vp), ("n_steps", i32), ("n_rows", i32)]


class _Layout(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("n_img", i32), ("c", i32), ("hw", i32), ("other_c", i32),
                ("other_c_off", i32), ("dtype", i32), ("clamp01", i32), ("scale", f32), ("shift", f32), ("img_map", vp)]


class _Rays(C.Structure):
    _fields_ = [("extrinsics", vp), ("intrinsics", vp), ("out_nchw", vp), ("out_nhwc", vp), ("img_map", vp),
                ("n_cam", i32), ("h", i32), ("w", i32), ("nhwc_c", i32), ("nhwc_c_off", i32), ("nhwc_dtype", i32),
                ("mode", i32), ("n_origin_octaves", i32), ("n_dir_octaves", i32), ("plucker", i32)]


class _Posterior(C.Structure):
    _fields_ = [("moments", vp), ("noise", vp), ("out", vp), ("n", i32), ("c", i32), ("hw", i32), ("scale", f32)]


class _Memcpy(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("bytes", sz)]


class _AttnMerge(C.Structure):
    _fields_ = [("oa", vp), ("ob", vp), ("out", vp), ("lse_a", vp), ("lse_b", vp), ("a_img", vp), ("b_img", vp), ("out_img", vp),
                ("n_img", i32), ("tokens", i32), ("heads", i32), ("head_dim", i32), ("ld_a", i32), ("ld_b", i32), ("ld_o", i32),
                ("lse_ld_a", i32), ("lse_ld_b", i32), ("dtype", i32)]


class _Gather(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("src_index", vp), ("dst_index", vp), ("row_bytes", sz), ("n_rows", i32)]


class PackJob(C.Structure):
    """mvldm_pack_job: the arguments of one mvldm_pack_weight() call as a row of the batched re-pack's job list"""
    _fields_ = [("src", vp), ("dst", vp), ("n_out", i32), ("c_in", i32), ("ksize", i32), ("c_pad", i32), ("n_pad", i32), ("k_pad", i32),
                ("geglu", i32), ("k_order", i32), ("transpose", i32), ("c_off", i32), ("n_rows", i32), ("kind", i32), ("blocks", i32),
                ("block0", i32)]


class WgradDesc(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("dy", vp), ("grad", vp), ("workspace", vp), ("workspace_bytes", sz),
                ("c0", i32), ("c1", i32), ("c_in", i32),
                ("n_img", i32), ("h_in", i32), ("w_in", i32), ("h_out", i32), ("w_out", i32),
                ("ksize", i32), ("stride", i32), ("pad", i32), ("upsample", i32),
                ("n_out", i32), ("dy_ld", i32), ("act_dtype", i32), ("accumulate", i32)]


class AttnBwdDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp), ("dout", vp), ("dq", vp), ("dk", vp), ("dv", vp),
                ("lse", vp), ("delta", vp), ("seg", vp),
                ("ld_q", i32), ("ld_k", i32), ("ld_v", i32), ("ld_o", i32), ("ld_do", i32), ("ld_dq", i32), ("ld_dk", i32), ("ld_dv", i32),
                ("heads", i32), ("head_dim", i32), ("n_seg", i32), ("max_q_len", i32), ("max_kv_len", i32), ("total_q_rows", i32),
                ("stat_ld", i32), ("dtype", i32), ("scale", f32)]


class _GroupNormBwd(C.Structure):
    _fields_ = [("x0", vp), ("x1", vp), ("dy", vp), ("dx0", vp), ("dx1", vp), ("gamma", vp), ("beta", vp), ("stats", vp),
                ("dgamma", vp), ("dbeta", vp), ("workspace", vp), ("workspace_bytes", sz),
                ("n_img", i32), ("hw", i32), ("c0", i32), ("c1", i32), ("groups", i32), ("silu", i32), ("dtype", i32)]


class _LayerNormBwd(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("dx", vp), ("gamma", vp), ("dgamma", vp), ("dbeta", vp), ("workspace", vp),
                ("workspace_bytes", sz), ("rows", i32), ("c", i32), ("dtype", i32), ("eps", f32)]


class _Colsum(C.Structure):
    _fields_ = [("x", vp), ("dst", vp), ("workspace", vp), ("workspace_bytes", sz),
                ("n_seg", i32), ("rows_per_seg", i32), ("n", i32), ("ld", i32), ("ld_dst", i32), ("per_seg", i32),
                ("accumulate", i32), ("dtype", i32)]


class _TrainEltwise(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("out", vp), ("rows", sz), ("op", i32), ("d", i32), ("a_dtype", i32), ("dtype", i32)]


class _Resample(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("n_img", i32), ("h", i32), ("w", i32), ("c", i32), ("dtype", i32)]


class _AddNoise(C.Structure):
    _fields_ = [("x0", vp), ("noise", vp), ("coef", vp), ("dst", vp), ("img_map", vp),
                ("n", i32), ("c", i32), ("hw", i32), ("dst_c", i32), ("dst_c_off", i32), ("dst_dtype", i32)]


class _Mse(C.Structure):
    _fields_ = [("pred", vp), ("noise", vp), ("tgt_img", vp), ("loss", vp), ("dpred", vp), ("workspace", vp),
                ("n_tgt", i32), ("hw", i32), ("c", i32), ("accumulate", i32), ("dpred_c", i32), ("dpred_dtype", i32),
                ("loss_scale", f32), ("grad_scale", f32)]


class _Fill(C.Structure):
    _fields_ = [("dst", vp), ("bytes", sz)]


class _OpUnion(C.Union):
    _fields_ = [("igemm", IgemmDesc), ("groupnorm", _GroupNorm), ("layernorm", _LayerNorm), ("attention", _Attention),
                ("temb", _Temb), ("eltwise", _Eltwise), ("ddim", _Ddim), ("advance", _Advance), ("layout", _Layout),
                ("memcpy_", _Memcpy), ("gather", _Gather), ("attn_merge", _AttnMerge), ("rays", _Rays), ("posterior", _Posterior),
                ("wgrad", WgradDesc), ("attention_bwd", AttnBwdDesc), ("groupnorm_bwd", _GroupNormBwd),
                ("layernorm_bwd", _LayerNormBwd), ("colsum", _Colsum), ("train_eltwise", _TrainEltwise), ("resample", _Resample),
                ("add_noise", _AddNoise), ("mse", _Mse), ("fill", _Fill)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("tag", i32), ("u", _OpUnion)]


# name -> (restype, argtypes); mirrors include/mvldm.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    "mvldm_abi_version": (C.c_int, []),
    "mvldm_build_flags": (C.c_int, []),
    "mvldm_last_error": (C.c_char_p, []),
    "mvldm_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(sz), C.c_char_p, C.c_int]),
    "mvldm_igemm_fwd": (C.c_int, [C.POINTER(IgemmDesc), vp]),
    "mvldm_igemm_workspace_bytes": (sz, [C.POINTER(IgemmDesc)]),
    "mvldm_pack_weight": (C.c_int, [vp, vp] + [C.c_int] * 12 + [vp]),
    "mvldm_pack_skinny": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mvldm_igemm_skinny_config": (C.c_int, [C.POINTER(IgemmDesc)]),
    "mvldm_pack_job_prepare": (C.c_int, [C.POINTER(PackJob), C.c_int]),
    "mvldm_pack_weight_batch": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "mvldm_groupnorm_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32, C.c_int, C.c_int, vp, vp, vp]),
    "mvldm_groupnorm_passes": (C.c_int, [C.c_int] * 5),
    "mvldm_layernorm_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, f32, C.c_int, vp]),
    "mvldm_attention_fwd": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 6 + [vp, C.c_int, C.c_int, f32, C.c_int, vp, C.c_int, vp]),
    "mvldm_igemm_wgrad": (C.c_int, [C.POINTER(WgradDesc), vp]),
    "mvldm_colsum": (C.c_int, [vp, vp, vp, sz] + [C.c_int] * 8 + [vp]),
    "mvldm_groupnorm_bwd": (C.c_int, [vp] * 10 + [C.c_int] * 7 + [vp, sz, vp]),
    "mvldm_layernorm_bwd": (C.c_int, [vp] * 6 + [C.c_int, C.c_int, f32, C.c_int, vp, sz, vp]),
    "mvldm_attention_bwd": (C.c_int, [C.POINTER(AttnBwdDesc), vp]),
    "mvldm_train_eltwise": (C.c_int, [C.c_int, vp, vp, vp, sz, C.c_int, C.c_int, C.c_int, vp]),
    "mvldm_pool2x2_sum": (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp]),
    "mvldm_zero_insert2x": (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp]),
    "mvldm_add_noise": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 6 + [vp, vp]),
    "mvldm_mse_loss": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, f32, vp, C.c_int, C.c_int, f32, vp, vp]),
    "mvldm_grad_norm": (C.c_int, [vp, sz, vp, f32, vp, vp, vp]),
    "mvldm_adamw_step": (C.c_int, [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, C.c_int, f32, vp, vp]),
    "mvldm_timestep_embed_fwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mvldm_eltwise_fwd": (C.c_int, [vp, vp, sz, C.c_int, C.c_int, C.c_int, vp]),
    "mvldm_ddim_cfg_step": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, C.c_int, C.c_int, C.c_int, f32, vp]),
    "mvldm_ddim_advance": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, vp]),
    "mvldm_gather_rows": (C.c_int, [vp, vp, vp, vp, C.c_int, sz, vp]),
    "mvldm_attention_merge": (C.c_int, [vp] * 8 + [C.c_int] * 10 + [vp]),
    "mvldm_ddpm_cfg_step": (C.c_int, [vp, vp, vp, vp, vp, sz, f32, vp, f32, vp]),
    "mvldm_ema_update": (C.c_int, [vp, vp, sz, f32, vp]),
    "mvldm_nchw_to_nhwc": (C.c_int, [vp, vp] + [C.c_int] * 6 + [f32, f32, vp, vp]),
    "mvldm_ray_channels": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mvldm_ray_encode": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp] + [C.c_int] * 4 + [vp]),
    "mvldm_posterior_sample": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, f32, vp]),
    "mvldm_nhwc_to_nchw": (C.c_int, [vp, vp] + [C.c_int] * 6 + [f32, f32, C.c_int, vp]),
    "mvldm_op_run": (C.c_int, [C.POINTER(Op), vp]),
    "mvldm_plan_create": (C.c_int, [C.POINTER(Op), C.c_int, C.POINTER(vp)]),
    "mvldm_plan_num_ops": (C.c_int, [vp]),
    "mvldm_plan_run": (C.c_int, [vp, vp]),
    "mvldm_plan_run_range": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "mvldm_plan_capture": (C.c_int, [vp, vp]),
    "mvldm_plan_replay": (C.c_int, [vp, vp]),
    "mvldm_plan_profile": (C.c_int, [vp, vp, C.c_int, C.POINTER(f32)]),
    "mvldm_plan_destroy": (None, [vp]),
}

_lib = None


class MvldmError(RuntimeError):
    pass


_EXPERIMENT_KNOBS = ("MVLDM_ADAMW_NT", "MVLDM_ATTN_QB", "MVLDM_ATTN_WIDE_VALU", "MVLDM_GN_NTHR", "MVLDM_GN_SPAN", "MVLDM_GN_TWOPASS", "MVLDM_GN_WIDE",
                     "MVLDM_IGEMM_FAKE", "MVLDM_IGEMM_GROUP", "MVLDM_IGEMM_NOSTAGE", "MVLDM_IGEMM_PX", "MVLDM_IGEMM_SYNC", "MVLDM_IGEMM_TARGET",
                     "MVLDM_LPP_CPT", "MVLDM_LPP_FAKE", "MVLDM_PLAN_SERIAL", "MVLDM_PW_FAKE", "MVLDM_PW_GM", "MVLDM_PW_TN", "MVLDM_RS_FAKE", "MVLDM_SK_FAKE",
                     "MVLDM_STREAM_STORES", "MVLDM_WGRAD_DIRECT", "MVLDM_WGRAD_TARGET", "MVLDM_WGRAD_WIDE", "MVLDM_WGRAD_WIDE_BP",
                     "MVLDM_WGRAD_WIDE_TARGET", "MVLDM_WGRAD_XCD", "MVLDM_WS_FAKE")


def load(required: bool = True):
    """dlopen the in-tree shared library and attach prototypes.  No fallback: raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        if required:
            raise MvldmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU / PyTorch fallback for the HIP path)")
        return None
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.mvldm_abi_version() != ABI_VERSION:
        raise MvldmError("libmvldm_hip.so ABI version mismatch")
    # The product library reads NO environment variable (csrc/common.h knob_int): a tool that sets one of the kernel-level A/B knobs
    # against it would compare a path with itself and pass vacuously -- refuse instead.  Experiment builds report
    # bit 0 of mvldm_build_flags; the single-file experiment libraries of tools/*_probe.sh are recognised by their file name.
    if not (lib.mvldm_build_flags() & 1) and "_exp" not in LIB_PATH.name:
        import os
        stale = sorted(k for k in _EXPERIMENT_KNOBS if k in os.environ)
        if stale:
            raise MvldmError(f"{', '.join(stale)} set, but {LIB_PATH.name} was built without -DMVLDM_EXPERIMENTS and ignores it: build an "
                             "experiment library (MVLDM_EXPERIMENTS=1 python -m mv_ldm_amd._build --force, or tools/*_probe.sh + --lib) or unset it")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise MvldmError(f"mvldm error {rc}: {load().mvldm_last_error().decode()}")
