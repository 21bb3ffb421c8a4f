"""STUDY (test infrastructure, CPU only; not collected by pytest): would an fp32 residual trunk bring bf16 inside the 1e-3 latent tolerance?

VERDICT round 5, weak #1 / next #5: the bf16 path drifts 5.9e-3 from the f32 path over 50 DDIM steps (north star: 1e-3; f16: 7.7e-4).  The
proposed lever: keep the residual trunk (and the skip tensors) in fp32 -- they are only ever read by GroupNorm / LayerNorm and by residual
epilogues -- while every MFMA operand stays bf16.  Building that (fp32 residual epilogues, fp32-reading norms, bf16 shadows wherever a conv
reads the trunk directly) is days of kernel plumbing, so this script first PRICES it: the CPU oracle (fp32 torch restatement of the same
network, `oracle/`) runs the configs[1] sampling loop with every storage rounding of the bf16 HIP path emulated (`x.bfloat16().float()` at
the points where the HIP path writes a 16-bit tensor: GroupNorm+SiLU / LayerNorm outputs, conv / Linear outputs, q / k / v, the attention
output, the GEGLU product, every residual sum; weights rounded once; accumulation stays fp32 like the MFMA's), under three policies:
    all_bf16        every stored activation rounded                      (what the library does today)
    trunk_fp32      residual sums stay fp32, the skip copies are rounded  ("trunk only")
    trunk_skip_fp32 residual sums and skips stay fp32                     ("trunk + skips")
and reports the relative L2 drift of the final latents against the unrounded fp32 run, same seeds.  MFMA operands that would read an fp32
trunk tensor directly (shortcut / down- / upsample convs, proj_in of the first step) are rounded at the operand in every policy.

    python tests/study_bf16_trunk.py [steps=50] [threads=8] [out=profiles/r06_bf16_trunk_ablation.json]

(Emulation, not the kernels: the attention probabilities and the in-kernel epilogue orders are not modelled; `all_bf16` is the control --
it has to land near the measured 5.9e-3 for the other two rows to mean anything.)"""
import copy
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import blocks as OB          # noqa: E402
from oracle import multiview as OMV      # noqa: E402
from oracle import pipeline as OP        # noqa: E402
from oracle.scheduler import DDIMScheduler as ODDIM   # noqa: E402

POLICY = {"trunk": False, "skip": False, "branch": False}      # which classes of stored tensors are rounded to bf16


def _r(t):
    return t.bfloat16().float()


def T(t):        # a residual-trunk tensor is stored
    return _r(t) if POLICY["trunk"] else t


def S(t):        # a skip tensor is kept for the up path
    return _r(t) if POLICY["skip"] else t


def Bq(t):       # a branch tensor is stored (always 16-bit in the library)
    return _r(t) if POLICY["branch"] else t


def MM(t):       # an MFMA operand read straight from a (possibly fp32) trunk tensor
    return _r(t) if POLICY["branch"] else t


# ---- the oracle's forwards with the storage points of the HIP path made explicit -------------------------------------------------------
def resnet_forward(self, x, temb=None):
    # gn_fused (GroupNorm + SiLU, one 16-bit store) -> conv1 (+ bias + time row, one store) -> gn_fused -> conv2 (+ bias + residual, one store)
    h = self.conv1(Bq(self.nonlinearity(self.norm1(x))))
    if self.time_emb_proj is not None:
        h = h + self.time_emb_proj(self.nonlinearity(temb.to(h.dtype)))[:, :, None, None]
    h = self.conv2(Bq(self.nonlinearity(self.norm2(Bq(h)))))
    if self.conv_shortcut is not None:
        x = T(self.conv_shortcut(MM(x)))          # the 1x1 shortcut's output is the residual operand of conv2's epilogue
    return T((x + h) / self.output_scale_factor)


def attention_forward(self, hidden_states, encoder_hidden_states=None, temb=None):
    # (the transformer-block form: no group_norm, no residual -- the caller adds it in the to_out epilogue)
    assert self.group_norm is None and not self.residual_connection and hidden_states.ndim == 3
    ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states.to(hidden_states.dtype)
    B, L, _ = hidden_states.shape
    q = Bq(self.to_q(hidden_states)).view(B, L, self.heads, self.dim_head).transpose(1, 2)
    k = Bq(self.to_k(ctx)).view(B, -1, self.heads, self.dim_head).transpose(1, 2)
    v = Bq(self.to_v(ctx)).view(B, -1, self.heads, self.dim_head).transpose(1, 2)
    o = Bq(OB.sdpa(q, k, v, self.scale)).transpose(1, 2).reshape(B, L, self.heads * self.dim_head)
    return self.to_out[1](self.to_out[0](o)) / self.rescale_output_factor


def geglu_forward(self, x):
    x, gate = self.proj(x).chunk(2, dim=-1)
    return Bq(x * F.gelu(gate))                   # the GEGLU epilogue stores the product


def basic_block_forward(self, x, encoder_hidden_states=None):
    x = T(self.attn1(Bq(self.norm1(x))) + x)
    x = T(self.attn2(Bq(self.norm2(x)), encoder_hidden_states) + x)
    x = T(self.ff(Bq(self.norm3(x))) + x)
    return x


def transformer2d_forward(self, hidden_states, encoder_hidden_states=None, **_unused):
    b, _, h, w = hidden_states.shape
    residual = hidden_states
    x = Bq(self.norm(hidden_states))
    if not self.use_linear_projection:
        x = self.proj_in(x)
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
    else:
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        x = self.proj_in(x)
    x = T(x)                                       # the token stream the block's residual sums continue
    for blk in self.transformer_blocks:
        x = blk(x, encoder_hidden_states)
    x = MM(x)
    if not self.use_linear_projection:
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        x = self.proj_out(x)
    else:
        x = self.proj_out(x)
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
    return SimpleNamespace(sample=T(x + residual))


def mv_attention_forward(self, x, context=None):
    ctx = x if context is None else context
    Bn, L, _ = x.shape
    split = lambda t: t.view(Bn, -1, self.heads, self.dim_head).transpose(1, 2)   # noqa: E731
    q, k, v = split(Bq(self.to_q(x))), split(Bq(self.to_k(ctx))), split(Bq(self.to_v(ctx)))
    o = Bq(OB.sdpa(q, k, v, self.scale)).transpose(1, 2).reshape(Bn, L, self.heads * self.dim_head)
    return self.to_out(o)


def mv_block_forward(self, x, num_frames: int):
    bf, l, c = x.shape
    x = x.reshape(bf // num_frames, num_frames * l, c)
    x = T(self.attn1(Bq(self.norm1(x))) + x)
    x = x.reshape(bf, l, c)
    x = T(self.attn2(Bq(self.norm2(x))) + x)
    x = T(self.ff(Bq(self.norm3(x))) + x)
    return x


def spatial3d_forward(self, x):
    b, v, c, h, w = x.shape
    x = x.reshape(b * v, c, h, w)
    x_in = x
    x = T(self.proj_in(Bq(self.norm(x))))
    x = x.permute(0, 2, 3, 1).reshape(b * v, h * w, c)
    for blk in self.transformer_blocks:
        x = blk(x, num_frames=v)
    x = MM(x).reshape(b * v, h, w, c).permute(0, 3, 1, 2)
    x = T(self.proj_out(x) + x_in)
    return x.reshape(b, v, c, h, w)


def mvunet_forward(self, latents, timestep, cond_state=None):
    b, views = latents.shape[:2]
    t = timestep.reshape(b, -1)
    t = (t.expand(b, views) if t.shape[1] == 1 else t).reshape(b * views)
    emb = self.unet.time_embedding(self.unet.time_proj(t))
    h = T(self.unet.conv_in(MM(latents.reshape(b * views, *latents.shape[2:]))))
    skips = [S(h)]
    for lvl, blk in enumerate(self.unet.down_blocks):
        for i, resnet in enumerate(blk.resnets):
            h = resnet(h, emb)
            if getattr(blk, "has_cross_attention", False):
                h = blk.attentions[i](h, encoder_hidden_states=self._zero_context(h, b * views)).sample
            skips.append(S(h))
        if h.shape[-2] <= 32 and h.shape[-1] <= 32 and self.cfg.encoder_conditioning:
            h = self._mv(self.cross_attn_blocks_encoder[lvl], h, views)
        if blk.downsamplers is not None:
            for d in blk.downsamplers:
                h = T(d(MM(h)))
            skips.append(S(h))
    mid = self.unet.mid_block
    h = mid.resnets[0](h, emb)
    for attn, resnet in zip(mid.attentions, mid.resnets[1:]):
        h = attn(h, encoder_hidden_states=self._zero_context(h, b * views)).sample
        h = resnet(h, emb)
    if self.cfg.mid_conditioning:
        h = self._mv(self.cross_attn_blocks_mid[0], h, views)
    for lvl, blk in enumerate(self.unet.up_blocks):
        for i, resnet in enumerate(blk.resnets):
            h = resnet(torch.cat([h, skips.pop()], dim=1), emb)
        if h.shape[-2] <= 32 and h.shape[-1] <= 32 and self.cfg.decoder_conditioning:
            h = self._mv(self.cross_attn_blocks_decoder[lvl], h, views)
        if blk.upsamplers is not None:
            for u in blk.upsamplers:
                h = T(u(MM(h)))
    h = self.unet.conv_out(Bq(self.unet.conv_act(self.unet.conv_norm_out(h))))      # eps leaves in fp32
    return h.reshape(b, views, *h.shape[1:])


def install():
    OB.ResnetBlock2D.forward = resnet_forward
    OB.Attention.forward = attention_forward
    OB.GEGLU.forward = geglu_forward
    OB.BasicTransformerBlock.forward = basic_block_forward
    OB.Transformer2DModel.forward = transformer2d_forward
    OMV.MVCrossAttention.forward = mv_attention_forward
    OMV.MVBlock3D.forward = mv_block_forward
    OMV.SpatialTransformer3D.forward = spatial3d_forward
    OMV.MultiViewUNet.forward = mvunet_forward


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r06_bf16_trunk_ablation.json")
    hl = int(os.environ.get("STUDY_LATENT", "32"))
    torch.set_grad_enabled(False)
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    g = torch.Generator().manual_seed(1234)
    den = OMV.MultiViewUNet(OMV.MVUNetCfg(pretrained_from="sd21"), 11, 4).eval()
    assert den.pretrained_from is not None          # (the patched walk assumes the released model's zero-context cross-attention)
    for blk in [*den.cross_attn_blocks_encoder, *den.cross_attn_blocks_mid, *den.cross_attn_blocks_decoder]:
        torch.nn.init.normal_(blk.proj_out.weight, std=0.02)
    den_r = copy.deepcopy(den)
    for prm in den_r.parameters():                  # 16-bit weight packs; biases / norm parameters stay fp32 in the library
        if prm.ndim >= 2:
            prm.copy_(_r(prm))
    install()
    v_c, v_t = 1, 4
    x_T = torch.randn(1, v_t, 4, hl, hl, generator=g)
    ctx_in = torch.cat([torch.randn(1, v_c, 4, hl, hl, generator=g), torch.zeros(1, v_c, 1, hl, hl)], dim=2)
    rays = torch.randn(1, v_c + v_t, 6, hl, hl, generator=g)
    mask = torch.ones(1, v_t, 1, hl, hl)

    def run(model, policy):
        POLICY.update(policy)
        sch = ODDIM(clip_sample=False)
        sch.set_timesteps(steps)
        x = x_T.clone()
        t0 = time.perf_counter()
        for i, t in enumerate(sch.timesteps):
            x = OP.step(model, sch, x, t, ctx_in, rays, mask, True, 3.0)
            if i in (0, 4, 9, 24, steps - 1):
                print(f"   step {i + 1}/{steps}  {time.perf_counter() - t0:.0f}s", flush=True)
        return x

    res = {"what": "relative L2 drift of the final latents against the unrounded fp32 oracle run; CPU emulation of the bf16 storage points "
                   "(tests/study_bf16_trunk.py); 1 scene, 1 + 4 views, CFG 3.0, SD-2.1 widths, seeded default-init weights",
           "ddim_steps": steps, "latent": hl, "north_star_latent_tolerance": 1e-3, "measured_hip": {"bf16": 5.9e-3, "f16": 7.7e-4}, "rows": {}}
    print("fp32 reference", flush=True)
    ref = run(den, {"trunk": False, "skip": False, "branch": False})
    del den
    for name, pol in (("all_bf16", {"trunk": True, "skip": True, "branch": True}),
                      ("trunk_fp32", {"trunk": False, "skip": True, "branch": True}),
                      ("trunk_skip_fp32", {"trunk": False, "skip": False, "branch": True}),
                      ("weights_only", {"trunk": False, "skip": False, "branch": False})):
        print(name, flush=True)
        x = run(den_r, pol)
        err = float((x - ref).norm() / ref.norm())
        res["rows"][name] = {"latent_rel_err": round(err, 6), "policy": pol}
        print(f"   -> {err:.3e}", flush=True)
        json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res["rows"]))


if __name__ == "__main__":
    main()
