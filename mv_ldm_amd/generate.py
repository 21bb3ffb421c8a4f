"""Generation harness: the build's counterpart of `src/scripts/generate_mvldm.py:29-87`.

Builds the object graph (denoiser, VAE, DDIM scheduler, pipeline), optionally loads a checkpoint, feeds SYNTHETIC
scenes shaped like the reference's `BatchedExample` (`src/dataset/types.py:16-28`: image [b,v,3,H,W] in [0,1],
extrinsics [b,v,4,4] camera-to-world, intrinsics [b,v,3,3] normalised), runs the anchored or autoregressive
schedule of `DiffusionWrapper.test_video_*`, optionally writes the frames as PNG, and reports views/s.

    python -m mv_ldm_amd.generate --mode anchored --frames 80 --steps 25 --scenes 8 [--ckpt last.ckpt] [--out frames/]
"""
from __future__ import annotations

import argparse
import json
import time

import torch


def synthetic_trajectory(n_frames: int, seed: int) -> torch.Tensor:
    """a smooth camera path: small per-frame rotations about a random axis + forward drift (camera-to-world)"""
    g = torch.Generator().manual_seed(seed)
    axis = torch.randn(3, generator=g)
    axis = axis / axis.norm()
    K = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    step = 0.2 / max(n_frames - 1, 1)
    extr = torch.eye(4).repeat(n_frames, 1, 1)
    for i in range(n_frames):
        th = torch.tensor(step * i)
        extr[i, :3, :3] = torch.eye(3) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
        extr[i, :3, 3] = torch.tensor([0.01 * i, 0.0, 0.02 * i]) + 0.002 * torch.randn(3, generator=g)
    return extr


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="anchored", choices=["anchored", "autoregressive"])
    ap.add_argument("--frames", type=int, default=80, help="target frames per scene")
    ap.add_argument("--steps", type=int, default=25, help="DDIM steps")
    ap.add_argument("--scenes", type=int, default=8, help="scenes generated together (batch dimension of every sample() call)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--ckpt", default=None, help="DiffusionWrapper checkpoint (.ckpt / .safetensors); default: random init")
    ap.add_argument("--out", default=None, help="directory for PNG frames")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)

    import mv_ldm_amd
    from mv_ldm_amd import _lib
    from mv_ldm_amd.image_io import save_image
    from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
    from mv_ldm_amd.pipeline import MVLDMPipeline, SamplerCfg
    from mv_ldm_amd.scheduler import DDIMScheduler
    from mv_ldm_amd.schedules import anchored_schedule, autoregressive_schedule, run_schedule_batched
    from mv_ldm_amd.vae import AutoencoderKL
    if not torch.cuda.is_available():
        raise SystemExit("generate needs a GPU (the HIP path has no CPU fallback)")
    _lib.load()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda")
    mv_ldm_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype])
    with torch.device(dev):
        den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1"), 11, 4)
        vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1")
    pipe = MVLDMPipeline(den, vae, DDIMScheduler(clip_sample=False), SamplerCfg(True, 3.0, args.steps))
    if args.ckpt:
        from mv_ldm_amd.checkpoint import load_pipeline_checkpoint
        rep = load_pipeline_checkpoint(pipe, args.ckpt)
        print({k: v.loaded for k, v in rep.items()}, flush=True)
    else:   # random init of the shape of a trained model (no network for checkpoints)
        g = torch.Generator(device=dev).manual_seed(args.seed)
        for m in (den, vae):
            for p in m.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g, device=dev) * (1.0 / max(p[0].numel(), 1)) ** 0.5)
    pipe.set_timesteps(args.steps)

    n_frames = args.frames + 1                                   # frame 0 is the context view
    intr = torch.tensor([[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]])
    fn = anchored_schedule if args.mode == "anchored" else autoregressive_schedule
    scene_calls, scene_imgs = [], []
    for s in range(args.scenes):
        extr = synthetic_trajectory(n_frames, args.seed + s)
        scene_calls.append(fn([0], extr[:1], list(range(1, n_frames)), extr[1:], limit_frames=args.frames))
        scene_imgs.append({0: torch.rand(3, args.res, args.res, generator=torch.Generator().manual_seed(1000 + s)).to(dev)})
    n_calls = len(scene_calls[0])

    torch.manual_seed(args.seed)
    run_schedule_batched(pipe, [c[:2] for c in scene_calls], scene_imgs, intrinsics_default=intr)      # warm-up: records both plan shapes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run_schedule_batched(pipe, scene_calls, scene_imgs, intrinsics_default=intr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    views = sum(len(o) for o in outs)
    if args.out:
        for s, o in enumerate(outs):
            for f, im in o.items():
                save_image(im, f"{args.out}/scene{s:03d}/frame{f:04d}.png")
    print(json.dumps({"mode": args.mode, "scenes": args.scenes, "frames_per_scene": args.frames, "ddim_steps": args.steps,
                      "sample_calls": n_calls, "views": views, "seconds": round(dt, 3), "views_per_s": round(views / dt, 3),
                      "dtype": args.dtype, "data": "synthetic trajectory, " + ("checkpoint" if args.ckpt else "random-init weights")}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
