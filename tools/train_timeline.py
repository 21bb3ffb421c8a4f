"""where one optimizer step of the training path spends its wall time (configs[3] shape): prepare (VAE encode + input staging),
weight re-pack after an optimizer step, the forward+backward plan, the optimizer step.  python tools/train_timeline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mv_ldm_amd
from mv_ldm_amd import _lib
from mv_ldm_amd.mvunet import MultiViewUNet, MultiViewUNetCfg
from mv_ldm_amd.scheduler import DDIMScheduler
from mv_ldm_amd.train import MVLDMTrainer
from mv_ldm_amd.vae import AutoencoderKL

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
_lib.load()
mv_ldm_amd.set_compute_dtype(torch.bfloat16)
with torch.device(dev):
    den = MultiViewUNet(MultiViewUNetCfg(pretrained_from="stabilityai/stable-diffusion-2-1", allow_random_init=True), 11, 4)
    vae = AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1", allow_random_init=True)
bench.random_init_(den, 1234)
bench.random_init_(vae, 1235)
tr = MVLDMTrainer(den, vae, DDIMScheduler(clip_sample=False), dtype=torch.bfloat16)
b = 4
batch = bench.synthetic_batch(b, 1, 3, 256, 4000, dev)
batch["target"]["image"] = torch.rand(b, 3, 3, 256, 256).to(dev)
for _ in range(4):
    tr.training_step(batch, index=1, unconditional=False)
torch.cuda.synchronize()

def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t)

acc = {}
def add(k, t):
    acc[k] = acc.get(k, 0) + t

N = 4
window = [batch, batch]
if os.environ.get("MVLDM_TIMELINE_PER_MICRO") != "1":
    # the accumulation window as ONE plan (MVLDMTrainer.training_window), phase by phase with a device sync between phases
    for _ in range(2):
        tr.training_window(window, [dict(index=1, unconditional=False)] * 2)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for _ in range(N):
        tr.training_window(window, [dict(index=1, unconditional=False)] * 2)
    torch.cuda.synchronize()
    whole = 1e3 * (time.perf_counter() - t_all) / N
    for it in range(N):
        parts, t = timed(lambda: [tr._host_part(bt, index=1, unconditional=False) for bt in window]); add("host part (image upload, cameras)", t)
        lats, t = timed(lambda: tr._encode([p_["x"] for p_ in parts], [p_["encode_noise"] for p_ in parts])); add("VAE encode (one call, 32 views)", t)
        parts, t = timed(lambda: [tr._finish_part(p_, lat) for p_, lat in zip(parts, lats)]); add("draws (noise, timesteps)", t)
        _, t = timed(lambda: (tr.flat.zero_grad(), [p.loss.zero_() for p in tr.plans.values()])); add("zero_grad", t)
        hl, wl = parts[0]["lat"].shape[-2:]
        tp = tr.plan_for_parts([(p_["b"], p_["vc_eff"], p_["v_t"]) for p_ in parts], hl, wl)
        _, t = timed(lambda: [tr._stage_part(tp, i, p_) for i, p_ in enumerate(parts)]); add("stage inputs into the plan's buffers", t)
        _, t = timed(lambda: tr._fresh(tp)); add("wait for / do the weight re-pack", t)
        _, t = timed(tp.run); add("plan (fwd + loss + bwd, 2 micro-batches)", t)
        tr.micro += 2
        _, t = timed(tr.opt.step); add("optimizer step (norm, clip, AdamW)", t)
        tr.global_step += 1
        tr._weights_gen += 1
        _, t = timed(lambda: tr._repack_ahead(tp)); add("weight re-pack (side stream, here waited for)", t)
    print({k: round(v / N, 2) for k, v in acc.items()}, "ms per optimizer step, phases serialised;", round(sum(acc.values()) / N, 2), "ms summed;",
          round(whole, 2), "ms per training_window() unserialised")
    _, t1 = timed(lambda: [fn() for fn in tp.repack])
    _, t2 = timed(lambda: tp._pack_batch.run() if tp._pack_batch is not None else None)
    print(f"re-pack: {len(tp.repack)} fp32 refresh closures {t1:.2f} ms, {len(tp.pack_jobs)} pack jobs in one batch {t2:.2f} ms")
    sys.exit(0)
for it in range(3):
    for micro in range(2):
        if micro == 0:
            _, t = timed(lambda: (tr.flat.zero_grad(), [p.loss.zero_() for p in tr.plans.values()]))
            acc["zero_grad"] = acc.get("zero_grad", 0) + t
        tp, t = timed(lambda: tr.prepare(batch, index=1, unconditional=False))
        acc["prepare (VAE encode + staging)"] = acc.get("prepare (VAE encode + staging)", 0) + t
        if tp.weights_gen != tr._weights_gen:
            _, t = timed(tp.refresh_weights)
            acc["refresh_weights (re-pack)"] = acc.get("refresh_weights (re-pack)", 0) + t
            tp.weights_gen = tr._weights_gen
        _, t = timed(tp.run)
        acc["plan (fwd + loss + bwd)"] = acc.get("plan (fwd + loss + bwd)", 0) + t
        tr.micro += 1
    _, t = timed(tr.opt.step)
    acc["optimizer step"] = acc.get("optimizer step", 0) + t
    tr._weights_gen += 1
print({k: round(v / 3, 2) for k, v in acc.items()}, "ms per optimizer step (2 micro-batches)")
