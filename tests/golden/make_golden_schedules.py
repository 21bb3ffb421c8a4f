"""G7: capture the outer generation schedules of the reference (build container only).

Runs `DiffusionWrapper.test_video_anchored` (src/model/diffusion_wrapper.py:644-902) and
`.test_video_autoregressive` (:904-1055) from /root/reference with `sample()` replaced by a
recorder: each call's context / target frame indices, the frame each context IMAGE came from (images
carry their frame index as pixel value) and the relative extrinsics handed to `sample()` are stored.
"""
from __future__ import annotations

import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

import ref_import as R  # noqa: E402
from seeded import random_cameras  # noqa: E402


def _tagged(indices):
    """images [1, n, 3, 2, 2] whose channel 0 is frame_index / 1000 (channels 1, 2: zero = "original image")."""
    t = torch.as_tensor(indices, dtype=torch.float32) / 1000.0
    out = torch.zeros(1, len(indices), 3, 2, 2)
    out[:, :, 0] = t.reshape(1, -1, 1, 1)
    return out


def _batch(n_ctx, ctx_idx, tgt_idx, seed):
    n_t = len(tgt_idx)
    extr, intr = random_cameras(1, n_ctx + n_t, seed)
    # give the first camera a non-identity pose as well
    extr = extr.roll(1, dims=1)
    mk = lambda sl, idx: {"image": _tagged(idx), "extrinsics": extr[:, sl].clone(), "intrinsics": intr[:, sl].clone(),
                          "near": torch.ones(1, len(idx)), "far": torch.full((1, len(idx)), 100.0),
                          "index": torch.tensor([idx], dtype=torch.int64)}
    return {"context": mk(slice(0, n_ctx), ctx_idx), "target": mk(slice(n_ctx, None), tgt_idx), "scene": ["synthetic"]}


def _run(mode, n_ctx, ctx_idx, tgt_idx, limit_frames, seed, num_anchors_views=4):
    W = R.ref("src.model.diffusion_wrapper")
    w = W.DiffusionWrapper.__new__(W.DiffusionWrapper)
    nn.Module.__init__(w)
    w.test_cfg = SimpleNamespace(num_anchors_views=num_anchors_views, sampling_mode=mode, limit_frames=limit_frames)
    w.step_tracker = SimpleNamespace(get_step=lambda: 0)
    w.output_dir = Path("/tmp/mvldm_golden_unused")
    calls = []

    def rec_sample(batch):
        c, t = batch["context"], batch["target"]
        # generated images carry frame index / 1000 in channel 0 (as before) and, in channel 1, the provenance
        # (call number * 8 + target slot + 1) / 1000 -- 0 for an original context image
        calls.append(dict(ctx_idx=c["index"][0].tolist(), tgt_idx=t["index"][0].tolist(),
                          ctx_tag=[int(round(float(x) * 1000)) for x in c["image"][0, :, 0, 0, 0]],
                          ctx_prov=[int(round(float(x) * 1000)) for x in c["image"][0, :, 1, 0, 0]],
                          ctx_extr=c["extrinsics"][0].clone(), tgt_extr=t["extrinsics"][0].clone()))
        out = _tagged(t["index"][0].tolist())
        k = len(calls) - 1
        for j in range(out.shape[1]):
            out[0, j, 1] = (k * 8 + j + 1) / 1000.0
        return out, batch
    w.sample = rec_sample
    o_save = W.save_image
    W.save_image = lambda *a, **k: None
    batch = _batch(n_ctx, ctx_idx, tgt_idx, seed)
    abs_extr = torch.cat([batch["context"]["extrinsics"], batch["target"]["extrinsics"]], dim=1)[0].clone()
    try:
        with R.cpu_cuda():
            if mode == "anchored":
                w.test_video_anchored(batch, 0, limit_frames=limit_frames)
            else:
                w.test_video_autoregressive(batch, 0, limit_frames=limit_frames)
    finally:
        W.save_image = o_save
    return calls, abs_extr


def g7():
    out = {}
    cases = [("anchored", 1, [0], list(range(1, 279)), None),
             ("anchored", 1, [0], list(range(1, 279)), 80),
             ("autoregressive", 1, [0], list(range(1, 81)), None),
             ("anchored", 2, [0, 90], list(range(1, 90)) + list(range(91, 121)), None),
             ("autoregressive", 2, [0, 40], list(range(1, 40)), None),
             ("anchored", 1, [5], list(range(6, 30)), None),
             # chained anchor calls (num_anchors_views > 4, diffusion_wrapper.py:744-792)
             ("anchored", 1, [0], list(range(1, 279)), None, 7),
             ("anchored", 1, [0], list(range(1, 126)), None, 10),
             ("anchored", 2, [0, 200], list(range(1, 200)), None, 7)]
    for i, case in enumerate(cases):
        mode, n_ctx, ctx_idx, tgt_idx, limit = case[:5]
        n_anch = case[5] if len(case) > 5 else 4
        calls, abs_extr = _run(mode, n_ctx, ctx_idx, tgt_idx, limit, seed=70 + i, num_anchors_views=n_anch)
        n = len(calls)
        ci = -np.ones((n, 2), np.int64); ti = -np.ones((n, 4), np.int64); ct = -np.ones((n, 2), np.int64)
        cp = -np.ones((n, 2), np.int64)
        ce = np.zeros((n, 2, 4, 4), np.float32); te = np.zeros((n, 4, 4, 4), np.float32)
        for k, c in enumerate(calls):
            ci[k, :len(c["ctx_idx"])] = c["ctx_idx"]; ti[k, :len(c["tgt_idx"])] = c["tgt_idx"]
            ct[k, :len(c["ctx_tag"])] = c["ctx_tag"]
            cp[k, :len(c["ctx_prov"])] = c["ctx_prov"]
            ce[k, :len(c["ctx_idx"])] = c["ctx_extr"].numpy(); te[k, :len(c["tgt_idx"])] = c["tgt_extr"].numpy()
        p = f"c{i}_"
        out.update({p + "mode": mode, p + "ctx_index": np.array(ctx_idx), p + "tgt_index": np.array(tgt_idx),
                    p + "limit_frames": -1 if limit is None else limit, p + "abs_extr": abs_extr,
                    p + "num_anchors_views": n_anch, p + "calls_ctx_prov": cp,
                    p + "calls_ctx_idx": ci, p + "calls_tgt_idx": ti, p + "calls_ctx_tag": ct,
                    p + "calls_ctx_extr": ce, p + "calls_tgt_extr": te})
        nviews = int((ti >= 0).sum())
        print(f"  g7 case {i}: {mode} n_ctx={n_ctx} N={len(tgt_idx)} limit={limit}: {n} calls, {nviews} views")
    out["n"] = len(cases)
    np.savez_compressed(HERE / "g7_schedules.npz", **out)
    print("wrote g7_schedules.npz")


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    g7()
