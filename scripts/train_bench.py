#!/usr/bin/env python3
"""Config C4: one TrajUNet training step (q_sample -> forward -> MSE -> backward -> AdamW), fp32, B=1024, L=200.

    python scripts/train_bench.py [--batch 1024] [--steps 5] [--warmup 2] [--breakdown]

Prints ONE JSON line: training steps/s and trajectories/s, the time split between forward / backward / optimizer and (with
--breakdown) the per-entry-point device time of the two programs.  Synthetic data, random-init weights.  Not the headline
metric (bench.py is); this is the measurement for SURVEY.md 8 row a13.
"""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCHEDULE = [128, 128, 256, 512, 1024]
FLOP_PER_TRAJ_FWD = 2 * 1_015_085_824


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--L", type=int, default=200)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--breakdown", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from _synth import synth_inputs, synth_noise
    from trajectorydiffusion_amd import DiffusionManager, TrajUNet
    from trajectorydiffusion_amd.training import TrainStep
    torch.manual_seed(0)
    model = TrajUNet(SCHEDULE, diffusion_steps=a.T, res_blocks=2).to(dev)
    dm = DiffusionManager(1e-4, 0.05, a.T, device=str(dev))
    B, L = a.batch, a.L
    x0, t, cat, num = synth_inputs(B, L, a.T, seed=7)
    x0, t, cat, num = x0.to(dev), t.to(dev), cat.to(dev), num.to(dev)
    eps = synth_noise(0, (B, 2, L), seed=7).to(dev)
    ts = TrainStep(model, B, L, dev, lr=5e-4)
    torch.cuda.synchronize()
    mem_gb = torch.cuda.max_memory_allocated() / 2 ** 30

    def step():
        x_t = dm.diffusionForward(x0, t, eps)
        loss = ts.forward_backward(x_t, t, cat, num, eps)
        ts.optimizer_step()
        return loss

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.steps
    assert torch.isfinite(loss).all()

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    x_t = dm.diffusionForward(x0, t, eps)
    ev[0].record()
    ts.forward(x_t, t, cat, num)
    ev[1].record()
    ts.backward(eps)
    ev[2].record()
    ts.optimizer_step()
    ev[3].record()
    torch.cuda.synchronize()
    out = {"metric": "training steps/sec, TrajUNet fp32 step (forward+backward+AdamW), L=%d" % L, "value": round(1 / el, 4), "unit": "steps/s",
           "trajectories_per_s": round(B / el, 2), "ms_per_step": round(1e3 * el, 2), "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "dtype": "fp32", "data": "synthetic", "config": {"workload": f"C4: TrajUNet training step, B={B}, L={L}, schedule {SCHEDULE}, dropout 0.1, AdamW"},
           "forward_ms": round(ev[0].elapsed_time(ev[1]), 2), "backward_ms": round(ev[1].elapsed_time(ev[2]), 2),
           "optimizer_ms": round(ev[2].elapsed_time(ev[3]), 2),
           "model_tflops": round(3 * FLOP_PER_TRAJ_FWD * B / el / 1e12, 2), "peak_tflops_fp32_mfma": 157.3,
           "frac_of_fp32_mfma_peak": round(3 * FLOP_PER_TRAJ_FWD * B / el / 1e12 / 157.3, 4),
           "device_memory_gib": round(mem_gb, 2), "loss": float(loss)}
    if a.breakdown:
        for tag, pg in (("forward", ts.fw), ("backward", ts.bw)):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for name, fn, ms, fl in pg.profile(repeats=2):
                agg[fn][0] += 1
                agg[fn][1] += ms
            out[tag + "_by_entry_point"] = {k: {"calls": v[0], "ms": round(v[1], 3)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
