#!/usr/bin/env python3
"""Evaluation metrics (eval_trajunet.ipynb cell 12) on B sampled trajectories: HIP-event time per metric, achieved HBM rate
against the algorithmic bytes (point histogram: every coordinate read twice; lengths: once).  (No CPU leg: the oracle is for
tests/, smoke() and bench.py's cpu_baseline only.)

    python scripts/metrics_bench.py [--batch 65536] [--L 200]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--L", type=int, default=200)
    a = ap.parse_args()
    from _synth import synth_metric_trajectories
    from trajectorydiffusion_amd import metrics as m
    t0, _ = synth_metric_trajectories(a.batch, a.L, 11)
    d = t0.cuda()
    nbytes = d.numel() * 4
    rows = {}
    for name, fn, passes in (("getPointDistribution", m.getPointDistribution, 2), ("getStartDistribution", m.getStartDistribution, None),
                             ("getLengthDistribution", m.getLengthDistribution, 1)):
        fn(d)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            ev0.record()
            fn(d)
            ev1.record()
            torch.cuda.synchronize()
            best = min(best, ev0.elapsed_time(ev1))
        rows[name] = {"ms": round(best, 4)}
        if passes:
            rows[name]["algorithmic_GBps"] = round(passes * nbytes / best / 1e6, 1)
            rows[name]["frac_of_8TBps"] = round(passes * nbytes / best / 1e6 / 8000, 4)
    gpu_ms = rows["getPointDistribution"]["ms"] + rows["getLengthDistribution"]["ms"]
    print(json.dumps({"workload": f"evaluation metrics, B={a.batch}, L={a.L}, fp32", "per_metric": rows,
                      "trajectories_per_s_point_plus_length": round(a.batch / (gpu_ms * 1e-3))}))


if __name__ == "__main__":
    main()
