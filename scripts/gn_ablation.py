#!/usr/bin/env python3
"""Per-layer GroupNorm ablation of one TrajUNet denoise step (HIP events, same process, same box): every conv whose consumer is a
GroupNorm(+Swish), as conv + stand-alone tdx_groupnorm_act ("pass") against the GroupNorm in the conv's epilogue ("fused", wherever
the library can), and what engine.FUSE_GN_DEFAULT = "auto" chooses.
usage: python scripts/gn_ablation.py [--batch 2048] [--dtype bf16] > profiles/r3_gn_per_layer_ablation.txt"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import TrajUNet, engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2048, help="trajectories per launch (the sampler runs B=4096 as two slices of 2048)")
ap.add_argument("--L", type=int, default=200)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32


def measure(mode):
    engine.FUSE_GN_DEFAULT = mode
    torch.manual_seed(0)
    m = TrajUNet([128, 128, 256, 512, 1024], diffusion_steps=500).cuda().eval().set_compute_dtype(dt)
    prog = m.program(a.batch, a.L, torch.device("cuda"), split_embed=True)
    prog.x.normal_()
    prog.cat.zero_()
    prog.num.zero_()
    prog.time.zero_()
    prog.embed_pg.run()
    prog.tbias.zero_()
    prog.run()
    rows = prog.pg.profile(repeats=5, inner=4)
    out, order = {}, []
    for name, fn, ms, fl in rows:
        base = name[:-3] if name.endswith(".gn") else name
        e = out.setdefault(base, {"conv": 0.0, "gn": 0.0})
        if base not in order:
            order.append(base)
        e["gn" if name.endswith(".gn") or fn == "tdx_groupnorm_act" else "conv"] += ms
    total = sum(r[2] for r in rows)
    del prog, m
    torch.cuda.empty_cache()
    return out, order, total


res = {mode: measure(mode) for mode in (False, True, "auto")}
p0, order, t0 = res[False]
p1, _, t1 = res[True]
pa, _, ta = res["auto"]
print(f"# TrajUNet denoise step, B={a.batch} per launch, L={a.L}, {a.dtype}; ms per launch (HIP events, 4 launches per event pair, best of 5)")
print(f"# pass = conv + stand-alone GroupNorm pass; fused = GroupNorm in the conv's epilogue wherever tdx_conv_gemm_gn_fusable() > 0; auto = engine's choice")
print(f"{'layer':34s} {'conv':>8s} {'+ gn pass':>10s} {'= pass':>8s} {'fused':>8s} {'fused-pass':>11s} {'auto':>8s}  auto picks")
for k in order:
    e0, e1, ea = p0[k], p1.get(k, {"conv": 0, "gn": 0}), pa.get(k, {"conv": 0, "gn": 0})
    if e0["gn"] == 0.0:
        continue
    s0, s1, sa = e0["conv"] + e0["gn"], e1["conv"] + e1["gn"], ea["conv"] + ea["gn"]
    pick = "fused" if ea["gn"] == 0.0 and e0["gn"] > 0 else "pass"
    was = "" if e1["gn"] == 0.0 else "  (not fusable)"
    print(f"{k:34s} {e0['conv']:8.4f} {e0['gn']:10.4f} {s0:8.4f} {s1:8.4f} {s1 - s0:+11.4f} {sa:8.4f}  {pick}{was}")
print(f"# whole step (sum of all launches): pass {t0:.3f} ms, fused wherever possible {t1:.3f} ms, auto {ta:.3f} ms")
