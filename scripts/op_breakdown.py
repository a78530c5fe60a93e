#!/usr/bin/env python3
"""Per-op device-time table of one denoise step (HIP events), for choosing what to optimise next.
usage: python scripts/op_breakdown.py [--batch 4096] [--dtype bf16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import TrajUNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--L", type=int, default=200)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--force", type=int, nargs="*", default=[0], help="tdx_conv_gemm_force_small settings, one ms column each (first = baseline of the table)")
ap.add_argument("--fuse-gn", action="store_true", help="GroupNorm of a conv's consumer in the conv's epilogue where the shape allows (engine.FUSE_GN_DEFAULT)")
a = ap.parse_args()
if a.fuse_gn:
    from trajectorydiffusion_amd import engine
    engine.FUSE_GN_DEFAULT = True
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
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
from trajectorydiffusion_amd import _lib  # noqa: E402
cols = []
for f in a.force:
    _lib.lib().tdx_conv_gemm_force_small(f)
    cols.append(prog.pg.profile(repeats=5))
_lib.lib().tdx_conv_gemm_force_small(0)
rows = cols[-1]
tot = sum(r[2] for r in rows)
print(f"{'op':40s} {'kernel':20s} {'ms':>9s} {'TFLOP/s':>9s}" + "".join(f"  ms@force={f}" for f in a.force[:-1]))
agg = {}
for i, (name, fn, ms, fl) in enumerate(rows):
    print(f"{name:40s} {fn:20s} {ms:9.4f} {fl / ms / 1e9 if fl else 0:9.1f}" + "".join(f" {c[i][2]:12.4f}" for c in cols[:-1]))
    k = agg.setdefault(fn, [0.0, 0.0, 0])
    k[0] += ms
    k[1] += fl
    k[2] += 1
print("-" * 80)
for fn, (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{fn:30s} n={n:3d} {ms:9.3f} ms {100 * ms / tot:5.1f}%  {fl / ms / 1e9 if fl else 0:8.1f} TFLOP/s")
print(f"total {tot:.3f} ms per denoise step, B={a.batch}" + "".join(f"   (force={f}: {sum(r[2] for r in c):.3f})" for f, c in zip(a.force[:-1], cols[:-1])))
