#!/usr/bin/env python3
"""Micro-benchmark of tdx_conv_gemm on the TrajUNet conv shapes at B=4096 (bf16), with ablation bits.
usage: python scripts/conv_bench.py [--debug BITS] [--small]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import _lib, engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--debug", type=int, nargs="*", default=[0])
ap.add_argument("--small", action="store_true")
ap.add_argument("--force", type=int, nargs="*", default=None, help="tdx_conv_gemm_force_small settings to compare (one column each)")
ap.add_argument("--gn", action="store_true", help="fuse a consumer GroupNorm+Swish into the epilogue")
ap.add_argument("--noraw", action="store_true")
ap.add_argument("--gnact", type=int, default=2)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default="", help="run the shapes whose name contains this")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
B = a.batch
SHAPES = [  # (name, L, Cin, Cout, taps)
    ("128->128@200", 200, 128, 128, 3), ("256->256@100", 100, 256, 256, 3), ("512->512@50", 50, 512, 512, 3),
    ("1024->1024@25", 25, 1024, 1024, 3), ("1024->1024@13", 13, 1024, 1024, 3), ("2048->512@13", 13, 2048, 512, 3),
    ("qkv 1024->3072@13", 13, 1024, 3072, 1), ("128->256@100", 100, 128, 256, 3), ("256->512@50", 50, 256, 512, 3),
    ("512->1024@25", 25, 512, 1024, 3), ("1x1 1024->1024@13", 13, 1024, 1024, 1),
]
lib = _lib.lib()
forces = a.force if a.force is not None else [1 if a.small else 0]
print(f"{'shape':22s} " + " ".join(f"f{f}/dbg{d}: ms / TF  " for f in forces for d in a.debug))
for name, L, ci, co, k in SHAPES:
    if a.only not in name:
        continue
    x = torch.randn((B, L, ci), device="cuda").to(dt)
    w = (torch.randn((co, k * ci), device="cuda") / (k * ci) ** 0.5).to(dt)
    bias = torch.randn(co, device="cuda")
    out = torch.empty((B, L, co), device="cuda", dtype=dt)
    pg = engine.Program(torch.device("cuda"), dt)
    gn = None
    if a.gn:
        class H:
            weight, bias, eps = torch.ones(co, device="cuda"), torch.zeros(co, device="cuda"), 1e-6
        y = torch.empty((B, L, co), device="cuda", dtype=dt)
        gn = engine.GnSpec(H, "h", 0, co // 32, a.gnact, y, co, 0)
    pg.conv([engine.Seg(x, ci, ci, L, taps=k)], w, bias, None if a.noraw else out, co, B, L, co, gn=gn)
    if len(pg.ops) != 1:
        continue
    fl = 2.0 * B * L * co * k * ci
    combos = [(f, d) for f in forces for d in a.debug]
    best = [1e9] * len(combos)
    for rnd in range(4):  # interleaved rounds, best of 4: clocks drift by a few % within a process
        for cidx, (f, d) in enumerate(combos):
            lib.tdx_conv_gemm_force_small(f)
            lib.tdx_conv_gemm_debug(d)
            for _ in range(2):
                pg.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                pg.run()
            e1.record()
            torch.cuda.synchronize()
            best[cidx] = min(best[cidx], e0.elapsed_time(e1) / 10)
    cells = [f"{ms:7.3f} / {fl / ms / 1e9:6.0f}" for ms in best]
    lib.tdx_conv_gemm_debug(0)
    lib.tdx_conv_gemm_force_small(0)
    print(f"{name:22s} " + "   ".join(cells))
