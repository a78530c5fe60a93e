#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of the persistent conv kernel (wave 0 of every workgroup), from a library built with
-DTDX_STAMPS=1 (never a product build).  usage: TDX_LIB_PATH=<stamps build> python scripts/conv_stamps.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import _lib, engine  # noqa: E402

B, BF = 4096, torch.bfloat16
lib = _lib.lib()
fn = lib.tdx_conv_gemm_stamps
fn.argtypes = [C.c_void_p]
dev = torch.device("cuda")
lib.tdx_conv_gemm_force_small(16)   # no tail launch (it would overwrite the main launch's slots)
NAMES = ["s1 vmcnt", "s1 barrier", "s1 compute", "s2 vmcnt", "s2 barrier", "s2 compute", "sN vmcnt", "sN barrier", "sN compute",
         "pin+issue", "stage+bar+issue2", "finish"]
for name, L, ci, co, k in (("128->256@100", 100, 128, 256, 3), ("256->256@100", 100, 256, 256, 3), ("1024->1024@25", 25, 1024, 1024, 3),
                           ("128->128@200", 200, 128, 128, 3)):
    x = torch.randn((B, L, ci), device=dev).to(BF)
    w = (torch.randn((co, k * ci), device=dev) / (k * ci) ** 0.5).to(BF)
    bias = torch.randn(co, device=dev)
    out = torch.empty((B, L, co), device=dev, dtype=BF)
    pg = engine.Program(dev, BF)
    pg.conv([engine.Seg(x, ci, ci, L, taps=k)], w, bias, out, co, B, L, co)
    for _ in range(3):
        pg.run()
    torch.cuda.synchronize()
    dbg = torch.zeros(4096 * 16, dtype=torch.int32, device=dev)
    fn(dbg.data_ptr())
    pg.run()
    torch.cuda.synchronize()
    fn(None)
    d = dbg.view(-1, 16).double().cpu()
    d = d[d[:, 15] > 1]          # workgroups of the main launch (the tail launch runs one tile per workgroup)
    tiles = d[:, 15].mean()
    nsteps = k * ci // 64
    per = {n: float(d[:, i].mean() / tiles) for i, n in enumerate(NAMES)}   # (slots 12, 13: whole-kernel cycle / 100 MHz counters)
    rest = max(nsteps - (2 if per["s2 compute"] > 0 else 1), 1)
    print(f"{name}: {len(d)} workgroups, {tiles:.1f} tiles each, {nsteps} K-steps; cycles per tile (wave 0):")
    print("   " + "  ".join(f"{n}={v:.0f}" for n, v in per.items()))
    clk = (d[:, 12] / d[:, 13].clamp(min=1)).median().item() * 100.0
    print(f"   in-kernel clock (s_memtime / s_memrealtime x 100 MHz, median over workgroups): {clk:.0f} MHz")
    print(f"   per middle step: vmcnt {per['sN vmcnt'] / rest:.0f}  barrier {per['sN barrier'] / rest:.0f}  compute {per['sN compute'] / rest:.0f};  total per tile {sum(per.values()):.0f}")
