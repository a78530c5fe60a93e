#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of the narrow conv kernel (needs a library built with -DNARROW_STAMPS)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import _lib, engine  # noqa: E402

B, L, BF = 4096, 200, torch.bfloat16
lib = _lib.lib()
dev = torch.device("cuda")
x = torch.randn(B, L, 128, device=dev).to(BF)
res = torch.randn(B, L, 128, device=dev).to(BF)
w = (torch.randn(128, 384, device=dev) / 20).to(BF)
bias = torch.randn(128, device=dev)
eb = torch.randn(B, 6912, device=dev)
gn = torch.nn.GroupNorm(32, 128, eps=1e-6).to(dev)
stats = torch.zeros(B, 13, 32, 2, device=dev)
mr = torch.zeros(B, 32, 2, device=dev)
out = torch.empty(B, L, 128, device=dev, dtype=BF)
norm = engine.NormIn(gn, "gn", mr, 4, _lib.ACT_SWISH)
names = ["wait", "barrier1", "dma issue", "transform", "barrier2", "offsets+K loop", "epilogue", "-"]
fn = lib.tdx_narrow_conv_stamps
fn.argtypes = [C.c_void_p]
for tag, kw in (("plain", {}), ("+res", dict(res=res, res_ld=128)), ("+norm+ebias+stats", dict(norm=norm, ebias=eb, ebias_ld=6912, ebias_off=128, stats_out=stats)),
                ("+norm+res+stats", dict(norm=norm, res=res, res_ld=128, stats_out=stats))):
    dbg = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
    pg = engine.Program(dev, BF)
    pg.conv([engine.Seg(x, 128, 128, L)], w, bias, out, 128, B, L, 128, **kw)
    pg.run()
    torch.cuda.synchronize()
    fn(dbg.data_ptr())
    pg.run()
    torch.cuda.synchronize()
    fn(None)
    d = dbg.view(512, 8).double().cpu()
    tiles = 100.0
    print(tag, " cycles per tile (mean over workgroups, wave 0):", {n: round(float(d[:, i].mean()) / tiles) for i, n in enumerate(names) if n != "-"},
          "total", round(float(d.sum(1).mean()) / tiles))
