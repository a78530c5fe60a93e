#!/usr/bin/env python3
"""Micro-benchmark of tdx_groupnorm_act on the TrajUNet shapes at B=4096 (bf16): ms and achieved GB/s (read + write)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import _lib
B = 4096
dt = torch.bfloat16
lib = _lib.lib()
print(f"{'shape':18s} {'ms':>8s} {'TB/s':>7s}")
for L, C in [(200, 128), (100, 256), (50, 512), (25, 1024), (13, 1024), (13, 2048), (100, 128), (25, 512), (24, 128), (12, 256), (6, 512), (3, 1024), (6, 1024), (12, 512), (24, 256)]:
    x = torch.randn((B, L, C), device="cuda").to(dt)
    y = torch.empty_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    def run():
        _lib.check(lib.tdx_groupnorm_act(x.data_ptr(), C, y.data_ptr(), C, 0, g.data_ptr(), b.data_ptr(), B, L, C, 32, 1e-5, 2, _lib.dtype_code(dt), _lib.stream_ptr()))
    best = 1e9
    for _ in range(4):
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"L={L:3d} C={C:4d}      {best:8.4f} {2 * x.numel() * 2 / best / 1e9:7.2f}")
x = torch.randn((B, 200, 128), device="cuda").to(dt); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"torch copy 210 MB   {ms:8.4f} {2 * x.numel() * 2 / ms / 1e9:7.2f}")
