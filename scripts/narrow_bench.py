#!/usr/bin/env python3
"""Device time of the narrow (128 -> 128, k=3) conv kernel by feature, against the general kernel (+ the GroupNorm pass it replaces).
usage: python scripts/narrow_bench.py [--batch 4096] [--L 200]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import _lib, engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--L", type=int, default=200)
a = ap.parse_args()
B, L, BF = a.batch, a.L, torch.bfloat16
lib = _lib.lib()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, L, 128, device=dev, generator=g).to(BF)
res = torch.randn(B, L, 128, device=dev, generator=g).to(BF)
w = (torch.randn(128, 384, device=dev, generator=g) / 20).to(BF)
bias = torch.randn(128, device=dev, generator=g)
eb = torch.randn(B, 6912, device=dev, generator=g)
gn = torch.nn.GroupNorm(32, 128, eps=1e-6).to(dev)
nblk = (L + 15) // 16
stats = torch.zeros(B, nblk, 32, 2, device=dev)
mr = torch.zeros(B, 32, 2, device=dev)
mr[..., 0] = 1.0
out = torch.empty(B, L, 128, device=dev, dtype=BF)
a_n = torch.empty_like(x)


def timed(pg, reps=20):
    for _ in range(3):
        pg.run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        pg.run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def conv_pg(force=0, **kw):
    pg = engine.Program(dev, BF)
    lib.tdx_conv_gemm_force_small(force)
    pg.conv([engine.Seg(x, 128, 128, L)], w, bias, out, 128, B, L, 128, **kw)
    lib.tdx_conv_gemm_force_small(0)
    return pg


norm = engine.NormIn(gn, "gn", mr, 4, _lib.ACT_SWISH)
cases = [("narrow plain", dict()), ("narrow +ebias", dict(ebias=eb, ebias_ld=6912, ebias_off=128)), ("narrow +res", dict(res=res, res_ld=128)),
         ("narrow +stats", dict(stats_out=stats)), ("narrow +norm", dict(norm=norm)), ("narrow +norm +ebias +stats", dict(norm=norm, ebias=eb, ebias_ld=6912, ebias_off=128, stats_out=stats)),
         ("narrow +norm +res +stats", dict(norm=norm, res=res, res_ld=128, stats_out=stats))]
flop = 2.0 * B * L * 128 * 384
for name, kw in cases:
    ms = timed(conv_pg(**kw))
    print(f"{name:34s} {ms:8.4f} ms  {flop / ms / 1e9:8.1f} TFLOP/s  {(2 * B * L * 128 * 2 + (B * L * 256 if 'res' in kw else 0)) / ms / 1e6:8.1f} GB/s")
NOFF = 1 << 16
lib.tdx_conv_gemm_force_small(NOFF)
for name, kw in (("general plain", dict()), ("general +ebias", dict(ebias=eb, ebias_ld=6912, ebias_off=128)), ("general +res", dict(res=res, res_ld=128))):
    pg = engine.Program(dev, BF)
    pg.conv([engine.Seg(x, 128, 128, L)], w, bias, out, 128, B, L, 128, **kw)
    ms = 0.0
    for _ in range(3):
        pg.run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        pg.run()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print(f"{name:34s} {ms:8.4f} ms  {flop / ms / 1e9:8.1f} TFLOP/s")
lib.tdx_conv_gemm_force_small(0)
pg = engine.Program(dev, BF)
pg.groupnorm(x, 128, a_n, 128, 0, gn.weight.detach(), gn.bias.detach(), 0, B, L, 128, 32, 1e-6, _lib.ACT_SWISH)
print(f"{'stand-alone GroupNorm+Swish':34s} {timed(pg):8.4f} ms")
