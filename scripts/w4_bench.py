#!/usr/bin/env python3
"""Experiment: the four-wave 128 x 128 wave-tile conv kernel (tdx_conv_gemm_force_small bit 17) against the default persistent kernel:
bit-identity of the outputs, then time.  usage: python scripts/s3_bench.py [--check-only]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import _lib, engine  # noqa: E402

lib = _lib.lib()
S3 = 1 << 17
dt = torch.bfloat16
SHAPES = [("small 256->256@25 B=64", 64, 25, 256, 256), ("512->512@50", 4096, 50, 512, 512), ("1024->1024@25", 4096, 25, 1024, 1024),
          ("1024->1024@13", 4096, 13, 1024, 1024), ("2048->512@13", 4096, 13, 2048, 512), ("1024->1024@25 B=2048", 2048, 25, 1024, 1024)]
for name, B, L, ci, co in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, L, ci), device="cuda", generator=g).to(dt)
    w = (torch.randn((co, 3 * ci), device="cuda", generator=g) / (3 * ci) ** 0.5).to(dt)
    bias = torch.randn(co, device="cuda", generator=g)
    outs = []
    pgs = []
    for f in (16, 16 | S3):          # 16: no tail split on either side (the experiment has none)
        out = torch.zeros((B, L, co), device="cuda", dtype=dt)
        pg = engine.Program(torch.device("cuda"), dt)
        pg.conv([engine.Seg(x, ci, ci, L, taps=3)], w, bias, out, co, B, L, co)
        lib.tdx_conv_gemm_force_small(f)
        pg.run()
        torch.cuda.synchronize()
        outs.append(out)
        pgs.append(pg)
    same = torch.equal(outs[0], outs[1])
    diff = float((outs[0].float() - outs[1].float()).abs().max())
    line = f"{name:26s} identical={same} maxdiff={diff:.3e}"
    if "--check-only" not in sys.argv and (same or "--time-anyway" in sys.argv):
        fl = 2.0 * B * L * co * 3 * ci
        best = [1e9, 1e9]
        for rnd in range(4):
            for k, f in enumerate((16, 16 | S3)):
                lib.tdx_conv_gemm_force_small(f)
                for _ in range(2):
                    pgs[k].run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    pgs[k].run()
                e1.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], e0.elapsed_time(e1) / 10)
        line += f"   default {best[0]:.4f} ms {fl / best[0] / 1e9:7.0f} TF   w4 {best[1]:.4f} ms {fl / best[1] / 1e9:7.0f} TF"
    print(line, flush=True)
    lib.tdx_conv_gemm_force_small(0)
