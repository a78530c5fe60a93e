#!/usr/bin/env python3
"""Per-launch HIP-event times of one SDUNet_CAR denoise step (C5: B=4096, latents (B, 4, 24)), grouped by op kind and shape."""
import os
import re
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectorydiffusion_amd import SDUNet_CAR  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = torch.float32 if "fp32" in sys.argv else torch.bfloat16
torch.manual_seed(0)
if "fuse_gn" in sys.argv:
    from trajectorydiffusion_amd import engine
    engine.FUSE_GN_DEFAULT = True
m = SDUNet_CAR(4, [128, 256, 512, 1024], 500, res_blocks=4).cuda().eval().set_compute_dtype(dtype)
prog = m.program(B, 24, torch.device("cuda"), split_embed=True)
prog.cat.zero_()
prog.time.zero_()
prof = prog.pg.profile(repeats=5)
total = sum(ms for _, _, ms, _ in prof)
print(f"B={B} {dtype}: {len(prof)} launches, {total:.3f} ms per denoise step")
groups = defaultdict(lambda: [0, 0.0, 0.0])
for name, fn, ms, fl in prof:
    kind = re.sub(r"^(down_blocks|up_blocks)\.(\d+)\.\d+\.", r"\1.\2.*.", name)
    g = groups[(kind, fn)]
    g[0] += 1
    g[1] += ms
    g[2] += fl
for (kind, fn), (n, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    tf = f"{fl / ms / 1e9:8.1f} TF/s" if fl else " " * 13
    print(f"{ms:8.3f} ms  {n:3d}x  {tf}  {kind:40s} {fn}")
