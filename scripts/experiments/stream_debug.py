"""Locate mismatches between the persistent conv kernel and the one-tile-per-workgroup kernel (debug aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from trajectorydiffusion_amd import _lib, engine
import test_gpu_ops as T
dt = torch.float32 if len(sys.argv) < 2 or sys.argv[1] == "fp32" else torch.bfloat16
case = sys.argv[2] if len(sys.argv) > 2 else "win"
g = torch.Generator().manual_seed(1)
B, L, Ci, Co, k = 1457, 100, 128, 256, 3
kw = {}
if case == "plain":
    pass
elif case == "win":
    kw.update(res=torch.randn(B, Co, L, generator=g), ebias=torch.randn(B, Co + 8, generator=g), act=2, tbias=torch.randn(Co + 8, generator=g))
x = torch.randn(B, Ci, L, generator=g); w = torch.randn(Co, Ci, k, generator=g) / 20; b = torch.randn(Co, generator=g)
_lib.lib().tdx_conv_gemm_force_small(1)
ref = T.run_conv(dt, x, w, b, **kw)
for force in (0, 16, 0, 16, 0, 32, 48, 0, 0):
    _lib.lib().tdx_conv_gemm_force_small(force)
    got = T.run_conv(dt, x, w, b, **kw)
    _lib.lib().tdx_conv_gemm_force_small(0)
    d = (got != ref) | torch.isnan(got)
    d = d.permute(0, 2, 1).reshape(B * L, Co).cpu()   # rows x channels
    bad_rows = d.any(1).nonzero().flatten()
    print(f"force={force}: mismatching elements {int(d.sum())}, rows {len(bad_rows)} of {B*L}")
    if len(bad_rows):
        tiles = torch.unique(bad_rows // 256)
        print("  bad 256-row tiles:", tiles[:40].tolist(), "... count", len(tiles))
        r = int(bad_rows[0]); cols = d[r].nonzero().flatten()
        print("  first bad row", r, "in-tile", r % 256, "bad cols", cols[:16].tolist(), "n", len(cols))
        gr = got.permute(0, 2, 1).reshape(B * L, Co).cpu(); rr = ref.permute(0, 2, 1).reshape(B * L, Co).cpu()
        print("  got", gr[r, cols[:6]].tolist(), "ref", rr[r, cols[:6]].tolist())
        rows_in_tile = torch.unique(bad_rows % 256)
        print("  in-tile rows affected:", rows_in_tile[:64].tolist(), "count", len(rows_in_tile))
        per_tile = [(int(t), int((bad_rows // 256 == t).sum())) for t in tiles[:10]]
        print("  rows per bad tile:", per_tile)
