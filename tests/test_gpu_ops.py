"""GPU: every C-ABI kernel against a plain fp32 PyTorch-CPU statement of the same op (op-level parity)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import DiffusionOracle, nearest_index_map  # noqa: E402
from oracle.trajunet_oracle import swish  # noqa: E402

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2}  # max |err| relative to max |ref|


def _eng():
    from trajectorydiffusion_amd import engine, _lib
    return engine, _lib


def rows(x, dt):  # (B, C, L) cpu -> (B, L, C) cuda
    return x.permute(0, 2, 1).contiguous().to("cuda", dt)


def unrows(r):
    return r.float().cpu().permute(0, 2, 1).contiguous()


def q(x, dt):  # what the kernel actually sees after the dtype cast
    return x.to(dt).float()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def run_conv(dt, x, w, b, stride=1, upmap_to=None, act=0, res=None, ebias=None, tbias=None, split_k_floats=0):
    engine, _lib = _eng()
    B, Ci, L = x.shape
    Co = w.shape[0]
    pg = engine.Program(torch.device("cuda"), dt)
    if split_k_floats:
        pg.split_k_ws = torch.full((split_k_floats,), float("nan"), device="cuda")
    src = rows(x, dt)
    if upmap_to is not None:
        lo = upmap_to
        seg = engine.Seg(src, Ci, Ci, L, taps=w.shape[2], lvirt=lo)
    else:
        lo = (L - 1) // stride + 1 if w.shape[2] == 3 else L
        seg = engine.Seg(src, Ci, Ci, L, taps=w.shape[2], stride=stride)
    out = torch.full((B, lo, Co), float("nan"), device="cuda", dtype=dt)
    kw = {}
    if res is not None:
        kw.update(res=rows(res, dt), res_ld=Co)
    if ebias is not None or tbias is not None:
        kw.update(ebias=None if ebias is None else ebias.cuda(), tbias=None if tbias is None else tbias.cuda(),
                  ebias_ld=Co + 8, ebias_off=8)
    pg.conv([seg], engine.pack_conv_weight(w.cuda(), dt), None if b is None else b.cuda(), out, Co, B, lo, Co, act=act, **kw)
    pg.run()
    torch.cuda.synchronize()
    return unrows(out)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,L,Ci,Co", [(3, 13, 64, 128), (2, 200, 128, 128), (5, 25, 256, 256), (1, 7, 96, 96), (130, 1, 128, 256),
                                       (2, 50, 32, 64)])
def test_conv3_stride1(dt, B, L, Ci, Co):
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, 3, generator=g) / (3 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv1d(q(x, dt), q(w, dt), b, padding=1)
    assert rel(run_conv(dt, x, w, b), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("L", [25, 200, 13, 14, 2])
def test_conv3_stride2(dt, L):
    g = torch.Generator().manual_seed(L)
    x = torch.randn(3, 128, L, generator=g)
    w = torch.randn(128, 128, 3, generator=g) / 20
    b = torch.randn(128, generator=g)
    ref = F.conv1d(q(x, dt), q(w, dt), b, stride=2, padding=1)
    assert rel(run_conv(dt, x, w, b, stride=2), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("li,lo", [(13, 25), (25, 50), (100, 200), (12, 24), (24, 47)])
def test_conv3_on_nearest_upsample(dt, li, lo):
    g = torch.Generator().manual_seed(li)
    x = torch.randn(2, 128, li, generator=g)
    w = torch.randn(128, 128, 3, generator=g) / 20
    b = torch.randn(128, generator=g)
    up = F.interpolate(q(x, dt), size=lo, mode="nearest")
    assert torch.equal(up, q(x, dt)[:, :, nearest_index_map(li, lo)])
    ref = F.conv1d(up, q(w, dt), b, padding=1)
    assert rel(run_conv(dt, x, w, b, upmap_to=lo), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_conv1_epilogue_bias_act_residual(dt):
    g = torch.Generator().manual_seed(7)
    B, L, Ci, Co = 4, 13, 128, 384
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, 1, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g)
    res = torch.randn(B, Co, L, generator=g)
    eb = torch.randn(B, Co + 8, generator=g)
    tb = torch.randn(Co + 8, generator=g)
    pre = F.conv1d(q(x, dt), q(w, dt), b) + eb[:, 8:, None] + tb[None, 8:, None]
    for act, fn in ((0, lambda v: v), (1, lambda v: F.leaky_relu(v, 0.1)), (2, swish)):
        ref = fn(pre) + q(res, dt)
        assert rel(run_conv(dt, x, w, b, act=act, res=res, ebias=eb, tbias=tb), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_conv_concat_plus_shortcut_segments(dt):
    """conv3 over an activated tensor + 1x1 shortcut over two raw concat sources, one launch (decoder ResnetBlock conv2)."""
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(11)
    B, L, C, Co = 3, 25, 64, 128
    a2 = torch.randn(B, Co, L, generator=g)
    xa, xb = torch.randn(B, C, L, generator=g), torch.randn(B, C, L, generator=g)
    w2 = torch.randn(Co, Co, 3, generator=g) / 20
    wsc = torch.randn(Co, 2 * C, 1, generator=g) / 12
    b = torch.randn(Co, generator=g)
    ref = F.conv1d(q(a2, dt), q(w2, dt), b, padding=1) + F.conv1d(torch.cat([q(xa, dt), q(xb, dt)], 1), q(wsc, dt))
    pg = engine.Program(torch.device("cuda"), dt)
    wp = torch.cat([engine.pack_conv_weight(w2.cuda(), dt), engine.pack_conv_weight(wsc.cuda(), dt)], dim=1).contiguous()
    out = torch.empty((B, L, Co), device="cuda", dtype=dt)
    segs = [engine.Seg(rows(a2, dt), Co, Co, L), engine.Seg(rows(xa, dt), C, C, L, taps=1), engine.Seg(rows(xb, dt), C, C, L, taps=1)]
    pg.conv(segs, wp, b.cuda(), out, Co, B, L, Co)
    pg.run()
    assert rel(unrows(out), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("case", ["k3", "k3_ragged", "s2", "up", "k1_epilogue", "tiny_ws", "one_row"])
def test_conv_split_k_small_batches(dt, case):
    """tdx_conv_desc.ws: few output tiles and a long K loop are cut into K slices (concurrent workgroups) and summed in order by
    a second kernel.  Same result as the unsplit kernel up to fp32 summation order, and identical on repetition."""
    g = torch.Generator().manual_seed(len(case))
    B, L, Ci, Co, taps, stride, up = 1, 13, 1024, 1024, 3, 1, None
    ws = 4 << 20
    if case == "k3_ragged":
        B, L, Ci, Co = 3, 7, 480, 200
    elif case == "s2":
        B, L, Ci, Co, stride = 2, 25, 512, 256, 2
    elif case == "up":
        B, L, Ci, Co, up = 2, 13, 512, 384, 25
    elif case == "k1_epilogue":
        B, L, Ci, Co, taps = 4, 13, 2048, 512, 1
    elif case == "tiny_ws":
        ws = B * L * Co * 3            # room for three slices only
    elif case == "one_row":
        B, L, Ci, Co, taps = 1, 1, 1280, 256, 1     # the embedding MLP at batch 1
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, taps, generator=g) / (taps * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    xin = F.interpolate(q(x, dt), size=up, mode="nearest") if up else q(x, dt)
    pre = F.conv1d(xin, q(w, dt), b, stride=stride, padding=taps // 2)
    kw = {}
    if case == "k1_epilogue":
        res, eb, tb = torch.randn(pre.shape, generator=g), torch.randn(B, Co + 8, generator=g), torch.randn(Co + 8, generator=g)
        pre = swish(pre + eb[:, 8:, None] + tb[None, 8:, None]) + q(res, dt)
        kw = dict(act=2, res=res, ebias=eb, tbias=tb)
    plain = run_conv(dt, x, w, b, stride=stride, upmap_to=up, **kw)
    split = run_conv(dt, x, w, b, stride=stride, upmap_to=up, split_k_floats=ws, **kw)
    assert rel(plain, pre) < TOL[dt] and rel(split, pre) < TOL[dt]
    assert rel(split, plain) < (2e-6 if dt == torch.float32 else 1e-2)
    assert not torch.equal(split, plain) or case == "one_row" or dt == torch.bfloat16      # the split path really ran (fp32: other rounding)
    assert torch.equal(split, run_conv(dt, x, w, b, stride=stride, upmap_to=up, split_k_floats=ws, **kw))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("seed", range(16))
def test_conv_seeded_random_shapes(dt, seed):
    """Random (seeded) shapes across every routing decision of tdx_conv_gemm: small / large row counts (register-staged, one-tile
    and persistent kernels), ragged last tiles, channel counts that are and are not whole 128-byte blocks, taps, stride 2,
    nearest upsampling, and every combination of the optional epilogue operands.  The output is pre-filled with NaN."""
    rng = np.random.default_rng(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    mode = ["plain", "plain", "s2", "up"][seed % 4]
    taps = 3 if mode != "plain" or rng.random() < 0.7 else 1
    Ci = int(rng.choice([32, 64, 96, 128, 192, 256, 512]))
    Co = int(rng.choice([8, 64, 96, 128, 256, 384, 512]))
    L = int(rng.integers(1, 60)) if rng.random() < 0.7 else int(rng.integers(150, 211))
    big = seed % 2 == 1                                         # odd seeds: enough rows for the LDS-DMA kernels' >= 128 tiles
    rows_target = int(rng.integers(33000, 60000)) if big else int(rng.integers(1, 3000))
    B = max(1, min(rows_target // max(L, 1), 4000))
    if B * L * Ci > 12_000_000:
        B = max(1, 12_000_000 // (L * Ci))
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, taps, generator=g) / (taps * Ci) ** 0.5
    b = torch.randn(Co, generator=g) if rng.random() < 0.8 else None
    stride, up = (2, None) if mode == "s2" else (1, None)
    if mode == "up":
        up = 2 * L - int(rng.integers(0, 2))
    xin = F.interpolate(q(x, dt), size=up, mode="nearest") if up else q(x, dt)
    pre = F.conv1d(xin, q(w, dt), b, stride=stride, padding=taps // 2)
    lo = pre.shape[2]
    res = torch.randn(B, Co, lo, generator=g) if rng.random() < 0.4 else None
    eb = torch.randn(B, Co + 8, generator=g) if rng.random() < 0.4 else None
    tb = torch.randn(Co + 8, generator=g) if rng.random() < 0.4 else None
    act = int(rng.choice([0, 0, 1, 2]))
    if eb is not None:
        pre = pre + eb[:, 8:, None]
    if tb is not None:
        pre = pre + tb[None, 8:, None]
    ref = (pre, F.leaky_relu(pre, 0.1), swish(pre))[act]
    if res is not None:
        ref = ref + q(res, dt)
    got = run_conv(dt, x, w, b, stride=stride, upmap_to=up, act=act, res=res, ebias=eb, tbias=tb)
    assert got.shape == ref.shape and bool(torch.isfinite(got).all()), (mode, B, L, Ci, Co, taps)
    assert rel(got, ref) < TOL[dt], (mode, B, L, Ci, Co, taps, act, res is not None, eb is not None, tb is not None)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("case", ["s1_c128", "s1_c256", "s2", "up", "k1_res", "tail_rows", "stream_win", "stream_s2", "stream_up", "stream_c128"])
def test_conv_large_tiles_dma_kernel(dt, case):
    """Shapes big enough for the LDS-DMA kernels (window and gather variants, persistent and one-tile-per-workgroup forms,
    with and without the small-tile tail launch): checked against F.conv1d AND bit-for-bit against each other and the
    128x128 register-staged kernel (same K order, same MFMA chain).  The stream_* cases give every workgroup of the
    persistent kernel two or more tiles plus a partial last round (569 tiles of 256 rows on 256 CUs)."""
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(len(case))
    kw = {}
    if case == "s1_c128":
        B, L, Ci, Co, k = 170, 200, 128, 128, 3
    elif case == "s1_c256":
        B, L, Ci, Co, k = 1400, 25, 256, 256, 3
    elif case == "s2":
        B, L, Ci, Co, k = 1400, 50, 128, 256, 3
        kw["stride"] = 2
    elif case == "up":
        B, L, Ci, Co, k = 1400, 13, 256, 256, 3
        kw["upmap_to"] = 25
    elif case == "k1_res":
        B, L, Ci, Co, k = 2600, 13, 256, 768, 1
    elif case == "stream_win":
        B, L, Ci, Co, k = 1457, 100, 128, 256, 3
    elif case == "stream_s2":
        B, L, Ci, Co, k = 1457, 200, 128, 256, 3
        kw["stride"] = 2
    elif case == "stream_up":
        B, L, Ci, Co, k = 2914, 25, 128, 256, 3
        kw["upmap_to"] = 50
    elif case == "stream_c128":
        B, L, Ci, Co, k = 1300, 200, 128, 128, 3   # 2032 tiles of 128 rows on 512 workgroup slots
    else:
        B, L, Ci, Co, k = 331, 101, 64, 384, 3   # N = 33431: ragged last row tile; Co = 384 -> 128-wide channel tiles
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / (k * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    lo = kw.get("upmap_to", (L - 1) // kw.get("stride", 1) + 1)
    if case in ("k1_res", "stream_win", "stream_s2", "stream_c128"):
        kw.update(res=torch.randn(B, Co, lo, generator=g), ebias=torch.randn(B, Co + 8, generator=g), act=2)
    if case in ("stream_win", "stream_up"):
        kw.update(tbias=torch.randn(Co + 8, generator=g))
    got = run_conv(dt, x, w, b, **kw)
    # 1: register-staged 128x128 kernel; 2: LDS-DMA without the shared tap window; 8: no 128-row tiles; 16: no tail launch;
    # 32: one tile per workgroup instead of the persistent kernel; 128: 128-channel tiles for every Cout
    for bits in (1, 2, 8, 16, 32, 48, 34, 128):
        _lib.lib().tdx_conv_gemm_force_small(bits)
        try:
            other = run_conv(dt, x, w, b, **kw)
        finally:
            _lib.lib().tdx_conv_gemm_force_small(0)
        assert torch.equal(got, other), bits
    xq = q(x, dt)
    if "upmap_to" in kw:
        xq = F.interpolate(xq, size=kw["upmap_to"], mode="nearest")
    ref = F.conv1d(xq, q(w, dt), b, stride=kw.get("stride", 1), padding=k // 2)
    if "tbias" in kw:
        ref = ref + kw["tbias"][None, 8:, None]
    if "ebias" in kw:
        ref = swish(ref + kw["ebias"][:, 8:, None]) + q(kw["res"], dt)
    assert rel(got, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,L,Ci,Co,cpg,act,raw", [(40, 13, 128, 1024, 32, 2, True), (9, 200, 128, 128, 4, 2, False), (300, 25, 64, 256, 8, 0, True),
                                                  (7, 100, 256, 256, 16, 2, False), (3, 256, 64, 128, 4, 2, True), (21, 50, 128, 512, 64, 2, True)])
def test_conv_with_fused_consumer_groupnorm(dt, B, L, Ci, Co, cpg, act, raw):
    """conv -> (+bias, +embed bias, +residual) -> raw store AND GroupNorm(+Swish) of the consumer in the same launch."""
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(L * 7 + Co)
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, 3, generator=g) / (3 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    eb = torch.randn(B, Co, generator=g)
    res = torch.randn(B, Co, L, generator=g)
    gamma, beta = torch.randn(2 * Co, generator=g), torch.randn(2 * Co, generator=g)
    pre = F.conv1d(q(x, dt), q(w, dt), b, padding=1) + eb[:, :, None] + q(res, dt)
    ref = F.group_norm(pre, Co // cpg, gamma[Co:], beta[Co:], eps=1e-6)
    ref = swish(ref) if act == 2 else ref

    class Holder:  # stands in for an nn.GroupNorm: parameters + eps
        weight, bias, eps = gamma.cuda(), beta.cuda(), 1e-6

    def run():
        pg = engine.Program(torch.device("cuda"), dt)
        pg.fuse_gn = True
        out = torch.full((B, L, Co), float("nan"), device="cuda", dtype=dt) if raw else None
        y = torch.zeros((B, L, Co + 16), device="cuda", dtype=dt)
        spec = engine.GnSpec(Holder, "t", Co, cpg, act, y, Co + 16, 16)
        pg.conv([engine.Seg(rows(x, dt), Ci, Ci, L)], engine.pack_conv_weight(w.cuda(), dt), b.cuda(), out, Co, B, L, Co,
                res=rows(res, dt), res_ld=Co, ebias=eb.cuda(), ebias_ld=Co, gn=spec)
        pg.run()
        torch.cuda.synchronize()
        return pg, out, y
    pg, out, y = run()
    assert [n for n, _, _ in pg.ops] == ["conv"], "expected the GroupNorm to be fused into the conv launch"
    got = unrows(y)
    assert float(got[:, :16].abs().max()) == 0.0
    tol = 5e-5 if dt == torch.float32 else 1.5e-2
    assert rel(got[:, 16:], ref) < tol
    if raw:
        assert rel(unrows(out), pre) < TOL[dt]
    _, _, y2 = run()
    assert torch.equal(y, y2)  # no atomics: bitwise reproducible
    _lib.lib().tdx_conv_gemm_force_small(4)  # unfused fallback: conv + tdx_groupnorm_act
    try:
        pg3, _, y3 = run()
    finally:
        _lib.lib().tdx_conv_gemm_force_small(0)
    assert [n for n, _, _ in pg3.ops] == ["conv", "conv.gn"]
    assert rel(unrows(y3)[:, 16:], ref) < tol


@pytest.mark.parametrize("B,L,Ci,Co,cpg,act,raw,mode", [
    (700, 25, 128, 1024, 32, 2, False, 1),    # ResnetBlock conv1 at the 1024-channel level: embed rows, normalised output only; 280 tiles (> one per CU)
    (640, 13, 256, 1024, 32, 0, True, 2),     # the attention's proj: residual, raw rows too, no activation
    (300, 50, 128, 512, 16, 2, False, 0),
    (260, 25, 128, 256, 8, 2, True, 1),
    (200, 13, 128, 512, 64, 2, False, 0),     # groups of 64 channels (the decoder's concat GroupNorm over one half)
    (100, 24, 128, 128, 8, 2, False, 2),      # 128-channel tiles
    (2000, 3, 128, 1024, 32, 2, False, 2),    # latent path: 85 trajectories per tile
    (5, 13, 128, 1024, 32, 2, False, 1),      # one partly filled tile
])
def test_conv_fused_groupnorm_persistent_form(B, L, Ci, Co, cpg, act, raw, mode):
    """The consumer's GroupNorm(+Swish) in the epilogue of the PERSISTENT conv kernel (bf16; tdx_conv_gemm_gn_fusable() == 2): against the
    torch statement, against conv + stand-alone tdx_groupnorm_act, bitwise repeatable, and invariant to how the batch is cut."""
    engine, _lib = _eng()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(L * 11 + Co + mode)
    x = torch.randn(B, Ci, L, generator=g)
    w = torch.randn(Co, Ci, 3, generator=g) / (3 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    eb = torch.randn(B, Co, generator=g) if mode == 1 else None
    res = torch.randn(B, Co, L, generator=g) if mode == 2 else None
    gamma, beta = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    pre = F.conv1d(q(x, dt), q(w, dt), b, padding=1)
    if eb is not None:
        pre = pre + eb[:, :, None]
    if res is not None:
        pre = pre + q(res, dt)
    ref = F.group_norm(pre, Co // cpg, gamma, beta, eps=1e-6)
    ref = swish(ref) if act == 2 else ref

    class Holder:
        weight, bias, eps = gamma.cuda(), beta.cuda(), 1e-6

    def run(lo=0, hi=B, fuse=True):
        n = hi - lo
        pg = engine.Program(torch.device("cuda"), dt)
        pg.fuse_gn = fuse
        out = torch.full((n, L, Co), float("nan"), device="cuda", dtype=dt) if raw else None
        y = torch.zeros((n, L, Co + 8), device="cuda", dtype=dt)
        spec = engine.GnSpec(Holder, "t", 0, cpg, act, y, Co + 8, 8)
        pg.conv([engine.Seg(rows(x[lo:hi], dt), Ci, Ci, L)], engine.pack_conv_weight(w.cuda(), dt), b.cuda(), out, Co, n, L, Co,
                res=None if res is None else rows(res[lo:hi], dt), res_ld=Co, ebias=None if eb is None else eb[lo:hi].contiguous().cuda(), ebias_ld=Co, gn=spec)
        pg.run()
        torch.cuda.synchronize()
        return pg, out, y
    pg, out, y = run()
    assert [n for n, _, _ in pg.ops] == ["conv"]
    d = pg.keep[[type(k).__name__ for k in pg.keep].index("ConvDesc")]
    assert _lib.lib().tdx_conv_gemm_gn_fusable(d) == 2, "expected the persistent kernel's GroupNorm epilogue"
    got = unrows(y)
    assert float(got[:, :8].abs().max()) == 0.0
    assert rel(got[:, 8:], ref) < 1.5e-2
    if raw:
        assert rel(unrows(out), pre) < TOL[dt]
    _, _, y2 = run()
    assert torch.equal(y, y2)                       # fixed summation order: bitwise reproducible
    cut = max(1, B // 3)
    _, _, ya = run(0, cut)
    _, _, yb = run(cut, B)
    assert torch.equal(torch.cat([ya, yb]), y)      # a trajectory's statistics do not depend on its place in a tile or in the batch
    _, _, y3 = run(fuse=False)                      # conv + tdx_groupnorm_act: statistics of the bf16-rounded rows instead of the fp32 ones
    assert rel(unrows(y3)[:, 8:], ref) < 1.5e-2
    assert rel(unrows(y3)[:, 8:], got[:, 8:]) < 1.5e-2


def test_conv_fused_groupnorm_falls_back_when_trajectory_exceeds_tile():
    engine, _lib = _eng()
    B, L, C = 2, 300, 128
    x = torch.randn(B, C, L)
    w = torch.randn(C, C, 3) / 20

    class Holder:
        weight, bias, eps = torch.ones(C).cuda(), torch.zeros(C).cuda(), 1e-6
    pg = engine.Program(torch.device("cuda"), torch.float32)
    pg.fuse_gn = True
    y = torch.zeros((B, L, C), device="cuda")
    pg.conv([engine.Seg(rows(x, torch.float32), C, C, L)], engine.pack_conv_weight(w.cuda(), torch.float32), None, None, C, B, L, C,
            gn=engine.GnSpec(Holder, "t", 0, 4, 0, y, C, 0))
    pg.run()
    assert len(pg.ops) == 2
    assert rel(unrows(y), F.group_norm(F.conv1d(x, w, padding=1), 32, eps=1e-6)) < 5e-5


def test_conv_rejects_bad_arguments():
    engine, _lib = _eng()
    pg = engine.Program(torch.device("cuda"), torch.float32)
    x = torch.zeros((2, 10, 32), device="cuda")
    out = torch.zeros((2, 10, 128), device="cuda")
    w = torch.zeros((128, 96), device="cuda")
    pg.conv([engine.Seg(x, 32, 32, 10, taps=3, stride=3)], w, None, out, 128, 2, 10, 128)
    with pytest.raises(_lib.TdxError, match="stride"):
        pg.run()
    pg = engine.Program(torch.device("cuda"), torch.float32)
    pg.conv([engine.Seg(x, 32, 32, 10, taps=3)], w, None, out, 128, 2, 11, 128)  # wrong lout
    with pytest.raises(_lib.TdxError, match="output rows"):
        pg.run()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,L,C,act", [(3, 200, 128, 2), (2, 13, 1024, 2), (2, 13, 2048, 2), (4, 25, 256, 0), (1, 100, 512, 2), (2, 7, 4096, 0),
                                       # sweeps split over 2 / 4 waves of a workgroup (L > 64 / 128): odd lengths, a last part that is
                                       # short or a single row, idle waves in the last workgroup, and lengths past the split's reach
                                       (1, 65, 128, 2), (3, 129, 128, 0), (1, 199, 128, 2), (5, 97, 256, 2), (1, 128, 256, 0), (2, 513, 128, 2),
                                       (1, 600, 128, 0),
                                       # fewer than 8 rows: the window is widened (fewer rows per sweep) as far as C allows
                                       (5, 3, 1024, 2), (4, 6, 512, 2), (3, 5, 256, 0), (2, 1, 2048, 2), (3, 2, 128, 0)])
def test_groupnorm_act(dt, B, L, C, act):
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(C + L)
    x = torch.randn(B, C, L, generator=g) * 2 + 0.7
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.group_norm(q(x, dt), 32, gamma, beta, eps=1e-6)
    ref = swish(ref) if act == 2 else ref
    pg = engine.Program(torch.device("cuda"), dt)
    out = torch.zeros((B, L, C + 64), device="cuda", dtype=dt)
    pg.groupnorm(rows(x, dt), C, out, C + 64, 64, gamma.cuda(), beta.cuda(), 0, B, L, C, 32, 1e-6, act)
    pg.run()
    got = unrows(out)
    assert rel(got[:, 64:], ref) < (3e-5 if dt == torch.float32 else 1.2e-2)
    assert float(got[:, :64].abs().max()) == 0.0  # columns outside [out_off, out_off + C) untouched


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,L,C", [(3, 13, 1024), (2, 16, 256), (1, 64, 128), (2, 1, 128), (9, 13, 96), (5, 7, 1024), (2, 17, 64)])
def test_attention(dt, B, L, C):
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 3 * C, generator=g)
    qq = q(qkv, dt)
    qv, kv, vv = torch.split(qq, C, dim=2)
    ref = torch.bmm(torch.softmax(torch.bmm(qv, kv.transpose(1, 2)) / C ** 0.5, dim=2), vv)
    out = torch.empty((B, L, C), device="cuda", dtype=dt)
    qd = qkv.to("cuda", dt)
    _lib.check(_lib.lib().tdx_attention(qd.data_ptr(), out.data_ptr(), B, L, C, _lib.dtype_code(dt), _lib.stream_ptr()))
    assert rel(out.float().cpu(), ref) < (2e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("L", [200, 187, 1])
def test_stem_and_head(dt, L):
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(L)
    B, C0 = 3, 128
    x = torch.randn(B, 2, L, generator=g)
    w, b = torch.randn(C0, 2, 3, generator=g), torch.randn(C0, generator=g)
    out = torch.empty((B, L, C0), device="cuda", dtype=dt)
    code = _lib.dtype_code(dt)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()  # keep the device copies alive while raw pointers are in flight
    _lib.check(_lib.lib().tdx_stem_conv(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, L, C0, code, _lib.stream_ptr()))
    ref = F.conv1d(x, w, b, padding=1)
    assert rel(unrows(out), ref) < (1e-6 if dt == torch.float32 else 5e-3)
    a = torch.randn(B, C0, L, generator=g)
    wh, bh = torch.randn(2, C0, 3, generator=g) / 20, torch.randn(2, generator=g)
    eps = torch.empty((B, 2, L), device="cuda")
    ad, whd, bhd = rows(a, dt), wh.cuda(), bh.cuda()
    _lib.check(_lib.lib().tdx_head_conv(ad.data_ptr(), whd.data_ptr(), bhd.data_ptr(), eps.data_ptr(), B, L, C0, code, _lib.stream_ptr()))
    assert rel(eps.cpu(), F.conv1d(q(a, dt), wh, bh, padding=1)) < 2e-5


def test_ddpm_update_injected_noise_matches_reference_formula():
    from trajectorydiffusion_amd import DiffusionManager
    dm = DiffusionManager(1e-4, 0.05, 500, device="cuda")
    o = DiffusionOracle(1e-4, 0.05, 500)
    g = torch.Generator().manual_seed(0)
    for shape in [(16, 2, 200), (3, 2, 187), (1, 2, 1)]:
        x, e, z = (torch.randn(shape, generator=g) for _ in range(3))
        for t in (499, 250, 1, 0):
            got = dm.diffusionBackwardStep(x.cuda(), t, e.cuda(), noise=z.cuda()).cpu()
            ref = o.posterior_step(x, t, e, z)
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-6 * float(ref.abs().max()))
    with pytest.raises(IndexError):
        dm.diffusionBackwardStep(x.cuda(), 500, e.cuda())


def test_ddpm_update_philox_noise_is_standard_normal_and_shard_invariant():
    from trajectorydiffusion_amd import _lib
    lib = _lib.lib()
    n = 1 << 22
    coef = torch.tensor([0.0, 1.0, 1.0], device="cuda")  # x = 0 + 1 * z
    ctr = torch.tensor([5, 123, 456, 5], dtype=torch.int32, device="cuda")   # {step, seed_lo, seed_hi, Philox stream id (= t)}
    zeros = torch.zeros(n, device="cuda")

    def draw(first, count, seed=99):
        x = torch.zeros(count, device="cuda")
        _lib.check(lib.tdx_ddpm_update(x.data_ptr(), zeros.data_ptr(), None, coef.data_ptr(), count, seed, ctr.data_ptr(), first, _lib.stream_ptr()))
        return x
    z = draw(0, n)
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.var()) - 1) < 5e-3
    assert abs(float((z ** 4).mean()) - 3) < 0.05 and float(z.abs().max()) < 7
    half = n // 2
    assert torch.equal(torch.cat([draw(0, half), draw(half, half)]), z)  # batch shards draw what the whole batch draws
    assert not torch.equal(draw(0, 1024, seed=100), z[:1024])
    ctr[0] = 6
    assert torch.equal(draw(0, 1024), z[:1024])          # the step counter itself is not part of the key ...
    ctr[3] = 6
    assert not torch.equal(draw(0, 1024), z[:1024])      # ... the stream id (the diffusion step t) is
    # tdx_step_prepare writes the stream id: step_to_t[step], then advances the step
    step_to_t = torch.tensor([499, 498, 7], dtype=torch.int32, device="cuda")
    coef_tab = torch.zeros(3, 4, device="cuda"); coef_cur = torch.zeros(4, device="cuda")
    c2 = torch.tensor([2, 0, 0, 0], dtype=torch.int32, device="cuda")
    _lib.check(lib.tdx_step_prepare(c2.data_ptr(), coef_tab.data_ptr(), 4, coef_cur.data_ptr(), None, 0, None, step_to_t.data_ptr(), None, 0, 3, _lib.stream_ptr()))
    assert c2.tolist() == [3, 0, 0, 7]


def test_q_sample():
    from trajectorydiffusion_amd import DiffusionManager
    dm = DiffusionManager(1e-4, 0.05, 500, device="cuda")
    o = DiffusionOracle(1e-4, 0.05, 500)
    g = torch.Generator().manual_seed(1)
    x0, e = torch.randn(8, 2, 200, generator=g), torch.randn(8, 2, 200, generator=g)
    t = torch.tensor([0, 1, 17, 250, 499, 3, 3, 498])
    got = dm.diffusionForward(x0.cuda(), t.cuda(), e.cuda()).cpu()
    np.testing.assert_array_equal(got.numpy(), o.q_sample(x0, t, e).numpy())  # mul, mul, add in fp32: bit-exact
    with pytest.raises(IndexError):
        dm.diffusionForward(x0.cuda(), torch.full((8,), 500).cuda(), e.cuda())


def test_didi_items_match_golden(golden_dir, tmp_path):
    import os
    from _synth import synth_raw_trajectories
    from trajectorydiffusion_amd import DidiTrajectoryDataset, collectFunc
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    root = tmp_path / "nov"
    root.mkdir()
    torch.save(synth_raw_trajectories(6, 200, seed=4), str(root / "gps_20161101.pt"))
    ds = DidiTrajectoryDataset(str(root), 200, [21599.4980, 108.950773428688, 34.24354179925547],
                               [12470.9102, 0.02129110045580343, 0.019358855648211895])
    assert ds.loadNextFiles(1) and len(ds) == g["lon_lat"].shape[0]
    ll, cat, num, tm = collectFunc([ds[i] for i in range(len(ds))])
    np.testing.assert_array_equal(ll.cpu().numpy(), g["lon_lat"])   # z-score is (x - mean) / std in fp32: bit-exact
    np.testing.assert_array_equal(tm.cpu().numpy(), g["times"])
    np.testing.assert_array_equal(cat.cpu().numpy(), g["cat"])
    np.testing.assert_allclose(num.cpu().numpy(), g["num"], rtol=2e-6)  # 199-term fp32 sum, different association
    deg = ds.denormalize(ll)
    raw = torch.stack([a for a, _ in ds.dataset_part])
    assert float((deg.cpu() - raw).abs().max()) < 2e-5  # degrees; fp32 ulp at 108.95 deg is 7.6e-6 (~0.8 m)
    # getCellIndex (reference :127-135) on the normalised start / end points reproduces the golden categorical attributes bit for bit
    for i in range(len(ds)):
        src = ds.getCellIndex(ll[i, :, 0])
        dst = ds.getCellIndex(ll[i, :, -1])
        assert src.dtype == torch.long and src.shape == (2,)
        np.testing.assert_array_equal(torch.cat([src, dst]).cpu().numpy(), g["cat"][i, :4])
    # ... and the searchsorted(right=False) - 1 rule at its edges: a value ON a boundary belongs to the cell below it
    b = torch.from_numpy(g["boundaries"])
    probe = torch.tensor([float(b[1]), float(b[128]), -10.0, 10.0, 0.0049, float(b[255]) + 1e-3])
    want = torch.searchsorted(b, probe) - 1
    got = torch.stack([ds.getCellIndex(torch.stack([probe[j], probe[j + 1]]).cuda()) for j in (0, 2, 4)]).reshape(-1).cpu()
    np.testing.assert_array_equal(got.numpy(), want.numpy())
    with pytest.raises(Exception):
        ds.getCellIndex(torch.zeros(2))   # CPU tensor: no fallback
    with pytest.raises(ValueError):
        DidiTrajectoryDataset(str(tmp_path), 200, [0, 0, 0], [1, 1, 1])
