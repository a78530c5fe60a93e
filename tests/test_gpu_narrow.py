"""GPU: the narrow (128 -> 128, k = 3) conv kernel of csrc/narrow_conv.hip and the split GroupNorm around it (block partial sums
-> tdx_groupnorm_finalize -> normalise-on-load), through the C ABI, against plain fp32 PyTorch statements of the same ops and,
bit for bit, against the general conv kernels."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
NARROW_OFF = 1 << 16     # tdx_conv_gemm_force_small bit: never take the narrow kernel


def _eng():
    from trajectorydiffusion_amd import engine, _lib
    return engine, _lib


def rows(x):  # (B, C, L) cpu fp32 -> (B, L, C) cuda bf16
    return x.permute(0, 2, 1).contiguous().to("cuda", BF)


def unrows(r):
    return r.float().cpu().permute(0, 2, 1).contiguous()


def swish(x):
    return x * torch.sigmoid(x)


def conv_once(x, w, b, B, L, stride=1, up_to=None, res=None, ebias=None, tbias=None, norm=None, want_stats=False, force=0):
    """One tdx_conv_gemm call on (B, 128, L) -> rows out, optional stats; `norm` = (mr tensor, gn module, act)."""
    engine, _lib = _eng()
    lib = _lib.lib()
    pg = engine.Program(torch.device("cuda"), BF)
    src = rows(x)
    if up_to is not None:
        lo, seg = up_to, engine.Seg(src, 128, 128, L, lvirt=up_to)
    else:
        lo = (L - 1) // stride + 1
        seg = engine.Seg(src, 128, 128, L, stride=stride)
    out = torch.full((B, lo, 128), float("nan"), device="cuda", dtype=BF)
    kw = {}
    if res is not None:
        kw.update(res=rows(res), res_ld=128)
    if ebias is not None or tbias is not None:
        kw.update(ebias=None if ebias is None else ebias.cuda(), tbias=None if tbias is None else tbias.cuda(), ebias_ld=136, ebias_off=8)
    stats = torch.full((B, (lo + 15) // 16, 32, 2), float("nan"), device="cuda") if want_stats else None
    if norm is not None:
        mr, gn, act = norm
        kw["norm"] = engine.NormIn(gn, "gn", mr, 128 // gn.num_groups, act)
    lib.tdx_conv_gemm_force_small(force)
    try:
        pg.conv([seg], engine.pack_conv_weight(w.cuda(), BF), None if b is None else b.cuda(), out, 128, B, lo, 128, stats_out=stats, **kw)
        pg.run()
        torch.cuda.synchronize()
    finally:
        lib.tdx_conv_gemm_force_small(0)
    return out, stats


@pytest.mark.parametrize("L,stride,up_to", [(200, 1, None), (100, 1, None), (187, 1, None), (37, 1, None), (200, 2, None), (187, 2, None),
                                            (50, 2, None), (100, 1, 200), (50, 1, 100), (94, 1, 187)])
def test_narrow_kernel_is_bit_identical_to_the_general_kernels(L, stride, up_to):
    """Same K order per output element and the same epilogue association: the narrow kernel must reproduce the general
    LDS-DMA / register-staged kernels bit for bit (bias + per-step row + per-trajectory rows, then the residual)."""
    g = torch.Generator().manual_seed(L * 7 + stride)
    lo = up_to if up_to is not None else (L - 1) // stride + 1
    B = max(3, (300 * 64) // ((lo + 15) // 16 * 16) + 1)        # >= 256 tiles of 64 padded rows
    x = torch.randn(B, 128, L, generator=g)
    w = torch.randn(128, 128, 3, generator=g) / 20
    b = torch.randn(128, generator=g)
    eb, tb = torch.randn(B, 136, generator=g), torch.randn(136, generator=g)
    res = torch.randn(B, 128, lo, generator=g)
    for kw in (dict(), dict(ebias=eb, tbias=tb), dict(res=res), dict(ebias=eb)):
        if stride == 2 and "res" in kw:
            continue                                              # (no residual on the stride-2 path of the narrow kernel)
        # stats_out routes the call to the narrow kernel whatever the dispatcher would choose by itself
        new, _ = conv_once(x, w, b, B, L, stride=stride, up_to=up_to, want_stats=True, **kw)
        old, _ = conv_once(x, w, b, B, L, stride=stride, up_to=up_to, force=NARROW_OFF, **kw)
        assert torch.isfinite(new.float()).all()
        assert torch.equal(new, old), (L, stride, up_to, list(kw))
    if stride == 2:   # the dispatcher takes the narrow kernel for stride-2 convs by itself
        auto, _ = conv_once(x, w, b, B, L, stride=2)
        forced, _ = conv_once(x, w, b, B, L, stride=2, want_stats=True)
        assert torch.equal(auto, forced)
    # ... and the torch statement of the op (bf16-quantised operands, fp32 accumulation)
    xq, wq = x.to(BF).float(), w.to(BF).float()
    if up_to is not None:
        xq = F.interpolate(xq, size=up_to, mode="nearest")
    ref = F.conv1d(xq, wq, b, stride=stride, padding=1)
    got = unrows(conv_once(x, w, b, B, L, stride=stride, up_to=up_to, want_stats=True)[0])
    assert float((got - ref).abs().max() / ref.abs().max()) < 1.2e-2


@pytest.mark.parametrize("B,L", [(90, 200), (5, 200), (170, 100), (3, 37), (1, 200)])
def test_block_statistics_and_finalize(B, L):
    """stats_out = {sum, sum of squares} per (trajectory, 16-row block, 4-channel granule) of the finished values;
    tdx_groupnorm_finalize turns them into {rstd, -mean * rstd} per (trajectory, group)."""
    engine, _lib = _eng()
    lib = _lib.lib()
    g = torch.Generator().manual_seed(B + L)
    x = torch.randn(B, 128, L, generator=g)
    w = torch.randn(128, 128, 3, generator=g) / 20
    b = torch.randn(128, generator=g)
    res = torch.randn(B, 128, L, generator=g)
    out, stats = conv_once(x, w, b, B, L, res=res, want_stats=True)
    again, stats2 = conv_once(x, w, b, B, L, res=res, want_stats=True)
    assert torch.equal(out, again) and torch.equal(stats, stats2)                  # fixed summation order
    plain, _ = conv_once(x, w, b, B, L, res=res)
    assert torch.equal(out, plain)                                                # emitting statistics does not change the output
    o = out.float().cpu()                                                         # (B, L, 128): the kernel summed the fp32 values before rounding
    nblk = (L + 15) // 16
    pad = torch.zeros(B, nblk * 16, 128)
    pad[:, :L] = o
    blocks = pad.view(B, nblk, 16, 32, 4)
    s1, s2 = blocks.sum(dim=(2, 4)), (blocks * blocks).sum(dim=(2, 4))
    mag = blocks.abs().sum(dim=(2, 4)) + 1e-3
    st = stats.cpu()
    assert torch.isfinite(st).all()
    assert float(((st[..., 0] - s1).abs() / mag).max()) < 8e-3                    # bf16 rounding of the stored values: 2^-9 relative each
    assert float(((st[..., 1] - s2).abs() / (s2 + 1e-3)).max()) < 1.6e-2
    # shard invariance: the statistics of a sub-batch equal the corresponding rows of the full batch bit for bit
    if B >= 4:
        h = B // 2
        _, part = conv_once(x[h:], w, b, B - h, L, res=res[h:], want_stats=True)
        assert torch.equal(part, stats[h:])
    for groups, cpg, eps in ((32, 4, 1e-6), (16, 8, 1e-5), (8, 16, 1e-6)):
        mr = torch.full((B, groups + 3, 2), float("nan"), device="cuda")
        _lib.check(lib.tdx_groupnorm_finalize(stats.data_ptr(), 32, 0, mr.data_ptr(), groups + 3, 2, B, L, groups, cpg, eps, _lib.stream_ptr()))
        torch.cuda.synchronize()
        og = o.view(B, L, groups, cpg).double()
        mean = og.mean(dim=(1, 3))
        var = og.var(dim=(1, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + eps)
        got = mr.cpu().double()
        assert torch.isnan(got[:, :2]).all() and torch.isnan(got[:, groups + 2:]).all()       # only its own columns are written
        assert float(((got[:, 2:groups + 2, 0] - rstd).abs() / rstd).max()) < 1e-2
        assert float((got[:, 2:groups + 2, 1] + mean * rstd).abs().max()) < 2e-2


@pytest.mark.parametrize("B,L,act", [(90, 200, 2), (170, 100, 2), (7, 200, 0), (40, 53, 2)])
def test_norm_on_load_matches_groupnorm_then_conv(B, L, act):
    """conv(act(GroupNorm(x))) with the normalisation applied inside the conv's load path, against (a) the torch statement in
    fp32 and (b) the two-launch dataflow (tdx_groupnorm_act, then the same conv on the normalised tensor)."""
    engine, _lib = _eng()
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3 * B + L)
    raw_in = torch.randn(B, 128, L, generator=g)
    w0 = torch.randn(128, 128, 3, generator=g) / 20
    w = torch.randn(128, 128, 3, generator=g) / 20
    b = torch.randn(128, generator=g)
    eb = torch.randn(B, 136, generator=g)
    gn = torch.nn.GroupNorm(32, 128, eps=1e-6)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(128, generator=g) * 0.5 + 1.0)
        gn.bias.copy_(torch.randn(128, generator=g) * 0.3)
    gn = gn.cuda()
    # producer: a conv that leaves x (bf16 rows) and its block sums
    x_rows, stats = conv_once(raw_in, w0, None, B, L, want_stats=True)
    x = unrows(x_rows)                                       # (B, 128, L) fp32 view of the stored bf16 tensor
    mr = torch.empty((B, 32, 2), device="cuda")
    _lib.check(lib.tdx_groupnorm_finalize(stats.data_ptr(), 32, 0, mr.data_ptr(), 32, 0, B, L, 32, 4, 1e-6, _lib.stream_ptr()))
    fused, _ = conv_once(x, w, b, B, L, ebias=eb, norm=(mr, gn, act))
    # (a) torch, fp32
    with torch.no_grad():
        a = F.group_norm(x, 32, gn.weight.cpu(), gn.bias.cpu(), 1e-6)
        a = swish(a) if act == 2 else a
        ref = F.conv1d(a.to(BF).float(), w.to(BF).float(), b, padding=1) + eb[:, 8:].unsqueeze(2)
    got = unrows(fused)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1.5e-2
    assert float((got - ref).norm() / ref.norm()) < 5e-3
    # (b) two launches: stand-alone GroupNorm kernel (exact two-pass statistics on the bf16 tensor), then the plain conv
    a_rows = torch.empty_like(x_rows)
    _lib.check(lib.tdx_groupnorm_act(x_rows.data_ptr(), 128, a_rows.data_ptr(), 128, 0, gn.weight.data_ptr(), gn.bias.data_ptr(), B, L, 128, 32,
                                     1e-6, act, _lib.TDX_BF16, _lib.stream_ptr()))
    torch.cuda.synchronize()
    two, _ = conv_once(unrows(a_rows), w, b, B, L, ebias=eb)
    d = (fused.float() - two.float()).abs().cpu()
    assert float(d.max() / ref.abs().max()) < 1.5e-2 and float((d > 0).float().mean()) < 0.5     # statistics differ only by summation order
    # deterministic and shard-invariant
    fused2, _ = conv_once(x, w, b, B, L, ebias=eb, norm=(mr, gn, act))
    assert torch.equal(fused, fused2)
    h = B // 2
    part, _ = conv_once(x[h:], w, b, B - h, L, ebias=eb[h:], norm=(mr[h:].contiguous(), gn, act))
    assert torch.equal(part, fused[h:])


def test_norm_fields_are_refused_where_unsupported():
    engine, _lib = _eng()
    pg = engine.Program(torch.device("cuda"), BF)
    pg.fuse_norm = True
    assert engine.norm_fusable_shape(pg, 64, 200, 200, 128, 128)
    assert engine.norm_fusable_shape(pg, 64, 100, 200, 128, 128, norm=False)           # resampled source: statistics only
    assert not engine.norm_fusable_shape(pg, 64, 100, 200, 128, 128, norm=True)
    assert not engine.norm_fusable_shape(pg, 64, 200, 200, 256, 256)
    p32 = engine.Program(torch.device("cuda"), torch.float32)
    p32.fuse_norm = True
    assert not engine.norm_fusable_shape(p32, 64, 200, 200, 128, 128)
    x = torch.zeros(4, 100, 256, device="cuda", dtype=BF)
    w = torch.zeros(256, 768, device="cuda", dtype=BF)
    st = torch.zeros(4, 7, 64, 2, device="cuda")
    with pytest.raises(_lib.TdxError):
        pg.conv([engine.Seg(x, 256, 256, 100)], w, None, torch.empty_like(x), 256, 4, 100, 256, stats_out=st)


@pytest.mark.parametrize("B,L", [(33, 200), (4, 187), (2, 16)])
def test_stem_with_statistics_and_head_with_norm(B, L):
    engine, _lib = _eng()
    lib, st = _lib.lib(), _lib.stream_ptr()
    g = torch.Generator().manual_seed(B * L)
    x = torch.randn(B, 2, L, generator=g).cuda()
    w, b = (torch.randn(128, 2, 3, generator=g) / 2).cuda(), torch.randn(128, generator=g).cuda()
    ref = torch.empty((B, L, 128), device="cuda", dtype=BF)
    _lib.check(lib.tdx_stem_conv(x.data_ptr(), w.data_ptr(), b.data_ptr(), ref.data_ptr(), B, L, 128, _lib.TDX_BF16, st))
    out = torch.full((B, L, 128), float("nan"), device="cuda", dtype=BF)
    nblk = (L + 15) // 16
    stats = torch.full((B, nblk, 32, 2), float("nan"), device="cuda")
    _lib.check(lib.tdx_stem_conv_stats(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(), 32, 0, B, L, 128, st))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    o = out.float().cpu()
    pad = torch.zeros(B, nblk * 16, 128)
    pad[:, :L] = o
    blocks = pad.view(B, nblk, 16, 32, 4)
    s1, s2 = blocks.sum(dim=(2, 4)), (blocks * blocks).sum(dim=(2, 4))
    stc = stats.cpu()
    assert float(((stc[..., 0] - s1).abs() / (blocks.abs().sum(dim=(2, 4)) + 1e-3)).max()) < 8e-3
    assert float(((stc[..., 1] - s2).abs() / (s2 + 1e-3)).max()) < 1.6e-2
    # head: GroupNorm(32, 128, eps 1e-5, no activation) + conv to 2 channels, normalised on load vs the two-kernel dataflow
    gam, bet = (torch.randn(128, generator=g) * 0.5 + 1).cuda(), (torch.randn(128, generator=g) * 0.2).cuda()
    hw, hb = (torch.randn(2, 128, 3, generator=g) / 10).cuda(), torch.randn(2, generator=g).cuda()
    mr = torch.empty((B, 32, 2), device="cuda")
    _lib.check(lib.tdx_groupnorm_finalize(stats.data_ptr(), 32, 0, mr.data_ptr(), 32, 0, B, L, 32, 4, 1e-5, st))
    got = torch.full((B, 2, L), float("nan"), device="cuda")
    _lib.check(lib.tdx_head_conv_norm(out.data_ptr(), mr.data_ptr(), 32, 0, 4, gam.data_ptr(), bet.data_ptr(), hw.data_ptr(), hb.data_ptr(),
                                      got.data_ptr(), B, L, 128, st))
    a = torch.empty_like(out)
    _lib.check(lib.tdx_groupnorm_act(out.data_ptr(), 128, a.data_ptr(), 128, 0, gam.data_ptr(), bet.data_ptr(), B, L, 128, 32, 1e-5, 0, _lib.TDX_BF16, st))
    two = torch.empty((B, 2, L), device="cuda")
    _lib.check(lib.tdx_head_conv(a.data_ptr(), hw.data_ptr(), hb.data_ptr(), two.data_ptr(), B, L, 128, _lib.TDX_BF16, st))
    torch.cuda.synchronize()
    with torch.no_grad():
        an = F.group_norm(o.permute(0, 2, 1), 32, gam.cpu(), bet.cpu(), 1e-5)
        ref_h = F.conv1d(an, hw.cpu(), hb.cpu(), padding=1)
    assert float((got.cpu() - ref_h).abs().max() / ref_h.abs().max()) < 1e-2
    assert float((got - two).abs().max() / ref_h.abs().max()) < 1e-2


def test_trajunet_forward_with_split_groupnorm_dataflow(golden_dir, monkeypatch):
    """Whole TrajUNet forward (bf16) with the first encoder level and the head on the split GroupNorm dataflow (statistics from
    the producers, normalise-on-load in the consumers: engine.FUSE_NORM_DEFAULT) against the reference's golden output, the
    default dataflow, and itself on a batch shard."""
    import json
    import os
    import numpy as np
    from _synth import FULL_SCHEDULE, synth_inputs, synth_state_dict
    from trajectorydiffusion_amd import TrajUNet, engine
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    m = TrajUNet(FULL_SCHEDULE, diffusion_steps=500, res_blocks=2)
    m.load_state_dict(synth_state_dict([(k, s) for k, s in spec], seed=0), strict=True)
    m = m.cuda().eval().set_compute_dtype(BF)
    g = np.load(os.path.join(golden_dir, "forward.npz"))
    x, t, cat, num = synth_inputs(4, 200, 500, seed=1)
    args = (x.cuda(), t.cuda(), cat.cuda(), num.cuda())
    base = m(*args).clone()
    monkeypatch.setattr(engine, "FUSE_NORM_DEFAULT", True)
    m._programs.clear()
    fused = m(*args).clone()
    prog = m.program(4, 200, torch.device("cuda", 0))
    names = [n for n, _, _ in prog.pg.ops]
    assert "stem.norm" in names and "head.norm" in names and "down_blocks.0.0.gn1" not in names and "down_blocks.0.1.conv1.gn" not in names
    ref = torch.as_tensor(g["B4_L200/eps"])
    assert float((fused.cpu() - ref).norm() / ref.norm()) < 1e-2 and float((fused.cpu() - ref).abs().max() / ref.abs().max()) < 3e-2
    assert float((fused - base).abs().max() / ref.abs().max()) < 2e-2
    part = m(*(a[2:] for a in args))
    assert torch.equal(part, fused[2:])                               # batch shards reproduce the unsharded batch bit for bit
    m._programs.clear()
