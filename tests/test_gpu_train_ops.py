"""GPU: training-step building blocks through the C ABI against torch autograd / torch.optim on the CPU."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _eng():
    from trajectorydiffusion_amd import _lib, engine
    return engine, _lib


def q(x, dt):
    return x.to(dt).float()


def rows(x, dt):  # (B, C, L) -> channels-last rows (B, L, C) on the GPU
    return x.permute(0, 2, 1).contiguous().to("cuda", dt)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def workspace(n=None):
    _, _lib = _eng()
    return torch.empty(int(n or _lib.lib().tdx_train_workspace_floats()), device="cuda", dtype=torch.float32)


def test_mse_loss_and_gradient():
    _, _lib = _eng()
    g = torch.Generator().manual_seed(0)
    for n in (5, 1600, 2 * 200 * 1024):
        pred, tgt = torch.randn(n, generator=g), torch.randn(n, generator=g)
        p = pred.clone().requires_grad_(True)
        loss = F.mse_loss(p, tgt)
        loss.backward()
        pd, td = pred.cuda(), tgt.cuda()
        dp, lo, ws = torch.empty_like(pd), torch.zeros(1, device="cuda"), workspace()
        _lib.check(_lib.lib().tdx_mse_loss_grad(pd.data_ptr(), td.data_ptr(), dp.data_ptr(), lo.data_ptr(), n, ws.data_ptr(), _lib.stream_ptr()))
        assert abs(float(lo) - float(loss.detach())) < 2e-6 * float(loss.detach())
        assert rel(dp.cpu(), p.grad) < 1e-6
        lo2 = torch.zeros(1, device="cuda")
        _lib.check(_lib.lib().tdx_mse_loss_grad(pd.data_ptr(), td.data_ptr(), dp.data_ptr(), lo2.data_ptr(), n, ws.data_ptr(), _lib.stream_ptr()))
        assert torch.equal(lo, lo2)  # fixed-order reduction


def test_adamw_matches_torch_optim():
    _, _lib = _eng()
    g = torch.Generator().manual_seed(1)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)   # train.ipynb:131
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (0.1 if step % 2 else 3.0)
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        _lib.check(_lib.lib().tdx_adamw_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 5e-4, 0.9, 0.999, 1e-8, 0.01, step,
                                             _lib.stream_ptr()))
        assert rel(p.cpu(), ref.detach()) < 1e-6, step
    st = opt.state[ref]
    assert rel(m.cpu(), st["exp_avg"]) < 1e-6 and rel(v.cpu(), st["exp_avg_sq"]) < 1e-6


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n,c", [(1, 8), (100, 4), (300, 128), (5000, 64), (33431, 384), (1024, 2048), (4096 * 13, 1024), (1024 * 200, 128)])
def test_colsum(dt, n, c):
    _, _lib = _eng()
    x = torch.randn(n, c + 8, generator=torch.Generator().manual_seed(n))
    xd = x.to("cuda", dt)
    out, ws = torch.full((c,), 7.0, device="cuda"), workspace()
    for acc in (0, 1):
        _lib.check(_lib.lib().tdx_colsum(xd.data_ptr() + 8 * xd.element_size(), c + 8, n, c, out.data_ptr(), acc, _lib.dtype_code(dt),
                                         ws.data_ptr(), _lib.stream_ptr()))
        ref = (acc + 1) * q(x, dt)[:, 8:].double().sum(0).float()
        assert rel(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("n,c", [(1024, 128), (1024, 1024), (37, 32), (4096, 2048), (100, 6)])
def test_colsum_pair_is_two_colsums(n, c):
    _, _lib = _eng()
    lib, st = _lib.lib(), _lib.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(n + c)
    xa, xb = torch.randn(n, c, device="cuda", generator=g), torch.randn(n, c, device="cuda", generator=g)
    ws = workspace()
    for acc in (0, 1):
        oa, ob = torch.full((c,), 3.0, device="cuda"), torch.full((c,), -1.0, device="cuda")
        pa, pb = oa.clone(), ob.clone()
        _lib.check(lib.tdx_colsum(xa.data_ptr(), c, n, c, oa.data_ptr(), acc, _lib.TDX_F32, ws.data_ptr(), st))
        _lib.check(lib.tdx_colsum(xb.data_ptr(), c, n, c, ob.data_ptr(), acc, _lib.TDX_F32, ws.data_ptr(), st))
        _lib.check(lib.tdx_colsum_pair(xa.data_ptr(), xb.data_ptr(), c, n, c, pa.data_ptr(), pb.data_ptr(), acc, _lib.TDX_F32, ws.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(oa, pa) and torch.equal(ob, pb)
        assert rel(pa.cpu().double(), xa.cpu().double().sum(0) + (3.0 if acc else 0.0)) < 1e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("case", ["k3", "k3_s2", "k3_up", "k1", "concat_half", "ragged", "big_k3", "big_up", "big_k1", "big_concat", "big_short",
                                  "big_one_row", "big_s2", "big_s2_odd"])
def test_conv_weight_gradient(dt, case):
    """dW of nn.Conv1d from autograd, for every source map the forward conv supports."""
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(len(case))
    B, L, Ci, Co, k, stride, up, ci_off, cin_total = 37, 50, 128, 192, 3, 1, None, 0, None
    if case == "k3_s2":
        stride = 2
    elif case == "k3_up":
        L, up = 25, 50
    elif case == "k1":
        k = 1
    elif case == "concat_half":
        ci_off, cin_total = 64, 128 + 64 + 32   # this source is channels 64..191 of a wider (concatenated) conv input
    elif case == "ragged":
        B, L, Ci, Co = 5, 13, 40, 72
    elif case == "big_k3":          # big_*: shapes the 128 x 128 x all-taps kernel takes (fp32 only; bf16 stays on the small-tile kernel)
        Ci, Co = 128, 256
    elif case == "big_up":
        L, up, Ci, Co = 25, 50, 256, 128
    elif case == "big_k1":
        k, Ci, Co = 1, 256, 256
    elif case == "big_concat":
        Ci, Co, ci_off, cin_total = 128, 128, 128, 384
    elif case == "big_short":       # several trajectory boundaries inside every 32-row chunk
        B, L, Ci, Co = 70, 13, 128, 128
    elif case == "big_s2":          # stride 2, even length: the even / odd row window
        stride, Ci, Co = 2, 128, 256
    elif case == "big_s2_odd":      # odd length (25 -> 13): the last output row has no odd source row
        stride, B, L, Ci, Co = 2, 41, 25, 256, 128
    elif case == "big_one_row":     # the embedding MLP's linears: lout = 1, every row is first and last of its trajectory
        B, L, Ci, Co, k = 100, 1, 128, 256, 1
    cin_total = cin_total or Ci
    x = torch.randn(B, Ci, L, generator=g)
    xq = q(x, dt)
    xin = F.interpolate(xq, size=up, mode="nearest") if up else xq
    w = torch.randn(Co, Ci, k, generator=g, requires_grad=True)
    y = F.conv1d(xin, w, stride=stride, padding=k // 2)
    dy = torch.randn(y.shape, generator=g)
    dyq = q(dy, dt)
    y.backward(dyq)
    lout = y.shape[2]
    src, dyr = rows(x, dt), rows(dy, dt)
    seg = _lib.KSeg(src.data_ptr(), float(L) / up if up else 0.0, 0, Ci, 0, Ci, k, stride, L, up or L, 0)
    dw = torch.full((Co, cin_total, k), 0.5, device="cuda")
    ws = workspace((512 if case.startswith("big") else 64) * Co * Ci * k)
    for acc in (0, 1):
        _lib.check(_lib.lib().tdx_conv_wgrad(dyr.data_ptr(), Co, 0, Co, ctypes.byref(seg), B, lout, dw.data_ptr(), cin_total, ci_off, acc,
                                             _lib.dtype_code(dt), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        got = dw.cpu()[:, ci_off:ci_off + Ci, :]
        assert rel(got, (acc + 1) * w.grad) < 2e-5, acc
    untouched = torch.ones(cin_total, dtype=torch.bool)
    untouched[ci_off:ci_off + Ci] = False
    assert bool((dw.cpu()[:, untouched, :] == 0.5).all())
    if case.startswith("big") and dt == torch.float32:     # and bit-for-bit repeatable; the small-tile kernel agrees to rounding
        again = torch.empty_like(dw)
        for force, tol in ((0, 0.0), (1, 2e-5)):
            old = _lib.lib().tdx_conv_wgrad_debug(force)
            _lib.check(_lib.lib().tdx_conv_wgrad(dyr.data_ptr(), Co, 0, Co, ctypes.byref(seg), B, lout, again.data_ptr(), cin_total, ci_off, 0,
                                                 _lib.dtype_code(dt), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
            _lib.lib().tdx_conv_wgrad_debug(old)
            a, b = again.cpu()[:, ci_off:ci_off + Ci, :], dw.cpu()[:, ci_off:ci_off + Ci, :] / 2
            assert rel(a, b) <= tol, force
    small = workspace(Co * Ci * k)   # room for a single slab: still correct
    _lib.check(_lib.lib().tdx_conv_wgrad(dyr.data_ptr(), Co, 0, Co, ctypes.byref(seg), B, lout, dw.data_ptr(), cin_total, ci_off, 0,
                                         _lib.dtype_code(dt), small.data_ptr(), small.numel(), _lib.stream_ptr()))
    assert rel(dw.cpu()[:, ci_off:ci_off + Ci, :], w.grad) < 2e-5
    with pytest.raises(_lib.TdxError):
        _lib.check(_lib.lib().tdx_conv_wgrad(dyr.data_ptr(), Co, 0, Co, ctypes.byref(seg), B, lout, dw.data_ptr(), cin_total, ci_off, 0,
                                             _lib.dtype_code(dt), small.data_ptr(), 16, _lib.stream_ptr()))


@pytest.mark.parametrize("B,L,cin,cout,taps", [(8, 200, 128, 128, 3), (64, 25, 256, 384, 3), (16, 13, 128, 256, 1), (4, 50, 64, 96, 3), (3, 7, 128, 128, 3)])
def test_conv_weight_gradient_with_bias_gradient(B, L, cin, cout, taps):
    """tdx_conv_wgrad_bias: dW bit-identical to tdx_conv_wgrad, dbias = column sums of dY (tdx_colsum's value up to the order of the sum),
    accumulate honoured; large-tile shapes take the sums from the weight-gradient kernel, the others from tdx_colsum."""
    _, _lib = _eng()
    lib, st = _lib.lib(), _lib.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(cin + L)
    x = torch.randn(B, L, cin, device="cuda", generator=g)
    dy = torch.randn(B, L, cout + 8, device="cuda", generator=g)
    k = _lib.KSeg(x.data_ptr(), 0.0, 0, cin, 0, cin, taps, 1, L, L, 0)
    ws = workspace(64 << 20)
    dw0, dw1 = torch.zeros(cout, cin * taps, device="cuda"), torch.zeros(cout, cin * taps, device="cuda")
    db = torch.full((cout,), 2.0, device="cuda")
    _lib.check(lib.tdx_conv_wgrad(dy.data_ptr(), cout + 8, 8, cout, ctypes.byref(k), B, L, dw0.data_ptr(), cin, 0, 0, _lib.TDX_F32, ws.data_ptr(), ws.numel(), st))
    _lib.check(lib.tdx_conv_wgrad_bias(dy.data_ptr(), cout + 8, 8, cout, ctypes.byref(k), B, L, dw1.data_ptr(), cin, 0, 0, db.data_ptr(), 1, _lib.TDX_F32,
                                       ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1)
    want = dy[:, :, 8:].double().sum(dim=(0, 1)) + 2.0
    assert float((db.double() - want).abs().max()) < 1e-4 * float(want.abs().max())
    db2 = torch.empty(cout, device="cuda")
    _lib.check(lib.tdx_conv_wgrad_bias(dy.data_ptr(), cout + 8, 8, cout, ctypes.byref(k), B, L, dw1.data_ptr(), cin, 0, 0, db2.data_ptr(), 0, _lib.TDX_F32,
                                       ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
    assert float((db2.double() - (want - 2.0)).abs().max()) < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_data_gradient_is_the_forward_kernel_on_flipped_weights(dt, k):
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(k)
    B, L, Ci, Co = 130, 25, 128, 256
    x = torch.randn(B, Ci, L, generator=g, requires_grad=True)
    w = torch.randn(Co, Ci, k, generator=g) / (k * Ci) ** 0.5
    y = F.conv1d(x, q(w, dt), padding=k // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(q(dy, dt))
    pg = engine.Program(torch.device("cuda"), dt)
    dx = torch.empty((B, L, Ci), device="cuda", dtype=dt)
    pg.conv([engine.Seg(rows(dy, dt), Co, Co, L, taps=k)], engine.pack_conv_weight_dgrad(w.cuda(), dt), None, dx, Ci, B, L, Ci)
    pg.run()
    torch.cuda.synchronize()
    got = dx.float().cpu().permute(0, 2, 1)
    assert rel(got, x.grad) < (1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,L,C,act", [(3, 200, 128, 2), (2, 25, 1024, 2), (5, 13, 256, 0), (2, 7, 2048, 2)])
def test_groupnorm_swish_backward(dt, B, L, C, act):
    _, _lib = _eng()
    g = torch.Generator().manual_seed(L)
    x = torch.randn(B, C, L, generator=g) * 2 + 0.3
    xq = q(x, dt).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    u = F.group_norm(xq, 32, gamma, beta, 1e-5)
    y = u * torch.sigmoid(u) if act == 2 else u
    dy = torch.randn(y.shape, generator=g)
    y.backward(q(dy, dt))
    xr, dyr = rows(x, dt), rows(dy, dt)
    dx = torch.zeros((B, L, C + 16), device="cuda", dtype=dt)
    dgr, dbr = torch.empty((B, C), device="cuda"), torch.empty((B, C), device="cuda")
    gd, bd = gamma.detach().cuda(), beta.detach().cuda()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    for acc in (0, 1):
        _lib.check(_lib.lib().tdx_groupnorm_act_bwd(xr.data_ptr(), C, dyr.data_ptr(), C, 0, dx.data_ptr(), C + 16, 16, gd.data_ptr(), bd.data_ptr(),
                                                    dgr.data_ptr(), dbr.data_ptr(), B, L, C, 32, 1e-5, act, acc, _lib.dtype_code(dt), _lib.stream_ptr()))
        torch.cuda.synchronize()
        got = dx.float().cpu()[:, :, 16:].permute(0, 2, 1)
        assert rel(got, (acc + 1) * xq.grad) < tol, acc
    assert float(dx[:, :, :16].abs().max()) == 0.0
    ws = workspace()
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    for src, dst in ((dgr, dgam), (dbr, dbet)):
        _lib.check(_lib.lib().tdx_colsum(src.data_ptr(), C, B, C, dst.data_ptr(), 0, _lib.TDX_F32, ws.data_ptr(), _lib.stream_ptr()))
    assert rel(dgam.cpu(), gamma.grad) < 5e-5 and rel(dbet.cpu(), beta.grad) < 5e-5


@pytest.mark.parametrize("B,L,C", [(3, 200, 128), (2, 100, 256), (5, 64, 128), (2, 300, 128), (3, 13, 1024), (2, 3, 2048)])
def test_groupnorm_backward_register_and_rereading_forms_agree(B, L, C):
    """Rows held in registers (split over 2 / 4 waves from L > 64) against the one-wave form that re-reads its rows (test hook bit 8):
    the same arithmetic; only the association of the sums over the rows differs when a trajectory is split."""
    _, _lib = _eng()
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(L)
    x = torch.randn(B, L, C, device="cuda", generator=g) * 2 + 0.3
    dy = torch.randn(B, L, C, device="cuda", generator=g)
    gam, bet = 1 + 0.2 * torch.randn(C, device="cuda", generator=g), 0.2 * torch.randn(C, device="cuda", generator=g)
    out = []
    for force in (0, 8):
        old = lib.tdx_conv_wgrad_debug(force)
        try:
            dx = torch.randn(B, L, C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
            r0, r1 = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
            _lib.check(lib.tdx_groupnorm_act_bwd(x.data_ptr(), C, dy.data_ptr(), C, 0, dx.data_ptr(), C, 0, gam.data_ptr(), bet.data_ptr(), r0.data_ptr(),
                                                 r1.data_ptr(), B, L, C, 32, 1e-6, 2, 1, _lib.TDX_F32, _lib.stream_ptr()))
            torch.cuda.synchronize()
            out.append((dx, r0, r1))
        finally:
            lib.tdx_conv_wgrad_debug(old)
    for a, b in zip(*out):
        assert rel(a, b) < 2e-6
    if L <= 64 or L == 300:   # one wave per window either way (300 rows: too long for the register form): identical order
        assert all(torch.equal(a, b) for a, b in zip(*out))


@pytest.mark.parametrize("B,L,C,p", [(3, 200, 128, 0.1), (4, 25, 1024, 0.1), (2, 100, 256, 0.5), (5, 13, 1024, 0.0)])
def test_groupnorm_dropout_in_one_pass_equals_the_two_passes(B, L, C, p):
    """tdx_groupnorm_act_dropout == tdx_groupnorm_act then tdx_dropout; tdx_groupnorm_act_dropout_bwd == tdx_dropout on the gradient then
    tdx_groupnorm_act_bwd: bit for bit (same mask, same arithmetic)."""
    _, _lib = _eng()
    lib, st = _lib.lib(), _lib.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(L)
    x = torch.randn(B, L, C, device="cuda", generator=g) * 2 + 0.3
    dy = torch.randn(B, L, C, device="cuda", generator=g)
    gam, bet = 1 + 0.2 * torch.randn(C, device="cuda", generator=g), 0.2 * torch.randn(C, device="cuda", generator=g)
    ctr = torch.tensor([7, 0, 0, 0], dtype=torch.int32, device="cuda")
    seed, layer = 0x1234_5678_9ABC_DEF0, 5
    y2, y1 = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.tdx_groupnorm_act(x.data_ptr(), C, y2.data_ptr(), C, 0, gam.data_ptr(), bet.data_ptr(), B, L, C, 32, 1e-6, 2, _lib.TDX_F32, st))
    _lib.check(lib.tdx_dropout(y2.data_ptr(), C, B * L, C, p, seed, layer, ctr.data_ptr(), st))
    _lib.check(lib.tdx_groupnorm_act_dropout(x.data_ptr(), C, y1.data_ptr(), C, 0, gam.data_ptr(), bet.data_ptr(), B, L, C, 32, 1e-6, 2, _lib.TDX_F32, p, seed,
                                             layer, ctr.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    if p > 0:
        zero = float((y1 == 0).float().mean())
        assert abs(zero - p) < 0.01
    res = []
    for fused in (False, True):
        dx = torch.ones(B, L, C, device="cuda")
        r0, r1 = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
        args = (x.data_ptr(), C, None, C, 0, dx.data_ptr(), C, 0, gam.data_ptr(), bet.data_ptr(), r0.data_ptr(), r1.data_ptr(), B, L, C, 32, 1e-6, 2, 1, _lib.TDX_F32)
        if fused:
            a = list(args)
            a[2] = dy.data_ptr()
            _lib.check(lib.tdx_groupnorm_act_dropout_bwd(*a, p, seed, layer, ctr.data_ptr(), st))
        else:
            dyd = dy.clone()
            _lib.check(lib.tdx_dropout(dyd.data_ptr(), C, B * L, C, p, seed, layer, ctr.data_ptr(), st))
            a = list(args)
            a[2] = dyd.data_ptr()
            _lib.check(lib.tdx_groupnorm_act_bwd(*a, st))
        torch.cuda.synchronize()
        res.append((dx, r0, r1))
    assert all(torch.equal(a, b) for a, b in zip(*res))
    # shapes the fused forms refuse: the mask is keyed by the element index of a plain tensor
    wide = torch.empty(B, L, C + 16, device="cuda")
    assert lib.tdx_groupnorm_act_dropout(x.data_ptr(), C, wide.data_ptr(), C + 16, 0, gam.data_ptr(), bet.data_ptr(), B, L, C, 32, 1e-6, 2, _lib.TDX_F32, p, seed,
                                         layer, ctr.data_ptr(), st) != 0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("L", [50, 25, 200])
def test_conv_data_gradient_stride2(dt, L):
    """Stride-2 down conv (TrajUNet.py:44): two forward-kernel launches on dy (one per input-row parity) + tdx_interleave2."""
    engine, _lib = _eng()
    g = torch.Generator().manual_seed(L)
    B, Ci, Co = 40, 128, 128
    x = torch.randn(B, Ci, L, generator=g, requires_grad=True)
    w = torch.randn(Co, Ci, 3, generator=g) / (3 * Ci) ** 0.5
    y = F.conv1d(x, q(w, dt), stride=2, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(q(dy, dt))
    we, wo = engine.pack_conv_weight_dgrad_stride2(w.cuda(), dt)
    tol = 1e-5 if dt == torch.float32 else 1e-2
    dx = torch.empty((B, L, Ci), device="cuda", dtype=dt)
    for acc in (False, True):
        pg = engine.Program(torch.device("cuda"), dt)
        engine.conv_dgrad_stride2(pg, rows(dy, dt), we, wo, dx, B, L, Co, Ci, accumulate=acc)
        pg.run()
        torch.cuda.synchronize()
        assert rel(dx.float().cpu().permute(0, 2, 1), (2 if acc else 1) * x.grad) < (tol if not acc else 2 * tol)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("li,lo", [(13, 25), (25, 50), (100, 200), (12, 24), (24, 47), (7, 7)])
def test_upsample_fold_is_the_gradient_of_nearest_interpolate(dt, li, lo):
    _, _lib = _eng()
    g = torch.Generator().manual_seed(li)
    B, C = 3, 64
    x = torch.randn(B, C, li, generator=g, requires_grad=True)
    up = F.interpolate(x, size=lo, mode="nearest")
    du = torch.randn(up.shape, generator=g)
    up.backward(q(du, dt))
    dud = rows(du, dt)
    dx = torch.zeros((B, li, C + 8), device="cuda", dtype=dt)
    for acc in (0, 1):
        _lib.check(_lib.lib().tdx_upsample_fold(dud.data_ptr(), C, 0, dx.data_ptr(), C + 8, 8, B, li, lo, C, acc, _lib.dtype_code(dt), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert rel(dx.float().cpu()[:, :, 8:].permute(0, 2, 1), (acc + 1) * x.grad) < (1e-6 if dt == torch.float32 else 2e-2)
    assert float(dx[:, :, :8].abs().max()) == 0.0


def test_small_backward_pieces():
    """Activation fwd/bwd, per-trajectory row sums, nn.Embedding backward, attention backward against autograd (fp32)."""
    _, _lib = _eng()
    lib, st = _lib.lib(), _lib.stream_ptr()
    g = torch.Generator().manual_seed(3)
    # swish / leaky
    z = torch.randn(37, 256, generator=g, requires_grad=True)
    dy = torch.randn(37, 256, generator=g)
    for act, fn in ((2, lambda v: v * torch.sigmoid(v)), (1, lambda v: F.leaky_relu(v, 0.1))):
        z.grad = None
        y = fn(z)
        y.backward(dy)
        zd, dyd = z.detach().cuda(), dy.cuda()
        yd, dz = torch.empty_like(zd), torch.ones_like(zd)
        _lib.check(lib.tdx_act_fwd(zd.data_ptr(), 256, yd.data_ptr(), 256, 37, 256, act, st))
        assert rel(yd.cpu(), y.detach()) < 2e-6
        ref = zd if act == 2 else yd          # leaky: from the output
        _lib.check(lib.tdx_act_bwd(dyd.data_ptr(), 256, ref.data_ptr(), 256, dz.data_ptr(), 256, 37, 256, act, 1, st))
        assert rel(dz.cpu() - 1, z.grad) < 5e-6
    # per-trajectory row sums
    x = torch.randn(5, 13, 40, generator=g)
    out = torch.zeros(5, 48, device="cuda")
    xd = x.cuda()  # (device tensors stay referenced while the asynchronous launches read them)
    _lib.check(lib.tdx_traj_rowsum(xd.data_ptr(), 40, out.data_ptr() + 8 * 4, 48, 5, 13, 40, 0, st))
    assert rel(out.cpu()[:, 8:], x.sum(1)) < 1e-6 and float(out[:, :8].abs().max()) == 0.0
    # embedding backward
    table = torch.randn(33, 16, generator=g, requires_grad=True)
    idx = torch.randint(0, 33, (50, 4), generator=g)
    dvec = torch.randn(50, 4 * 16 + 8, generator=g)
    F.embedding(idx, table).flatten(1).backward(dvec[:, :64])
    dt_ = torch.empty(33, 16, device="cuda")
    idxd, dvecd = idx.cuda(), dvec.cuda()
    _lib.check(lib.tdx_embedding_bwd(idxd.data_ptr(), 4, 4, dvecd.data_ptr(), 72, dt_.data_ptr(), 33, 16, 50, 0, st))
    assert rel(dt_.cpu(), table.grad) < 1e-6
    # attention backward
    for B, L, C in ((3, 13, 1024), (2, 7, 64), (1, 16, 128)):
        qkv = torch.randn(B, L, 3 * C, generator=g, requires_grad=True)
        qv, kv, vv = torch.split(qkv, C, dim=2)
        o = torch.bmm(torch.softmax(torch.bmm(qv, kv.transpose(1, 2)) / C ** 0.5, dim=2), vv)
        do = torch.randn(o.shape, generator=g)
        o.backward(do)
        dq = torch.empty(B, L, 3 * C, device="cuda")
        qkvd, dod = qkv.detach().cuda(), do.cuda()
        _lib.check(lib.tdx_attention_bwd(qkvd.data_ptr(), dod.data_ptr(), dq.data_ptr(), B, L, C, st))
        assert rel(dq.cpu(), qkv.grad) < 2e-5, (B, L, C)
