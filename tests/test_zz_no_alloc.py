"""GPU: the one entry point that allocates nothing (tdx_mse_loss_grad_static: scratch in a file-scope device array) against the
existing entry point it is a variant of (tdx_mse_loss_grad, caller workspace).  Collected last on purpose."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, _lib, pred, target, want_grad=True):
    n = pred.numel()
    st = _lib.stream_ptr()
    ws = torch.empty(int(lib.tdx_train_workspace_floats()), device="cuda")
    d_ref, l_ref = torch.full_like(pred, float("nan")), torch.full((1,), float("nan"), device="cuda")
    d_new, l_new = torch.full_like(pred, float("nan")), torch.full((1,), float("nan"), device="cuda")
    _lib.check(lib.tdx_mse_loss_grad(pred.data_ptr(), target.data_ptr(), d_ref.data_ptr() if want_grad else None, l_ref.data_ptr(), n, ws.data_ptr(), st))
    _lib.check(lib.tdx_mse_loss_grad_static(pred.data_ptr(), target.data_ptr(), d_new.data_ptr() if want_grad else None, l_new.data_ptr(), n, st))
    torch.cuda.synchronize()
    return d_ref, l_ref, d_new, l_new


def test_static_scratch_entry_point_matches_the_workspace_one_twice_in_a_row():
    from trajectorydiffusion_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(123)
    # first call: more elements than 1024 blocks x 256 threads (every partial slot used); second: smaller (39 blocks), so the
    # scratch still holds 985 stale partials of the first call that must not leak into the sum
    for n in (1024 * 256 * 3 + 77, 9999):
        pred, target = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
        d_ref, l_ref, d_new, l_new = _both(lib, _lib, pred, target)
        assert torch.equal(l_new, l_ref) and torch.equal(d_new, d_ref)           # same kernels, same grid: bit-exact
        want = torch.nn.functional.mse_loss(pred.double(), target.double())
        assert abs(float(l_new) - float(want)) < 1e-6 * float(want)
        assert torch.allclose(d_new, 2 * (pred - target) / n, rtol=1e-6, atol=0)
    # loss only (dpred = NULL), and the size limit
    pred, target = torch.randn(4096, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
    _, l_ref, _, l_new = _both(lib, _lib, pred, target, want_grad=False)
    assert torch.equal(l_new, l_ref)
    out = torch.zeros(1, device="cuda")
    rc = lib.tdx_mse_loss_grad_static(pred.data_ptr(), target.data_ptr(), None, out.data_ptr(), (1 << 28) + 1, _lib.stream_ptr())
    assert rc == 1 and b"exceeds" in lib.tdx_last_error()
    assert lib.tdx_mse_loss_grad_static(None, target.data_ptr(), None, out.data_ptr(), 16, _lib.stream_ptr()) == 1
