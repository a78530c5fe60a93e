"""Evaluation metrics (eval_trajunet.ipynb cell 12) on the HIP kernels against the CPU restatement oracle/metrics_oracle.py.
PARITY UNPINNED by the reference (no fixtures; see the oracle's header): these tests pin the kernels to the restatement."""
import numpy as np
import pytest
import torch

from _synth import synth_metric_trajectories

pytestmark = pytest.mark.gpu


def _m():
    from trajectorydiffusion_amd import metrics
    return metrics


@pytest.mark.parametrize("B,L,seed", [(64, 200, 1), (37, 50, 2), (1, 2, 3), (4096, 200, 4)])
def test_point_start_end_distributions_are_bit_exact(B, L, seed):
    from oracle import metrics_oracle as mo
    m = _m()
    t0, t1 = synth_metric_trajectories(B, L, seed)
    for t in (t0, t1):
        d = t.cuda()
        assert np.array_equal(m.getPointDistribution(d).cpu().numpy(), mo.point_distribution(t))
        assert np.array_equal(m.getPointDistribution(d, n_grids=7, normalize=False).cpu().numpy(), mo.point_distribution(t, 7, False))
        assert np.array_equal(m.getStartDistribution(d).cpu().numpy(), mo.start_distribution(t))
        assert np.array_equal(m.getEndDistribution(d).cpu().numpy(), mo.end_distribution(t))
    raw = m.getPointDistribution(t0.cuda(), normalize=False)
    assert float(raw.sum()) == B * L                          # every point lands in exactly one cell


@pytest.mark.parametrize("B,L,seed", [(64, 200, 1), (37, 50, 2), (4096, 200, 4)])
def test_length_distribution(B, L, seed):
    from oracle import metrics_oracle as mo
    m = _m()
    t0, _ = synth_metric_trajectories(B, L, seed)
    dist, lengths = m.getLengthDistribution(t0.cuda(), return_lengths=True)
    ref_len = mo.path_lengths(t0)
    assert np.allclose(lengths.cpu().numpy(), ref_len, rtol=2e-6, atol=0)       # fp32 sums in a different order
    bars = mo.linspace_f32(0.0, 10.0, 64)
    clear = np.abs(ref_len[:, None] - bars[None, :]).min(axis=1) > 1e-4        # lengths not within rounding of a bin edge
    got, ref = dist.cpu().numpy(), mo.length_distribution(t0)
    if clear.all():
        assert np.array_equal(got, ref)
    else:
        assert np.abs(got - ref).max() <= (~clear).sum() / B + 1e-7
    assert abs(float(dist.sum()) - (B + 6.4) / B) < 1e-4                      # (count + 0.1) / sum over 64 bins
    far = t0.clone()
    far[0, 0, :] = torch.arange(L) * 1.0                                       # one path of length > 10
    with pytest.raises(IndexError):
        m.getLengthDistribution(far.cuda())
    with pytest.raises(IndexError):
        mo.length_distribution(far)


def test_jsd_matches_restatement_and_is_zero_on_identical_sets():
    from oracle import metrics_oracle as mo
    m = _m()
    t0, t1 = synth_metric_trajectories(512, 200, 9)
    for fn, ofn in ((m.getPointDistribution, mo.point_distribution), (m.getStartDistribution, mo.start_distribution),
                    (m.getEndDistribution, mo.end_distribution), (m.getLengthDistribution, mo.length_distribution)):
        P, Q = fn(t0.cuda()), fn(t1.cuda())
        ref = mo.jsd(ofn(t0), ofn(t1))
        got = m.JSD(P, Q)
        assert got > 0 and abs(got - ref) <= 1e-5 * abs(ref) + 1e-9, (got, ref)
        assert abs(m.JSD(P, P)) < 1e-9
        assert abs(m.JSD(P, Q) - m.JSD(Q, P)) <= 1e-6 * got      # symmetric


def test_metrics_reject_cpu_tensors_and_bad_shapes():
    from trajectorydiffusion_amd._lib import TdxError
    m = _m()
    t0, _ = synth_metric_trajectories(4, 10, 5)
    with pytest.raises(TdxError):
        m.getPointDistribution(t0)
    with pytest.raises(TdxError):
        m.getLengthDistribution(t0.cuda()[:, :1, :])
    with pytest.raises(TdxError):
        m.JSD(torch.ones(4, 4).cuda(), torch.ones(4, 5).cuda())
    with pytest.raises(TdxError):
        m.getPointDistribution(t0.cuda(), n_grids=1)
