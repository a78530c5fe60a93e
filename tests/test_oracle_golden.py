"""CPU: the oracle restatement against the golden vectors produced by the reference itself."""
import json
import os

import numpy as np
import torch

from _synth import FULL_SCHEDULE, sample_view, synth_inputs, synth_noise, synth_raw_trajectories, synth_state_dict
from oracle import DiffusionOracle, didi_item, gaussian_cell_boundaries, nearest_index_map, trajunet_forward

T_MODEL = 500


def _sd(golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    return synth_state_dict([(k, s) for k, s in spec], seed=0)


def test_schedules(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    for tag, args in {"T500": (1e-4, 0.05, 500, False), "T50": (1e-4, 0.05, 50, False),
                      "default": (1e-4, 0.002, 100, False), "scaled": (1e-4, 0.02, 300, True)}.items():
        o = DiffusionOracle(*args)
        for k in ("betas", "alphas", "alpha_bars", "sqrt_alpha_bars", "sqrt_1_minus_alphas_bars"):
            np.testing.assert_array_equal(getattr(o, k).numpy(), g[f"{tag}/{k}"])  # bit-exact: same fp32 op order
    assert abs(float(DiffusionOracle(1e-4, 0.05, 500).alpha_bars[499]) - 2.9334e-06) < 1e-9


def test_forward_and_taps(golden_dir):
    g = np.load(os.path.join(golden_dir, "forward.npz"))
    sd = _sd(golden_dir)
    for tag, (B, L) in {"B4_L200": (4, 200), "B3_L187": (3, 187), "B2_L64": (2, 64)}.items():
        x, t, cat, num = synth_inputs(B, L, T_MODEL, seed=1)
        taps = {}
        with torch.no_grad():
            eps = trajunet_forward(sd, x, t, cat, num, FULL_SCHEDULE, T_MODEL, taps=taps)
        ref = g[f"{tag}/eps"]
        assert np.abs(eps.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
        for k in ("embed", "stem", "down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3"):
            r = g[f"{tag}/tap/{k}"]
            assert tuple(g[f"{tag}/tapshape/{k}"]) == tuple(taps[k].shape)
            assert np.abs(sample_view(taps[k]) - r).max() <= 2e-5 * np.abs(r).max(), k


def test_nearest_map_matches_interpolate():
    for li, lo in [(13, 25), (25, 50), (50, 100), (100, 200), (12, 24), (24, 47), (47, 94), (94, 187), (4, 8), (7, 13)]:
        x = torch.arange(li, dtype=torch.float32).view(1, 1, li)
        ref = torch.nn.functional.interpolate(x, size=lo, mode="nearest").view(-1).long()
        assert torch.equal(nearest_index_map(li, lo), ref)


def test_c1_trace(golden_dir):
    """Free-running 50-step chain with injected noise reproduces the reference's x_t / eps_hat / x_0."""
    g = np.load(os.path.join(golden_dir, "trace_c1.npz"))
    sd = _sd(golden_dir)
    B, L, T = 16, 200, 50
    x_T, _, cat, num = synth_inputs(B, L, T_MODEL, seed=2)
    o = DiffusionOracle(1e-4, 0.05, T)
    trace = []
    with torch.no_grad():
        x0 = o.sample(x_T, lambda x, t: trajunet_forward(sd, x, t, cat, num, FULL_SCHEDULE, T_MODEL),
                      lambda t: synth_noise(t, x_T.shape, seed=2), trace=trace)
    for i, (t, x_t, eps) in enumerate(trace):
        assert t == T - 1 - i
        assert np.abs(x_t.numpy() - g["x_t"][i]).max() <= 1e-4 * max(1.0, np.abs(g["x_t"][i]).max())
        assert np.abs(eps.numpy() - g["eps"][i]).max() <= 1e-4 * np.abs(g["eps"][i]).max()
    assert np.abs(x0.numpy() - g["x0"]).max() <= 1e-4 * np.abs(g["x0"]).max()


def test_q_sample_and_loss(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    sd = _sd(golden_dir)
    x0, t, cat, num = synth_inputs(4, 200, T_MODEL, seed=3)
    eps = synth_noise(0, x0.shape, seed=3)
    o = DiffusionOracle(1e-4, 0.05, 500)
    x_t = o.q_sample(x0, t, eps)
    np.testing.assert_array_equal(x_t.numpy(), g["x_t"])
    with torch.no_grad():
        loss = torch.mean((trajunet_forward(sd, x_t, t, cat, num, FULL_SCHEDULE, T_MODEL) - eps) ** 2)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    # dead decoder blocks never receive gradients
    dead = [k for k in g.files if k.startswith("nograd/")]
    assert len(dead) == 40 and all(".1." in k and "up_blocks" in k for k in dead)


def test_dataset_items(golden_dir):
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    mean = [21599.4980, 108.950773428688, 34.24354179925547]
    std = [12470.9102, 0.02129110045580343, 0.019358855648211895]
    np.testing.assert_array_equal(gaussian_cell_boundaries(256).numpy(), g["boundaries"])
    recs = [r for r in synth_raw_trajectories(6, 200, seed=4) if r[2].shape[0] >= 200]
    assert len(recs) == g["lon_lat"].shape[0]
    for i, (_, _, ll, ts) in enumerate(recs):
        ll32 = ll[:200].to(torch.float32)  # DidiDataset.py:83
        mins = ((ts[:200] - 1477958400) / 60).to(torch.float32)  # DidiDataset.py:84
        a, c, n, tm = didi_item(ll32, mins, mean, std)
        np.testing.assert_array_equal(a.numpy(), g["lon_lat"][i])
        np.testing.assert_array_equal(c.numpy(), g["cat"][i])
        np.testing.assert_array_equal(n.numpy(), g["num"][i])
        np.testing.assert_array_equal(tm.numpy(), g["times"][i])


def test_metrics_oracle_self_consistency():
    """oracle/metrics_oracle.py is PARITY UNPINNED (no reference fixtures exist for eval_trajunet.ipynb cell 12): checked here are
    the properties the reference's code implies, and the restated linspace against torch's (CPU kernel: <= 1 ulp apart)."""
    import numpy as np
    import torch
    from _synth import synth_metric_trajectories
    from oracle import metrics_oracle as mo
    for a, b, n in ((-1.7, 2.9, 16), (0.0, 10.0, 64), (3.0, 3.5, 7)):
        g, t = mo.linspace_f32(a, b, n), torch.linspace(a, b, n).numpy()
        assert g[0] == np.float32(a) and g[-1] == np.float32(b) and np.all(np.diff(g) > 0)
        assert np.abs(g - t).max() <= 2 * np.spacing(np.float32(max(abs(a), abs(b))))
    t0, t1 = synth_metric_trajectories(64, 200, 1)
    raw = mo.point_distribution(t0, 16, False)
    assert raw.sum() == 64 * 200 and raw[0].sum() >= 1 and raw[-1].sum() >= 1      # min lands in row 0, max in row 15
    p, q = mo.point_distribution(t0), mo.point_distribution(t1)
    assert abs(p.sum() - (64 * 200 + 256) / (64 * 200)) < 1e-5
    assert mo.jsd(p, p) == 0.0 and mo.jsd(p, q) > 0 and abs(mo.jsd(p, q) - mo.jsd(q, p)) < 1e-9
    assert mo.start_distribution(t0).shape == (16, 16) and mo.length_distribution(t0).shape == (64,)
    one = torch.zeros(1, 2, 3)
    one[0, 0] = torch.tensor([0.0, 3.0, 3.0])
    one[0, 1] = torch.tensor([0.0, 4.0, 4.0])
    assert mo.path_lengths(one)[0] == 5.0
