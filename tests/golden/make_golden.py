"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference is imported unmodified; weights/inputs/noise come from tests/_synth.py (numpy PCG64)
so the tests can rebuild them anywhere.  Only inputs' seeds and expected OUTPUTS are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
for name in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
    sys.modules.setdefault(name, types.ModuleType(name))  # imported but unused by Dataset/DidiDataset.py:6-9

from _synth import (FULL_SCHEDULE, sample_view, synth_inputs, synth_noise, synth_raw_trajectories,  # noqa: E402
                    synth_state_dict)
from DiffusionManager import DiffusionManager  # noqa: E402  (reference)
from Models.TrajUNet import TrajUNet  # noqa: E402  (reference)
from Dataset.DidiDataset import DidiTrajectoryDataset, collectFunc  # noqa: E402  (reference)

torch.set_num_threads(8)
T_MODEL = 500


def build(schedule):
    m = TrajUNet(channel_schedule=schedule, diffusion_steps=T_MODEL, res_blocks=2).eval()
    spec = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict(synth_state_dict(spec, seed=0), strict=True)
    return m, spec


def hook_taps(m, taps):
    hs = []

    def add(mod, name):
        hs.append(mod.register_forward_hook(lambda _m, _i, o: taps.__setitem__(name, o.detach())))

    add(m.embed_block, "embed")
    add(m.stem, "stem")
    for i in range(m.stages):
        add(m.down_blocks[i][0], f"res_d{i}0")
        add(m.down_blocks[i][m.res_blocks], f"down{i}")
        add(m.up_blocks[i][0], f"res_u{i}0")
        add(m.up_blocks[i][m.res_blocks], f"up{i}")
    add(m.mid_attn_block.res1, "mid_res1")
    add(m.mid_attn_block, "mid")
    return hs


def main():
    model, spec = build(FULL_SCHEDULE)
    with open(os.path.join(HERE, "state_dict_spec.json"), "w") as f:
        json.dump(spec, f)

    # (i) schedules -----------------------------------------------------------------------------
    out = {}
    for tag, args in {"T500": (1e-4, 0.05, 500, False), "T50": (1e-4, 0.05, 50, False),
                      "default": (1e-4, 0.002, 100, False), "scaled": (1e-4, 0.02, 300, True)}.items():
        dm = DiffusionManager(args[0], args[1], args[2], device="cpu", scaled_linear=args[3])
        for k in ("betas", "alphas", "alpha_bars", "sqrt_alpha_bars", "sqrt_1_minus_alphas_bars"):
            v = getattr(dm, k)
            assert tuple(v.shape) == (args[2], 1, 1)
            out[f"{tag}/{k}"] = v.reshape(-1).numpy()
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **out)

    # (ii) one forward, per-sample t, with block taps -----------------------------------------------
    out = {}
    for tag, (B, L) in {"B4_L200": (4, 200), "B3_L187": (3, 187), "B2_L64": (2, 64)}.items():
        x, t, cat, num = synth_inputs(B, L, T_MODEL, seed=1)
        taps = {}
        hs = hook_taps(model, taps)
        with torch.no_grad():
            eps = model(x, t, cat, num)
        for h in hs:
            h.remove()
        out[f"{tag}/eps"] = eps.numpy()
        for k, v in taps.items():
            out[f"{tag}/tap/{k}"] = sample_view(v)
            out[f"{tag}/tapshape/{k}"] = np.array(v.shape)
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **out)

    # (iii) C1 trace: B=16, L=200, T=50 DDPM with injected noise ----------------------------------------
    B, L, T = 16, 200, 50
    x_T, _, cat, num = synth_inputs(B, L, T_MODEL, seed=2)
    dm = DiffusionManager(1e-4, 0.05, T, device="cpu")
    xs, epss = [], []
    state = {"t": T - 1}
    real_randn_like = torch.randn_like

    def injected(x, *a, **k):
        z = synth_noise(state["t"], x.shape, seed=2)
        return z

    def traced_model(x_t, t_vec, c, n):
        assert int(t_vec[0]) == state["t"]
        e = model(x_t, t_vec, c, n)
        xs.append(x_t.numpy().copy())
        epss.append(e.numpy().copy())
        return e

    class Wrapper(torch.nn.Module):
        def forward(self, *a):
            return traced_model(*a)

    torch.randn_like = injected
    try:
        # the reference loop itself (DiffusionManager.py:57-79); step() is wrapped to track t for the noise key
        orig_step = dm.diffusionBackwardStep

        def step(x_t, t, e):
            state["t"] = t
            r = orig_step(x_t, t, e)
            state["t"] = t - 1
            return r

        dm.diffusionBackwardStep = step
        x0 = dm.diffusionBackward(x_T, cat, num, Wrapper())
    finally:
        torch.randn_like = real_randn_like
    np.savez_compressed(os.path.join(HERE, "trace_c1.npz"), x_t=np.stack(xs), eps=np.stack(epss), x0=x0.numpy())

    # (iv) q-sample + eps-MSE train step (dropout off) -----------------------------------------------
    B, L = 4, 200
    x0_, t, cat, num = synth_inputs(B, L, T_MODEL, seed=3)
    dm = DiffusionManager(1e-4, 0.05, 500, device="cpu")
    eps = synth_noise(0, x0_.shape, seed=3)
    x_t = dm.diffusionForward(x0_, t, eps)
    model.zero_grad()
    loss = torch.nn.MSELoss()(model(x_t, t, cat, num), eps)
    loss.backward()
    out = {"x_t": x_t.detach().numpy(), "loss": np.array(loss.item(), dtype=np.float64)}
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + k] = sample_view(p.grad, 512)
            out["gradnorm/" + k] = np.array(p.grad.double().norm().item())
        else:
            out["nograd/" + k] = np.array(0)
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **out)

    # (v) dataset items -----------------------------------------------------------------------------
    import tempfile
    mean = [21599.4980, 108.950773428688, 34.24354179925547]
    std = [12470.9102, 0.02129110045580343, 0.019358855648211895]
    with tempfile.TemporaryDirectory() as d:
        root = os.path.join(d, "nov")
        os.makedirs(root)
        torch.save(synth_raw_trajectories(6, 200, seed=4), os.path.join(root, "gps_20161101.pt"))
        ds = DidiTrajectoryDataset(root, 200, mean, std)
        ds.loadNextFiles(1)
        ll, c, n, tm = collectFunc([ds[i] for i in range(len(ds))])
        np.savez_compressed(os.path.join(HERE, "dataset.npz"), lon_lat=ll.numpy(), cat=c.numpy(), num=n.numpy(),
                            times=tm.numpy(), boundaries=ds.city_cell_boundaries.numpy())
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
