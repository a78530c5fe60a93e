"""Probe for the host-side crash seen in round 3: latent-path work first (mode), then the multi-slice TrajUNet sampler sequence of
tests/test_gpu_model.py::test_concurrent_batch_slices_are_bit_identical_to_the_whole_batch.
usage: python graph_crash_probe.py <mode>   mode: car_graph | car_eager | car_forward | none"""
import faulthandler, os, sys
faulthandler.enable()
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _synth import synth_inputs
from trajectorydiffusion_amd import DiffusionManager, SDUNet_CAR, TrajUNet
from trajectorydiffusion_amd.sampler import GraphSampler
mode = sys.argv[1]
torch.manual_seed(0)
if mode != "none":
    car = SDUNet_CAR(4, [128, 256, 512, 1024], 500, res_blocks=4).cuda().eval().set_compute_dtype(torch.bfloat16)
    B = 256
    x = torch.randn(B, 4, 24).cuda(); t = torch.randint(0, 500, (B,)).cuda()
    cat = torch.cat([torch.randint(0, 256, (B, 4)), torch.randint(0, 33, (B, 1))], 1).cuda(); num = torch.randn(B, 3).cuda()
    if mode == "car_forward":
        y = car(x, t, cat, num)
    else:
        dm5 = DiffusionManager(1e-4, 0.05, 4, device="cuda"); dm5.seed = 1
        y = dm5.diffusionBackward(x, cat, num, car, use_graph=(mode == "car_graph"))
    torch.cuda.synchronize(); print("latent part done", float(y.abs().mean()), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "drop":
        del car, y
        if mode != "car_forward": dm5._samplers.clear()
        torch.cuda.empty_cache()
model = TrajUNet([128, 128, 256, 512, 1024], diffusion_steps=500, res_blocks=2).cuda().eval().set_compute_dtype(torch.bfloat16)
B, L, T = 2048, 200, 3
x, _, cat, num = synth_inputs(B, L, 500, seed=77)
x, cat, num = x.cuda(), cat.cuda(), num.cuda()
dm = DiffusionManager(1e-4, 0.05, 500, device="cuda"); dm.seed = 99
outs = []
for shards in (1, 2, 4):
    smp = GraphSampler(dm, model, B, L, x.device, T, shards=shards)
    outs.append(smp.sample(x, cat, num, first_sample=4096)); print("shards", shards, "ok", flush=True)
noise = torch.randn((T, B, 2, L), generator=torch.Generator().manual_seed(5)).cuda()
a = GraphSampler(dm, model, B, L, x.device, T, shards=1).sample(x, cat, num, noise=noise, use_graph=False); print("eager inject ok", flush=True)
b = GraphSampler(dm, model, B, L, x.device, T, shards=2).sample(x, cat, num, noise=noise); print("graph inject ok", flush=True)
torch.cuda.synchronize()
print("PROBE OK", torch.equal(a, b), torch.equal(outs[0], outs[1]), flush=True)
