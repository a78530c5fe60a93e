#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec, TrajUNet 500-step DDPM sampling, L=200 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--batch B] [--T 500]

One "step" = one full T-step diffusionBackward over a batch of B synthetic trajectories per GPU
(config C2: B=4096, L=200, T=500).  N>1 (launched by torch.distributed.run, one rank per GPU) shards
trajectories across ranks with no data-path collective; the final all_gather of x_0 is inside the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCHEDULE = [128, 128, 256, 512, 1024]
FLOP_PER_TRAJ_STEP = 2 * 1_015_085_824  # algorithmic MACs of one TrajUNet forward at L=200
PEAK = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md


def cpu_baseline(L, T_total, seconds_budget=20.0):
    """The oracle (CPU restatement of the reference, stock torch ops) on this host's cores, bounded sample."""
    from _synth import synth_inputs, synth_noise
    from oracle import DiffusionOracle, trajunet_forward
    from trajectorydiffusion_amd import TrajUNet
    torch.manual_seed(0)
    sd = {k: v.detach() for k, v in TrajUNet(SCHEDULE, diffusion_steps=T_total, res_blocks=2).state_dict().items()}
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    B = 16
    x, _, cat, num = synth_inputs(B, L, T_total, seed=0)
    o = DiffusionOracle(1e-4, 0.05, T_total)
    steps, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for t in range(T_total - 1, -1, -1):
            tv = torch.full((B,), t, dtype=torch.long)
            eps = trajunet_forward(sd, x, tv, cat, num, SCHEDULE, T_total)
            x = o.posterior_step(x, t, eps, synth_noise(t, x.shape) if t > 0 else None)
            steps += 1
            if time.perf_counter() - t0 > seconds_budget and steps >= 3:
                break
    dt = time.perf_counter() - t0
    return {"value": round(B / (dt / steps * T_total), 4), "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32) B={B} L={L}: {steps} of {T_total} DDPM steps timed in {dt:.1f}s, scaled to {T_total} steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--L", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sampler", default="ddpm", choices=["ddpm", "ddim"], help="ddim = config C3 (eta=0, --ddim-steps evenly spaced steps)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--low-latency", action="store_true", help="small-batch serving mode: split-K convs (TrajUNet.set_low_latency)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from _synth import synth_inputs
    from trajectorydiffusion_amd import DiffusionManager, TrajUNet
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)  # same random-init weights on every rank
    model = TrajUNet(SCHEDULE, diffusion_steps=a.T, res_blocks=2).to(dev).eval().set_compute_dtype(dtype).set_low_latency(a.low_latency)
    dm = DiffusionManager(1e-4, 0.05, a.T, device=str(dev))
    dm.seed = 2024
    B, L = a.batch, a.L
    x_T, _, cat, num = synth_inputs(B, L, a.T, seed=100 + rank)
    x_T, cat, num = x_T.to(dev), cat.to(dev), num.to(dev)
    gathered = torch.empty((world * B, 2, L), device=dev) if world > 1 else None

    def one_step():
        if a.sampler == "ddim":
            x0 = dm.ddimBackward(x_T, cat, num, model, n_steps=a.ddim_steps, use_graph=not a.no_graph)
        else:
            x0 = dm.diffusionBackward(x_T, cat, num, model, first_sample=rank * B, use_graph=not a.no_graph)
        if world > 1:
            dist.all_gather_into_tensor(gathered, x0.contiguous())
        return x0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        x0 = one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        x0 = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(x0).all()
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        value = world * B * a.steps / elapsed
        n_denoise = a.ddim_steps if a.sampler == "ddim" else a.T
        # roofline of the dominant kernel (conv_gemm): per-launch HIP-event times of one denoise step
        prog = model.program(B, L, dev, split_embed=True)
        prof = prog.pg.profile(repeats=3)
        gemm = [(n, ms, fl) for n, fn, ms, fl in prof if fn == "tdx_conv_gemm"]
        gemm_ms, gemm_fl = sum(m for _, m, _ in gemm), sum(f for _, _, f in gemm)
        step_ms = sum(ms for _, _, ms, _ in prof)
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12
        # HBM traffic of the same kernel from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), stored per round
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            if pm["workload"] == f"B={B}, L={L}, {a.dtype}":
                traffic = round(pm["conv_gemm_bytes_per_launch_avg"])
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "trajectories/sec, TrajUNet 500-step DDPM sampling L=200" if a.sampler == "ddpm" and a.T == 500 else f"trajectories/sec, TrajUNet {a.sampler} sampling ({n_denoise} denoise steps) L={L}", "value": round(value, 3), "unit": "trajectories/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"C2: TrajUNet DDPM {a.T}-step sampling" if a.sampler == "ddpm" else f"C3: TrajUNet DDIM {a.ddim_steps}-step sampling (T={a.T} table)")
                       + f", B={B} per GPU, L={L}, schedule {SCHEDULE}, random-init weights",
                       "batch_per_gpu": B, "global_batch": world * B, "diffusion_steps": a.T, "parallelism": f"batch-shard x{world}"},
            "ms_per_denoise_step": round(1e3 * elapsed / a.steps / n_denoise, 4),
            "model_tflops": round(value * n_denoise * FLOP_PER_TRAJ_STEP / 1e12 / world, 2),
            "roofline": {"bound": "mfma", "kernel": "conv_gemm_stream_kernel<...> via tdx_conv_gemm (a launch = one tdx_conv_gemm call = main grid + optional small-tile tail grid; all %d calls of one denoise step)" % len(gemm),
                         "achieved": round(achieved, 2), "peak": PEAK[a.dtype], "unit": "TFLOP/s", "frac": round(achieved / PEAK[a.dtype], 4),
                         "traffic": traffic, "traffic_source": "profiles/r1_pmc_traffic.json (rocprofv3 --pmc, bytes per launch, avg of 38)",
                         "algorithmic_bytes_per_launch_avg": round((920976 + 831376) * 2 * B / 38) if a.dtype == "bf16" else None, "launch_ms_avg": round(gemm_ms / len(gemm), 4), "launches": len(gemm),
                         "flop_per_launch_avg": gemm_fl / len(gemm), "gemm_ms_per_step": round(gemm_ms, 3),
                         "all_kernels_ms_per_step": round(step_ms, 3)},
        }
        if not a.no_cpu_baseline and world == 1:   # rank 0 at N=1 only: the other ranks must not idle at the final barrier for it
            out["cpu_baseline"] = cpu_baseline(L, a.T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
