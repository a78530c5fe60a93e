#!/usr/bin/env python3
"""Config C4: one TrajUNet training step (q_sample -> forward -> MSE -> backward -> AdamW), fp32, B=1024, L=200.

    python scripts/train_bench.py [--batch 1024] [--steps 5] [--warmup 2] [--breakdown] [--no-graph]
    python scripts/train_bench.py --gpus 8 --overlap                # data parallel: starts the 8 ranks itself (child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_bench.py   # the same, launched by hand

With N ranks every rank steps its own B trajectories (weak scaling) and the gradients are averaged over RCCL
(TrainStep.sync_gradients) between backward and the optimizer; the time is the max over ranks.

Prints ONE JSON line: training steps/s and trajectories/s, the time split between forward / backward / optimizer and (with
--breakdown) the per-entry-point device time of the two programs.  Synthetic data, random-init weights.  Not the headline
metric (bench.py is); this is the measurement for SURVEY.md 8 row a13.
"""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCHEDULE = [128, 128, 256, 512, 1024]
FLOP_PER_TRAJ_FWD = 2 * 1_015_085_824


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without RANK in the environment: spawn N ranks (one per GPU, RCCL) as a child process")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--L", type=int, default=200)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="weight re-layout ops launched eagerly (needed under rocprofv3 --kernel-trace)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--per-op", type=int, default=0, help="with --breakdown: list the N slowest launches of each program on stderr")
    ap.add_argument("--overlap", action="store_true", help="N > 1: reduce gradient slices under the backward program instead of after it")
    ap.add_argument("--check-overlap", action="store_true", help="N > 1: assert that both reduction schedules give identical gradients")
    ap.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK)")
    ap.add_argument("--force-collectives", action="store_true", help="run the process group and every collective even at world size 1 "
                    "(rehearses the RCCL code path on a one-GPU box; needs RANK / MASTER_* from torch.distributed.run)")
    a = ap.parse_args()
    if a.force_collectives:
        os.environ["TDX_FORCE_COLLECTIVES"] = "1"
    from trajectorydiffusion_amd.distributed import spawn_ranks_if_needed
    rc = spawn_ranks_if_needed(os.path.abspath(__file__), sys.argv[1:], a.gpus)   # before any GPU call
    if rc is not None:
        sys.exit(rc)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or a.force_collectives
    dist = None
    if multi:
        import faulthandler
        import torch.distributed as dist
        faulthandler.dump_traceback_later(90, exit=True)      # a stuck collective names its line instead of hanging the box
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    from _synth import synth_inputs, synth_noise
    from trajectorydiffusion_amd import DiffusionManager, TrajUNet
    from trajectorydiffusion_amd.training import TrainStep
    torch.manual_seed(0)
    model = TrajUNet(SCHEDULE, diffusion_steps=a.T, res_blocks=2).to(dev)
    dm = DiffusionManager(1e-4, 0.05, a.T, device=str(dev))
    B, L = a.batch, a.L
    x0, t, cat, num = synth_inputs(B, L, a.T, seed=7 + rank)
    x0, t, cat, num = x0.to(dev), t.to(dev), cat.to(dev), num.to(dev)
    eps = synth_noise(0, (B, 2, L), seed=7 + rank).to(dev)
    ts = TrainStep(model, B, L, dev, lr=5e-4, graph_repack=not a.no_graph, seed=rank)   # dropout masks differ per rank
    torch.cuda.synchronize()
    mem_gb = torch.cuda.max_memory_allocated() / 2 ** 30

    def step():
        x_t = dm.diffusionForward(x0, t, eps)
        if multi and a.overlap:
            loss = ts.forward_backward(x_t, t, cat, num, eps, sync_group=None)
        else:
            loss = ts.forward_backward(x_t, t, cat, num, eps)
            if multi:
                ts.sync_gradients()
        ts.optimizer_step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    el = (time.perf_counter() - t0) / a.steps
    assert torch.isfinite(loss).all()
    if multi:
        cdev = dev if a.backend == "nccl" else torch.device("cpu")
        tmax = torch.tensor([el], device=cdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
        # replicas must still hold identical parameters after synchronised steps
        probe = torch.stack([p.detach().double().sum() for p in model.parameters()]).to(cdev)
        lo, hi = probe.clone(), probe.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged"

    if multi and a.check_overlap:
        x_t = dm.diffusionForward(x0, t, eps)
        ts._passes = 1000                                   # replay the same dropout masks in both passes
        ts.forward_backward(x_t, t, cat, num, eps)
        local = ts.flat_gradients().clone()
        ts.sync_gradients()
        after = ts.flat_gradients().clone()
        ts._passes = 1000
        ts.forward(x_t, t, cat, num)                        # dpred of the pass above is still in ts.dpred (same inputs, same masks)
        ts.backward(ts.dpred, sync_group=None, bucket_bytes=3 << 20)   # many small slices: exercises the readiness plan
        assert torch.equal(ts.flat_gradients(), after), "overlapped reduction differs from reduce-after-backward"
        if world > 1:
            assert not torch.equal(local, after), "ranks hold different data, the mean must differ from the local gradient"
        else:
            assert torch.equal(local, after), "world size 1: the mean over ranks is the local gradient"
        plan = ts._bucket_plan(3 << 20)
        print(f"rank {rank}: overlap check ok, {len(plan)} slices, first ready after launch {plan[0][2]} of {len(ts.bw.ops)}", file=sys.stderr, flush=True)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    x_t = dm.diffusionForward(x0, t, eps)
    ev[0].record()
    ts.forward(x_t, t, cat, num)
    ev[1].record()
    ts.backward(eps)
    ev[2].record()
    ts.optimizer_step()
    ev[3].record()
    torch.cuda.synchronize()
    out = {"metric": "training steps/sec, TrajUNet fp32 step (forward+backward+AdamW), L=%d" % L, "value": round(1 / el, 4), "unit": "steps/s",
           "trajectories_per_s": round(world * B / el, 2), "ms_per_step": round(1e3 * el, 2), "n_gpus": world, "scaling": "weak", "steps": a.steps, "warmup": a.warmup,
           "dtype": "fp32", "data": "synthetic", "config": {"workload": f"C4: TrajUNet training step, B={B} per GPU, L={L}, schedule {SCHEDULE}, dropout 0.1, AdamW",
                      "parallelism": f"data parallel x{world}, gradient mean over {a.backend if multi else 'nothing'}" + (", overlapped with backward" if a.overlap and multi else "")},
           "forward_ms": round(ev[0].elapsed_time(ev[1]), 2), "backward_ms": round(ev[1].elapsed_time(ev[2]), 2),
           "optimizer_ms": round(ev[2].elapsed_time(ev[3]), 2),
           "model_tflops_per_gpu": round(3 * FLOP_PER_TRAJ_FWD * B / el / 1e12, 2), "peak_tflops_fp32_mfma": 157.3,
           "frac_of_fp32_mfma_peak": round(3 * FLOP_PER_TRAJ_FWD * B / el / 1e12 / 157.3, 4),
           "device_memory_gib": round(mem_gb, 2), "loss": float(loss)}
    if a.breakdown:
        for tag, pg in (("forward", ts.fw), ("backward", ts.bw)):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for name, fn, ms, fl in pg.profile(repeats=2):
                agg[fn][0] += 1
                agg[fn][1] += ms
            out[tag + "_by_entry_point"] = {k: {"calls": v[0], "ms": round(v[1], 3)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            if a.per_op:
                for name, fn, ms, fl in sorted(pg.profile(repeats=2), key=lambda r: -r[2])[: a.per_op]:
                    print(f"{tag:9s} {name:44s} {fn:24s} {ms:8.4f} ms", file=sys.stderr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        faulthandler.cancel_dump_traceback_later()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
