"""CPU: host-side logic, the C-ABI surface (symbols only, no compute) and the multi-process shard/gather path."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from trajectorydiffusion_amd import _lib
    header = open(os.path.join(ROOT, "include", "trajdiff_hip.h")).read()
    declared = set(re.findall(r"\b(tdx_[a-z0-9_]+)\s*\(", header))
    assert {"tdx_conv_gemm", "tdx_groupnorm_act", "tdx_ddpm_update", "tdx_attention", "tdx_didi_items"} <= declared
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/trajdiff_hip.h but not exported"
    assert declared <= set(_lib.EXPORTS) | {"tdx_last_error"}
    assert handle.tdx_version() >= 100 and isinstance(handle.tdx_last_error(), bytes)


def test_state_dict_keys_match_reference_and_default_init_is_deterministic(golden_dir):
    import json
    from trajectorydiffusion_amd import TrajUNet
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    torch.manual_seed(0)
    m = TrajUNet([128, 128, 256, 512, 1024], diffusion_steps=500, res_blocks=2)
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, s) for k, s in spec]
    assert len(list(m.buffers())) == 0 and not isinstance(m.embed_block.pos_enc, torch.nn.Parameter)
    assert sum(p.numel() for p in m.parameters()) == 50_039_170
    assert m.embed_cols == 6912 and len(dict(m.executed_resblocks())) == 14
    with pytest.raises(Exception):
        TrajUNet([128, 128], res_blocks=1)


def test_checkpoint_is_plain_tensors(tmp_path):
    """saveModel / loadModel (Utils.py:6-12): the file must hold tensors only, loadable with weights_only=True, strict keys."""
    import torch
    from trajectorydiffusion_amd import TrajUNet
    m = TrajUNet([16, 16, 32], diffusion_steps=20, res_blocks=1) if False else TrajUNet([128, 128, 256, 512, 1024], diffusion_steps=500, res_blocks=2)
    path = tmp_path / "TrajUNet_0.pth"
    torch.save(m.state_dict(), path)
    sd = torch.load(path, weights_only=True)
    assert list(sd) == list(m.state_dict()) and m.load_state_dict(sd).missing_keys == []


def test_no_cpu_fallback():
    from trajectorydiffusion_amd import DiffusionManager, TrajUNet
    from trajectorydiffusion_amd._lib import TdxError
    m = TrajUNet([32, 32, 64], diffusion_steps=10).eval()
    with pytest.raises(TdxError):
        m(torch.zeros(1, 2, 16), torch.zeros(1, dtype=torch.long), torch.zeros(1, 5, dtype=torch.long), torch.zeros(1, 3))
    dm = DiffusionManager(1e-4, 0.05, 10, device="cpu")
    with pytest.raises(TdxError):
        dm.diffusionForward(torch.zeros(1, 2, 16), torch.zeros(1, dtype=torch.long), torch.zeros(1, 2, 16))


def test_schedule_matches_golden_bit_exact(golden_dir):
    from trajectorydiffusion_amd import DiffusionManager
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    for tag, args in {"T500": (1e-4, 0.05, 500, False), "default": (1e-4, 0.002, 100, False), "scaled": (1e-4, 0.02, 300, True)}.items():
        dm = DiffusionManager(args[0], args[1], args[2], device="cpu", scaled_linear=args[3])
        for k in ("betas", "alphas", "alpha_bars", "sqrt_alpha_bars", "sqrt_1_minus_alphas_bars"):
            v = getattr(dm, k)
            assert tuple(v.shape) == (args[2], 1, 1)
            np.testing.assert_array_equal(v.reshape(-1).numpy(), g[f"{tag}/{k}"])


def test_weight_packing_order_and_nearest_map():
    from oracle import nearest_index_map
    from trajectorydiffusion_amd import engine
    w = torch.arange(4 * 96 * 3, dtype=torch.float32).reshape(4, 96, 3)
    p = engine.pack_conv_weight(w, torch.bfloat16).float()  # 64-channel blocks: [blk0: tap0 c0..63, tap1, tap2][blk1: 32 ch ...]
    assert p.shape == (4, 288)
    assert torch.equal(p[:, 0:64], w[:, 0:64, 0].bfloat16().float()) and torch.equal(p[:, 64:128], w[:, 0:64, 1].bfloat16().float())
    assert torch.equal(p[:, 192:224], w[:, 64:96, 0].bfloat16().float()) and torch.equal(p[:, 256:288], w[:, 64:96, 2].bfloat16().float())
    p32 = engine.pack_conv_weight(w, torch.float32)           # 32-channel blocks in fp32
    assert torch.equal(p32[:, 32:64], w[:, 0:32, 1]) and torch.equal(p32[:, 96:128], w[:, 32:64, 0])
    assert torch.equal(engine.pack_conv_weight(w[:, :, :1], torch.float32), w[:, :, 0])
    for li, lo in [(13, 25), (25, 50), (100, 200), (24, 47)]:
        assert torch.equal(engine.nearest_map(li, lo).long(), nearest_index_map(li, lo))
    assert [engine.conv_out_len(l, 2) for l in (200, 100, 50, 25, 13)] == [100, 50, 25, 13, 7]


def test_shard_bounds_cover_batch_exactly():
    from trajectorydiffusion_amd.distributed import shard_bounds
    for total in (0, 1, 7, 4096, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from _synth import synth_inputs, synth_noise
from oracle import DiffusionOracle
from trajectorydiffusion_amd.distributed import shard_bounds, gather_samples
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B, L, T = 7, 40, 6
x_T, _, cat, num = synth_inputs(B, L, 500, seed=11)
o = DiffusionOracle(1e-4, 0.05, T)
toy = lambda x, t: 0.1 * x + 0.01 * t.view(-1, 1, 1).float()
full_noise = lambda t: synth_noise(t, (B, 2, L), seed=11)
ref = o.sample(x_T, toy, full_noise)
lo, hi = shard_bounds(B, rank, world)
mine = o.sample(x_T[lo:hi], toy, lambda t: full_noise(t)[lo:hi])   # noise keyed by GLOBAL sample index
got = gather_samples(mine, B)
assert got.shape == ref.shape and torch.equal(got, ref), (rank, (got - ref).abs().max())
dist.barrier(); dist.destroy_process_group(); print("rank", rank, "ok")
'''


def _free_port():
    from trajectorydiffusion_amd.distributed import free_port
    return free_port()


def test_two_process_shard_and_gather_equals_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from trajectorydiffusion_amd.distributed import shard_bounds, allreduce_mean_
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# the data-parallel training step on a stand-in model: equal shards of one global batch, MSE 'mean' loss per shard
# (train.ipynb:200), gradient mean over ranks == the full-batch gradient
torch.manual_seed(5)
Bg, n_in, n_out = 8, 37, 29
W = torch.randn(n_out, n_in, dtype=torch.float64, requires_grad=True); b = torch.randn(n_out, dtype=torch.float64, requires_grad=True)
x = torch.randn(Bg, n_in, dtype=torch.float64); y = torch.randn(Bg, n_out, dtype=torch.float64)
loss = lambda xs, ys: torch.nn.functional.mse_loss(torch.tanh(xs @ W.T + b), ys)
gW, gb = torch.autograd.grad(loss(x, y), (W, b))
full = torch.cat([gW.flatten(), gb.flatten()])
lo, hi = shard_bounds(Bg, rank, world)
mW, mb = torch.autograd.grad(loss(x[lo:hi], y[lo:hi]), (W, b))
flat = torch.cat([mW.flatten(), mb.flatten()]).contiguous()
allreduce_mean_(flat, bucket_bytes=1000)          # 125-element slices: 9 buckets, the last one ragged
assert torch.allclose(flat, full, rtol=1e-12, atol=1e-14), (rank, (flat - full).abs().max())
f32 = torch.arange(1001, dtype=torch.float32) * (rank + 1)
allreduce_mean_(f32, bucket_bytes=64 << 20)       # one bucket
assert torch.equal(f32, torch.arange(1001, dtype=torch.float32) * 1.5)
try:
    allreduce_mean_(torch.zeros(4, 4))
    raise SystemExit("2-D buffer accepted")
except ValueError:
    pass
dist.barrier(); dist.destroy_process_group(); print("rank", rank, "ok")
'''


def test_two_process_gradient_mean_equals_full_batch_gradient(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_persistent_conv_kernels_have_no_spills_in_their_k_loops():
    """Build check (cross-compiles conv_gemm.hip for gfx950, no GPU needed): a spill reload inside a K loop is a vector-memory
    operation whose wait also waits for every LDS-DMA in flight, which would serialise the software pipeline."""
    import shutil
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc") and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_spills.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


def test_rank_launcher_builds_the_torchrun_child_command(monkeypatch):
    """`python bench.py --gpus N` must itself start N ranks: a CHILD `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>`, spawned before any GPU call."""
    from trajectorydiffusion_amd import distributed as D
    cmd = D.rank_launch_command("/x/bench.py", ["--gpus", "8", "--steps", "3"], 8, port=12345, python="py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "12345",
                   "/x/bench.py", "--gpus", "8", "--steps", "3"]
    calls = []

    class _R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return _R()
    monkeypatch.setattr(D.subprocess, "run", fake_run)
    monkeypatch.delenv("RANK", raising=False)
    assert D.spawn_ranks_if_needed("/x/bench.py", ["--gpus", "1"], 1) is None and not calls         # N == 1: run in this process
    assert D.spawn_ranks_if_needed("/x/bench.py", ["--gpus", "4"], 4) == 7
    (cmd, env), = calls
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4"] and cmd[-3:] == ["/x/bench.py", "--gpus", "4"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("RANK", "0")                                                                    # already a rank: never spawn again
    assert D.spawn_ranks_if_needed("/x/bench.py", ["--gpus", "4"], 4) is None and len(calls) == 1


@pytest.mark.parametrize("script", ["bench.py", os.path.join("scripts", "train_bench.py")])
def test_bench_gpus_flag_spawns_ranks_before_any_gpu_call(script, monkeypatch):
    """The driver runs `python bench.py --gpus N`; with N > 1 the script must hand over to the launcher (exit code relayed)
    without touching the GPU in the launching process."""
    import importlib.util
    import torch
    from trajectorydiffusion_amd import distributed as D
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, script))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen = []

    class _R:
        returncode = 0

    monkeypatch.setattr(D.subprocess, "run", lambda cmd, env=None, **kw: (seen.append(cmd), _R())[1])
    monkeypatch.delenv("RANK", raising=False)

    def boom(*a, **k):
        raise AssertionError("GPU touched before the ranks were spawned")
    monkeypatch.setattr(torch.cuda, "set_device", boom)
    monkeypatch.setattr(sys, "argv", [script, "--gpus", "8", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        mod.main()
    assert e.value.code == 0
    (cmd,), n = seen, 8
    assert f"--nproc-per-node={n}" in cmd and cmd[-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith(script)


def test_bench_reports_the_world_size_and_newest_traffic_file(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in ("r1_pmc_traffic.json", "r2_pmc_traffic.json", "r10_pmc_traffic.json", "r3_pmc_traffic_old.json"):
        (tmp_path / name).write_text("{}")
    assert os.path.basename(mod.newest_traffic_file(str(tmp_path))) == "r10_pmc_traffic.json"
    assert mod.newest_traffic_file(str(tmp_path / "missing")) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"n_gpus": world' in src and "world = dist.get_world_size()" in src       # the line reports the RCCL world, not the flag


def test_gradient_overlap_schedule_invariants():
    """The data-parallel training step starts the RCCL all-reduce of a gradient slice right after the backward launch that
    completes it (TrainStep.backward).  The schedule is pure host logic: slices cover the buffer exactly once, are issued in
    non-decreasing readiness order, every gradient has a writer, and a slice is never ready before its last writer."""
    from trajectorydiffusion_amd.distributed import bucket_plan, scan_gradient_writers
    rng = np.random.default_rng(0)
    base = 0x7F0000000000
    names, spans_addr, off = [], [], 0
    for i in range(23):                                   # 23 gradients, 64-float aligned like TrainStep's arena
        n = int(rng.integers(1, 5000))
        spans_addr.append((base + 4 * off, base + 4 * (off + n), f"p{i}"))
        names.append(f"p{i}")
        off += (n + 63) // 64 * 64
    used = off
    # a backward program of 40 launches: gradient i is written by launch 39 - i (reverse layer order) and, for some, by an
    # earlier launch too (accumulating writers: the LAST one counts); p22 only through a fold after the program
    launches = [[] for _ in range(40)]
    for i, (lo, hi, n) in enumerate(spans_addr[:-1]):
        launches[39 - i].append(lo)
        if i % 3 == 0:
            launches[max(0, 39 - i - 5)].append(lo + 4 * 7)        # pointer into the middle of the span
    for a in launches:
        a += [3, 17, True, base - 64, base + 4 * used + 128, None, 2.5]   # non-pointer arguments and addresses outside the buffer
    ready = scan_gradient_writers(spans_addr, launches, base, base + 4 * used, folded=["p22"])
    assert ready["p22"] == 40 and all(ready[f"p{i}"] == 39 - i for i in range(22))
    with pytest.raises(ValueError):
        scan_gradient_writers(spans_addr, launches, base, base + 4 * used)           # p22 has no writer
    spans = [((lo - base) // 4, (hi - base) // 4, ready[n]) for lo, hi, n in spans_addr]
    for bucket_bytes in (256, 3000, 1 << 20):
        plan = bucket_plan(spans, used, bucket_bytes)
        cover = sorted((lo, hi) for lo, hi, _ in plan)
        assert cover[0][0] == 0 and cover[-1][1] == used and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        assert all(hi - lo <= max(64, bucket_bytes // 4) for lo, hi, _ in plan)
        assert [r for _, _, r in plan] == sorted(r for _, _, r in plan)
        for lo, hi, r in plan:
            need = max((rd for a, b, rd in spans if a < hi and b > lo), default=0)
            assert r == need
    assert bucket_plan(spans, 0, 1024) == []
