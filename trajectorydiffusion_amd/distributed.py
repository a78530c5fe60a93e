"""Batch sharding for multi-GPU sampling: one process per GPU, no data-path collective, one final gather; and the one real
exchange step of the training path, the data-parallel gradient mean.

Trajectories are independent given the weights (reference DiffusionManager.py:57-79), so rank r of n samples
the contiguous slice shard_bounds(B, r, n) and passes first_sample=lo so that in-kernel Philox noise is keyed
by the GLOBAL sample index: the gathered result equals the unsharded run bit for bit.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def free_port() -> int:
    """A TCP port that is free on 127.0.0.1 right now (rendezvous of a one-node job)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_launch_command(script: str, argv: Sequence[str], gpus: int, port: Optional[int] = None, python: Optional[str] = None) -> List[str]:
    """The child command that runs `script argv...` as `gpus` ranks of ONE node, one rank per GPU over RCCL:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script argv..."""
    if gpus < 1:
        raise ValueError(f"gpus={gpus}")
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else free_port()), script, *argv]


def spawn_ranks_if_needed(script: str, argv: Sequence[str], gpus: int) -> Optional[int]:
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks as a CHILD process (never an
    exec: this must also be safe after the GPU has been touched, although callers call it before any GPU call), let the
    child's stdout/stderr through (rank 0 prints the JSON line) and return its exit code.  Returns None when this process
    is already a rank (launched by torch.distributed.run) or N == 1: the caller then runs the benchmark itself."""
    if gpus <= 1 or "RANK" in os.environ:
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL on this image needs dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // gpus))))
    return subprocess.run(rank_launch_command(script, argv, gpus), env=env).returncode


def force_collectives() -> bool:
    """TDX_FORCE_COLLECTIVES=1: run every collective of the multi-GPU paths even at world size 1 (where they are no-ops for the
    data).  For rehearsing the RCCL code path (process group, side-stream all-reduces, all_gather) on a one-GPU box."""
    return os.environ.get("TDX_FORCE_COLLECTIVES", "") == "1"


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of `rank`'s contiguous slice; sizes differ by at most one, earlier ranks take the remainder."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """all_gather of per-rank sample slices (possibly ragged by one row) into the (total, ...) tensor on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def sample_sharded(dm, model, x_T: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor, **kw) -> torch.Tensor:
    """diffusionBackward over this rank's slice of a replicated global batch, then gather."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(x_T.shape[0], rank, world)
    x0 = dm.diffusionBackward(x_T[lo:hi], cat_attr[lo:hi], num_attr[lo:hi], model, first_sample=lo, **kw)
    return gather_samples(x0, x_T.shape[0])


def allreduce_mean_(flat: torch.Tensor, group: Optional[dist.ProcessGroup] = None, bucket_bytes: int = 64 << 20) -> torch.Tensor:
    """In-place mean over ranks of one flat gradient buffer, issued as back-to-back asynchronous all-reduces over
    bucket_bytes-sized slices (no packing: TrainStep keeps every parameter gradient inside one flat buffer).

    xGMI is point-to-point, so RCCL's ring all-reduce is bound per link (~153 GB/s); the default 64 MiB slices are large
    enough to run at link rate and small enough that the reduction of slice k+1 overlaps the scaling of slice k.  The mean
    (not the sum) matches what torch's DistributedDataParallel hands the optimizer, so lr keeps its single-GPU meaning."""
    if flat.dim() != 1 or not flat.is_contiguous():
        raise ValueError("allreduce_mean_: expected one contiguous 1-D buffer")
    world = dist.get_world_size(group)
    if (world == 1 and not force_collectives()) or flat.numel() == 0:
        return flat
    step = max(1, bucket_bytes // flat.element_size())
    pending = [allreduce_slice_async(flat[lo:lo + step], group) for lo in range(0, flat.numel(), step)]
    inv = 1.0 / world
    for work, lo in zip(pending, range(0, flat.numel(), step)):
        if work is not None:
            work.wait()
        flat[lo:lo + step].mul_(inv)
    return flat


def allreduce_slice_async(piece: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
    """Start the in-place SUM over ranks of one slice on the CURRENT stream's timeline; returns the work handle to wait on
    (None when the reduction has already completed).  RCCL for device tensors; any other backend (the gloo rehearsals of
    the tests) stages the slice through the host, synchronously: gloo's own device path hung on this image."""
    if piece.is_cuda and dist.get_backend(group) != "nccl":
        host = piece.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        piece.copy_(host)
        return None
    return dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group, async_op=True)


# ---- overlap schedule of the gradient exchange (pure host logic: unit-tested on the CPU) ---------------------------------

def scan_gradient_writers(spans: Sequence[Tuple[int, int, str]], launches: Sequence[Sequence], arena_lo: int, arena_hi: int,
                          folded: Sequence[str] = ()) -> Dict[str, int]:
    """name -> index of the LAST launch of a recorded program that writes that gradient.

    spans    (address_lo, address_hi, name) of every parameter gradient inside the flat gradient buffer [arena_lo, arena_hi)
    launches per recorded launch, its argument tuple: any int argument that is an address inside a span marks a write
             (the program passes raw device pointers, so the schedule cannot drift from what the program really does)
    folded   names completed only AFTER the program (padded stand-ins folded back): ready index = len(launches)
    Raises ValueError when some gradient has no writer."""
    import bisect
    spans = sorted(spans)
    starts = [lo for lo, _, _ in spans]
    ready = {n: -1 for _, _, n in spans}
    for i, args in enumerate(launches):
        for a in args:
            if isinstance(a, int) and not isinstance(a, bool) and arena_lo <= a < arena_hi:
                k = bisect.bisect_right(starts, a) - 1
                if k >= 0 and a < spans[k][1]:
                    ready[spans[k][2]] = i
    for n in folded:
        if n in ready:
            ready[n] = len(launches)
    missing = [n for n, r in ready.items() if r < 0]
    if missing:
        raise ValueError(f"no backward launch writes the gradient of {missing[:3]}")
    return ready


def bucket_plan(spans: Sequence[Tuple[int, int, int]], used: int, bucket_bytes: int, elem_bytes: int = 4) -> List[Tuple[int, int, int]]:
    """Slices (lo, hi, ready) of a flat gradient buffer of `used` elements, at most bucket_bytes each, sorted by the launch
    index after which each is complete (ties: by position).  spans = (lo, hi, ready) per gradient, in elements.  The slices
    cover [0, used) exactly once; a slice is ready when the last gradient overlapping it is (alignment gaps belong to no
    gradient and are ready at once)."""
    step = max(64, int(bucket_bytes) // elem_bytes)
    plan = []
    for lo in range(0, used, step):
        hi = min(lo + step, used)
        rdy = max((r for a, b, r in spans if a < hi and b > lo), default=0)
        plan.append((lo, hi, rdy))
    plan.sort(key=lambda t: (t[2], t[0]))
    return plan
