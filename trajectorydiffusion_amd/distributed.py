"""Batch sharding for multi-GPU sampling: one process per GPU, no data-path collective, one final gather; and the one real
exchange step of the training path, the data-parallel gradient mean.

Trajectories are independent given the weights (reference DiffusionManager.py:57-79), so rank r of n samples
the contiguous slice shard_bounds(B, r, n) and passes first_sample=lo so that in-kernel Philox noise is keyed
by the GLOBAL sample index: the gathered result equals the unsharded run bit for bit.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


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
    if world == 1 or flat.numel() == 0:
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
