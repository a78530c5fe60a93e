"""Evaluation metrics of sampled trajectories on the GPU: the functions of the reference's eval_trajunet.ipynb cell 12 under
their own names (getPointDistribution, getStartDistribution, getEndDistribution, getLengthDistribution, JSD), each one or
two C-ABI launches (csrc/metrics.hip) instead of a Python loop over B*L points.

Differences a caller can see: results stay on the trajectories' device (the reference builds the point histograms on the
CPU); a path longer than 10 raises IndexError after the launch (the reference raises it inside its loop).
No CPU fallback: CPU tensors raise TdxError.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import TdxError


def _prep(traj: torch.Tensor) -> torch.Tensor:
    _lib.require_device(traj, "traj_dataset")
    if traj.dim() != 3 or traj.shape[1] != 2 or traj.shape[0] == 0 or traj.shape[2] == 0:
        raise TdxError(f"traj_dataset must be (B, 2, L) with B, L > 0; got {tuple(traj.shape)}")
    return traj.detach().to(torch.float32).contiguous()


def _workspace(dev: torch.device, bins: int) -> torch.Tensor:
    return torch.empty(int(_lib.lib().tdx_metric_workspace_bytes(bins)), dtype=torch.uint8, device=dev)


def _points(traj: torch.Tensor, l_first: int, l_count: int, n_grids: int, normalize: bool) -> torch.Tensor:
    x = _prep(traj)
    B, _, L = x.shape
    out = torch.empty((n_grids, n_grids), dtype=torch.float32, device=x.device)
    ws = _workspace(x.device, n_grids * n_grids)
    _lib.check(_lib.lib().tdx_point_distribution(x.data_ptr(), B, L, l_first, l_count, n_grids, 1 if normalize else 0, out.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "tdx_point_distribution")
    return out


def getPointDistribution(traj_dataset: torch.Tensor, n_grids: int = 16, normalize: bool = True) -> torch.Tensor:
    """(B, 2, L) -> (n_grids, n_grids): occupancy of all points on a grid spanning their own bounding box."""
    return _points(traj_dataset, 0, traj_dataset.shape[2], n_grids, normalize)


def getStartDistribution(traj_dataset: torch.Tensor) -> torch.Tensor:
    return _points(traj_dataset, 0, 1, 16, True)


def getEndDistribution(traj_dataset: torch.Tensor) -> torch.Tensor:
    return _points(traj_dataset, traj_dataset.shape[2] - 1, 1, 16, True)


def getLengthDistribution(traj_dataset: torch.Tensor, return_lengths: bool = False):
    """(B, 2, L) -> (64,): histogram of path lengths on linspace(0, 10, 64), (count + 0.1) / count.sum()."""
    x = _prep(traj_dataset)
    B, _, L = x.shape
    out = torch.empty(64, dtype=torch.float32, device=x.device)
    lengths = torch.empty(B, dtype=torch.float32, device=x.device)
    err = torch.zeros(1, dtype=torch.int32, device=x.device)
    ws = _workspace(x.device, 64)
    _lib.check(_lib.lib().tdx_length_distribution(x.data_ptr(), B, L, 0.0, 10.0, 64, out.data_ptr(), lengths.data_ptr(), err.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "tdx_length_distribution")
    if int(err.item()):
        raise IndexError("getLengthDistribution: a trajectory is longer than 10 (index 64 is out of bounds for 64 bins)")
    return (out, lengths) if return_lengths else out


def JSD(P: torch.Tensor, Q: torch.Tensor) -> float:
    """Jensen-Shannon divergence as the reference computes it (kl_div 'batchmean': the sum is divided by P.shape[0])."""
    _lib.require_device(P, "P")
    _lib.require_device(Q, "Q")
    if P.shape != Q.shape or P.numel() == 0:
        raise TdxError(f"JSD: shapes differ or are empty: {tuple(P.shape)} vs {tuple(Q.shape)}")
    p, q = P.detach().to(torch.float32).contiguous(), Q.detach().to(torch.float32).contiguous()
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().tdx_jsd(p.data_ptr(), q.data_ptr(), p.numel(), p.shape[0], out.data_ptr(), _lib.stream_ptr()), "tdx_jsd")
    return out.item()
