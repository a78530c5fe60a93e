"""CPU restatement of the reference's evaluation metrics (eval_trajunet.ipynb, cell 12).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference holds no fixtures for these functions and running its notebook cell here was refused by
the environment, so this restatement is checked by reading only.  Each function names the lines of
cell 12 it follows.  The reference loops over points in Python; the loops are vectorised here, the arithmetic is the same.

Grid boundaries: the reference calls torch.linspace on the trajectories' device (a GPU in the notebook), whose kernel is
the elementwise two-half formula restated in `linspace_f32`; torch's CPU kernel differs from it by <= 1 ulp at interior
boundaries, which only matters for a coordinate lying within an ulp of a boundary.
"""
import numpy as np
import torch

f32 = np.float32


def linspace_f32(start, end, n: int) -> np.ndarray:
    """torch.linspace(start, end, n) in fp32, ATen's device formula: start + step*i below n/2, end - step*(n-1-i) above."""
    start, end = f32(start), f32(end)
    step = f32(f32(end - start) / f32(n - 1))
    i = np.arange(n)
    lo = (start + (step * i.astype(f32)).astype(f32)).astype(f32)
    hi = (end - (step * (n - 1 - i).astype(f32)).astype(f32)).astype(f32)
    return np.where(i < n // 2, lo, hi).astype(f32)


def point_distribution(traj: torch.Tensor, n_grids: int = 16, normalize: bool = True) -> np.ndarray:
    """getPointDistribution (cell 12): (B, 2, L) -> (n_grids, n_grids) occupancy of all points, indexed [lon bin, lat bin]."""
    pts = traj.transpose(1, 2).reshape(-1, 2).numpy().astype(f32)          # (B*L, 2)
    lon_grid = linspace_f32(pts[:, 0].min(), pts[:, 0].max(), n_grids)     # STEP 1-2
    lat_grid = linspace_f32(pts[:, 1].min(), pts[:, 1].max(), n_grids)
    a = np.searchsorted(lon_grid, pts[:, 0], side="left")                  # STEP 3: torch.searchsorted default = left
    b = np.searchsorted(lat_grid, pts[:, 1], side="left")
    if a.max() >= n_grids or b.max() >= n_grids:
        raise IndexError("point outside the grid")                          # the reference's point_count[lon_idx, lat_idx] would raise
    count = np.zeros((n_grids, n_grids), dtype=f32)
    np.add.at(count, (a, b), f32(1))
    if normalize:                                                           # STEP 4: (point_count + 1) / point_count.sum()
        count = ((count + f32(1)) / count.sum(dtype=f32)).astype(f32)
    return count


def start_distribution(traj: torch.Tensor) -> np.ndarray:
    """getStartDistribution (cell 12): traj[:, :, [0]]."""
    return point_distribution(traj[:, :, [0]])


def end_distribution(traj: torch.Tensor) -> np.ndarray:
    """getEndDistribution (cell 12): traj[:, :, [-1]]."""
    return point_distribution(traj[:, :, [-1]])


def path_lengths(traj: torch.Tensor) -> np.ndarray:
    """getLengthDistribution, first loop: sqrt(sum over lon/lat of squared steps).sum() per trajectory, fp32."""
    d = (traj[:, :, 1:] - traj[:, :, :-1]) ** 2
    return torch.sqrt(d.sum(dim=1)).sum(dim=1).numpy().astype(f32)


def length_distribution(traj: torch.Tensor) -> np.ndarray:
    """getLengthDistribution (cell 12): 64 bins on linspace(0, 10, 64), (count + 0.1) / count.sum()."""
    lengths = path_lengths(traj)
    bars = linspace_f32(0.0, 10.0, 64)
    idx = np.searchsorted(bars, lengths, side="left")
    if idx.max() >= 64:
        raise IndexError("trajectory longer than 10")                       # length_dist[idx] += 1 with idx == 64
    count = np.zeros(64, dtype=f32)
    np.add.at(count, idx, f32(1))
    return ((count + f32(0.1)) / count.sum(dtype=f32)).astype(f32)


def jsd(P, Q) -> float:
    """JSD (cell 12), torch.nn.functional.kl_div(log-input, target, reduction='batchmean') as the reference calls it."""
    P, Q = torch.as_tensor(P), torch.as_tensor(Q)
    avg = 0.5 * (P + Q)
    kp = torch.nn.functional.kl_div(P.log(), avg, reduction="batchmean")
    kq = torch.nn.functional.kl_div(Q.log(), avg, reduction="batchmean")
    return (0.5 * (kp + kq)).item()
