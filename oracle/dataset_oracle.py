"""CPU restatement of DidiTrajectoryDataset's per-item math (test oracle; see oracle/__init__.py)."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import scipy.stats as stats
import torch

MINUTES_PER_DAY = 1440


def gaussian_cell_boundaries(n_segments: int = 256) -> torch.Tensor:
    """Dataset/DidiDataset.py:92-114: [-inf, ppf(1/n) .. ppf((n-1)/n), +inf] as fp32 (n+1 values)."""
    b = [-np.inf] + [stats.norm(0, 1).ppf((i + 1) / n_segments) for i in range(n_segments)]
    b[-1] = np.inf
    return torch.tensor(b, dtype=torch.float32)


def didi_item(lon_lat: torch.Tensor, minutes: torch.Tensor, feature_mean: Sequence[float],
              feature_std: Sequence[float]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Dataset/DidiDataset.py:138-174 for one trajectory already cropped and cast by loadNextFiles (:83-84).

    lon_lat (L, 2) fp32 raw degrees, minutes (L,) fp32 since month start.
    Returns lon_lat (2, L) fp32 z-scored, cat (5,) int64, num (3,) fp32, times (L,) fp32 z-scored.
    """
    mean = torch.tensor(feature_mean, dtype=torch.float32).view(1, 3)
    std = torch.tensor(feature_std, dtype=torch.float32).view(1, 3)
    bounds = gaussian_cell_boundaries(256)
    depart_day = torch.div(minutes[0], MINUTES_PER_DAY, rounding_mode="floor").view(1) + 1  # :150
    depart_time = (minutes[0] % MINUTES_PER_DAY) / MINUTES_PER_DAY  # :151
    times = (minutes - mean[:, 0]) / std[:, 0]  # :152
    traj_length = torch.sqrt(torch.sum((lon_lat[1:] - lon_lat[:-1]) ** 2, dim=1)).sum()  # :155
    traj_length = traj_length / std[:, 1:].sum()  # :156
    avg_move = traj_length / (len(lon_lat) - 1)  # :159
    ll = ((lon_lat - mean[:, 1:]) / std[:, 1:]).transpose(0, 1).contiguous()  # :162-163

    def cell(pt):  # :127-135
        return torch.stack([torch.searchsorted(bounds, pt[0]) - 1, torch.searchsorted(bounds, pt[1]) - 1])

    cat = torch.cat([cell(ll[:, 0]), cell(ll[:, -1]), depart_day.to(torch.long)]).to(torch.long)  # :171
    num = torch.tensor([depart_time, traj_length, avg_move]).to(torch.float32)  # :172
    return ll, cat, num, times.to(torch.float32)
