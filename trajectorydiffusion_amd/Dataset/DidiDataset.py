"""DidiTrajectoryDataset with the reference's constructor / methods / item protocol
(reference Dataset/DidiDataset.py:15-189).  The per-item z-score + attribute derivation runs as ONE
batched HIP kernel (tdx_didi_items) over all loaded trajectories when files are loaded; __getitem__
then just slices device tensors.
"""
from __future__ import annotations

import os
import random
from typing import Any, List, Tuple

import scipy.stats as stats
import torch
import torch.utils.data as data

from .. import _lib
from .._lib import TdxError


def gaussian_cell_boundaries(n_segments: int = 256) -> torch.Tensor:
    """[-inf, Phi^-1(1/n) .. Phi^-1((n-1)/n), +inf] as fp32 (reference :92-114)."""
    dist = stats.norm(0, 1)
    b = [float("-inf")] + [float(dist.ppf((i + 1) / n_segments)) for i in range(n_segments - 1)] + [float("inf")]
    return torch.tensor(b, dtype=torch.float32)


class DidiTrajectoryDataset(data.Dataset):
    Unix_time_20161001 = 1475276400  # constants kept verbatim from the reference (:16-19)
    seconds_per_day = 86400
    minutes_per_day = 1440
    Unix_time_20161101 = 1477958400

    device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

    def __init__(self, dataset_root: str, traj_length: int, feature_mean: List[float], feature_std: List[float]):
        if "oct" in dataset_root:
            self.time_shift = self.Unix_time_20161001
        elif "nov" in dataset_root:
            self.time_shift = self.Unix_time_20161101
        else:
            raise ValueError(f"dataset_root should contain 'oct' or 'nov', but got {dataset_root}")
        self.traj_mean = torch.tensor(feature_mean, dtype=torch.float32, device=self.device).view(1, 3)
        self.traj_std = torch.tensor(feature_std, dtype=torch.float32, device=self.device).view(1, 3)
        self.dataset_root = dataset_root
        self.file_paths = [os.path.join(dataset_root, f) for f in os.listdir(dataset_root) if f.endswith('.pt')]
        self.part_idx = 0
        self.traj_length = traj_length
        self.dataset_part: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.city_cell_boundaries = gaussian_cell_boundaries(256).to(self.device)
        self._items = None

    def loadNextFiles(self, load_n: int) -> bool:
        self.dataset_part = []
        for _ in range(load_n):
            if self.part_idx >= len(self.file_paths):
                self.part_idx = 0
                break
            file_path = self.file_paths[self.part_idx]
            print(f'Loading {file_path}')
            for (_, _, lon_lat, times) in torch.load(file_path, weights_only=True):   # list of (id, id, Tensor, Tensor): nothing in the file is executed
                if lon_lat.shape[0] >= self.traj_length:
                    self.dataset_part.append((lon_lat[:self.traj_length].to(torch.float32),  # fp32 BEFORE normalising (:83)
                                              ((times[:self.traj_length] - self.time_shift) / 60).to(torch.float32)))
            self.part_idx += 1
        self._items = None
        return len(self.dataset_part) > 0

    def shuffleAllFiles(self):
        if self.part_idx != 0:
            raise RuntimeError('You should call loadNextFiles() to load all files before shuffling')
        random.shuffle(self.file_paths)

    def __len__(self):
        return len(self.dataset_part)

    @property
    def n_files(self) -> int:
        return len(self.file_paths)

    def getCellIndex(self, point: torch.Tensor) -> torch.Tensor:
        """City cell index of a normalised point (reference :127-135): point (2,) (lon, lat) -> (2,) long (lon_idx, lat_idx),
        `searchsorted(city_cell_boundaries, v) - 1` per coordinate, by the search __getitem__'s kernel runs (tdx_cell_index)."""
        _lib.require_device(point, "getCellIndex point")
        v = point.reshape(-1)[:2].to(torch.float32).contiguous()
        if v.numel() != 2:
            raise TdxError(f"getCellIndex: point must hold (lon, lat), got shape {tuple(point.shape)}")
        out = torch.empty(2, dtype=torch.long, device=v.device)
        _lib.check(_lib.lib().tdx_cell_index(v.data_ptr(), self.city_cell_boundaries.to(v.device).data_ptr(), out.data_ptr(), 2,
                                             _lib.stream_ptr()), "tdx_cell_index")
        return out

    def normalizeBatch(self, lon_lat_raw: torch.Tensor, minutes: torch.Tensor):
        """lon_lat_raw (B, L, 2) fp32 degrees, minutes (B, L) fp32 -> (lon_lat (B,2,L), cat (B,5) i64, num (B,3), times (B,L))."""
        _lib.require_device(lon_lat_raw, "normalizeBatch lon_lat_raw")
        raw = lon_lat_raw.to(torch.float32).contiguous()
        mins = minutes.to(device=raw.device, dtype=torch.float32).contiguous()
        B, L = mins.shape
        if tuple(raw.shape) != (B, L, 2):
            raise TdxError(f"normalizeBatch: lon_lat_raw {tuple(raw.shape)} vs minutes {tuple(mins.shape)}")
        dev = raw.device
        stats_vec = torch.cat([self.traj_mean.view(-1), self.traj_std.view(-1)]).to(dev).contiguous()
        bounds = self.city_cell_boundaries.to(dev)
        lon_lat = torch.empty((B, 2, L), dtype=torch.float32, device=dev)
        cat = torch.empty((B, 5), dtype=torch.int64, device=dev)
        num = torch.empty((B, 3), dtype=torch.float32, device=dev)
        times = torch.empty((B, L), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().tdx_didi_items(raw.data_ptr(), mins.data_ptr(), stats_vec.data_ptr(), bounds.data_ptr(), lon_lat.data_ptr(),
                                             cat.data_ptr(), num.data_ptr(), times.data_ptr(), B, L, _lib.stream_ptr()), "tdx_didi_items")
        return lon_lat, cat, num, times

    def denormalize(self, lon_lat: torch.Tensor) -> torch.Tensor:
        """Inverse z-score (not in the reference): (B, 2, L) -> degrees (B, L, 2)."""
        _lib.require_device(lon_lat, "denormalize lon_lat")
        x = lon_lat.to(torch.float32).contiguous()
        B, _, L = x.shape
        stats_vec = torch.cat([self.traj_mean.view(-1), self.traj_std.view(-1)]).to(x.device).contiguous()
        out = torch.empty((B, L, 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().tdx_didi_denorm(x.data_ptr(), stats_vec.data_ptr(), out.data_ptr(), B, L, _lib.stream_ptr()), "tdx_didi_denorm")
        return out

    def _materialise(self):
        if self._items is None:
            raw = torch.stack([a for a, _ in self.dataset_part]).to(self.device)
            mins = torch.stack([b for _, b in self.dataset_part]).to(self.device)
            self._items = self.normalizeBatch(raw, mins)
        return self._items

    def __getitem__(self, index: Any) -> Any:
        lon_lat, cat, num, times = self._materialise()
        return lon_lat[index], cat[index], num[index], times[index]


def collectFunc(batch):
    """(reference :182-189) list of items -> (B,2,N), (B,5), (B,3), (B,N)."""
    traj_list, categorial_attr_list, numeric_attr_list, time_list = zip(*batch)
    return (torch.stack(traj_list, dim=0), torch.stack(categorial_attr_list, dim=0),
            torch.stack(numeric_attr_list, dim=0), torch.stack(time_list, dim=0))
