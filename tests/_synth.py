"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

Everything is a pure function of (name, shape, seed) through numpy's PCG64, so the GPU box can
rebuild exactly the tensors the golden vectors were produced with, without the reference and
without depending on torch's RNG stream.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

FULL_SCHEDULE = [128, 128, 256, 512, 1024]


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """fp32 parameter values: GN/embedding tables O(1), conv/linear weights ~U(-1/sqrt(fan_in), +)."""
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = len(shape) == 1 and any(s in name for s in (".stage1.0.", ".stage2.0.", "qkv_getter.0.", "head.0."))
    if is_norm and leaf == "weight":
        a = 1.0 + 0.2 * g.standard_normal(shape)
    elif is_norm and leaf == "bias":
        a = 0.1 * g.standard_normal(shape)
    elif "embedder" in name:
        a = g.standard_normal(shape)
    elif leaf == "bias":
        a = 0.05 * g.standard_normal(shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        bound = (3.0 / fan_in) ** 0.5  # unit-gain uniform
        a = g.uniform(-bound, bound, size=shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state_dict(spec: List[Tuple[str, Sequence[int]]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {name: synth_tensor(name, shape, seed) for name, shape in spec}


def synth_inputs(B: int, L: int, T: int, seed: int = 0, same_t: int = -1):
    """x (B,2,L) fp32 ~N(0,1); time (B,) int64; cat (B,5) int64; num (B,3) fp32."""
    g = _rng(f"inputs/{B}/{L}/{T}", seed)
    x = torch.from_numpy(g.standard_normal((B, 2, L)).astype(np.float32))
    if same_t >= 0:
        t = torch.full((B,), same_t, dtype=torch.long)
    else:
        t = torch.from_numpy(g.integers(0, T, size=(B,)).astype(np.int64))
    cat = np.empty((B, 5), dtype=np.int64)
    cat[:, :4] = g.integers(0, 256, size=(B, 4))
    cat[:, 4] = g.integers(1, 32, size=(B,))
    num = np.empty((B, 3), dtype=np.float32)
    num[:, 0] = g.uniform(0, 1, size=B)
    num[:, 1] = np.abs(g.normal(0.6, 0.2, size=B))
    num[:, 2] = num[:, 1] / (L - 1)
    return x, t, torch.from_numpy(cat), torch.from_numpy(num)


def synth_noise(step: int, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """The z injected at denoise step ``step`` (t > 0)."""
    g = _rng(f"noise/{step}", seed)
    return torch.from_numpy(g.standard_normal(tuple(shape)).astype(np.float32))


def synth_raw_trajectories(n: int, L: int, seed: int = 0):
    """Raw Didi-like records: lon/lat float64 degrees (N,2) random walk near Xi'an, unix seconds (N,) int64."""
    g = _rng(f"didi/{n}/{L}", seed)
    out = []
    for i in range(n):
        N = L + int(g.integers(0, 40))
        start = np.array([108.95, 34.24]) + g.normal(0, 0.02, size=2)
        steps = g.normal(0, 2.0e-4, size=(N, 2))
        steps[0] = 0
        ll = start + np.cumsum(steps, axis=0)
        t0 = 1477958400 + int(g.integers(0, 29 * 86400))
        ts = t0 + np.cumsum(g.integers(2, 5, size=N)).astype(np.int64)
        out.append((f"driver{i}", f"order{i}", torch.from_numpy(ll.astype(np.float64)), torch.from_numpy(ts)))
    return out


def sample_view(t: torch.Tensor, n: int = 2048) -> np.ndarray:
    """A deterministic strided sample of a tensor (golden files keep these instead of whole activations)."""
    flat = t.detach().reshape(-1).to(torch.float32)
    stride = max(1, flat.numel() // n)
    return flat[::stride][:n].numpy().copy()


def synth_metric_trajectories(B: int, L: int, seed: int):
    """Two (B, 2, L) fp32 sets ("real" and "generated") of random walks in z-scored lon/lat units for the evaluation metrics;
    step sizes spread the path lengths over the reference's 0..10 histogram range without leaving it."""
    import numpy as np
    import torch
    sets = []
    for k in range(2):
        rng = np.random.Generator(np.random.PCG64(1000 * seed + k))
        start = rng.normal(0.0, 1.0 + 0.2 * k, size=(B, 2, 1))
        scale = rng.uniform(0.2, 8.0, size=(B, 1, 1)) / (1.6 * (L - 1))
        steps = rng.normal(0.0, 1.0, size=(B, 2, L - 1)) * scale
        sets.append(torch.from_numpy(np.concatenate([start, start + np.cumsum(steps, axis=2)], axis=2).astype(np.float32)))
    return sets[0], sets[1]
