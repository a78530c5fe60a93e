"""DiffusionManager with the reference's constructor, attributes and four methods
(reference DiffusionManager.py:5-79).  On MI355X the per-step arithmetic runs in ONE fused HIP
kernel (tdx_ddpm_update) and, when the model is this package's TrajUNet, the whole T-step loop is
a single captured HIP graph replayed T times with the step index kept on the device.
"""
from __future__ import annotations

from typing import Callable, Optional, Union

import torch

from . import _lib
from ._lib import TdxError

NoiseArg = Union[None, torch.Tensor, Callable[[int], torch.Tensor]]


class DiffusionManager:
    def __init__(self, min_beta: float = 0.0001, max_beta: float = 0.002, max_diffusion_step: int = 100, device: str = 'cuda',
                 scaled_linear: bool = False):
        # Schedule tables are T scalars: built on the host in fp32 with the reference's exact op order
        # (linspace -> 1-beta -> sequential running product, reference :7-16), then moved to `device`.
        if scaled_linear:
            betas = torch.linspace(min_beta ** 0.5, max_beta ** 0.5, max_diffusion_step) ** 2
        else:
            betas = torch.linspace(min_beta, max_beta, max_diffusion_step)
        alphas = 1 - betas
        alpha_bars = torch.empty_like(alphas)
        product = 1
        for i, alpha in enumerate(alphas):
            product *= alpha
            alpha_bars[i] = product
        self.T = max_diffusion_step
        self.device = torch.device(device)
        to = lambda t: t.to(self.device).view(-1, 1, 1)  # noqa: E731  (T, 1, 1) like the reference
        self.betas = to(betas)
        self.alphas = to(alphas)
        self.alpha_bars = to(alpha_bars)
        self.sqrt_alpha_bars = to(torch.sqrt(alpha_bars))
        self.sqrt_1_minus_alphas_bars = to(torch.sqrt(1 - alpha_bars))
        # per-step coefficient rows for the fused update kernel: {beta/sqrt(1-abar), sqrt(alpha), sqrt(beta)}
        s1m = torch.sqrt(1 - alpha_bars)
        self._coef = torch.stack([betas / s1m, torch.sqrt(alphas), torch.sqrt(betas)], dim=1).contiguous()
        self._coef[0, 2] = 0.0  # t == 0: return mu, no noise (reference :51-52)
        self._coef_dev = self._coef.to(self.device)
        self._err = None
        self._samplers = {}
        self.seed: Optional[int] = None  # fixed Philox seed for in-kernel noise; None -> drawn from torch's CPU generator

    # ---- q-sample (reference :25-34) --------------------------------------------------------------
    def diffusionForward(self, x_0: torch.Tensor, t: torch.Tensor, epsilon: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x_0, "diffusionForward x_0")
        if x_0.dtype != torch.float32 or epsilon.dtype != torch.float32:
            raise TdxError("diffusionForward: x_0 and epsilon must be fp32")
        lib = _lib.lib()
        x0 = x_0.contiguous()
        eps = epsilon.contiguous()
        tt = t.to(device=x0.device, dtype=torch.int64).contiguous()
        B = x0.shape[0]
        if tt.numel() != B or eps.shape != x0.shape:
            raise TdxError(f"diffusionForward: shapes x_0 {tuple(x0.shape)}, t {tuple(t.shape)}, epsilon {tuple(eps.shape)}")
        out = torch.empty_like(x0)
        err = self._err_flag(x0.device)
        _lib.check(lib.tdx_q_sample(x0.data_ptr(), eps.data_ptr(), tt.data_ptr(), self.sqrt_alpha_bars.data_ptr(),
                                    self.sqrt_1_minus_alphas_bars.data_ptr(), self.T, out.data_ptr(), B, x0.numel() // B,
                                    err.data_ptr(), _lib.stream_ptr()), "tdx_q_sample")
        if int(err.item()):
            err.zero_()
            raise IndexError("diffusionForward: t out of [0, T)")
        return out

    def _err_flag(self, device) -> torch.Tensor:
        if self._err is None or self._err.device != device:
            self._err = torch.zeros(1, dtype=torch.int32, device=device)
        return self._err

    # ---- one posterior step (reference :37-55) -------------------------------------------------------
    def diffusionBackwardStep(self, x_t: torch.Tensor, t: int, epsilon_pred: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_{t-1} from x_t and eps_hat.  `noise` (optional, extension) injects z instead of drawing it."""
        _lib.require_device(x_t, "diffusionBackwardStep x_t")
        t = int(t)
        if not 0 <= t < self.T:
            raise IndexError(f"diffusionBackwardStep: t={t} out of [0, {self.T})")
        out = x_t.to(torch.float32).clone(memory_format=torch.contiguous_format)
        eps = epsilon_pred.to(torch.float32).contiguous()
        z = None
        if noise is not None and t > 0:
            z = noise.to(device=out.device, dtype=torch.float32).contiguous()
        seed = self._draw_seed() if z is None else 0
        # {step, seed_lo, seed_hi, Philox stream id = t}: the same words the graph-replayed loop keeps on the device, so both
        # paths draw the same noise for the same seed
        ctr = torch.tensor(self.seed_words(seed, t), dtype=torch.int32).to(out.device)
        _lib.check(_lib.lib().tdx_ddpm_update(out.data_ptr(), eps.data_ptr(), _lib.ptr(z), self._coef_dev[t].data_ptr(), out.numel(),
                                              0, ctr.data_ptr(), 0, _lib.stream_ptr()), "tdx_ddpm_update")
        return out

    def release_samplers(self) -> None:
        """Drop the cached graph-replayed sampler (workspaces + captured HIP graphs) NOW.  A sampler and this object reference each
        other, so without this call a replaced sampler lives until Python's cyclic collector finds it, and its captured graphs are
        then destroyed at an arbitrary point, possibly in the middle of another graph's capture or launch (round 3: a host-side crash
        inside a later hipGraphLaunch that went away once destruction happened at defined points)."""
        for smp in self._samplers.values():
            smp.close()
        self._samplers = {}

    @staticmethod
    def seed_words(seed: int, t: int = 0):
        """Device words {step, seed_lo, seed_hi, stream} of tdx_ddpm_update's step_ctr for a 62-bit seed."""
        return [int(t), seed & 0x7fffffff, (seed >> 31) & 0x7fffffff, int(t)]

    def _draw_seed(self) -> int:
        """62-bit Philox seed: dm.seed when set (rejected beyond 62 bits: the device words are two 31-bit halves), else
        drawn from torch's CPU generator (follows torch.manual_seed)."""
        if self.seed is not None:
            seed = int(self.seed)
            if not 0 <= seed < (1 << 62):
                raise ValueError(f"DiffusionManager.seed={seed} outside [0, 2**62)")
            return seed
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    # ---- the sampling loop (reference :57-79) ------------------------------------------------------------
    @torch.no_grad()
    def diffusionBackward(self, x_T: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor, model: torch.nn.Module,
                          max_t: int = None, noise: NoiseArg = None, first_sample: int = 0, use_graph: bool = True) -> torch.Tensor:
        """T-step DDPM sampling.  Extensions over the reference signature (all optional):
        noise        injected z: tensor (max_t, B, C, L) indexed by t, or callable t -> (B, C, L); default in-kernel Philox
        first_sample global index of sample 0 (batch shards draw the same noise as the unsharded batch)
        use_graph    capture one denoise step into a HIP graph and replay it (this package's TrajUNet / SDUNet_CAR)
        """
        if max_t is None:
            max_t = self.T
        if not 0 < max_t <= self.T:
            raise IndexError(f"diffusionBackward: max_t={max_t} out of (0, {self.T}]")
        from .Models.StableDiffusion import SDUNet_CAR
        from .Models.TrajUNet import TrajUNet
        if isinstance(model, (TrajUNet, SDUNet_CAR)):
            from .sampler import GraphSampler
            key = (id(model), x_T.shape[0], x_T.shape[2], model.compute_dtype, x_T.device, max_t, "ddpm", getattr(model, "low_latency", False))
            smp = self._samplers.get(key)
            if smp is None or not smp.valid():
                self.release_samplers()       # keep one: a sampler owns GBs of workspace (and captured graphs: released HERE, not by the GC)
                smp = GraphSampler(self, model, x_T.shape[0], x_T.shape[2], x_T.device, max_t)
                self._samplers = {key: smp}
            return smp.sample(x_T, cat_attr, num_attr, noise=noise, first_sample=first_sample, use_graph=use_graph)
        # generic callable model: the reference loop with the fused update kernel
        B = x_T.shape[0]
        x_t = x_T
        tensor_t = torch.arange(max_t, dtype=torch.long, device=x_t.device).repeat(B, 1)
        for t in range(max_t - 1, -1, -1):
            epsilon_pred = model(x_t, tensor_t[:, t], cat_attr, num_attr)
            z = None
            if noise is not None and t > 0:
                z = noise(t) if callable(noise) else noise[t]
            x_t = self.diffusionBackwardStep(x_t, t, epsilon_pred, noise=z)
        return x_t

    # ---- DDIM (eta = 0); not in the reference, SURVEY.md 8(f).1 ---------------------------------------------
    @torch.no_grad()
    def ddimBackward(self, x_T: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor, model: torch.nn.Module,
                     n_steps: int = 50, use_graph: bool = True) -> torch.Tensor:
        from .Models.StableDiffusion import SDUNet_CAR
        from .Models.TrajUNet import TrajUNet
        from .sampler import GraphSampler
        if not isinstance(model, (TrajUNet, SDUNet_CAR)):
            raise TdxError("ddimBackward: only this package's TrajUNet / SDUNet_CAR are supported")
        key = (id(model), x_T.shape[0], x_T.shape[2], model.compute_dtype, x_T.device, n_steps, "ddim", getattr(model, "low_latency", False))
        smp = self._samplers.get(key)
        if smp is None or not smp.valid():
            self.release_samplers()
            smp = GraphSampler(self, model, x_T.shape[0], x_T.shape[2], x_T.device, self.T, ddim_steps=n_steps)
            self._samplers = {key: smp}
        return smp.sample(x_T, cat_attr, num_attr, use_graph=use_graph)
