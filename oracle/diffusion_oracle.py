"""CPU restatement of DiffusionManager (test oracle; see oracle/__init__.py)."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


class DiffusionOracle:
    """Schedule + q-sample + DDPM posterior step + sampling loop.

    Follows /root/reference/DiffusionManager.py:6-79.  Noise is *injected* (a callable
    returning the z for each step) so that traces are reproducible without torch's RNG.
    """

    def __init__(self, min_beta: float = 0.0001, max_beta: float = 0.002, max_diffusion_step: int = 100,
                 scaled_linear: bool = False):
        # DiffusionManager.py:7-16: fp32 linspace, sequential fp32 running product
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
        self.betas = betas
        self.alphas = alphas
        self.alpha_bars = alpha_bars
        self.sqrt_alpha_bars = torch.sqrt(alpha_bars)
        self.sqrt_1_minus_alphas_bars = torch.sqrt(1 - alpha_bars)

    def q_sample(self, x0: torch.Tensor, t: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        # DiffusionManager.py:33
        return self.sqrt_alpha_bars[t].view(-1, 1, 1) * x0 + self.sqrt_1_minus_alphas_bars[t].view(-1, 1, 1) * eps

    def posterior_step(self, x_t: torch.Tensor, t: int, eps_pred: torch.Tensor, z: Optional[torch.Tensor]) -> torch.Tensor:
        # DiffusionManager.py:46-55: sigma_t^2 = beta_t, no noise at t == 0
        beta = self.betas[t]
        mu = (x_t - beta / self.sqrt_1_minus_alphas_bars[t] * eps_pred) / torch.sqrt(self.alphas[t])
        if t == 0:
            return mu
        return mu + torch.sqrt(beta) * z

    def sample(self, x_T: torch.Tensor, model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
               noise: Callable[[int], torch.Tensor], max_t: Optional[int] = None,
               trace: Optional[List] = None) -> torch.Tensor:
        # DiffusionManager.py:66-79.  model(x_t, t_vec) -> eps_hat
        if max_t is None:
            max_t = self.T
        B = x_T.shape[0]
        x_t = x_T
        for t in range(max_t - 1, -1, -1):
            t_vec = torch.full((B,), t, dtype=torch.long)
            eps = model(x_t, t_vec)
            if trace is not None:
                trace.append((t, x_t, eps))
            x_t = self.posterior_step(x_t, t, eps, noise(t) if t > 0 else None)
        return x_t

    def ddim_step(self, x_t: torch.Tensor, t: int, t_prev: int, eps_pred: torch.Tensor) -> torch.Tensor:
        """Deterministic DDIM (eta = 0) on the same alpha-bar table.

        NOT in the reference (DiffusionManager.py has DDPM only): parity unpinned by the reference;
        this is the definition the HIP path is checked against.
        """
        ab_t = self.alpha_bars[t]
        ab_p = self.alpha_bars[t_prev] if t_prev >= 0 else torch.tensor(1.0)
        x0 = (x_t - torch.sqrt(1 - ab_t) * eps_pred) / torch.sqrt(ab_t)
        return torch.sqrt(ab_p) * x0 + torch.sqrt(1 - ab_p) * eps_pred
