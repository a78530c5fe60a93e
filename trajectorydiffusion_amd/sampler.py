"""Graph-captured sampling loop for TrajUNet and SDUNet_CAR (the loop of reference DiffusionManager.py:66-79).

One denoise step = {step bookkeeping, the UNet program, the fused posterior update} is captured
ONCE into a HIP graph; the step index lives in device memory (tdx_step_prepare advances it), so the
same graph is replayed for every t with no host-side arguments changing.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, engine
from ._lib import TdxError


class GraphSampler:
    def __init__(self, dm, model, B: int, L: int, device: torch.device, max_t: int, ddim_steps: Optional[int] = None):
        self.dm, self.model, self.B, self.L, self.device = dm, model, B, L, device
        self.lib = _lib.lib()
        self.prog = model.program(B, L, device, split_embed=True)
        self._prog_key = (B, L, model.compute_dtype, device, True, model.low_latency)
        self.ddim = ddim_steps is not None
        ab = dm.alpha_bars.view(-1).cpu()
        if self.ddim:
            if not 0 < ddim_steps <= dm.T:
                raise TdxError(f"ddim steps {ddim_steps} out of (0, {dm.T}]")
            ts = torch.linspace(dm.T - 1, 0, ddim_steps).round().to(torch.int64)
            prev = torch.cat([ts[1:], torch.tensor([-1])])
            ab_t = ab[ts]
            ab_p = torch.where(prev >= 0, ab[prev.clamp(min=0)], torch.ones(()))
            coef = torch.stack([torch.sqrt(1 - ab_t), 1.0 / torch.sqrt(ab_t), torch.sqrt(ab_p), torch.sqrt(1 - ab_p)], dim=1)
        else:
            ts = torch.arange(max_t - 1, -1, -1, dtype=torch.int64)
            coef = torch.cat([dm._coef[ts], torch.zeros(len(ts), 1)], dim=1)
        self.ts = ts
        self.nsteps = len(ts)
        self.coef_table = coef.to(torch.float32).contiguous().to(device)
        self.step_to_t = ts.to(torch.int32).to(device)
        self.D = self.prog.x.shape[1]                                 # trajectory channels: 2 (TrajUNet) or 4 (latents)
        self.tbias_table = model.time_bias_table(ts.to(device))       # None: the model has no per-step time-bias row (SDUNet_CAR)
        self._epoch = model.weights_epoch
        self.coef_cur = torch.zeros(4, dtype=torch.float32, device=device)
        self.ctr = torch.zeros(4, dtype=torch.int32, device=device)  # {step, seed_lo, seed_hi, -}
        self.zbuf = torch.zeros((B, self.D, L), dtype=torch.float32, device=device)
        self.graphs = {}
        self.first_elem = 0

    def valid(self) -> bool:
        """False when the model dropped this sampler's program (parameters moved, dtype/mode changed).  Weight VALUES may
        change freely: the packed copies are refreshed in place (also after writes through `.data`, which only the norm
        digest sees) and the per-step time rows are re-derived."""
        self.model.packed_weights(self.model.compute_dtype, check_data=True)
        if self.model._programs.get(self._prog_key) is not self.prog:
            return False
        if self.model.weights_epoch != self._epoch:
            if self.tbias_table is not None:
                self.tbias_table.copy_(self.model.time_bias_table(self.ts.to(self.device)))
            self._epoch = self.model.weights_epoch
        return True

    def _step(self, inject: bool) -> None:
        p, st = self.prog, _lib.stream_ptr()
        _lib.check(self.lib.tdx_step_prepare(self.ctr.data_ptr(), self.coef_table.data_ptr(), 4, self.coef_cur.data_ptr(),
                                             _lib.ptr(self.tbias_table), self.model.embed_cols, _lib.ptr(p.tbias),
                                             self.step_to_t.data_ptr(), p.time.data_ptr(), self.B, self.nsteps, st), "tdx_step_prepare")
        p.run()
        n = p.x.numel()
        if self.ddim:
            _lib.check(self.lib.tdx_ddim_update(p.x.data_ptr(), p.eps.data_ptr(), self.coef_cur.data_ptr(), n, st), "tdx_ddim_update")
        else:
            _lib.check(self.lib.tdx_ddpm_update(p.x.data_ptr(), p.eps.data_ptr(), self.zbuf.data_ptr() if inject else None,
                                                self.coef_cur.data_ptr(), n, 0, self.ctr.data_ptr(), self.first_elem, st),
                       "tdx_ddpm_update")

    def _graph(self, inject: bool) -> torch.cuda.CUDAGraph:
        key = (inject, self.first_elem)
        g = self.graphs.get(key)
        if g is None:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._step(inject)  # warm-up outside capture (lazy kernel attributes, module loading)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step(inject)
            self.graphs[key] = g
        return g

    def sample(self, x_T: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor, noise=None, first_sample: int = 0,
               use_graph: bool = True, trace=None) -> torch.Tensor:
        p = self.prog
        for t, name in ((x_T, "x_T"), (cat_attr, "cat_attr"), (num_attr, "num_attr")):
            _lib.require_device(t, name)
        if tuple(x_T.shape) != (self.B, self.D, self.L):
            raise TdxError(f"sampler built for {(self.B, self.D, self.L)}, got x_T {tuple(x_T.shape)}")
        inject = noise is not None and not self.ddim
        self.first_elem = int(first_sample) * self.D * self.L
        if self.first_elem % 4:
            raise TdxError("first_sample * channels * L must be a multiple of 4")
        p.cat.copy_(cat_attr)   # (before the capture: a model that gathers its attributes inside the step must not see stale indices)
        p.num.copy_(num_attr)
        p.time.zero_()
        graph = self._graph(inject) if use_graph else None  # (its warm-up step scribbles on x / ctr: load x after)
        p.x.copy_(x_T)
        if p.embed_pg is not None:
            p.embed_pg.run()  # attribute half of every embed_proj, once per call
        engine.check_err_flag(p.err, "diffusionBackward")
        seed = self.dm._draw_seed()
        self.ctr.copy_(torch.tensor(self.dm.seed_words(seed), dtype=torch.int32))
        for s in range(self.nsteps):
            t = int(self.ts[s])
            if inject and t > 0:
                self.zbuf.copy_(noise(t) if callable(noise) else noise[t])
            if trace is not None:
                x_before = p.x.clone()
            if graph is not None:
                graph.replay()
            else:
                self._step(inject)
            if trace is not None:
                trace.append((t, x_before, p.eps.clone()))
        if p.embed_pg is None:   # attribute gathers ran inside the steps: report their range errors now
            engine.check_err_flag(p.err, "diffusionBackward")
        return p.x.clone()
