"""Graph-captured sampling loop for TrajUNet and SDUNet_CAR (the loop of reference DiffusionManager.py:66-79).

One denoise step = {step bookkeeping, the UNet program, the fused posterior update} is captured
ONCE into a HIP graph; the step index lives in device memory (tdx_step_prepare advances it), so the
same graph is replayed for every t with no host-side arguments changing.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, engine
from ._lib import TdxError


class _Shard:
    """One slice of the batch with its own program, step counter and stream."""
    __slots__ = ("lo", "n", "prog", "key", "ctr", "coef_cur", "zbuf", "stream", "first_elem")


def auto_shards(B: int, low_latency: bool) -> int:
    """Batch slices that run as concurrent streams inside one denoise step.  The conv GEMMs are bound by LDS / MFMA issue and occupy
    one workgroup per CU; the GroupNorm, attention and update passes between them are HBM streams that need neither.  With two
    slices on two streams the HBM passes of one slice run under the GEMMs of the other instead of in series with them.  Slices are
    independent (no cross-trajectory op) and draw the Philox noise of their global sample indices, so the result is bit-identical
    to the unsliced run.  Small batches stay whole: their GEMMs need every tile they have."""
    env = os.environ.get("TDX_SAMPLER_SHARDS")
    if env:
        return max(1, int(env))
    return 2 if (not low_latency and B % 2 == 0 and B >= 2048) else 1


class GraphSampler:
    def __init__(self, dm, model, B: int, L: int, device: torch.device, max_t: int, ddim_steps: Optional[int] = None,
                 shards: Optional[int] = None):
        self.dm, self.model, self.B, self.L, self.device = dm, model, B, L, device
        self.lib = _lib.lib()
        if shards is None:   # a model may pin its slice count (SDUNet_CAR: its GEMMs at 3-24 rows per trajectory lose more from halving than the overlap returns)
            shards = getattr(model, "sampler_slices", None) if not os.environ.get("TDX_SAMPLER_SHARDS") else None
        ns = auto_shards(B, getattr(model, "low_latency", False)) if shards is None else int(shards)
        if ns < 1 or B % ns:
            raise TdxError(f"sampler: {ns} shards do not divide a batch of {B}")
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
        self.tbias_table = model.time_bias_table(ts.to(device))       # None: the model has no per-step time-bias row (SDUNet_CAR)
        self._epoch = model.weights_epoch
        self.shards = []
        nb = B // ns
        for k in range(ns):
            sh = _Shard()
            sh.lo, sh.n = k * nb, nb
            sh.prog = model.program(nb, L, device, split_embed=True, slot=k)
            sh.key = (nb, L, model.compute_dtype, device, True, model.low_latency) + ((k,) if k else ())
            sh.ctr = torch.zeros(4, dtype=torch.int32, device=device)  # {step, seed_lo, seed_hi, Philox stream id}
            sh.coef_cur = torch.zeros(4, dtype=torch.float32, device=device)
            sh.stream = torch.cuda.Stream(device=device) if ns > 1 else None
            sh.first_elem = 0
            self.shards.append(sh)
        self.prog = self.shards[0].prog
        self.D = self.prog.x.shape[1]                                 # trajectory channels: 2 (TrajUNet) or 4 (latents)
        for sh in self.shards:
            sh.zbuf = torch.zeros((nb, self.D, L), dtype=torch.float32, device=device)
        self.graphs = {}
        self.first_elem = 0

    def close(self) -> None:
        """Drop the captured graphs at a defined point: after everything they may still be executing has finished, and never from inside
        another stream capture (DiffusionManager.release_samplers calls this when a sampler is replaced; a sampler that the cyclic
        garbage collector frees at some arbitrary later moment is what round 3 saw crash the host inside a later hipGraphLaunch)."""
        if self.graphs:
            try:
                if not torch.cuda.is_current_stream_capturing():
                    torch.cuda.synchronize(self.device)
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass
            self.graphs.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def valid(self) -> bool:
        """False when the model dropped this sampler's programs (parameters moved, dtype/mode changed).  Weight VALUES may
        change freely: the packed copies are refreshed in place (also after writes through `.data`, which only the norm
        digest sees) and the per-step time rows are re-derived."""
        self.model.packed_weights(self.model.compute_dtype, check_data=True)
        if any(self.model._programs.get(sh.key) is not sh.prog for sh in self.shards):
            return False
        if self.model.weights_epoch != self._epoch:
            if self.tbias_table is not None:
                self.tbias_table.copy_(self.model.time_bias_table(self.ts.to(self.device)))
            self._epoch = self.model.weights_epoch
        return True

    def _step_shard(self, sh: _Shard, inject: bool) -> None:
        p, st = sh.prog, _lib.stream_ptr()
        _lib.check(self.lib.tdx_step_prepare(sh.ctr.data_ptr(), self.coef_table.data_ptr(), 4, sh.coef_cur.data_ptr(),
                                             _lib.ptr(self.tbias_table), self.model.embed_cols, _lib.ptr(p.tbias),
                                             self.step_to_t.data_ptr(), p.time.data_ptr(), sh.n, self.nsteps, st), "tdx_step_prepare")
        p.run()
        n = p.x.numel()
        if self.ddim:
            _lib.check(self.lib.tdx_ddim_update(p.x.data_ptr(), p.eps.data_ptr(), sh.coef_cur.data_ptr(), n, st), "tdx_ddim_update")
        else:
            _lib.check(self.lib.tdx_ddpm_update(p.x.data_ptr(), p.eps.data_ptr(), sh.zbuf.data_ptr() if inject else None,
                                                sh.coef_cur.data_ptr(), n, 0, sh.ctr.data_ptr(), sh.first_elem, st),
                       "tdx_ddpm_update")

    def _step(self, inject: bool) -> None:
        if len(self.shards) == 1 or os.environ.get("TDX_SAMPLER_SERIAL"):   # (serial: the slices one after the other on one stream, for kernel traces)
            for sh in self.shards:
                self._step_shard(sh, inject)
            return
        cur = torch.cuda.current_stream()
        for sh in self.shards:     # fork: each slice's chain of launches on its own stream (also under graph capture: parallel branches)
            sh.stream.wait_stream(cur)
            with torch.cuda.stream(sh.stream):
                self._step_shard(sh, inject)
        for sh in self.shards:     # join
            cur.wait_stream(sh.stream)

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
        for t, name in ((x_T, "x_T"), (cat_attr, "cat_attr"), (num_attr, "num_attr")):
            _lib.require_device(t, name)
        if tuple(x_T.shape) != (self.B, self.D, self.L):
            raise TdxError(f"sampler built for {(self.B, self.D, self.L)}, got x_T {tuple(x_T.shape)}")
        inject = noise is not None and not self.ddim
        # Garbage that holds captured graphs or device buffers (programs of dropped models, replaced samplers: they sit in reference
        # cycles) is destroyed HERE, at a defined point, and the cyclic collector stays off while graphs are captured and replayed:
        # round 3 saw a host-side crash inside hipGraphLaunch that went away once no destructor could run in the middle of the loop.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._sample(x_T, cat_attr, num_attr, noise, inject, first_sample, use_graph, trace)
        finally:
            if gc_was_on:
                gc.enable()

    def _sample(self, x_T, cat_attr, num_attr, noise, inject, first_sample, use_graph, trace) -> torch.Tensor:
        self.first_elem = int(first_sample) * self.D * self.L
        for sh in self.shards:
            sh.first_elem = (int(first_sample) + sh.lo) * self.D * self.L
            if sh.first_elem % 4:
                raise TdxError("first_sample * channels * L must be a multiple of 4 (for every batch slice)")
            p = sh.prog
            p.cat.copy_(cat_attr[sh.lo:sh.lo + sh.n])   # (before the capture: a model that gathers its attributes inside the step must not see stale indices)
            p.num.copy_(num_attr[sh.lo:sh.lo + sh.n])
            p.time.zero_()
        graph = self._graph(inject) if use_graph else None  # (its warm-up step scribbles on x / ctr: load x after)
        seed = self.dm._draw_seed()
        words = torch.tensor(self.dm.seed_words(seed), dtype=torch.int32)
        for sh in self.shards:
            p = sh.prog
            p.x.copy_(x_T[sh.lo:sh.lo + sh.n])
            if p.embed_pg is not None:
                p.embed_pg.run()  # attribute half of every embed_proj, once per call
            engine.check_err_flag(p.err, "diffusionBackward")
            sh.ctr.copy_(words)
        for s in range(self.nsteps):
            t = int(self.ts[s])
            if inject and t > 0:
                z = noise(t) if callable(noise) else noise[t]
                for sh in self.shards:
                    sh.zbuf.copy_(z[sh.lo:sh.lo + sh.n])
            if trace is not None:
                x_before = torch.cat([sh.prog.x for sh in self.shards]).clone()
            if graph is not None:
                graph.replay()
            else:
                self._step(inject)
            if trace is not None:
                trace.append((t, x_before, torch.cat([sh.prog.eps for sh in self.shards]).clone()))
        for sh in self.shards:
            if sh.prog.embed_pg is None:   # attribute gathers ran inside the steps: report their range errors now
                engine.check_err_flag(sh.prog.err, "diffusionBackward")
        return torch.cat([sh.prog.x for sh in self.shards]) if len(self.shards) > 1 else self.prog.x.clone()
