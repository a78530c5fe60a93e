"""TrajUNet building blocks with the reference's constructor signatures and state_dict keys
(Models/TrajUNetBlocks.py:5-125 of the reference), executed by HIP kernels.

The torch.nn leaf modules below are *parameter containers only* (they give the reference's
parameter names, shapes and default initialisation); none of their forward() methods is ever
called.  Each block's own forward() keeps the reference's channels-first tensor protocol and
runs through trajectorydiffusion_amd.engine on the GPU; there is no CPU path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib, engine
from .._lib import ACT_NONE, TdxError


def _rows(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(B, C, L) channels-first -> (B, L, C) channels-last rows in the compute dtype (layout plumbing)."""
    return x.detach().permute(0, 2, 1).to(dtype).contiguous()


def _chan_first(rows: torch.Tensor) -> torch.Tensor:
    return rows.permute(0, 2, 1).to(torch.float32).contiguous()


class _HipBlock(nn.Module):
    """Shared plumbing: compute dtype selection and the no-fallback guards."""
    compute_dtype: torch.dtype = torch.float32

    def set_compute_dtype(self, dtype: torch.dtype) -> "_HipBlock":
        _lib.dtype_code(dtype)
        for m in self.modules():
            if isinstance(m, _HipBlock):
                m.compute_dtype = dtype
        return self

    def _guard(self, *tensors: torch.Tensor) -> None:
        if self.training:
            raise NotImplementedError(
                f"{type(self).__name__}: a block's stand-alone forward runs in eval mode only; call .eval() first. "
                "Training (dropout + backward kernels) runs through the whole model: TrajUNet.forward in train() mode "
                "(trajectorydiffusion_amd/training.py)")
        for t in tensors:
            _lib.require_device(t, f"{type(self).__name__} input")


class WideAndDeepEmbedBlock(_HipBlock):
    """Time + attribute embedding (reference Models/TrajUNetBlocks.py:5-62). forward -> (B, 2*embed_dim, 1)."""

    def __init__(self, max_time: int, hidden_dim: int = 256, embed_dim: int = 128) -> None:
        super().__init__()
        if hidden_dim != 256 or embed_dim != 128:
            raise TdxError("WideAndDeepEmbedBlock: the HIP kernels are specialised for hidden_dim=256, embed_dim=128 "
                           "(the only configuration TrajUNet constructs, Models/TrajUNet.py:18)")
        # sinusoid table: a plain attribute, not a buffer, exactly like the reference (:12-18) => not in state_dict
        position = torch.arange(max_time, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, embed_dim, 2, dtype=torch.float32) * -(math.log(1.0e4) / embed_dim))
        table = torch.zeros((max_time, embed_dim), dtype=torch.float32)
        table[:, 0::2] = torch.sin(position * div_term)
        table[:, 1::2] = torch.cos(position * div_term)
        self.pos_enc = table
        leaky = nn.Identity  # index placeholders for the parameter-free activations
        self.time_embed_layers = nn.Sequential(nn.Linear(embed_dim, embed_dim), leaky(), nn.Linear(embed_dim, embed_dim))
        self.cell_index_embedder = nn.Embedding(256, hidden_dim)
        self.day_embedder = nn.Embedding(33, hidden_dim)
        self.categorical_fc = nn.Linear(5 * hidden_dim, 3 * hidden_dim)
        self.wide_fc = nn.Linear(3, hidden_dim)
        self.deep_fc = nn.Sequential(leaky(), nn.Linear(4 * hidden_dim, 2 * hidden_dim), leaky(), nn.Linear(2 * hidden_dim, embed_dim))
        self.max_time = max_time
        self._progs: Dict[Tuple, Tuple] = {}

    def _apply(self, fn, *a, **k):  # keep pos_enc on the parameters' device (the reference pins it at import time)
        super()._apply(fn, *a, **k)
        self.pos_enc = self.pos_enc.to(self.wide_fc.weight.device)
        self._progs.clear()
        return self

    def forward(self, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor) -> torch.Tensor:
        self._guard(time, cat_attr, num_attr)
        B = time.shape[0]
        key = (B, time.device)
        if key not in self._progs:
            pg = engine.Program(time.device, torch.float32)
            t = pg.alloc((B,), torch.int64)
            c = pg.alloc((B, 5), torch.int64)
            n = pg.alloc((B, 3), torch.float32)
            err = torch.zeros(1, dtype=torch.int32, device=time.device)
            out = pg.alloc((B, 256), torch.float32)
            engine.emit_embed(pg, self, {}, B, t, c, n, err, out)
            self._progs[key] = (pg, t, c, n, err, out)
        pg, t, c, n, err, out = self._progs[key]
        t.copy_(time)
        c.copy_(cat_attr)
        n.copy_(num_attr.to(torch.float32))
        pg.run()
        engine.check_err_flag(err, "WideAndDeepEmbedBlock")
        return out.clone().unsqueeze(2)


class ResnetBlock(_HipBlock):
    """GN-Swish-Conv3 x2 + embed bias + shortcut (reference Models/TrajUNetBlocks.py:65-99)."""

    def __init__(self, in_c: int, out_c: Optional[int] = None, norm_groups: int = 32, embed_dim: int = 256, dropout: float = 0.1) -> None:
        super().__init__()
        out_c = in_c if out_c is None else out_c
        self.in_c, self.out_c, self.norm_groups, self.embed_dim, self.dropout = in_c, out_c, norm_groups, embed_dim, dropout
        hole = nn.Identity
        self.stage1 = nn.Sequential(nn.GroupNorm(norm_groups, in_c, eps=1e-6), hole(), nn.Conv1d(in_c, out_c, 3, padding=1))
        self.embed_proj = nn.Conv1d(embed_dim, out_c, 1, padding=0)
        self.stage2 = nn.Sequential(nn.GroupNorm(norm_groups, out_c, eps=1e-6), hole(), hole(), nn.Conv1d(out_c, out_c, 3, padding=1))
        self.has_shortcut = in_c != out_c
        self.shortcut = nn.Conv1d(in_c, out_c, 1, padding=0) if self.has_shortcut else nn.Identity()

    def forward(self, x: torch.Tensor, time_attr_embed: torch.Tensor) -> torch.Tensor:
        self._guard(x, time_attr_embed)
        B, _, L = x.shape
        dt = self.compute_dtype
        pg = engine.Program(x.device, dt)
        packed: Dict[str, torch.Tensor] = {}
        engine.pack_resblock(self, "", dt, packed)
        # embed_proj (a k=1 conv on a length-1 tensor, :98) as one fp32 GEMM producing the per-sample bias rows
        e = pg.alloc((B, self.out_c), torch.float32)
        emb = time_attr_embed.detach().reshape(B, self.embed_dim).to(torch.float32).contiguous()
        pg.conv([engine.Seg(emb, self.embed_dim, self.embed_dim, 1, taps=1)], self.embed_proj.weight.detach().reshape(self.out_c, -1),
                self.embed_proj.bias.detach(), e, self.out_c, B, 1, self.out_c, dtype_code=_lib.TDX_F32, name="embed_proj")
        xr = _rows(x, dt)
        gn1 = engine.block_input_gn(pg, self, "", B, L)
        engine.emit_groupnorm(pg, xr, self.in_c, B, L, gn1, "gn1")
        out = engine.emit_resblock(pg, self, packed, "", gn1.out, [(xr, self.in_c)], B, L, e, None, self.out_c, 0)
        pg.run()
        return _chan_first(out)


class AttnBlock(_HipBlock):
    """Res -> single-head self-attention -> Res (reference Models/TrajUNetBlocks.py:102-125)."""

    def __init__(self, in_c: int, norm_groups: int) -> None:
        super().__init__()
        self.in_c, self.norm_groups = in_c, norm_groups
        self.res1 = ResnetBlock(in_c, in_c, norm_groups)
        self.qkv_getter = nn.Sequential(nn.GroupNorm(norm_groups, in_c, eps=1e-6), nn.Conv1d(in_c, in_c * 3, 1, 1, 0))
        self.project_out = nn.Conv1d(in_c, in_c, 1, 1, 0)
        self.res2 = ResnetBlock(in_c, in_c, norm_groups)

    def pack(self, key: str, dtype: torch.dtype, out: Dict[str, torch.Tensor]) -> None:
        engine.pack_resblock(self.res1, key + "res1.", dtype, out)
        engine.pack_resblock(self.res2, key + "res2.", dtype, out)
        out[key + "qkv.w"] = engine.pack_conv_weight(self.qkv_getter[1].weight, dtype)
        out[key + "proj.w"] = engine.pack_conv_weight(self.project_out.weight, dtype)

    def emit(self, pg: engine.Program, packed: Dict[str, torch.Tensor], key: str, x: torch.Tensor, a1: torch.Tensor, B: int, L: int,
             ebias, tbias, e_ld: int, e_offs: Tuple[int, int], next_gn=None) -> torch.Tensor:
        """x raw (B, L, C) and a1 = Swish(GN1_res1(x)); returns the raw output; `next_gn` is applied by res2's last conv."""
        Cn = self.in_c
        gn, qkv_conv = self.qkv_getter[0], self.qkv_getter[1]
        a = pg.alloc((B, L, Cn))
        x1 = engine.emit_resblock(pg, self.res1, packed, key + "res1.", a1, [(x, Cn)], B, L, ebias, tbias, e_ld, e_offs[0],
                                  next_gn=engine.GnSpec(gn, key + "qkv_getter.0", 0, Cn // self.norm_groups, ACT_NONE, a, Cn, 0))
        qkv = pg.alloc((B, L, 3 * Cn))
        pg.conv([engine.Seg(a, Cn, Cn, L, taps=1)], packed[key + "qkv.w"], engine.fp32_param(qkv_conv.bias, key + "qkv.b"), qkv,
                3 * Cn, B, L, 3 * Cn, name=key + "qkv")
        pg.release(a)
        o = pg.alloc((B, L, Cn))
        pg.call(key + "attention", pg.lib.tdx_attention, qkv.data_ptr(), o.data_ptr(), B, L, Cn, pg.code, keep=[qkv, o])
        pg.release(qkv)
        x2 = pg.alloc((B, L, Cn))
        gn_r2 = engine.block_input_gn(pg, self.res2, key + "res2.", B, L)
        pg.conv([engine.Seg(o, Cn, Cn, L, taps=1)], packed[key + "proj.w"], engine.fp32_param(self.project_out.bias, key + "proj.b"),
                x2, Cn, B, L, Cn, res=x1, res_ld=Cn, name=key + "proj", gn=gn_r2)
        pg.release(o)
        pg.release(x1)
        out = engine.emit_resblock(pg, self.res2, packed, key + "res2.", gn_r2.out, [(x2, Cn)], B, L, ebias, tbias, e_ld, e_offs[1],
                                   next_gn=next_gn)
        pg.release(gn_r2.out)
        pg.release(x2)
        return out

    def forward(self, x: torch.Tensor, time_attr_embed: torch.Tensor) -> torch.Tensor:
        self._guard(x, time_attr_embed)
        B, Cn, L = x.shape
        dt = self.compute_dtype
        pg = engine.Program(x.device, dt)
        packed: Dict[str, torch.Tensor] = {}
        self.pack("", dt, packed)
        emb = time_attr_embed.detach().reshape(B, 256).to(torch.float32).contiguous()
        wcat = torch.cat([self.res1.embed_proj.weight.detach().reshape(Cn, -1), self.res2.embed_proj.weight.detach().reshape(Cn, -1)]).contiguous()
        bcat = torch.cat([self.res1.embed_proj.bias.detach(), self.res2.embed_proj.bias.detach()]).contiguous()
        e = pg.alloc((B, 2 * Cn), torch.float32)
        pg.conv([engine.Seg(emb, 256, 256, 1, taps=1)], wcat, bcat, e, 2 * Cn, B, 1, 2 * Cn, dtype_code=_lib.TDX_F32, name="embed_proj")
        xr = _rows(x, dt)
        gn1 = engine.block_input_gn(pg, self.res1, "res1.", B, L)
        engine.emit_groupnorm(pg, xr, Cn, B, L, gn1, "res1.gn1")
        out = self.emit(pg, packed, "", xr, gn1.out, B, L, e, None, 2 * Cn, (0, Cn))
        pg.run()
        return _chan_first(out)
