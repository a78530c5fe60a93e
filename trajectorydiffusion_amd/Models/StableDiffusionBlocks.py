"""Blocks of the latent denoiser SDUNet_CAR with the reference's constructor signatures and state_dict keys
(reference Models/StableDiffusionBlocks.py:43-142), executed by HIP kernels.

As in TrajUNetBlocks.py the torch.nn leaf modules are parameter containers only (names, shapes, default initialisation);
their forward() is never called.  Only the two blocks SDUNet_CAR constructs are provided (CrossAttentionRes, ConditionMixBlock).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _lib, engine
from .._lib import ACT_LEAKY, ACT_NONE, ACT_SWISH, TdxError
from ..engine import Seg, fp32_param
from .TrajUNetBlocks import _HipBlock

N_COND = 7   # condition tokens: time, 4 cells, day, numerical (StableDiffusionBlocks.py:54-55,140)


class ConditionMixBlock(_HipBlock):
    """Time + attribute condition vector (reference :85-142).  forward -> (B, 7 * embed_dim)."""

    def __init__(self, max_time: int, hidden_dim: int = 256, embed_dim: int = 128) -> None:
        super().__init__()
        if hidden_dim % 32 or embed_dim % 4:
            raise TdxError("ConditionMixBlock: hidden_dim must be a multiple of 32 and embed_dim of 4 (GEMM vector widths)")
        position = torch.arange(max_time, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, hidden_dim, 2, dtype=torch.float32) * -(math.log(1.0e4) / hidden_dim))
        table = torch.zeros((max_time, hidden_dim), dtype=torch.float32)
        table[:, 0::2] = torch.sin(position * div_term)
        table[:, 1::2] = torch.cos(position * div_term)
        self.pos_enc = table   # plain attribute, as in the reference (:95): not in state_dict
        hole = nn.Identity
        self.time_embed_layers = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), hole(), nn.Linear(hidden_dim, hidden_dim))
        self.cell_index_embedder = nn.Embedding(256, hidden_dim)
        self.day_embedder = nn.Embedding(33, hidden_dim)
        self.categorical_fc = nn.Linear(5 * hidden_dim, 5 * hidden_dim)
        self.numerical_fc = nn.Linear(3, hidden_dim)
        self.mix_fc = nn.Sequential(hole(), nn.Linear(7 * hidden_dim, 7 * hidden_dim), hole(), nn.Linear(7 * hidden_dim, 7 * embed_dim), hole())
        self.max_time, self.hidden_dim, self.embed_dim = max_time, hidden_dim, embed_dim
        self._progs: Dict[Tuple, Tuple] = {}

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.pos_enc = self.pos_enc.to(self.numerical_fc.weight.device)
        self._progs.clear()
        return self

    def emit(self, pg: engine.Program, B: int, time: torch.Tensor, cat: torch.Tensor, num: torch.Tensor, err: torch.Tensor,
             out: torch.Tensor) -> None:
        """Record forward() into `pg`: out (B, 7 * embed_dim) fp32.  All fp32 whatever the UNet's compute dtype.
        The LeakyReLU that opens mix_fc (:116) acts on the concatenation [time | categorical | numerical] (:140): each of the
        three producers applies it to its own slice of the shared (B, 7H) buffer.  The categorical slice is the gathered cell / day
        embeddings AS THEY ARE (:134-136): the reference constructs `categorical_fc` (:109) and never calls it, so it is kept here as
        a state_dict container only and changing its weights changes nothing."""
        H, E, f32, F32 = self.hidden_dim, self.embed_dim, torch.float32, _lib.TDX_F32
        pe = pg.alloc((B, H), f32)
        mix_in = pg.alloc((B, 7 * H), f32)
        pg.call("cond.gather", pg.lib.tdx_condmix_gather, time.data_ptr(), cat.data_ptr(), num.data_ptr(), self.pos_enc.data_ptr(), self.max_time,
                fp32_param(self.cell_index_embedder.weight, "cell").data_ptr(), fp32_param(self.day_embedder.weight, "day").data_ptr(),
                fp32_param(self.numerical_fc.weight, "num_w").data_ptr(), fp32_param(self.numerical_fc.bias, "num_b").data_ptr(),
                pe.data_ptr(), None, mix_in.data_ptr(), H, B, err.data_ptr(), keep=[time, cat, num, self.pos_enc, err, pe, mix_in])

        def lin(src, k, mod, name, dst, dst_ld, off, act):
            pg.conv([Seg(src, k, k, 1, taps=1)], fp32_param(mod.weight, name + ".weight"), fp32_param(mod.bias, name + ".bias"), dst, dst_ld, B, 1,
                    mod.weight.shape[0], out_off=off, act=act, dtype_code=F32, name="cond." + name)

        h1 = pg.alloc((B, H), f32)
        lin(pe, H, self.time_embed_layers[0], "time0", h1, H, 0, ACT_LEAKY)
        lin(h1, H, self.time_embed_layers[2], "time2", mix_in, 7 * H, 0, ACT_LEAKY)
        m1 = pg.alloc((B, 7 * H), f32)
        lin(mix_in, 7 * H, self.mix_fc[1], "mix1", m1, 7 * H, 0, ACT_LEAKY)
        lin(m1, 7 * H, self.mix_fc[3], "mix3", out, 7 * E, 0, ACT_SWISH)
        for t in (pe, mix_in, h1, m1):
            pg.release(t)

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
            out = pg.alloc((B, N_COND * self.embed_dim), torch.float32)
            self.emit(pg, B, t, c, n, err, out)
            self._progs[key] = (pg, t, c, n, err, out)
        pg, t, c, n, err, out = self._progs[key]
        t.copy_(time)
        c.copy_(cat_attr)
        n.copy_(num_attr.to(torch.float32))
        pg.run()
        engine.check_err_flag(err, "ConditionMixBlock")
        return out.clone()


class CrossAttentionRes(_HipBlock):
    """GN-Swish-Conv3, cross-attention to the 7 condition tokens (+ residual), GN-Swish-Conv3, + shortcut (reference :43-82)."""

    def __init__(self, embed_dim: int, out_dim: int, num_heads: int, dropout: float = 0.1, norm_groups: int = 32) -> None:
        super().__init__()
        self.in_c, self.out_c, self.num_heads, self.norm_groups = embed_dim, out_dim, num_heads, norm_groups
        hole = nn.Identity
        self.stage1 = nn.Sequential(nn.GroupNorm(norm_groups, embed_dim, eps=1e-6), hole(), nn.Conv1d(embed_dim, out_dim, 3, padding=1))
        self.cond_proj = nn.Sequential(nn.Linear(N_COND * 128, N_COND * out_dim), hole())
        # parameter container only (in_proj_weight (3C, C), in_proj_bias, out_proj.{weight,bias}); dropout is inactive in eval
        self.multihead_attn = nn.MultiheadAttention(out_dim, num_heads, dropout=dropout, batch_first=True)
        self.stage2 = nn.Sequential(nn.GroupNorm(norm_groups, out_dim, eps=1e-6), hole(), nn.Conv1d(out_dim, out_dim, 3, padding=1))
        self.has_shortcut = embed_dim != out_dim
        self.shortcut = nn.Conv1d(embed_dim, out_dim, 1, padding=0) if self.has_shortcut else nn.Identity()
        self._progs: Dict[Tuple, Tuple] = {}

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._progs.clear()
        return self

    # ---- weights in GEMM layout ------------------------------------------------------------------------------------------------
    def pack(self, key: str, dtype: torch.dtype, out: Dict[str, torch.Tensor]) -> None:
        c = self.out_c
        mh = self.multihead_attn
        out[key + "conv1.w"] = engine.pack_conv_weight(self.stage1[2].weight, dtype)
        out[key + "q.w"] = mh.in_proj_weight.detach()[:c].to(dtype).contiguous()
        out[key + "q.b"] = mh.in_proj_bias.detach()[:c].float().contiguous()
        # cond_proj (:53-56,75) and the K | V input projections of the attention (:77) are both linear in the condition vector with
        # nothing in between: one composite map per token n, kv[b, n, :] = (W_kv W_cp[n]) cond[b] + (W_kv b_cp[n] + b_kv), formed in
        # fp64 and rounded once.  14 C x 896 instead of (7 C x 896) + (2 C x C): fewer FLOPs from C = 512 up, one GEMM instead of two,
        # and every block's map can be stacked into ONE GEMM per denoise step (SDUNet_CAR).
        wc = self.cond_proj[0].weight.detach().double().view(N_COND, c, -1)
        bc = self.cond_proj[0].bias.detach().double().view(N_COND, c)
        wkv, bkv = mh.in_proj_weight.detach()[c:].double(), mh.in_proj_bias.detach()[c:].double()
        out[key + "kvc.w"] = torch.einsum("jc,nck->njk", wkv, wc).reshape(N_COND * 2 * c, -1).to(dtype).contiguous()
        out[key + "kvc.b"] = (torch.einsum("jc,nc->nj", wkv, bc) + bkv).reshape(-1).float().contiguous()
        out[key + "o.w"] = mh.out_proj.weight.detach().to(dtype).contiguous()
        w2 = engine.pack_conv_weight(self.stage2[2].weight, dtype)
        if self.has_shortcut:
            out[key + "conv2.w"] = torch.cat([w2, engine.pack_conv_weight(self.shortcut.weight, dtype)], dim=1).contiguous()
            out[key + "conv2.b"] = (self.stage2[2].bias.detach() + self.shortcut.bias.detach()).float().contiguous()
        else:
            out[key + "conv2.w"] = w2

    # ---- program ---------------------------------------------------------------------------------------------------------------
    def emit(self, pg: engine.Program, packed: Dict[str, torch.Tensor], key: str, srcs: Sequence[Tuple[torch.Tensor, int]],
             cond: Optional[torch.Tensor], B: int, L: int, kv_all: Optional[Tuple[torch.Tensor, int]] = None) -> torch.Tensor:
        """Record forward() over channels-last rows.  srcs = the raw input as [(tensor (B, L, C_k), C_k), ...] (two entries = the
        decoder's channel concat, StableDiffusion.py:200, never materialised).  The K | V rows of the 7 condition tokens come either
        from kv_all = (matrix (B, cols), first column of this block's 14 C columns), filled by the caller's stacked GEMM, or from
        this block's own composite GEMM on cond (B, 896) in the program's dtype.  Returns the raw output (B, L, out_c)."""
        cin, c = sum(ck for _, ck in srcs), self.out_c
        if cin != self.in_c:
            raise TdxError(f"{key}: sources carry {cin} channels, block expects {self.in_c}")
        if self.num_heads != 8 or c not in (128, 256, 512, 1024):
            raise TdxError(f"{key}: tdx_cross_attention handles 8 heads and 128/256/512/1024 channels (got {self.num_heads}, {c})")
        gn1, conv1, gn2, conv2 = self.stage1[0], self.stage1[2], self.stage2[0], self.stage2[2]
        mh = self.multihead_attn
        # stage1: Swish(GroupNorm(x)) over the (virtually concatenated) input, then the k=3 conv (:74)
        a1 = pg.alloc((B, L, cin))
        off = 0
        for src, ck in srcs:
            if ck % (cin // self.norm_groups):
                raise TdxError(f"{key}: a {ck}-channel source does not hold whole GroupNorm groups")
            pg.groupnorm(src, ck, a1, cin, off, fp32_param(gn1.weight, key + "stage1.0.weight"), fp32_param(gn1.bias, key + "stage1.0.bias"), off,
                         B, L, ck, ck // (cin // self.norm_groups), gn1.eps, ACT_SWISH, name=key + "gn1")
            off += ck
        h = pg.alloc((B, L, c))
        pg.conv([Seg(a1, cin, cin, L)], packed[key + "conv1.w"], fp32_param(conv1.bias, key + "stage1.2.bias"), h, c, B, L, c, name=key + "conv1")
        pg.release(a1)
        # K | V rows of the condition tokens (:75,77)
        if kv_all is None:
            kvm, kv_off = pg.alloc((B, N_COND * 2 * c)), 0
            pg.conv([Seg(cond, cond.shape[1], cond.shape[1], 1, taps=1)], packed[key + "kvc.w"], packed[key + "kvc.b"], kvm, N_COND * 2 * c, B, 1,
                    N_COND * 2 * c, name=key + "kv_proj")
        else:
            kvm, kv_off = kv_all
        # q projection, attention core, out projection + residual (:77)
        q = pg.alloc((B, L, c))
        pg.conv([Seg(h, c, c, L, taps=1)], packed[key + "q.w"], packed[key + "q.b"], q, c, B, L, c, name=key + "q_proj")
        att = pg.alloc((B, L, c))
        pg.call(key + "attn", pg.lib.tdx_cross_attention, q.data_ptr(), c, kvm.data_ptr() + kv_off * kvm.element_size(), 2 * c, kvm.shape[1],
                att.data_ptr(), c, B, L, N_COND, c, self.num_heads, pg.code, keep=[q, kvm, att])
        pg.release(q)
        if kv_all is None:
            pg.release(kvm)
        # out projection + residual (:77), with stage2's GroupNorm + Swish (:80) as the GEMM's consumer: x2 itself is never needed
        a2 = pg.alloc((B, L, c))
        pg.conv([Seg(att, c, c, L, taps=1)], packed[key + "o.w"], fp32_param(mh.out_proj.bias, key + "multihead_attn.out_proj.bias"), None, c, B, L, c,
                res=h, res_ld=c, name=key + "out_proj", gn=engine.GnSpec(gn2, key + "stage2.0", 0, c // self.norm_groups, ACT_SWISH, a2, c, 0))
        pg.release(att)
        pg.release(h)
        # stage2's conv + shortcut (:72,80,82)
        out = pg.alloc((B, L, c))
        segs = [Seg(a2, c, c, L)]
        if self.has_shortcut:
            segs += [Seg(src, ck, ck, L, taps=1) for src, ck in srcs]
            pg.conv(segs, packed[key + "conv2.w"], packed[key + "conv2.b"], out, c, B, L, c, name=key + "conv2")
        else:
            pg.conv(segs, packed[key + "conv2.w"], fp32_param(conv2.bias, key + "stage2.2.bias"), out, c, B, L, c, res=srcs[0][0], res_ld=cin,
                    name=key + "conv2")
        pg.release(a2)
        return out

    def forward(self, x: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """x (B, embed_dim, L), cond (B, 896) -> (B, out_dim, L) (stand-alone use; SDUNet_CAR records emit() into its own program)."""
        self._guard(x, cond)
        B, _, L = x.shape
        dt = self.compute_dtype
        key = (B, L, dt, x.device)
        if key not in self._progs:
            pg = engine.Program(x.device, dt)
            packed: Dict[str, torch.Tensor] = {}
            with torch.no_grad():
                self.pack("", dt, packed)
            xin = pg.alloc((B, self.in_c, L), torch.float32)
            cin = pg.alloc((B, cond.shape[1]), torch.float32)
            xr = pg.alloc((B, L, self.in_c))
            cr = pg.alloc((B, cond.shape[1]))
            pg.call("x.rows", pg.lib.tdx_chanfirst_to_rows, xin.data_ptr(), xr.data_ptr(), B, self.in_c, L, self.in_c, pg.code)
            pg.call("cond.cast", pg.lib.tdx_chanfirst_to_rows, cin.data_ptr(), cr.data_ptr(), B, cond.shape[1], 1, cond.shape[1], pg.code)
            o = self.emit(pg, packed, "", [(xr, self.in_c)], cr, B, L)
            of = pg.alloc((B, self.out_c, L), torch.float32)
            pg.call("out.chanfirst", pg.lib.tdx_rows_to_chanfirst, o.data_ptr(), self.out_c, of.data_ptr(), B, self.out_c, L, pg.code, keep=[o])
            self._progs[key] = (pg, xin, cin, of, self._weights_key())
        pg, xin, cin, of, wk = self._progs[key]
        if wk != self._weights_key():   # a stand-alone block keeps no refreshable weight cache: rebuild
            del self._progs[key]
            return self.forward(x, cond)
        xin.copy_(x)
        cin.copy_(cond)
        pg.run()
        return of.clone()

    def _weights_key(self):
        ps = list(self.parameters())
        return sum(p._version for p in ps), tuple(p.data_ptr() for p in ps)
