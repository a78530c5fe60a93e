"""VAE_InvTrans, the trajectory <-> latent codec of eval_sd.ipynb, with the reference's constructor, encode / decode and state_dict
keys (reference Models/VAE.py:135-192, blocks Models/VAEBlocks.py:102-121, Models/ModelUtils.py:17-33), executed as static
programs of HIP kernel launches.  fp32 (the codec runs once per sampling call, next to 500 denoiser steps).

encode(x (B, 2, L)) -> (B, 4, L // 8);  decode(z (B, 4, l)) -> (B, 2, l): the reference pools the decoder's 256 positions down to
the LATENT length (`_, _, L = z.shape`, VAE.py:191-192), not to 8 l; that behaviour is kept.
Inference only (BatchNorm uses its running statistics): forward() with reparameterisation belongs to the VAE training notebook,
which is outside the SURVEY section 8 scope.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn

from .. import _lib, engine
from .._lib import ACT_NONE, ACT_SWISH, TdxError
from ..engine import Seg, fp32_param
from ._packed import _PackedModel

F32 = torch.float32


class ConvNormAct(nn.Sequential):
    """Conv1d(bias=False) + BatchNorm1d + Swish (reference Models/ModelUtils.py:17-33): parameter container; the program folds the
    BatchNorm's running statistics into the conv's weight and bias."""

    def __init__(self, in_c: int, out_c: int, k: int, s: int, p: int = 0, d: int = 1, g: int = 1):
        if k != 3 or p != 1 or d != 1 or g != 1 or s not in (1, 2):
            raise TdxError("ConvNormAct: the HIP path covers k=3, p=1, d=1, g=1, s in {1, 2} (every use in VAE_InvTrans)")
        super().__init__(nn.Conv1d(in_c, out_c, k, s, p, d, g, bias=False), nn.BatchNorm1d(out_c), nn.Identity())
        self.stride = s

    def pack(self, key: str, out: Dict[str, torch.Tensor]) -> None:
        conv, bn = self[0], self[1]
        scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        w = conv.weight.detach() * scale[:, None, None]
        pad = (-w.shape[1]) % 4
        if pad:   # input rows are padded with zero channels up to the GEMM's 16-byte vectors
            w = torch.cat([w, w.new_zeros(w.shape[0], pad, w.shape[2])], dim=1)
        out[key + "w"] = engine.pack_conv_weight(w, F32)
        out[key + "b"] = (bn.bias.detach() - bn.running_mean.detach() * scale).float().contiguous()

    def emit(self, pg: engine.Program, packed: Dict[str, torch.Tensor], key: str, x: torch.Tensor, B: int, L: int) -> Tuple[torch.Tensor, int]:
        cin, cout = x.shape[2], self[0].out_channels
        lo = engine.conv_out_len(L, self.stride)
        out = pg.alloc((B, lo, cout), F32)
        pg.conv([Seg(x, cin, cin, L, stride=self.stride)], packed[key + "w"], packed[key + "b"], out, cout, B, lo, cout, act=ACT_SWISH, name=key + "conv")
        return out, lo


class InvTransformerBlock(nn.Module):
    """Self-attention ACROSS CHANNELS (tokens = channels, embedding = sequence positions) + conv residual
    (reference Models/VAEBlocks.py:102-121)."""

    def __init__(self, in_c: int, in_seq_len: int, num_heads: int, dropout: float = 0.1):
        super().__init__()
        self.in_c, self.seq_len, self.num_heads = in_c, in_seq_len, num_heads
        self.attn = nn.MultiheadAttention(embed_dim=in_seq_len, num_heads=num_heads, dropout=dropout, batch_first=True)   # parameters only
        self.layers = nn.Sequential(ConvNormAct(in_c, in_c * 2, 3, 1, 1), nn.Conv1d(in_c * 2, in_c, 3, 1, 1))
        self.final_act = nn.Identity()

    def pack(self, key: str, out: Dict[str, torch.Tensor]) -> None:
        out[key + "in.w"] = self.attn.in_proj_weight.detach().float().contiguous()
        out[key + "o.w"] = self.attn.out_proj.weight.detach().float().contiguous()
        self.layers[0].pack(key + "l0.", out)
        out[key + "l1.w"] = engine.pack_conv_weight(self.layers[1].weight, F32)

    def emit(self, pg: engine.Program, packed: Dict[str, torch.Tensor], key: str, x: torch.Tensor, B: int, L: int) -> torch.Tensor:
        """x rows (B, L, C) fp32 -> rows (B, L, C)."""
        C, E = self.in_c, self.seq_len
        if L != E or x.shape[2] != C:
            raise TdxError(f"{key}: built for (C={C}, L={E}), got rows {tuple(x.shape)}")
        lib, code = pg.lib, _lib.TDX_F32
        xc = pg.alloc((B, C, L), F32)    # the channels-first view: one row per token (= channel)
        pg.call(key + "to_tokens", lib.tdx_rows_to_chanfirst, x.data_ptr(), C, xc.data_ptr(), B, C, L, code, keep=[x, xc])
        qkv = pg.alloc((B, C, 3 * E), F32)
        pg.conv([Seg(xc, E, E, C, taps=1)], packed[key + "in.w"], fp32_param(self.attn.in_proj_bias, key + "attn.in_proj_bias"), qkv, 3 * E, B, C, 3 * E,
                name=key + "in_proj")
        att = pg.alloc((B, C, E), F32)
        pg.call(key + "attn", lib.tdx_mha_small, qkv.data_ptr(), att.data_ptr(), B, C, E, self.num_heads, keep=[qkv, att])
        pg.release(qkv)
        x1c = pg.alloc((B, C, L), F32)   # attn(x, x, x)[0] + x  (:120)
        pg.conv([Seg(att, E, E, C, taps=1)], packed[key + "o.w"], fp32_param(self.attn.out_proj.bias, key + "attn.out_proj.bias"), x1c, E, B, C, E,
                res=xc, res_ld=E, name=key + "out_proj")
        pg.release(att)
        pg.release(xc)
        x1 = pg.alloc((B, L, C), F32)
        pg.call(key + "to_rows", lib.tdx_chanfirst_to_rows, x1c.data_ptr(), x1.data_ptr(), B, C, L, C, code, keep=[x1c, x1])
        pg.release(x1c)
        h, _ = self.layers[0].emit(pg, packed, key + "l0.", x1, B, L)
        y = pg.alloc((B, L, C), F32)     # final_act(x + layers(x))  (:121): the conv adds x, the activation is its own pass
        pg.conv([Seg(h, 2 * C, 2 * C, L)], packed[key + "l1.w"], fp32_param(self.layers[1].bias, key + "layers.1.bias"), y, C, B, L, C, res=x1, res_ld=C,
                name=key + "l1")
        pg.release(h)
        pg.release(x1)
        pg.call(key + "act", lib.tdx_act_fwd, y.data_ptr(), C, y.data_ptr(), C, B * L, C, ACT_SWISH, keep=[y])
        return y


class _Codec:
    """One compiled encode or decode program for a fixed (B, length)."""

    def __init__(self, pg: engine.Program, inp: torch.Tensor, out: torch.Tensor):
        self.pg, self.inp, self.out = pg, inp, out


class VAE_InvTrans(_PackedModel):
    def __init__(self, no_noise_ratio: float = 0) -> None:
        super().__init__()
        self.noise_ratio = 1 - no_noise_ratio   # BaseVAE, VAE.py:7 (training-time reparameterisation only)
        hole = nn.Identity
        self.encoder = nn.Sequential(                       # VAE.py:141-153
            hole(),                                         # nn.Upsample(size=256, mode='linear', align_corners=True)
            ConvNormAct(2, 16, 3, 1, 1),
            ConvNormAct(16, 32, 3, 2, 1), InvTransformerBlock(in_c=32, in_seq_len=128, num_heads=8),
            ConvNormAct(32, 64, 3, 2, 1), InvTransformerBlock(in_c=64, in_seq_len=64, num_heads=8),
            ConvNormAct(64, 128, 3, 2, 1), InvTransformerBlock(in_c=128, in_seq_len=32, num_heads=8))
        self.mu_head = nn.Conv1d(128, 4, 1, 1, 0)           # VAE.py:13-14 (BaseVAE)
        self.logvar_head = nn.Conv1d(128, 4, 1, 1, 0)
        self.decoder = nn.Sequential(                       # VAE.py:155-172
            hole(),                                         # nn.Upsample(size=32, mode='linear', align_corners=True)
            ConvNormAct(4, 128, 3, 1, 1),
            InvTransformerBlock(in_c=128, in_seq_len=32, num_heads=8), hole(), ConvNormAct(128, 64, 3, 1, 1),
            InvTransformerBlock(in_c=64, in_seq_len=64, num_heads=8), hole(), ConvNormAct(64, 32, 3, 1, 1),
            InvTransformerBlock(in_c=32, in_seq_len=128, num_heads=8), hole(), ConvNormAct(32, 16, 3, 1, 1),
            nn.Conv1d(16, 2, 3, 1, 1))
        self._init_cache()

    def _build_packed(self, dtype: torch.dtype) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for name, seq in (("encoder.", self.encoder), ("decoder.", self.decoder)):
                for i, m in enumerate(seq):
                    if isinstance(m, (ConvNormAct, InvTransformerBlock)):
                        m.pack(f"{name}{i}.", out)
            out["mu.w"] = self.mu_head.weight.detach().reshape(4, 128).float().contiguous()
            last = self.decoder[11]   # Conv1d(16 -> 2): two zero output channels pad the GEMM's N to a multiple of 4
            w = torch.cat([last.weight.detach(), last.weight.new_zeros(2, 16, 3)], dim=0)
            out["last.w"] = engine.pack_conv_weight(w, F32)
            out["last.b"] = torch.cat([last.bias.detach(), last.bias.new_zeros(2)]).float().contiguous()
        return out

    # ---- programs --------------------------------------------------------------------------------------------------------------
    def _resample(self, pg, x, B, lin, lout, C, mode, name):
        out = pg.alloc((B, lout, C), F32)
        pg.call(name, pg.lib.tdx_resample_rows, x.data_ptr(), out.data_ptr(), B, lin, lout, C, mode, _lib.TDX_F32, keep=[x, out])
        pg.release(x)
        return out

    def _encode_program(self, B: int, L: int, device) -> _Codec:
        if L < 8:
            raise TdxError(f"VAE_InvTrans.encode: length {L} gives an empty latent (L // 8 == 0)")
        pg = engine.Program(device, F32)
        packed = self.packed_weights(F32)
        lib, code = pg.lib, _lib.TDX_F32
        xin = pg.alloc((B, 2, L), F32)
        r = pg.alloc((B, L, 4), F32)     # two zero channels pad the rows to one 16-byte vector
        pg.call("enc.rows", lib.tdx_chanfirst_to_rows, xin.data_ptr(), r.data_ptr(), B, 2, L, 4, code, keep=[xin, r])
        cur, l = self._resample(pg, r, B, L, 256, 4, 1, "enc.upsample"), 256   # VAE.py:142
        for i in range(1, 8):
            m = self.encoder[i]
            if isinstance(m, ConvNormAct):
                nxt, l = m.emit(pg, packed, f"encoder.{i}.", cur, B, l)
            else:
                nxt = m.emit(pg, packed, f"encoder.{i}.", cur, B, l)
            pg.release(cur)
            cur = nxt
        lat = L // 8
        cur = self._resample(pg, cur, B, l, lat, 128, 2, "enc.pool")           # F.adaptive_avg_pool1d(., L // 8), VAE.py:181
        mu = pg.alloc((B, lat, 4), F32)
        pg.conv([Seg(cur, 128, 128, lat, taps=1)], packed["mu.w"], fp32_param(self.mu_head.bias, "mu_head.bias"), mu, 4, B, lat, 4, name="mu_head")
        out = pg.alloc((B, 4, lat), F32)
        pg.call("enc.chanfirst", lib.tdx_rows_to_chanfirst, mu.data_ptr(), 4, out.data_ptr(), B, 4, lat, code, keep=[mu, out])
        return _Codec(pg, xin, out)

    def _decode_program(self, B: int, lat: int, device) -> _Codec:
        pg = engine.Program(device, F32)
        packed = self.packed_weights(F32)
        lib, code = pg.lib, _lib.TDX_F32
        zin = pg.alloc((B, 4, lat), F32)
        r = pg.alloc((B, lat, 4), F32)
        pg.call("dec.rows", lib.tdx_chanfirst_to_rows, zin.data_ptr(), r.data_ptr(), B, 4, lat, 4, code, keep=[zin, r])
        cur, l = self._resample(pg, r, B, lat, 32, 4, 1, "dec.upsample0"), 32  # VAE.py:156
        for i in range(1, 11):
            m = self.decoder[i]
            if isinstance(m, ConvNormAct):
                nxt, l = m.emit(pg, packed, f"decoder.{i}.", cur, B, l)
                pg.release(cur)
                cur = nxt
            elif isinstance(m, InvTransformerBlock):
                nxt = m.emit(pg, packed, f"decoder.{i}.", cur, B, l)
                pg.release(cur)
                cur = nxt
            else:   # nn.Upsample(scale_factor=2, mode='linear', align_corners=True), VAE.py:160,164,168
                cur, l = self._resample(pg, cur, B, l, 2 * l, cur.shape[2], 1, f"decoder.{i}.upsample"), 2 * l
        y = pg.alloc((B, l, 4), F32)
        pg.conv([Seg(cur, 16, 16, l)], packed["last.w"], packed["last.b"], y, 4, B, l, 4, name="decoder.11")
        pg.release(cur)
        y = self._resample(pg, y, B, l, lat, 4, 2, "dec.pool")                 # F.adaptive_avg_pool1d(., L) with L = the LATENT length, VAE.py:191-192
        out = pg.alloc((B, 2, lat), F32)
        pg.call("dec.chanfirst", lib.tdx_rows_to_chanfirst, y.data_ptr(), 4, out.data_ptr(), B, 2, lat, code, keep=[y, out])
        return _Codec(pg, zin, out)

    def _codec(self, kind: str, B: int, n: int, device) -> _Codec:
        self.packed_weights(F32)   # refresh (drops stale programs) before the cache lookup
        key = (kind, B, n, device)
        c = self._programs.get(key)
        if c is None:
            c = self._encode_program(B, n, device) if kind == "enc" else self._decode_program(B, n, device)
            self._programs[key] = c
        return c

    # ---- the reference's methods ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        self._guard(x)
        if x.dim() != 3 or x.shape[1] != 2:
            raise TdxError(f"VAE_InvTrans.encode: x must be (B, 2, L), got {tuple(x.shape)}")
        c = self._codec("enc", x.shape[0], x.shape[2], x.device)
        c.inp.copy_(x)
        c.pg.run()
        return c.out.clone()

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        self._guard(z)
        if z.dim() != 3 or z.shape[1] != 4:
            raise TdxError(f"VAE_InvTrans.decode: z must be (B, 4, l), got {tuple(z.shape)}")
        c = self._codec("dec", z.shape[0], z.shape[2], z.device)
        c.inp.copy_(z)
        c.pg.run()
        return c.out.clone()

    def forward(self, x: torch.Tensor):
        raise NotImplementedError("VAE_InvTrans.forward (reparameterised training pass, VAE.py:195-216) is outside the hot path; "
                                  "use encode() / decode()")
