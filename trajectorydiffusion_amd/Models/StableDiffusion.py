"""SDUNet_CAR, the latent denoiser of eval_sd.ipynb, with the reference's constructor / forward signature / state_dict keys
(reference Models/StableDiffusion.py:114-222), executed as a static program of HIP kernel launches.

forward(x (B, traj_dim, L) fp32, time (B,) int64, cat_attr (B, 5) int64, num_attr (B, 3) fp32) -> (B, traj_dim, L) fp32.
L must halve cleanly `stages` times (the reference's torch.cat of the decoder fails otherwise, :200).
Inference only: the reference trains this model in train_sd.ipynb, which is outside the SURVEY section 8 scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib, engine
from .._lib import ACT_NONE, TdxError
from ..engine import Seg, fp32_param
from ._packed import _PackedModel
from .StableDiffusionBlocks import N_COND, ConditionMixBlock, CrossAttentionRes


class SDProgram:
    """One compiled (B, L, dtype) instance of the forward pass plus its static I/O buffers (same face as UNetProgram, so the
    graph-replayed sampler drives both)."""

    def __init__(self, model: "SDUNet_CAR", B: int, L: int, dtype: torch.dtype, device: torch.device):
        self.B, self.L, self.dtype = B, L, dtype
        pg = engine.Program(device, dtype)
        if engine.FUSE_GN_ENV is None and engine.FUSE_GN_DEFAULT is False:
            pg.fuse_gn = "auto"   # the out projections' GroupNorm in their epilogue: 1.9 % per denoise step (profiles/r3_c5_gn_fusion.txt)
        self.pg = pg
        f32 = torch.float32
        D = model.traj_dim
        self.x = pg.alloc((B, D, L), f32)
        self.eps = pg.alloc((B, D, L), f32)
        self.time = pg.alloc((B,), torch.int64)
        self.cat = pg.alloc((B, 5), torch.int64)
        self.num = pg.alloc((B, 3), f32)
        self.err = torch.zeros(1, dtype=torch.int32, device=device)
        # the condition vector depends on t through mix_fc (a nonlinear mix of time and attributes, StableDiffusionBlocks.py:140-142),
        # so unlike TrajUNet's embedding nothing of it can be hoisted out of the step: no tbias table, no per-call embed program
        self.tbias: Optional[torch.Tensor] = None
        self.embed_pg: Optional[engine.Program] = None
        self._emit(model, model.packed_weights(dtype))

    def _emit(self, m: "SDUNet_CAR", packed: Dict[str, torch.Tensor]) -> None:
        pg, B, L = self.pg, self.B, self.L
        cs, R, S = m.channel_schedule, m.res_blocks, m.stages
        lens = [L]
        for _ in range(S):
            if lens[-1] % 2:
                raise TdxError(f"SDUNet_CAR: length {L} does not halve cleanly {S} times (the decoder's concat needs it, StableDiffusion.py:200)")
            lens.append(lens[-1] // 2)
        E = N_COND * m.embed_block.embed_dim
        cond32 = pg.alloc((B, E), torch.float32)
        m.embed_block.emit(pg, B, self.time, self.cat, self.num, self.err, cond32)
        if pg.dtype == torch.float32:
            cond = cond32
        else:
            cond = pg.alloc((B, E))
            pg.call("cond.cast", pg.lib.tdx_chanfirst_to_rows, cond32.data_ptr(), cond.data_ptr(), B, E, 1, E, pg.code, keep=[cond32, cond])
        # K | V rows of the 7 condition tokens for all blocks at once: one (B x 896) x (896 x 14 sum C) GEMM per denoise step
        kvm = pg.alloc((B, m.kv_cols))
        pg.conv([Seg(cond, E, E, 1, taps=1)], packed["kv_all.w"], packed["kv_all.b"], kvm, m.kv_cols, B, 1, m.kv_cols, name="kv_all")
        kv = lambda key: (kvm, m.kv_offsets[key])  # noqa: E731
        cur = pg.alloc((B, L, cs[0]))
        pg.call("stem", pg.lib.tdx_stem_conv_n, self.x.data_ptr(), fp32_param(m.stem.weight, "stem.weight").data_ptr(),
                fp32_param(m.stem.bias, "stem.bias").data_ptr(), cur.data_ptr(), B, L, m.traj_dim, cs[0], pg.code, keep=[cur])
        downs: List[torch.Tensor] = []
        cur_is_skip = False   # skip tensors stay live until the decoder has consumed them
        for i in range(S):   # encoder, :181-187
            li, co = lens[i], cs[i + 1]
            for j in range(R):
                key = f"down_blocks.{i}.{j}."
                blk = m.down_blocks[i][j]
                nxt = blk.emit(pg, packed, key, [(cur, blk.in_c)], None, B, li, kv_all=kv(key))
                if not cur_is_skip:
                    pg.release(cur)
                cur, cur_is_skip = nxt, False
            dconv = m.down_blocks[i][R]
            d = pg.alloc((B, lens[i + 1], co))
            pg.conv([Seg(cur, co, co, li, stride=2)], packed[f"down_blocks.{i}.down.w"], fp32_param(dconv.bias, f"down_blocks.{i}.{R}.bias"), d, co, B,
                    lens[i + 1], co, name=f"down_blocks.{i}.down")
            pg.release(cur)
            downs.append(d)
            cur, cur_is_skip = d, True
        cur = m.mid_attn_block.emit(pg, packed, "mid_attn_block.", [(downs[-1], cs[-1])], None, B, lens[-1], kv_all=kv("mid_attn_block."))   # :220 (downs[-1] stays live: skip)
        for i in range(S):   # decoder, :198-205
            skip = downs[-i - 1]
            li, lo = lens[S - i], lens[S - i - 1]
            cskip, co = cs[S - i], cs[S - i - 1]
            srcs: List[Tuple[torch.Tensor, int]] = [(cur, cskip), (skip, cskip)]
            for j in range(R - 1):
                key = f"up_blocks.{i}.{j}."
                nxt = m.up_blocks[i][j].emit(pg, packed, key, srcs, None, B, li, kv_all=kv(key))
                for t, _ in srcs:
                    pg.release(t)
                srcs = [(nxt, co)]
            cur = srcs[0][0]
            up = pg.alloc((B, lo, co))   # nn.Upsample(scale_factor=2, 'nearest'), :166: materialised (the next block's identity shortcut reads it)
            pg.call(f"up_blocks.{i}.upsample", pg.lib.tdx_resample_rows, cur.data_ptr(), up.data_ptr(), B, li, lo, co, 0, pg.code, keep=[cur, up])
            pg.release(cur)
            key = f"up_blocks.{i}.{R}."
            cur = m.up_blocks[i][R].emit(pg, packed, key, [(up, co)], None, B, lo, kv_all=kv(key))
            pg.release(up)
        gn = m.head[0]   # :148-151: default eps, no activation
        a = pg.alloc((B, L, cs[0]))
        pg.groupnorm(cur, cs[0], a, cs[0], 0, fp32_param(gn.weight, "head.0.weight"), fp32_param(gn.bias, "head.0.bias"), 0, B, L, cs[0], gn.num_groups,
                     gn.eps, ACT_NONE, name="head.gn")
        hconv = m.head[1]
        pg.call("head.conv", pg.lib.tdx_head_conv_n, a.data_ptr(), fp32_param(hconv.weight, "head.1.weight").data_ptr(),
                fp32_param(hconv.bias, "head.1.bias").data_ptr(), self.eps.data_ptr(), B, L, cs[0], m.traj_dim, pg.code, keep=[a])

    def load_inputs(self, x, time, cat_attr, num_attr) -> None:
        self.x.copy_(x)
        self.time.copy_(time)
        self.cat.copy_(cat_attr)
        self.num.copy_(num_attr)

    def run(self) -> None:
        self.pg.run()


class SDUNet_CAR(_PackedModel):
    def __init__(self, traj_dim: int, channel_schedule: List[int], diffusion_steps: int = 300, res_blocks: int = 2) -> None:
        super().__init__()
        if res_blocks < 2:
            raise TdxError("SDUNet_CAR: res_blocks must be >= 2 (the stage builders always create two blocks, StableDiffusion.py:155-156)")
        if not 1 <= traj_dim <= 8:
            raise TdxError("SDUNet_CAR: traj_dim must be in 1..8 (tdx_stem_conv_n / tdx_head_conv_n)")
        self.traj_dim = traj_dim
        self.channel_schedule = list(channel_schedule)
        self.stages = len(channel_schedule) - 1
        self.res_blocks = res_blocks
        self.diffusion_steps = diffusion_steps
        cs = self.channel_schedule
        self.embed_block = ConditionMixBlock(diffusion_steps, hidden_dim=128, embed_dim=128)
        self.stem = nn.Conv1d(traj_dim, cs[0], 3, 1, 1)
        car = lambda i, o: CrossAttentionRes(embed_dim=i, out_dim=o, num_heads=8)  # noqa: E731
        self.down_blocks = nn.ModuleList()
        for i in range(self.stages):   # :154-161: res_blocks CrossAttentionRes, then the stride-2 conv
            layers = [car(cs[i], cs[i + 1])] + [car(cs[i + 1], cs[i + 1]) for _ in range(res_blocks - 1)]
            layers.append(nn.Conv1d(cs[i + 1], cs[i + 1], 3, 2, 1))
            self.down_blocks.append(nn.ModuleList(layers))
        self.mid_attn_block = car(cs[-1], cs[-1])
        self.up_blocks = nn.ModuleList()
        for i in range(self.stages):   # :164-171: res_blocks - 1 CrossAttentionRes, Upsample (index res_blocks - 1), one more
            cin, cout = cs[self.stages - i] * 2, cs[self.stages - i - 1]
            layers = [car(cin, cout)] + [car(cout, cout) for _ in range(res_blocks - 2)]
            layers.append(nn.Identity())   # nn.Upsample(scale_factor=2, mode='nearest'): no parameters
            layers.append(car(cout, cout))
            self.up_blocks.append(nn.ModuleList(layers))
        self.head = nn.Sequential(nn.GroupNorm(32, cs[0]), nn.Conv1d(cs[0], traj_dim, 3, 1, 1))
        self.embed_cols = 0   # no per-step time-bias row (see SDProgram)
        self.sampler_slices = 1   # the sampler keeps the batch whole (measured: 2 slices cut the GEMMs' rate from 784 to 572 TFLOP/s for +2 % wall)
        # column layout of the stacked condition GEMM: 14 C columns (7 tokens x [K | V]) per executed block
        self.kv_offsets: Dict[str, int] = {}
        off = 0
        for key, blk in self.executed_blocks():
            self.kv_offsets[key] = off
            off += N_COND * 2 * blk.out_c
        self.kv_cols = off
        self._init_cache()

    def executed_blocks(self):
        R = self.res_blocks
        for i in range(self.stages):
            for j in range(R):
                yield f"down_blocks.{i}.{j}.", self.down_blocks[i][j]
        yield "mid_attn_block.", self.mid_attn_block
        for i in range(self.stages):
            for j in list(range(R - 1)) + [R]:
                yield f"up_blocks.{i}.{j}.", self.up_blocks[i][j]

    def _build_packed(self, dtype: torch.dtype) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for key, blk in self.executed_blocks():
                blk.pack(key, dtype, out)
            keys = [key for key, _ in self.executed_blocks()]
            out["kv_all.w"] = torch.cat([out.pop(key + "kvc.w") for key in keys], dim=0).contiguous()
            out["kv_all.b"] = torch.cat([out.pop(key + "kvc.b") for key in keys]).contiguous()
            for i in range(self.stages):
                out[f"down_blocks.{i}.down.w"] = engine.pack_conv_weight(self.down_blocks[i][self.res_blocks].weight, dtype)
        return out

    def program(self, B: int, L: int, device: torch.device, split_embed: bool = False, dtype: Optional[torch.dtype] = None, slot: int = 0) -> SDProgram:
        """split_embed is accepted for the sampler's sake and changes nothing here (see SDProgram)."""
        dtype = self.compute_dtype if dtype is None else dtype
        self.packed_weights(dtype)
        key = (B, L, dtype, device, split_embed, self.low_latency) + ((slot,) if slot else ())   # slot: distinct instances of one shape (batch slices that run concurrently)
        prog = self._programs.get(key)
        if prog is None:
            prog = SDProgram(self, B, L, dtype, device)
            self._programs[key] = prog
        return prog

    def time_bias_table(self, t_values: torch.Tensor) -> Optional[torch.Tensor]:
        return None

    def forward(self, x: torch.Tensor, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor) -> torch.Tensor:
        self._guard(x, time, cat_attr, num_attr)
        if x.dim() != 3 or x.shape[1] != self.traj_dim:
            raise TdxError(f"SDUNet_CAR.forward: x must be (B, {self.traj_dim}, L), got {tuple(x.shape)}")
        B, _, L = x.shape
        prog = self.program(B, L, x.device)
        prog.load_inputs(x, time, cat_attr, num_attr)
        prog.run()
        engine.check_err_flag(prog.err, "SDUNet_CAR.forward")
        return prog.eps.clone()
