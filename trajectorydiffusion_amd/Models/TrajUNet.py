"""TrajUNet with the reference's constructor / forward signature / state_dict keys
(reference Models/TrajUNet.py:4-112), executed as a static program of HIP kernel launches.

forward(x (B,2,L) fp32, time (B,) int64, cat_attr (B,5) int64, num_attr (B,3) fp32) -> (B,2,L) fp32.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib, engine
from .._lib import ACT_NONE, TdxError
from ._packed import _PackedModel
from .TrajUNetBlocks import AttnBlock, ResnetBlock, WideAndDeepEmbedBlock


class UNetProgram:
    """One compiled (B, L, dtype) instance of the forward pass plus its static I/O buffers."""

    def __init__(self, model: "TrajUNet", B: int, L: int, dtype: torch.dtype, device: torch.device, split_embed: bool):
        self.B, self.L, self.dtype, self.split_embed = B, L, dtype, split_embed
        pg = engine.Program(device, dtype)
        if model.low_latency:
            pg.split_k_ws = torch.empty(8 << 20, dtype=torch.float32, device=device)   # 32 MiB of split-K slices
        self.pg = pg
        f32 = torch.float32
        self.x = pg.alloc((B, 2, L), f32)          # x_t in, updated in place by the sampler
        self.eps = pg.alloc((B, 2, L), f32)        # eps_hat out
        self.time = pg.alloc((B,), torch.int64)
        self.cat = pg.alloc((B, 5), torch.int64)
        self.num = pg.alloc((B, 3), f32)
        self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self.emb = pg.alloc((B, 256), f32)         # swish([time_embed | attr_embed])  (B, 256)
        packed = model.packed_weights(dtype)
        NE = model.embed_cols
        self.ebias = pg.alloc((B, NE), f32)
        self.tbias: Optional[torch.Tensor] = None
        self.embed_pg: Optional[engine.Program] = None
        if split_embed:
            # sampling: every sample shares t, and attributes do not change between steps, so
            #   W_e . swish([time | attr]) = W_e[:, :128] . swish(time)  +  W_e[:, 128:] . swish(attr)
            # the attr half (+ bias) is computed once per call into ebias, the time half once per step table row.
            self.tbias = pg.alloc((NE,), f32)
            ep = engine.Program(device, f32)
            engine.emit_embed(ep, model.embed_block, packed, B, self.time, self.cat, self.num, self.err, self.emb, want_time=False)
            ep.conv([engine.Seg(self.emb, 256, 128, 1, taps=1, ch_off=128)], packed["embed_proj.w_attr"], packed["embed_proj.b"],
                    self.ebias, NE, B, 1, NE, dtype_code=_lib.TDX_F32, name="embed_proj.attr")
            self.embed_pg = ep
        else:
            engine.emit_embed(pg, model.embed_block, packed, B, self.time, self.cat, self.num, self.err, self.emb)
            pg.conv([engine.Seg(self.emb, 256, 256, 1, taps=1)], packed["embed_proj.w"], packed["embed_proj.b"], self.ebias, NE, B, 1,
                    NE, dtype_code=_lib.TDX_F32, name="embed_proj")
        self._emit_unet(model, packed)

    def _emit_unet(self, m: "TrajUNet", packed: Dict[str, torch.Tensor]) -> None:
        """Dataflow: every conv GEMM is recorded together with the GroupNorm(+Swish) of its consumer (engine.Program.conv,
        `gn=`).  By default that GroupNorm runs as its own tdx_groupnorm_act launch right behind the conv (measured faster
        on MI355X for every layer, also against the persistent kernel's fused epilogue: profiles/r3_gn_per_layer_ablation.txt);
        with Program.fuse_gn it is applied in the conv's epilogue where the library can.  Stand-alone GroupNorms with no conv producer: the one after the stem and the skip halves of the
        decoder concats."""
        pg, B, L = self.pg, self.B, self.L
        cs, NE, R = m.channel_schedule, m.embed_cols, m.res_blocks
        eb, tb = self.ebias, self.tbias
        lens = [L]
        for _ in range(m.stages):
            lens.append(engine.conv_out_len(lens[-1], 2))
        # stem (reference TrajUNet.py:107); only its LENGTH is used as a skip (:109 + :93)
        cur = pg.alloc((B, L, cs[0]))
        stem_w, stem_b = engine.fp32_param(m.stem.weight, "stem.weight"), engine.fp32_param(m.stem.bias, "stem.bias")
        # Split GroupNorm on the first encoder level when its convs are HBM-bound 128 -> 128 layers the library can run that way
        # (engine.FUSE_NORM_DEFAULT): the stem and every conv leave block partial sums, the consumers normalise on load.
        split0 = (not m.low_latency and cs[0] == cs[1] and cs[0] % 32 == 0 and cs[0] // 32 >= 4
                  and engine.norm_fusable_shape(pg, B, L, L, cs[0], cs[1]))
        downs: List[torch.Tensor] = []
        cur_is_skip = False  # skip tensors stay live until the decoder has consumed them
        if split0:
            st0 = pg.alloc((B, (L + 15) // 16, cs[0] // 4, 2), torch.float32)
            pg.call("stem", pg.lib.tdx_stem_conv_stats, self.x.data_ptr(), stem_w.data_ptr(), stem_b.data_ptr(), cur.data_ptr(), st0.data_ptr(),
                    cs[0] // 4, 0, B, L, cs[0], keep=[cur, st0])
            g1 = m.down_blocks[0][0].stage1[0]
            cur_mr = engine.emit_finalize(pg, st0, B, L, g1.num_groups, cs[0] // g1.num_groups, g1.eps, "stem.norm")
            pg.release(st0)
            for j in range(R):
                key = f"down_blocks.0.{j}."
                nxt_eps = m.down_blocks[0][j + 1].stage1[0].eps if j + 1 < R else None
                nxt, nxt_mr = engine.emit_resblock_split_norm(pg, m.down_blocks[0][j], packed, key, cur, cur_mr, B, L, eb, tb, NE,
                                                              m.embed_offsets[key], nxt_eps)
                pg.release(cur)
                pg.release(cur_mr)
                cur, cur_mr = nxt, nxt_mr
            gn_in = None
        else:
            pg.call("stem", pg.lib.tdx_stem_conv, self.x.data_ptr(), stem_w.data_ptr(), stem_b.data_ptr(), cur.data_ptr(), B, L, cs[0], pg.code,
                    keep=[cur])
            gn_in = engine.block_input_gn(pg, m.down_blocks[0][0], "down_blocks.0.0.", B, L)
            engine.emit_groupnorm(pg, cur, cs[0], B, L, gn_in, "down_blocks.0.0.gn1")
        for i in range(m.stages):  # encoder, :65-78
            li, co = lens[i], cs[i + 1]
            for j in range(R if not (split0 and i == 0) else 0):
                key = f"down_blocks.{i}.{j}."
                blk = m.down_blocks[i][j]
                nxt_gn = engine.block_input_gn(pg, m.down_blocks[i][j + 1], f"down_blocks.{i}.{j + 1}.", B, li) if j + 1 < R else None
                nxt = engine.emit_resblock(pg, blk, packed, key, gn_in.out, [(cur, blk.in_c)], B, li, eb, tb, NE, m.embed_offsets[key],
                                           next_gn=nxt_gn)
                pg.release(gn_in.out)
                if not cur_is_skip:
                    pg.release(cur)
                cur, cur_is_skip, gn_in = nxt, False, nxt_gn
            dconv = m.down_blocks[i][R]
            d = pg.alloc((B, lens[i + 1], co))
            if i + 1 < m.stages:
                gn_in = engine.block_input_gn(pg, m.down_blocks[i + 1][0], f"down_blocks.{i + 1}.0.", B, lens[i + 1])
            else:
                gn_in = engine.block_input_gn(pg, m.mid_attn_block.res1, "mid_attn_block.res1.", B, lens[i + 1])
            pg.conv([engine.Seg(cur, co, co, li, stride=2)], packed[f"down_blocks.{i}.down.w"],
                    engine.fp32_param(dconv.bias, f"down_blocks.{i}.{R}.bias"), d, co, B, lens[i + 1], co,
                    name=f"down_blocks.{i}.down", gn=gn_in)
            pg.release(cur)
            downs.append(d)
            cur, cur_is_skip = d, True
        lm = lens[-1]

        def decoder_input_gn(i: int, part_off: int, a1=None):
            blk = m.up_blocks[i][0]
            return engine.block_input_gn(pg, blk, f"up_blocks.{i}.0.", B, lens[m.stages - i], part_off, a1)

        key = "mid_attn_block."
        gn_up = decoder_input_gn(0, 0)  # first half of the decoder concat = the mid block's output
        cur = m.mid_attn_block.emit(pg, packed, key, downs[-1], gn_in.out, B, lm, eb, tb, NE,
                                    (m.embed_offsets[key + "res1."], m.embed_offsets[key + "res2."]), next_gn=gn_up)
        pg.release(gn_in.out)
        for i in range(m.stages):  # decoder, :81-95
            skip = downs[-i - 1]
            li = lens[m.stages - i]
            lo = lens[m.stages - i - 1]
            cskip = cs[m.stages - i]
            co = cs[m.stages - i - 1]
            # second half of the concat: GroupNorm+Swish of the skip tensor, written next to the first half
            engine.emit_groupnorm(pg, skip, cskip, B, li, decoder_input_gn(i, cskip, gn_up.out), f"up_blocks.{i}.0.gn1.skip")
            srcs: List[Tuple[torch.Tensor, int]] = [(cur, cskip), (skip, cskip)]
            a1 = gn_up.out
            for j in range(R - 1):  # up_stage[:-2]: the last ResnetBlock of each stage is never executed
                key = f"up_blocks.{i}.{j}."
                blk = m.up_blocks[i][j]
                nxt_gn = engine.block_input_gn(pg, m.up_blocks[i][j + 1], f"up_blocks.{i}.{j + 1}.", B, li) if j + 1 < R - 1 else None
                nxt = engine.emit_resblock(pg, blk, packed, key, a1, srcs, B, li, eb, tb, NE, m.embed_offsets[key], next_gn=nxt_gn)
                pg.release(a1)
                for t, _ in srcs:
                    pg.release(t)
                srcs = [(nxt, co)]
                a1 = nxt_gn.out if nxt_gn is not None else None
            cur = srcs[0][0]
            uconv = m.up_blocks[i][R]
            last = i + 1 == m.stages
            head_split = (last and not m.low_latency and co % 32 == 0 and co // m.head[0].num_groups >= 4 and (co & (co - 1)) == 0 and co <= 512
                          and engine.norm_fusable_shape(pg, B, li, lo, co, co, norm=False, stats=True))
            if head_split:
                # the last up conv leaves the block sums of its output; the head conv applies the head GroupNorm while it loads
                gn_head, hconv = m.head[0], m.head[1]
                nxt = pg.alloc((B, lo, co))
                st_h = pg.alloc((B, (lo + 15) // 16, co // 4, 2), torch.float32)
                pg.conv([engine.Seg(cur, co, co, li, lvirt=lo)], packed[f"up_blocks.{i}.up.w"],
                        engine.fp32_param(uconv.bias, f"up_blocks.{i}.{R}.bias"), nxt, co, B, lo, co, name=f"up_blocks.{i}.up", stats_out=st_h)
                pg.release(cur)
                cpg_h = co // gn_head.num_groups
                mr_h = engine.emit_finalize(pg, st_h, B, lo, gn_head.num_groups, cpg_h, gn_head.eps, "head.norm")
                pg.call("head.conv", pg.lib.tdx_head_conv_norm, nxt.data_ptr(), mr_h.data_ptr(), gn_head.num_groups, 0, cpg_h,
                        engine.fp32_param(gn_head.weight, "head.0.weight").data_ptr(), engine.fp32_param(gn_head.bias, "head.0.bias").data_ptr(),
                        engine.fp32_param(hconv.weight, "head.1.weight").data_ptr(), engine.fp32_param(hconv.bias, "head.1.bias").data_ptr(),
                        self.eps.data_ptr(), B, L, cs[0], keep=[nxt, mr_h])
                return
            if last:  # head GroupNorm (default eps 1e-5, no activation, TrajUNet.py:42-45): the raw tensor is not needed
                gn_head = m.head[0]
                a_head = pg.alloc((B, lo, co))
                gn_up = engine.GnSpec(gn_head, "head.0", 0, co // gn_head.num_groups, ACT_NONE, a_head, co, 0)
                nxt = None
            else:
                gn_up = decoder_input_gn(i + 1, 0)
                nxt = pg.alloc((B, lo, co))
            pg.conv([engine.Seg(cur, co, co, li, lvirt=lo)], packed[f"up_blocks.{i}.up.w"],
                    engine.fp32_param(uconv.bias, f"up_blocks.{i}.{R}.bias"), nxt, co, B, lo, co, name=f"up_blocks.{i}.up", gn=gn_up)
            pg.release(cur)
            cur = nxt
        hconv = m.head[1]
        pg.call("head.conv", pg.lib.tdx_head_conv, gn_up.out.data_ptr(), engine.fp32_param(hconv.weight, "head.1.weight").data_ptr(),
                engine.fp32_param(hconv.bias, "head.1.bias").data_ptr(), self.eps.data_ptr(), B, L, cs[0], pg.code, keep=[gn_up.out])

    def load_inputs(self, x, time, cat_attr, num_attr) -> None:
        self.x.copy_(x)
        self.time.copy_(time)
        self.cat.copy_(cat_attr)
        self.num.copy_(num_attr)

    def run(self) -> None:
        self.pg.run()


class TrajUNet(_PackedModel):
    def __init__(self, channel_schedule: List[int], traj_length: int = 200, diffusion_steps: int = 300, res_blocks: int = 2) -> None:
        super().__init__()
        if res_blocks < 2:
            raise TdxError("TrajUNet: res_blocks must be >= 2 (with 1 the reference's decoder feeds a 2C-channel tensor to a C-channel conv)")
        self.channel_schedule = list(channel_schedule)
        self.stages = len(channel_schedule) - 1
        self.res_blocks = res_blocks
        self.diffusion_steps = diffusion_steps  # traj_length is unused, as in the reference (:10)
        cs = self.channel_schedule
        self.embed_block = WideAndDeepEmbedBlock(diffusion_steps, hidden_dim=256, embed_dim=128)
        self.stem = nn.Conv1d(2, cs[0], 3, 1, 1)
        self.down_blocks = nn.ModuleList()
        for i in range(self.stages):
            layers = [ResnetBlock(cs[i], cs[i + 1], norm_groups=32)]
            layers += [ResnetBlock(cs[i + 1], norm_groups=32) for _ in range(res_blocks - 1)]
            layers.append(nn.Conv1d(cs[i + 1], cs[i + 1], 3, 2, 1))  # downsample
            self.down_blocks.append(nn.ModuleList(layers))
        self.mid_attn_block = AttnBlock(cs[-1], norm_groups=32)
        self.up_blocks = nn.ModuleList()
        for i in range(self.stages):
            cin, cout = cs[self.stages - i], cs[self.stages - i - 1]
            layers = [ResnetBlock(cin * 2, cout, norm_groups=32)]
            layers += [ResnetBlock(cout, norm_groups=32) for _ in range(res_blocks - 1)]  # last one: parameters only, never run
            layers.append(nn.Conv1d(cout, cout, 3, 1, 1))  # conv after the nearest upsample
            self.up_blocks.append(nn.ModuleList(layers))
        self.head = nn.Sequential(nn.GroupNorm(32, cs[0]), nn.Conv1d(cs[0], 2, 3, 1, 1))

        # column layout of the fused embed_proj GEMM: one (B, embed_cols) bias matrix for every executed ResnetBlock
        self.embed_offsets: Dict[str, int] = {}
        off = 0
        for key, blk in self.executed_resblocks():
            self.embed_offsets[key] = off
            off += blk.out_c
        self.embed_cols = off
        self._init_cache()   # packed weights, programs, weights_epoch, low_latency (Models/_packed.py)

    def set_low_latency(self, on: bool = True) -> "TrajUNet":
        """Small-batch serving mode (no reference counterpart): convs with too few output tiles to fill the 256 CUs are cut
        along K into concurrent slices (tdx_conv_desc.ws).  Off by default because the fp32 summation order, and with it the
        last bit of the result, then depends on the batch size: batch shards are no longer bit-identical to the unsharded run."""
        self.low_latency = bool(on)
        return self

    # ---- structure helpers ---------------------------------------------------------------------
    def executed_resblocks(self):
        for i in range(self.stages):
            for j in range(self.res_blocks):
                yield f"down_blocks.{i}.{j}.", self.down_blocks[i][j]
        yield "mid_attn_block.res1.", self.mid_attn_block.res1
        yield "mid_attn_block.res2.", self.mid_attn_block.res2
        for i in range(self.stages):
            for j in range(self.res_blocks - 1):
                yield f"up_blocks.{i}.{j}.", self.up_blocks[i][j]

    def _build_packed(self, dtype: torch.dtype) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for key, blk in self.executed_resblocks():
                engine.pack_resblock(blk, key, dtype, out)
            self.mid_attn_block.pack("mid_attn_block.", dtype, out)
            for i in range(self.stages):
                out[f"down_blocks.{i}.down.w"] = engine.pack_conv_weight(self.down_blocks[i][self.res_blocks].weight, dtype)
                out[f"up_blocks.{i}.up.w"] = engine.pack_conv_weight(self.up_blocks[i][self.res_blocks].weight, dtype)
            w = torch.cat([blk.embed_proj.weight.detach().reshape(blk.out_c, -1) for _, blk in self.executed_resblocks()])
            out["embed_proj.w"] = w.float().contiguous()
            out["embed_proj.w_time"] = w[:, :128].float().contiguous()
            out["embed_proj.w_attr"] = w[:, 128:].float().contiguous()
            out["embed_proj.b"] = torch.cat([blk.embed_proj.bias.detach() for _, blk in self.executed_resblocks()]).float().contiguous()
        return out

    def program(self, B: int, L: int, device: torch.device, split_embed: bool = False, dtype: Optional[torch.dtype] = None, slot: int = 0) -> UNetProgram:
        dtype = self.compute_dtype if dtype is None else dtype
        self.packed_weights(dtype)  # refresh (drops stale programs) before the cache lookup
        key = (B, L, dtype, device, split_embed, self.low_latency) + ((slot,) if slot else ())   # slot: distinct instances of one shape (batch slices that run concurrently)
        prog = self._programs.get(key)
        if prog is None:
            prog = UNetProgram(self, B, L, dtype, device, split_embed)
            self._programs[key] = prog
        return prog

    def time_bias_table(self, t_values: torch.Tensor) -> torch.Tensor:
        """(len(t_values), embed_cols) fp32 rows W_e[:, :128] . swish(time_embed(t)) for the sampler's step table."""
        n = t_values.numel()
        dev = t_values.device
        packed = self.packed_weights(self.compute_dtype)
        pg = engine.Program(dev, torch.float32)
        cat = torch.zeros((n, 5), dtype=torch.int64, device=dev)
        num = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        emb = pg.alloc((n, 256), torch.float32)
        engine.emit_embed(pg, self.embed_block, packed, n, t_values.to(torch.int64).contiguous(), cat, num, err, emb, want_attr=False)
        table = torch.empty((n, self.embed_cols), dtype=torch.float32, device=dev)
        pg.conv([engine.Seg(emb, 256, 128, 1, taps=1)], packed["embed_proj.w_time"], None, table, self.embed_cols, n, 1, self.embed_cols,
                dtype_code=_lib.TDX_F32, name="embed_proj.time")
        pg.run()
        engine.check_err_flag(err, "TrajUNet.time_bias_table")
        return table

    # ---- the reference's forward ----------------------------------------------------------------
    def forward(self, x: torch.Tensor, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor) -> torch.Tensor:
        if self.training:   # train.ipynb:204: fp32, dropout on, differentiable w.r.t. the parameters (training.py)
            from ..training import train_forward
            if x.dim() != 3 or x.shape[1] != 2:
                raise TdxError(f"TrajUNet.forward: x must be (B, 2, L), got {tuple(x.shape)}")
            return train_forward(self, x, time, cat_attr, num_attr)
        self._guard(x, time, cat_attr, num_attr)
        if x.dim() != 3 or x.shape[1] != 2:
            raise TdxError(f"TrajUNet.forward: x must be (B, 2, L), got {tuple(x.shape)}")
        B, _, L = x.shape
        prog = self.program(B, L, x.device)
        prog.load_inputs(x, time, cat_attr, num_attr)
        prog.run()
        engine.check_err_flag(prog.err, "TrajUNet.forward")
        return prog.eps.clone()
