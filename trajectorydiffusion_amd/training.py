"""One eps-MSE training step of TrajUNet on the HIP kernels (reference train.ipynb:198-209):

    x_t = q_sample(x0, t, eps);  loss = mean((model(x_t, t, cat, num) - eps)^2);  loss.backward();  AdamW.step()

fp32 throughout.  The forward pass is recorded op by op on a tape (every op = C-ABI launches in a forward `engine.Program`
plus a closure that appends its gradient launches to a backward Program), so the backward pass is the tape in reverse: conv
data gradients run on the forward conv kernel (tdx_conv_gemm on dy with repacked weights), weight gradients on
tdx_conv_wgrad, GroupNorm+Swish on tdx_groupnorm_act_bwd, and so on (include/trajdiff_hip.h, "Training-step building
blocks").  Activations are saved (no workspace reuse), statistics are recomputed.

torch does plumbing only: device memory, padded stand-ins and bias sums (small copy_ ops into fixed buffers), dtype /
contiguity fixes of the step's inputs, and zeroing of gradient buffers.  The embedding-row gathers and the layout changes of
the (B, 2, L) model input / output are library launches too (tdx_embed_gather, tdx_chanfirst_to_rows, tdx_rows_to_chanfirst).

Not covered: bf16 training (the reference trains in fp32).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, engine
from ._lib import ACT_LEAKY, ACT_NONE, ACT_SWISH, KSeg, TdxError

F32 = _lib.TDX_F32


class _Src:
    """One conv source: rows tensor (B, lin, ld), channel slice, taps / stride / nearest-upsample, and the slice of the weight
    parameter (Cout, cin_total, k) that multiplies it."""

    def __init__(self, t: torch.Tensor, cin: int, lin: int, W: torch.Tensor, taps: int = 3, stride: int = 1, ch_off: int = 0,
                 lvirt: Optional[int] = None, w_off: int = 0, needs_grad: bool = True):
        self.t, self.ld, self.cin, self.lin, self.W = t, t.shape[-1], cin, lin, W
        self.taps, self.stride, self.ch_off, self.lvirt, self.w_off, self.needs_grad = taps, stride, ch_off, lin if lvirt is None else lvirt, w_off, needs_grad

    def wslice(self) -> torch.Tensor:
        return self.W.detach()[:, self.w_off:self.w_off + self.cin, :]


def default_dropout_seed() -> int:
    """Dropout Philox seed when the caller gives none: drawn from torch's CPU generator (so torch.manual_seed controls the masks,
    and two runs without a manual seed differ) and mixed with the data-parallel rank (ranks that seed torch identically, as DDP
    scripts do, still draw different masks for their different batch shards)."""
    import torch.distributed as dist
    base = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return (base ^ (rank * 0x9E3779B97F4A7C15)) & ((1 << 64) - 1)


class TrainStep:
    """The training step of one (B, L) batch shape: forward and backward programs built once, then
    forward_backward() / optimizer_step() per batch (or forward() + backward() behind torch.autograd, see train_forward)."""

    def __init__(self, model, B: int, L: int, device: torch.device, lr: float = 5e-4, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.01, dropout: Optional[float] = None, seed: Optional[int] = None, graph_repack: bool = True):
        self.m, self.B, self.L, self.dev = model, B, L, device
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.lib = _lib.lib()
        self.fw = engine.Program(device, torch.float32)
        self.fw.release = lambda t: None            # saved activations: nothing is reused
        self.bw = engine.Program(device, torch.float32)
        self.bw.release = lambda t: None
        # the embedding-side GEMMs have B rows only (8 row tiles at B = 1024) and K up to 1280: with room for split-K slices they run as
        # ~64 workgroups instead of 8-16 (tdx_conv_desc.ws).  One buffer for both programs: they never run concurrently.
        if os.environ.get("TDX_TRAIN_SPLITK", "1") != "0":
            self.fw.split_k_ws = self.bw.split_k_ws = torch.empty(8 << 20, dtype=torch.float32, device=device)
        self._back: List[Callable[[], None]] = []
        self._repack: List[Callable[[], None]] = []        # torch plumbing refreshed every step: padded stand-ins, bias sums
        self._repack_torch: List[Callable[[], None]] = []  # the weight layouts as torch statements (tests check tdx_pack_weights against them)
        self._packs: List[Tuple["_Src", torch.Tensor, int, str]] = []
        self._pack_tab: Optional[torch.Tensor] = None
        self._agrad: Dict[int, torch.Tensor] = {}   # id(activation) -> gradient buffer
        # id(activation) -> channel slices [lo, hi) of its gradient buffer that the backward program has written so far, in emission
        # (= execution) order.  The FIRST writer of a slice overwrites it (accumulate = 0 / no residual operand), later ones add: a
        # buffer whose every channel is first written that way needs no zero fill per step, and its first data-gradient conv carries
        # no residual operand (one tensor read less, and the persistent conv kernel takes the launch)
        self._gwritten: Dict[int, List[Tuple[int, int]]] = {}
        self._gzero: List[torch.Tensor] = []        # gradient buffers that still need the per-step zero fill
        self._adamw_key, self._adamw_tab = None, None   # device table of the optimizer's tensors (tdx_adamw_step_multi)
        # Two streams for the backward program: a conv's weight gradient (side stream) and its data gradient (main stream) both read dY and
        # are independent; at B = 1024 every conv has ~400 tiles for 256 CUs, so a GEMM alone leaves a fifth of the CUs idle in its last
        # round and the other GEMM's workgroups fill them.  _side_idx: launches that go to the side stream; _join_before: launches in
        # front of which the main stream waits for it; _side_reads: gradient buffers the outstanding side launches read (while emitting).
        self._side_idx: set = set()
        self._join_before: set = set()
        self._side_reads: set = set()
        self._side_stream: Optional[torch.cuda.Stream] = None
        self.two_streams = os.environ.get("TDX_TRAIN_TWO_STREAMS", "1") != "0"
        self._keep: List[torch.Tensor] = []
        self.params = dict(model.named_parameters())
        for n, p in self.params.items():
            if p.dtype != torch.float32 or p.device.type != "cuda" or not p.is_contiguous():
                raise TdxError(f"TrainStep: parameter {n} must be a contiguous fp32 tensor on the GPU")
        self.grads: Dict[str, torch.Tensor] = {}    # parameter name -> gradient (only parameters the forward pass uses)
        # parameter gradients are views into ONE flat buffer (64-float aligned), handed out in first-use order: one memset
        # per step, and the data-parallel all-reduce (distributed.allreduce_mean_) runs over it in place with no packing copies
        self._garena = torch.zeros(sum((p.numel() + 63) // 64 * 64 for p in self.params.values()), device=device, dtype=torch.float32)
        self._garena_used = 0
        self._pname = {id(p): n for n, p in self.params.items()}
        self._padded: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Tuple[slice, ...]]] = []
        self.ws = torch.empty(64 * 1024 * 1024 * 3 + 16, device=device, dtype=torch.float32)   # wgrad slabs (largest: 1024 x 1024 x 3)
        self.red_ws = torch.empty(int(self.lib.tdx_train_workspace_floats()), device=device, dtype=torch.float32)
        self.step_count = 0
        self.state: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.dropout = dropout          # None: each ResnetBlock's own p (0.1, TrajUNetBlocks.py:66); 0.0 switches it off
        self.seed = default_dropout_seed() if seed is None else int(seed) & ((1 << 64) - 1)
        self.rng = torch.zeros(4, dtype=torch.int32, device=device)   # [0] = forward-pass counter, read by tdx_dropout
        self._passes = 0
        self._layer = 0
        self._plans: Dict[int, List[Tuple[int, int, int]]] = {}
        self._comm_stream = None
        self._repack_graph, self._repack_key, self._graph_repack = None, None, graph_repack   # (rocprofv3 cannot trace a capture)
        self._build()

    # ---- bookkeeping -----------------------------------------------------------------------------------------------------
    def _new(self, *shape, dtype=torch.float32) -> torch.Tensor:
        t = torch.zeros(shape, device=self.dev, dtype=dtype)
        self._keep.append(t)
        return t

    def G(self, t: torch.Tensor) -> torch.Tensor:
        g = self._agrad.get(id(t))
        if g is None:
            g = self._agrad[id(t)] = self._new(*t.shape)
        if id(g) in self._side_reads:
            # a weight-gradient launch on the side stream may still be reading this buffer: whatever is emitted next (it asked for the
            # buffer, so it may write it: an aliased shortcut gradient being added to) first joins the side stream
            self._join_before.add(len(self.bw.ops))
            self._side_reads.clear()
        return g

    def first_write(self, t: torch.Tensor, lo: int = 0, width: Optional[int] = None) -> bool:
        """Called while the backward program is emitted, by a launch that writes channels [lo, lo + width) of every row of G(t):
        True when no earlier launch has written any of them (the launch must then overwrite), False when it must accumulate."""
        hi = lo + (t.shape[-1] if width is None else width)
        spans = self._gwritten.setdefault(id(t), [])
        fresh = all(hi <= a or lo >= b for a, b in spans)
        if fresh:
            spans.append((lo, hi))
        elif not any(a <= lo and hi <= b for a, b in spans):
            raise TdxError("TrainStep: a gradient write overlaps an earlier one only partly")
        return fresh

    def PG(self, p: torch.Tensor) -> torch.Tensor:
        """Gradient buffer of a parameter (or of a padded stand-in registered with _pad)."""
        n = self._pname.get(id(p))
        if n is None:
            raise TdxError("TrainStep: unknown parameter")
        g = self.grads.get(n)
        if g is None:
            g = self.grads[n] = self._param_grad(n, p) if n in self.params else self._new(*p.shape)
        return g

    def _param_grad(self, n: str, p: torch.Tensor) -> torch.Tensor:
        lo = self._garena_used
        self._garena_used = lo + (p.numel() + 63) // 64 * 64
        return self._garena[lo:lo + p.numel()].view(p.shape)

    def flat_gradients(self) -> torch.Tensor:
        """Every parameter gradient as one flat fp32 tensor (the views in self.grads alias it)."""
        return self._garena[: self._garena_used]

    def sync_gradients(self, group=None, bucket_bytes: int = 64 << 20) -> None:
        """Data-parallel mean of the gradients over the ranks of `group` (RCCL all-reduce over xGMI), in place."""
        from .distributed import allreduce_mean_
        allreduce_mean_(self.flat_gradients(), group=group, bucket_bytes=bucket_bytes)

    def _pad(self, p: torch.Tensor, shape: Sequence[int], where: Tuple[slice, ...]) -> torch.Tensor:
        """A zero-padded copy of parameter p (channel counts the GEMM kernels can take), refreshed every step; its gradient is
        folded back into p's gradient after the backward pass."""
        q = self._new(*shape)
        self._pname[id(q)] = "__pad%d" % len(self._padded)
        self._repack.append(lambda: q[where].copy_(p.detach()))
        self._padded.append((q, p, None, where))
        return q

    def _pack(self, s: "_Src", dst: torch.Tensor, col: int, mode: str) -> None:
        """Register one tdx_pack_weights table entry: the layout engine.pack_conv_weight* would give the weight slice of s."""
        self._packs.append((s, dst, col, mode))

    def _pack_table(self) -> torch.Tensor:
        rows = []
        for s, dst, col, mode in self._packs:
            v = s.wslice()
            cout, cin, k = v.shape
            s0, s1, s2 = v.stride()
            if mode == "fwd":
                d = (cout, cin, k, s0, s1, list(range(k)))
            elif mode == "dgrad":          # engine.pack_conv_weight_dgrad: taps flipped, channel roles exchanged
                d = (cin, cout, k, s1, s0, [k - 1 - t for t in range(k)])
            elif mode == "dgrad_even":     # engine.pack_conv_weight_dgrad_stride2: dx[2m] = W_1^T dy[m]
                d = (cin, cout, 1, s1, s0, [1])
            else:                          # dx[2m + 1] = W_2^T dy[m] + W_0^T dy[m + 1]; taps read dy[m - 1], dy[m], dy[m + 1]
                d = (cin, cout, 3, s1, s0, [-1, 2, 0])
            nrow, nch, taps, sr, sc, tmap = d
            if dst.shape[0] != nrow or col + nch * taps > dst.shape[1] or v.dtype != torch.float32:
                raise TdxError("TrainStep: weight copy does not fit its buffer")
            tm = (C.c_int32 * 3)(*(tmap + [0] * (3 - len(tmap))))
            rows.append(_lib.PackDesc(v.data_ptr(), dst.data_ptr() + 4 * col, nrow, nch, taps, dst.shape[1], sr, sc, s2, tm, 0))
        arr = (_lib.PackDesc * len(rows))(*rows)
        return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)

    # ---- ops --------------------------------------------------------------------------------------------------------------
    def conv(self, name: str, srcs: Sequence[_Src], cout: int, lout: int, out: torch.Tensor, out_off: int = 0,
             biases: Sequence[torch.Tensor] = (), res: Optional[torch.Tensor] = None, ebias: Optional[Tuple[torch.Tensor, int]] = None,
             act: int = ACT_NONE) -> None:
        B, fw, lib = self.B, self.fw, self.lib
        ktot = sum(s.taps * s.cin for s in srcs)
        wp = self._new(cout, ktot)
        self._repack_torch.append(lambda: wp.copy_(torch.cat([engine.pack_conv_weight(s.wslice(), torch.float32) for s in srcs], dim=1)))
        koff = 0
        for s in srcs:
            self._pack(s, wp, koff, "fwd")
            koff += s.taps * s.cin
        bsum = None
        if biases:
            bsum = self._new(cout)
            self._repack.append(lambda: bsum.copy_(sum(b.detach() for b in biases)))
        fw.conv([engine.Seg(s.t, s.ld, s.cin, s.lin, taps=s.taps, stride=s.stride, ch_off=s.ch_off, lvirt=s.lvirt) for s in srcs], wp, bsum,
                out, out.shape[-1], B, lout, cout, out_off=out_off, act=act, res=res, res_ld=0 if res is None else res.shape[-1],
                ebias=None if ebias is None else ebias[0], ebias_ld=0 if ebias is None else ebias[0].shape[-1],
                ebias_off=0 if ebias is None else ebias[1], name=name)
        dgrad_w = []
        for s in srcs:
            if not s.needs_grad:
                dgrad_w.append(None)
            elif s.stride == 2:
                we, wo = self._new(s.cin, cout), self._new(s.cin, 3 * cout)

                def rp(s=s, we=we, wo=wo):
                    a, b = engine.pack_conv_weight_dgrad_stride2(s.wslice(), torch.float32)
                    we.copy_(a)
                    wo.copy_(b)
                self._repack_torch.append(rp)
                self._pack(s, we, 0, "dgrad_even")
                self._pack(s, wo, 0, "dgrad_odd")
                dgrad_w.append((we, wo))
            else:
                wd = self._new(s.cin, s.taps * cout)
                self._repack_torch.append(lambda s=s, wd=wd: wd.copy_(engine.pack_conv_weight_dgrad(s.wslice(), torch.float32)))
                self._pack(s, wd, 0, "dgrad")
                dgrad_w.append(wd)

        def backward():
            bw, N = self.bw, B * lout
            go, ld, off = self.G(out), out.shape[-1], out_off
            if act == ACT_LEAKY:   # (only the embedding MLP: lout == 1)
                dz = self._new(B, lout, cout)
                bw.call(name + ".act_bwd", lib.tdx_act_bwd, go.data_ptr() + 4 * off, ld, out.data_ptr() + 4 * off, ld, dz.data_ptr(), cout, N, cout,
                        ACT_LEAKY, 0)
                go, ld, off = dz, cout, 0
            elif act != ACT_NONE:
                raise TdxError(f"{name}: activation {act} in a conv epilogue has no backward here")
            gptr = go.data_ptr() + 4 * off
            if res is not None:
                # d(res) = dY.  When this is the first write of the residual's gradient and both are plain (B, L, cout) tensors, the residual's
                # gradient simply IS this buffer from here on: every reader of G(out) is emitted below, before any later launch adds to
                # G(res), and G(out) has no reader after its producer's backward (a 210 MB copy per identity shortcut otherwise)
                if (os.environ.get("TDX_TRAIN_ALIAS_DRES", "1") != "0" and ld == cout and off == 0 and tuple(res.shape) == tuple(go.shape)
                        and id(res) not in self._agrad and id(res) not in self._gwritten):
                    self.first_write(res, 0, cout)
                    self._agrad[id(res)] = go
                else:
                    gr = self.G(res)
                    bw.call(name + ".dres", lib.tdx_upsample_fold, go.data_ptr(), ld, off, gr.data_ptr(), gr.shape[-1], 0, B, lout, lout, cout,
                            0 if self.first_write(res, 0, cout) else 1, F32)
            # the first bias takes its gradient from the first segment's weight-gradient launch (tdx_conv_wgrad_bias); a second bias (the
            # 1x1 shortcut's, same gradient) keeps the stand-alone column sum
            fused_bias = biases[0] if biases and os.environ.get("TDX_TRAIN_FUSE_DBIAS", "1") != "0" else None
            for b in biases:
                if b is not fused_bias:
                    bw.call(name + ".dbias", lib.tdx_colsum, gptr, ld, N, cout, self.PG(b).data_ptr(), 1, F32, self.red_ws.data_ptr())
            if ebias is not None:
                ge = self.G(ebias[0])
                bw.call(name + ".debias", lib.tdx_traj_rowsum, gptr, ld, ge.data_ptr() + 4 * ebias[1], ge.shape[-1], B, lout, cout, 1)
            for s, wd in zip(srcs, dgrad_w):
                k = KSeg(s.t.data_ptr(), float(torch.tensor(float(s.lin), dtype=torch.float32) / torch.tensor(float(s.lvirt), dtype=torch.float32)),
                         0, s.ld, s.ch_off, s.cin, s.taps, s.stride, s.lin, s.lvirt, 0)
                self._keep.append(k)
                gw = self.PG(s.W)
                self._side_idx.add(len(bw.ops))
                self._side_reads.add(id(go))
                if fused_bias is not None:
                    bw.call(name + ".wgrad", lib.tdx_conv_wgrad_bias, go.data_ptr(), ld, off, cout, C.byref(k), B, lout, gw.data_ptr(), s.W.shape[1],
                            s.w_off, 1, self.PG(fused_bias).data_ptr(), 1, F32, self.ws.data_ptr(), self.ws.numel())
                    fused_bias = None
                else:
                    bw.call(name + ".wgrad", lib.tdx_conv_wgrad, go.data_ptr(), ld, off, cout, C.byref(k), B, lout, gw.data_ptr(), s.W.shape[1], s.w_off,
                            1, F32, self.ws.data_ptr(), self.ws.numel())
                if wd is None:
                    continue
                gs = self.G(s.t)
                if s.stride == 2:
                    if off != 0 or ld != cout or s.ch_off != 0 or s.ld != s.cin:
                        raise TdxError(f"{name}: stride-2 data gradient needs plain tensors")
                    engine.conv_dgrad_stride2(bw, go, wd[0], wd[1], gs, B, s.lin, cout, s.cin, accumulate=not self.first_write(s.t, 0, s.cin))
                elif s.lvirt != s.lin:
                    du = self._new(B, s.lvirt, s.cin)
                    bw.conv([engine.Seg(go, ld, cout, lout, taps=s.taps, ch_off=off)], wd, None, du, s.cin, B, s.lvirt, s.cin, name=name + ".dgrad")
                    bw.call(name + ".fold", lib.tdx_upsample_fold, du.data_ptr(), s.cin, 0, gs.data_ptr(), s.ld, s.ch_off, B, s.lin, s.lvirt, s.cin,
                            0 if self.first_write(s.t, s.ch_off, s.cin) else 1, F32)
                elif self.first_write(s.t, s.ch_off, s.cin):
                    bw.conv([engine.Seg(go, ld, cout, lout, taps=s.taps, ch_off=off)], wd, None, gs, s.ld, B, s.lin, s.cin, out_off=s.ch_off,
                            name=name + ".dgrad")
                else:
                    bw.conv([engine.Seg(go, ld, cout, lout, taps=s.taps, ch_off=off)], wd, None, gs, s.ld, B, s.lin, s.cin, out_off=s.ch_off,
                            res=gs[..., s.ch_off:], res_ld=s.ld, name=name + ".dgrad")
        self._back.append(backward)

    def gn(self, name: str, x: torch.Tensor, Cn: int, L: int, out: torch.Tensor, out_off: int, mod, p_off: int, groups: int, act: int,
           drop: Optional[Tuple[float, int]] = None) -> None:
        """GroupNorm (+ activation); drop = (p, layer): nn.Dropout(p) behind it in the same pass, forward and backward (tdx_dropout's mask
        for `out`, which must then be the plain (B, L, Cn) tensor)."""
        B, lib = self.B, self.lib
        gam, bet = mod.weight, mod.bias
        if drop is None:
            self.fw.groupnorm(x, x.shape[-1], out, out.shape[-1], out_off, gam.data, bet.data, p_off, B, L, Cn, groups, mod.eps, act, name=name)
        else:
            self.fw.call(name, lib.tdx_groupnorm_act_dropout, x.data_ptr(), x.shape[-1], out.data_ptr(), out.shape[-1], out_off,
                         gam.data.data_ptr() + 4 * p_off, bet.data.data_ptr() + 4 * p_off, B, L, Cn, groups, mod.eps, act, F32, drop[0], self.seed, drop[1],
                         self.rng.data_ptr(), keep=[x, out, gam.data, bet.data])

        def backward():
            ra, rb = self._new(B, Cn), self._new(B, Cn)
            go, gx = self.G(out), self.G(x)
            args = (x.data_ptr(), x.shape[-1], go.data_ptr(), out.shape[-1], out_off, gx.data_ptr(),
                    x.shape[-1], 0, gam.data.data_ptr() + 4 * p_off, bet.data.data_ptr() + 4 * p_off, ra.data_ptr(), rb.data_ptr(), B, L, Cn, groups,
                    mod.eps, act, 0 if self.first_write(x, 0, Cn) else 1, F32)
            if drop is None:
                self.bw.call(name + ".bwd", lib.tdx_groupnorm_act_bwd, *args)
            else:
                self.bw.call(name + ".bwd", lib.tdx_groupnorm_act_dropout_bwd, *args, drop[0], self.seed, drop[1], self.rng.data_ptr())
            self.bw.call(name + ".dparam", lib.tdx_colsum_pair, ra.data_ptr(), rb.data_ptr(), Cn, B, Cn, self.PG(gam).data_ptr() + 4 * p_off,
                         self.PG(bet).data_ptr() + 4 * p_off, 1, F32, self.red_ws.data_ptr())
        self._back.append(backward)

    # ---- the network ------------------------------------------------------------------------------------------------------
    def _resblock(self, key: str, blk, srcs: List[Tuple[torch.Tensor, int]], L: int) -> torch.Tensor:
        """ResnetBlock (TrajUNetBlocks.py:92-99); srcs = the (possibly concatenated) raw input as [(rows tensor, channels)]."""
        B = self.B
        cin, cout = blk.in_c, blk.out_c
        a1 = self._new(B, L, cin)
        off = 0
        for t, c in srcs:   # GroupNorm(32, cin) over a concat of equal halves = GroupNorm(16) over each half
            self.gn(key + "gn1", t, c, L, a1, off, blk.stage1[0], off, blk.norm_groups * c // cin, ACT_SWISH)
            off += c
        h = self._new(B, L, cout)
        self.conv(key + "conv1", [_Src(a1, cin, L, blk.stage1[2].weight)], cout, L, h, biases=[blk.stage1[2].bias],
                  ebias=(self.E, self.m.embed_offsets[key]))
        a2 = self._new(B, L, cout)
        p = blk.dropout if self.dropout is None else self.dropout
        fuse = os.environ.get("TDX_TRAIN_FUSE_DROPOUT", "1") != "0"
        if p > 0.0:
            self._layer += 1
        layer = self._layer
        self.gn(key + "gn2", h, cout, L, a2, 0, blk.stage2[0], 0, blk.norm_groups, ACT_SWISH, drop=(p, layer) if p > 0.0 and fuse else None)
        if p > 0.0 and not fuse:   # nn.Dropout in stage2 (TrajUNetBlocks.py:85) as its own pass; the same call on the gradient is its backward
            self.fw.call(key + "dropout", self.lib.tdx_dropout, a2.data_ptr(), cout, B * L, cout, p, self.seed, layer, self.rng.data_ptr())
            self._back.append(lambda: self.bw.call(key + "dropout.bwd", self.lib.tdx_dropout, self.G(a2).data_ptr(), cout, B * L, cout, p, self.seed,
                                                   layer, self.rng.data_ptr()))
        y = self._new(B, L, cout)
        s2 = [_Src(a2, cout, L, blk.stage2[3].weight)]
        if blk.has_shortcut:
            off = 0
            for t, c in srcs:
                s2.append(_Src(t, c, L, blk.shortcut.weight, taps=1, w_off=off))
                off += c
            self.conv(key + "conv2", s2, cout, L, y, biases=[blk.stage2[3].bias, blk.shortcut.bias])
        else:
            self.conv(key + "conv2", s2, cout, L, y, biases=[blk.stage2[3].bias], res=srcs[0][0])
        return y

    def _build(self) -> None:
        m, B, L, lib = self.m, self.B, self.L, self.lib
        cs, R, S = m.channel_schedule, m.res_blocks, m.stages
        emb = m.embed_block
        # ---- embedding (TrajUNetBlocks.py:45-62) ----
        self.pe, self.catvec, self.num4 = self._new(B, 1, 128), self._new(B, 1, 1280), self._new(B, 1, 4)
        # tdx_embed_gather also writes the inference path's LeakyReLU(wide_fc(num)) slice; training runs wide_fc as a GEMM (it needs the
        # weight gradient), so that slice lands in a scratch row buffer
        self._gather_scratch = torch.empty((B, 1024), dtype=torch.float32, device=self.dev)
        self._err = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._pos_enc = None
        t0, t2 = emb.time_embed_layers[0], emb.time_embed_layers[2]
        cf, d1, d3 = emb.categorical_fc, emb.deep_fc[1], emb.deep_fc[3]
        h1 = self._new(B, 1, 128)
        self._lin("embed.time0", self.pe, 128, t0, h1, 0, ACT_LEAKY, needs_grad=False)
        z = self._new(B, 1, 256)
        self._lin("embed.time2", h1, 128, t2, z, 0, ACT_NONE)
        deep_in = self._new(B, 1, 1024)
        self._lin("embed.cat_fc", self.catvec, 1280, cf, deep_in, 0, ACT_LEAKY)
        wide4 = self._pad(emb.wide_fc.weight, (256, 4, 1), (slice(None), slice(0, 3), 0))
        self._lin("embed.wide", self.num4, 4, emb.wide_fc, deep_in, 768, ACT_LEAKY, W=wide4, needs_grad=False)
        hd = self._new(B, 1, 512)
        self._lin("embed.deep1", deep_in, 1024, d1, hd, 0, ACT_LEAKY)
        self._lin("embed.deep3", hd, 512, d3, z, 128, ACT_NONE)
        e = self._new(B, 1, 256)
        self.fw.call("embed.swish", lib.tdx_act_fwd, z.data_ptr(), 256, e.data_ptr(), 256, B, 256, ACT_SWISH)
        self._back.append(lambda: self.bw.call("embed.swish.bwd", lib.tdx_act_bwd, self.G(e).data_ptr(), 256, z.data_ptr(), 256, self.G(z).data_ptr(), 256,
                                               B, 256, ACT_SWISH, 1))

        def emb_tables():
            gc = self.G(self.catvec)
            self.bw.call("embed.cell.bwd", lib.tdx_embedding_bwd, self.cat.data_ptr(), 5, 4, gc.data_ptr(), 1280,
                         self.PG(emb.cell_index_embedder.weight).data_ptr(), 256, 256, B, 1)
            self.bw.call("embed.day.bwd", lib.tdx_embedding_bwd, self.cat.data_ptr() + 8 * 4, 5, 1, gc.data_ptr() + 4 * 1024, 1280,
                         self.PG(emb.day_embedder.weight).data_ptr(), 33, 256, B, 1)
        self._back.insert(0, emb_tables)   # runs LAST in the backward pass (the list is replayed in reverse)
        self.cat = self._new(B, 5, dtype=torch.int64)
        # ---- per-block embed projections (TrajUNetBlocks.py:89,96): E[:, off : off + out_c] ----
        self.E = self._new(B, 1, m.embed_cols)
        for key, blk in m.executed_resblocks():
            self._lin(key + "embed_proj", e, 256, blk.embed_proj, self.E, m.embed_offsets[key], ACT_NONE, W=blk.embed_proj.weight)
        # ---- UNet (TrajUNet.py:98-112) ----
        lens = [L]
        for _ in range(S):
            lens.append(engine.conv_out_len(lens[-1], 2))
        self.x4 = self._new(B, L, 4)
        stem4 = self._pad(m.stem.weight, (cs[0], 4, 3), (slice(None), slice(0, 2), slice(None)))
        cur = self._new(B, L, cs[0])
        self.conv("stem", [_Src(self.x4, 4, L, stem4, needs_grad=False)], cs[0], L, cur, biases=[m.stem.bias])
        downs = []
        for i in range(S):
            for j in range(R):
                blk = m.down_blocks[i][j]
                cur = self._resblock(f"down_blocks.{i}.{j}.", blk, [(cur, blk.in_c)], lens[i])
            dconv = m.down_blocks[i][R]
            d = self._new(B, lens[i + 1], cs[i + 1])
            self.conv(f"down_blocks.{i}.down", [_Src(cur, cs[i + 1], lens[i], dconv.weight, stride=2)], cs[i + 1], lens[i + 1], d, biases=[dconv.bias])
            downs.append(d)
            cur = d
        ab, key, Cn, lm = m.mid_attn_block, "mid_attn_block.", cs[-1], lens[-1]
        x1 = self._resblock(key + "res1.", ab.res1, [(cur, Cn)], lm)
        a = self._new(B, lm, Cn)
        self.gn(key + "qkv_gn", x1, Cn, lm, a, 0, ab.qkv_getter[0], 0, ab.norm_groups, ACT_NONE)
        qkv = self._new(B, lm, 3 * Cn)
        self.conv(key + "qkv", [_Src(a, Cn, lm, ab.qkv_getter[1].weight, taps=1)], 3 * Cn, lm, qkv, biases=[ab.qkv_getter[1].bias])
        o = self._new(B, lm, Cn)
        self.fw.call(key + "attention", lib.tdx_attention, qkv.data_ptr(), o.data_ptr(), B, lm, Cn, F32)
        self._back.append(lambda: (self.first_write(qkv), self.bw.call(key + "attention.bwd", lib.tdx_attention_bwd, qkv.data_ptr(), self.G(o).data_ptr(),
                                                                         self.G(qkv).data_ptr(), B, lm, Cn)))
        x2 = self._new(B, lm, Cn)
        self.conv(key + "proj", [_Src(o, Cn, lm, ab.project_out.weight, taps=1)], Cn, lm, x2, biases=[ab.project_out.bias], res=x1)
        cur = self._resblock(key + "res2.", ab.res2, [(x2, Cn)], lm)
        for i in range(S):
            skip, li, lo = downs[-i - 1], lens[S - i], lens[S - i - 1]
            cskip, co = cs[S - i], cs[S - i - 1]
            srcs = [(cur, cskip), (skip, cskip)]
            for j in range(R - 1):
                cur = self._resblock(f"up_blocks.{i}.{j}.", m.up_blocks[i][j], srcs, li)
                srcs = [(cur, co)]
            uconv = m.up_blocks[i][R]
            nxt = self._new(B, lo, co)
            self.conv(f"up_blocks.{i}.up", [_Src(cur, co, li, uconv.weight, lvirt=lo)], co, lo, nxt, biases=[uconv.bias])
            cur = nxt
        gh, hc = m.head[0], m.head[1]
        ah = self._new(B, L, cs[0])
        self.gn("head.gn", cur, cs[0], L, ah, 0, gh, 0, gh.num_groups, ACT_NONE)
        head4 = self._pad(hc.weight, (4, cs[0], 3), (slice(0, 2), slice(None), slice(None)))
        hb4 = self._pad(hc.bias, (4,), (slice(0, 2),))
        self.out4 = self._new(B, L, 4)
        self.conv("head.conv", [_Src(ah, cs[0], L, head4)], 4, L, self.out4, biases=[hb4])
        # ---- backward program = the tape in reverse ----
        self.first_write(self.out4)                  # backward() itself writes the loss gradient there (all 4 padded channels)
        for f in reversed(self._back):
            f()
        # gradient buffers that some launch only ever adds to (per-trajectory embed rows, the embedding MLP's small ends) keep their
        # per-step zero fill; every other one is overwritten by its first writer
        for key, g in self._agrad.items():
            spans = sorted(self._gwritten.get(key, []))
            end = 0
            for a, b in spans:
                if a > end:
                    break
                end = max(end, b)
            if end < g.shape[-1]:
                self._gzero.append(g)
        self.pred = self._new(B, 2, L)
        self.dpred = self._new(B, 2, L)
        self.loss = self._new(1)
        for q, p, _, where in self._padded:          # folded targets get their slice of the flat buffer now, not at the first backward
            if self._pname[id(q)] in self.grads:
                self.PG(p)
        self._grad_ready = self._scan_gradient_writers()

    def _scan_gradient_writers(self) -> Dict[str, int]:
        """name -> index of the LAST backward launch that writes that parameter's gradient (len(ops) for gradients that
        are completed by the fold of a padded stand-in after the program).  Found by looking for pointers into the flat
        gradient buffer among the launch arguments, so it cannot drift from what the program really does."""
        from .distributed import scan_gradient_writers
        base, end = self._garena.data_ptr(), self._garena.data_ptr() + 4 * self._garena_used
        spans = [(g.data_ptr(), g.data_ptr() + 4 * g.numel(), n) for n, g in self.grads.items() if n in self.params]
        folded = [self._pname[id(p)] for q, p, _, where in self._padded if self._pname[id(q)] in self.grads]
        try:
            return scan_gradient_writers(spans, [args for _, _, args in self.bw.ops], base, end, folded)
        except ValueError as e:
            raise TdxError(f"TrainStep: {e}") from None

    def _bucket_plan(self, bucket_bytes: int) -> List[Tuple[int, int, int]]:
        """(lo, hi, ready) slices of the flat gradient buffer, at most bucket_bytes each, in the order they complete.  The
        buffer is laid out in backward order, so consecutive slices complete one after the other while the program runs."""
        key = int(bucket_bytes)
        plan = self._plans.get(key)
        if plan is None:
            from .distributed import bucket_plan
            base = self._garena.data_ptr()
            spans = [((g.data_ptr() - base) // 4, (g.data_ptr() - base) // 4 + g.numel(), self._grad_ready[n])
                     for n, g in self.grads.items() if n in self.params]
            plan = bucket_plan(spans, self._garena_used, bucket_bytes)
            self._plans[key] = plan
        return plan

    def _lin(self, name: str, src: torch.Tensor, k: int, mod, out: torch.Tensor, off: int, act: int, W: Optional[torch.Tensor] = None,
             needs_grad: bool = True) -> None:
        """nn.Linear / Conv1d(k=1) on (B, 1, k) rows."""
        if W is None:   # nn.Linear weight (out, in) as a Conv1d(k=1) weight through a registered view
            W = mod.weight.unsqueeze(2)
            self._keep.append(W)
            self._pname[id(W)] = self._pname[id(mod.weight)] + "::k1"
            self._padded.append((W, mod.weight, None, None))
        self.conv(name, [_Src(src, k, 1, W, taps=1, needs_grad=needs_grad)], W.shape[0], 1, out, out_off=off, biases=[mod.bias], act=act)

    # ---- one step -----------------------------------------------------------------------------------------------------------
    def _refresh_packed_weights(self) -> None:
        """Re-derive the GEMM-layout copies from the parameters' current values: padded stand-ins and bias sums are a few
        dozen small torch ops (captured once into a HIP graph and replayed), then ONE tdx_pack_weights launch writes every
        forward / data-gradient weight layout from a device-resident table.  Both are rebuilt if a parameter's storage moves."""
        key = tuple(p.data_ptr() for p in self.params.values())
        if self._repack_key != key:
            self._repack_graph, self._pack_tab, self._repack_key = None, self._pack_table(), key
        if not self._graph_repack:
            for f in self._repack:
                f()
        elif self._repack_graph is None:
            for f in self._repack:                      # eager once: warms the allocator and makes the copies valid now
                f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for f in self._repack:
                    f()
            self._repack_graph = g
        else:
            self._repack_graph.replay()
        _lib.check(self.lib.tdx_pack_weights(self._pack_tab.data_ptr(), len(self._packs), _lib.stream_ptr()), "tdx_pack_weights")

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, cat: torch.Tensor, num: torch.Tensor) -> torch.Tensor:
        """eps_hat (B, 2, L) of the training-mode forward pass; activations stay saved for backward()."""
        m, B, L = self.m, self.B, self.L
        emb = m.embed_block
        if tuple(x_t.shape) != (B, 2, L):
            raise TdxError(f"TrainStep: expected x_t of shape {(B, 2, L)}, got {tuple(x_t.shape)}")
        for v in (x_t, t, cat, num):
            _lib.require_device(v, "TrainStep input")
        F32, st, lib = _lib.TDX_F32, _lib.stream_ptr(), self.lib
        with torch.no_grad():
            self._refresh_packed_weights()
            # layouts and embedding-row gathers, through the library like everything else (dtype / contiguity fixes are the only
            # torch ops left): (B, 2, L) -> zero-padded rows (B, L, 4); pos_enc[t] and the five embedding rows; num -> (B, 4)
            x_c, num_c = x_t.to(torch.float32).contiguous(), num.to(torch.float32).contiguous()
            t_c = t.to(torch.int64).contiguous()
            self.cat.copy_(cat)
            if self._pos_enc is None or self._pos_enc.device != self.dev:
                self._pos_enc = emb.pos_enc.to(self.dev).contiguous()
            _lib.check(lib.tdx_chanfirst_to_rows(x_c.data_ptr(), self.x4.data_ptr(), B, 2, L, 4, F32, st), "x -> rows")
            _lib.check(lib.tdx_embed_gather(t_c.data_ptr(), self.cat.data_ptr(), num_c.data_ptr(), self._pos_enc.data_ptr(), self._pos_enc.shape[0],
                                            emb.cell_index_embedder.weight.data_ptr(), emb.day_embedder.weight.data_ptr(),
                                            emb.wide_fc.weight.data_ptr(), emb.wide_fc.bias.data_ptr(), self.pe.data_ptr(), self.catvec.data_ptr(),
                                            self._gather_scratch.data_ptr(), B, self._err.data_ptr(), st), "tdx_embed_gather")
            _lib.check(lib.tdx_chanfirst_to_rows(num_c.data_ptr(), self.num4.data_ptr(), B, 3, 1, 4, F32, st), "num -> rows")
            if int(self._err.item()):                       # the step's only host synchronisation
                self._err.zero_()
                raise IndexError("TrainStep: time step or categorical attribute out of range")
            self._passes += 1                               # (after the check: a rejected batch must not shift the dropout stream of later steps)
            self.rng[0] = self._passes
            self.fw.run()
            _lib.check(lib.tdx_rows_to_chanfirst(self.out4.data_ptr(), 4, self.pred.data_ptr(), B, 2, L, F32, st), "rows -> eps_hat")
        return self.pred

    def backward(self, dpred: torch.Tensor, sync_group=False, bucket_bytes: int = 32 << 20) -> Dict[str, torch.Tensor]:
        """Gradients of every used parameter (by name) for the gradient dpred (B, 2, L) of the last forward()'s output.

        sync_group (None = the default process group, or a group): data parallel.  The mean over ranks of each
        bucket_bytes slice of the flat gradient buffer is started on a side stream as soon as the backward launch that
        completes the slice has been issued, so the RCCL all-reduces run under the rest of the backward program; on return
        the gradients are the means (what backward() followed by sync_gradients() gives; bit for bit at world size 2, where
        the order of the sum is fixed)."""
        sync = sync_group is not False
        with torch.no_grad():
            for g in self._gzero:
                g.zero_()
            self.flat_gradients().zero_()
            for n, g in self.grads.items():
                if n not in self.params:
                    g.zero_()
            dp = dpred.to(torch.float32).contiguous()
            _lib.check(self.lib.tdx_chanfirst_to_rows(dp.data_ptr(), self.G(self.out4).data_ptr(), self.B, 2, self.L, 4, _lib.TDX_F32, _lib.stream_ptr()),
                       "dpred -> rows")
            if not sync:
                if self.two_streams:
                    self._run_backward_two_streams()
                else:
                    self.bw.run()
                self._fold_padded()
                return self.named_gradients()
            import torch.distributed as dist
            from .distributed import allreduce_slice_async, force_collectives
            world = dist.get_world_size(sync_group)
            plan = self._bucket_plan(bucket_bytes) if world > 1 or force_collectives() else []
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.dev)
            main, comm, flat = torch.cuda.current_stream(), self._comm_stream, self._garena
            pending, k = [], 0

            def start_ready(done: int) -> None:   # every slice whose last writer has index <= done
                nonlocal k
                if k < len(plan) and plan[k][2] <= done:
                    comm.wait_stream(main)        # orders the collective after the launches issued so far, not after later ones
                    if self.two_streams and self._side_stream is not None:
                        comm.wait_stream(self._side_stream)   # (the weight gradients are written on the side stream)
                    while k < len(plan) and plan[k][2] <= done:
                        lo, hi, _ = plan[k]
                        with torch.cuda.stream(comm):
                            pending.append((allreduce_slice_async(flat[lo:hi], sync_group), lo, hi))
                        k += 1
            if self.two_streams:
                self._run_backward_two_streams(after=start_ready)
            else:
                st = _lib.stream_ptr()
                for i, (name, fn, args) in enumerate(self.bw.ops):
                    rc = fn(*args, st)
                    if rc:
                        _lib.check(rc, name)
                    start_ready(i)
            self._fold_padded()
            start_ready(len(self.bw.ops))
            for work, lo, hi in pending:
                if work is not None:
                    work.wait()                   # main stream waits for the collective
            main.wait_stream(comm)
            if world > 1:
                flat[: self._garena_used].mul_(1.0 / world)
        return self.named_gradients()

    def _run_backward_two_streams(self, after: Optional[Callable[[int], None]] = None) -> None:
        """The backward program with its weight-gradient launches on a side stream (see __init__).  Order on each stream is program
        order; a side launch waits for everything the main stream was given before it (its dY is complete), the main stream waits for
        the side stream in front of the launches in _join_before and at the end.  after(i) is called once launch i has been issued
        (the data-parallel path starts the all-reduce of the gradient slices that launch completes)."""
        main = torch.cuda.current_stream()
        if self._side_stream is None or self._side_stream.device != self.dev:
            self._side_stream = torch.cuda.Stream(device=self.dev)
        side = self._side_stream
        st_main, st_side = main.cuda_stream, side.cuda_stream
        outstanding = False
        for i, (name, fn, args) in enumerate(self.bw.ops):
            if outstanding and i in self._join_before:
                main.wait_stream(side)
                outstanding = False
            if i in self._side_idx:
                side.wait_stream(main)
                rc = fn(*args, st_side)
                outstanding = True
            else:
                rc = fn(*args, st_main)
            if rc:
                _lib.check(rc, name)
            if after is not None:
                after(i)
        if outstanding:
            main.wait_stream(side)

    def _fold_padded(self) -> None:
        for q, p, _, where in self._padded:   # fold the gradients of padded / re-viewed stand-ins back into the parameters
            gq = self.grads.get(self._pname[id(q)])
            if gq is None:
                continue
            gp = self.PG(p)
            gp.add_(gq[where].reshape(p.shape) if where is not None else gq.reshape(p.shape))

    def forward_backward(self, x_t: torch.Tensor, t: torch.Tensor, cat: torch.Tensor, num: torch.Tensor, eps: torch.Tensor,
                         sync_group=False) -> torch.Tensor:
        """Loss (device scalar, this rank's) and gradients (self.grads, by parameter name) of mean((model(x_t, t, cat, num) - eps)^2);
        with sync_group (None = default group) the gradients are the data-parallel mean, reduced under the backward program."""
        pred = self.forward(x_t, t, cat, num)
        epsc = eps.contiguous()
        _lib.check(self.lib.tdx_mse_loss_grad(pred.data_ptr(), epsc.data_ptr(), self.dpred.data_ptr(), self.loss.data_ptr(), pred.numel(),
                                              self.red_ws.data_ptr(), _lib.stream_ptr()), "tdx_mse_loss_grad")
        self.backward(self.dpred, sync_group=sync_group)
        return self.loss

    def named_gradients(self) -> Dict[str, torch.Tensor]:
        return {n: g for n, g in self.grads.items() if n in self.params}

    def set_lr(self, lr: float) -> None:
        """Learning rate of the next optimizer_step() (a host double handed to tdx_adamw_step per call: a scheduler may change
        it every iteration, train.ipynb:209)."""
        lr = float(lr)
        if not lr >= 0.0:
            raise ValueError(f"TrainStep.set_lr: lr={lr}")
        self.lr = lr

    def optimizer(self) -> "FusedAdamW":
        """A torch.optim.Optimizer whose step() is optimizer_step(): plugs into torch.optim.lr_scheduler.*."""
        return FusedAdamW(self)

    def optimizer_step(self) -> None:
        """torch.optim.AdamW(lr, betas, eps, weight_decay) over every parameter that received a gradient (train.ipynb:131,208)."""
        self.step_count += 1
        rows = []
        for n, g in self.named_gradients().items():
            p = self.params[n]
            if n not in self.state:
                self.state[n] = (torch.zeros_like(p), torch.zeros_like(p))
            mm, vv = self.state[n]
            rows.append((p.data.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), p.numel()))
        # one launch over a device-resident table of (param, grad, exp_avg, exp_avg_sq, n) rows (tdx_adamw_desc); the table is rebuilt
        # when any of those buffers has moved (a parameter re-assigned, moments restored from a checkpoint)
        key = tuple(rows)
        if self._adamw_key != key:
            self._adamw_tab = torch.tensor(rows, dtype=torch.int64).to(self.dev)
            self._adamw_key = key
        _lib.check(self.lib.tdx_adamw_step_multi(self._adamw_tab.data_ptr(), len(rows), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                                 self.step_count, _lib.stream_ptr()), "tdx_adamw_step_multi")
        for n in self.named_gradients():
            torch.autograd.graph.increment_version(self.params[n])   # the inference side re-packs its weights when a parameter's version moves


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.Optimizer face of TrainStep.optimizer_step (tdx_adamw_step over every parameter that received a gradient),
    so that the reference's optimizer-side plumbing runs unchanged: `scheduler = ReduceLROnPlateau(optimizer, factor=0.5,
    patience=2000)` and `scheduler.step(float(mov_avg_loss))` per iteration (train.ipynb:133,209) write
    `param_groups[0]['lr']`, and step() hands that value to the kernel.  zero_grad() is a no-op: TrainStep zeroes its
    gradient arena at the start of every backward pass."""

    def __init__(self, train_step: "TrainStep"):
        self.ts = train_step
        super().__init__(list(train_step.params.values()), dict(lr=train_step.lr, betas=train_step.betas, eps=train_step.eps,
                                                                weight_decay=train_step.wd))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.ts.set_lr(g["lr"])
        self.ts.betas, self.ts.eps, self.ts.wd = tuple(g["betas"]), g["eps"], g["weight_decay"]
        self.ts.optimizer_step()
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        return None

    # The Adam moments and the step count live in TrainStep (device buffers the fused kernel updates in place), not in
    # torch.optim.Optimizer.state: checkpoint / resume through the standard optimizer face must carry them.
    def state_dict(self):
        sd = super().state_dict()
        sd["tdx_step_count"] = int(self.ts.step_count)
        sd["tdx_moments"] = {n: (m.detach().clone(), v.detach().clone()) for n, (m, v) in self.ts.state.items()}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        step = sd.pop("tdx_step_count", None)
        moments = sd.pop("tdx_moments", None)
        if step is None or moments is None:
            raise TdxError("FusedAdamW.load_state_dict: not a FusedAdamW state (no tdx_step_count / tdx_moments): the moments of a "
                           "torch.optim.AdamW checkpoint are keyed by parameter index, not by name")
        super().load_state_dict(sd)
        unknown = [n for n in moments if n not in self.ts.params]
        if unknown:
            raise TdxError(f"FusedAdamW.load_state_dict: moments for unknown parameters {unknown[:3]}")
        with torch.no_grad():
            self.ts.state = {}
            for n, (m, v) in moments.items():
                p = self.ts.params[n]
                if tuple(m.shape) != tuple(p.shape) or tuple(v.shape) != tuple(p.shape):
                    raise TdxError(f"FusedAdamW.load_state_dict: moment shapes of {n} do not match the parameter")
                self.ts.state[n] = (m.to(device=p.device, dtype=torch.float32).clone().contiguous(), v.to(device=p.device, dtype=torch.float32).clone().contiguous())
        self.ts.step_count = int(step)
        g = self.param_groups[0]
        self.ts.set_lr(g["lr"])


class _TrainForward(torch.autograd.Function):
    """Makes the training-mode forward pass a node of torch's autograd graph, so the reference's loop (train.ipynb:198-209:
    loss = MSELoss()(model(...), eps); loss.backward(); optimizer.step()) runs unchanged: backward() runs the HIP backward
    program and hands every parameter its gradient."""

    @staticmethod
    def forward(ctx, ts: "TrainStep", x_t, t, cat, num, *params):
        ctx.ts, ctx.names, ctx.version = ts, [ts._pname[id(p)] for p in params], ts._passes + 1
        return ts.forward(x_t, t, cat, num).clone()

    @staticmethod
    def backward(ctx, dpred):
        ts = ctx.ts
        if ts._passes != ctx.version:
            raise TdxError("TrajUNet (training mode): backward() must follow the forward() it belongs to (one graph at a time per batch shape)")
        grads = ts.backward(dpred.contiguous())
        return (None, None, None, None, None) + tuple(grads[n].clone() if n in grads else None for n in ctx.names)


def train_forward(model, x_t: torch.Tensor, t: torch.Tensor, cat: torch.Tensor, num: torch.Tensor) -> torch.Tensor:
    """model(x_t, t, cat, num) in training mode (dropout on), differentiable w.r.t. the parameters."""
    B, _, L = x_t.shape
    cache = model.__dict__.setdefault("_train_steps", {})
    ts = cache.get((B, L))
    if ts is None:
        ts = cache[(B, L)] = TrainStep(model, B, L, x_t.device)
    params = [p for p in model.parameters()]
    return _TrainForward.apply(ts, x_t, t, cat, num, *params)
