"""Host-side executor for the TrajUNet hot path: turns a TrajUNet (parameters only) plus a
(batch, length, dtype) into a static *program* — a flat list of C-ABI kernel launches over
pre-allocated channels-last workspaces — that can be run eagerly or captured into a HIP graph.

PyTorch is used for device memory, streams and graph capture only; every arithmetic op of
the path is a call into libtrajdiff_hip.so (trajectorydiffusion_amd/_lib.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_SWISH, ConvDesc, TdxError


# GroupNorm(+Swish) of a conv's consumer inside the conv's epilogue (tdx_conv_desc.gn_*):
#   False  never: every GroupNorm is a stand-alone tdx_groupnorm_act pass;
#   True   wherever the library can (both kernel forms; the ablation / test setting);
#   "auto" the persistent kernel's form only (tdx_conv_gemm_gn_fusable() == 2), for convs whose raw output nobody reads and whose
#          trajectories fill >= 93 % of a 256-row tile.
# Measured on MI355X, round 3 (profiles/r3_gn_per_layer_ablation.txt, profiles/r3_c5_gn_fusion.txt):
#   TrajUNet (B = 2048 per launch): the pass wins or ties on every layer.  The fused epilogue costs what the pass costs (the Swish of
#   128 values per lane, the statistics' LDS round trips and a second store sweep sit in series with the K loop, while the pass runs
#   at copy rate), and tiles of whole trajectories cannot use the small-tile tail launch (820 tiles = 4 rounds instead of 3.125).
#   SDUNet_CAR (3 .. 24 rows per trajectory, launch-latency-sized GroupNorm passes): "auto" is 1.9 % faster per denoise step.
# Hence: False by default, SDUNet_CAR asks for "auto" (Program.fuse_gn).  TDX_FUSE_GN=0|1|auto in the environment overrides both.
FUSE_GN_ENV = {"0": False, "1": True, "auto": "auto"}.get(os.environ.get("TDX_FUSE_GN", ""), None)
FUSE_GN_DEFAULT = False if FUSE_GN_ENV is None else FUSE_GN_ENV
FUSE_GN_MIN_FILL = 0.93
# Split GroupNorm on the full-resolution 128-channel level (bf16 programs): producers leave block partial sums (stats_out),
# tdx_groupnorm_finalize turns them into {rstd, -mean * rstd}, and the consuming conv normalises + activates its input while it
# loads it (tdx_conv_desc.norm): no stand-alone GroupNorm pass and no normalised copy of those activations in memory.
# Measured on MI355X (profiles/r2_split_norm_ablation.txt): the fused level costs 0.99 ms per denoise step against 1.02 ms for
# conv + stand-alone GroupNorm, because the fused conv is bound by vector-ALU issue (the Swish of the transform plus per-tile
# bookkeeping next to 96 MFMAs per wave), not by HBM.  Correct, deterministic and tested, but not a win yet: off by default.
FUSE_NORM_DEFAULT = False


def conv_out_len(l: int, stride: int) -> int:
    """Conv1d(k=3, p=1, stride) output length (Models/TrajUNet.py:52)."""
    return (l - 1) // stride + 1


def nearest_map(l_in: int, l_out: int) -> torch.Tensor:
    """F.interpolate(mode='nearest', size=l_out) source rows (Models/TrajUNet.py:93), int32 on CPU.

    ATen: src = min(floor(dst * float32(l_in / l_out)), l_in - 1).
    """
    scale = torch.tensor(l_in / l_out, dtype=torch.float32)
    idx = torch.floor(torch.arange(l_out, dtype=torch.float32) * scale).to(torch.int32)
    return idx.clamp_(max=l_in - 1)


def pack_conv_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(Cout, Cin, k) -> (Cout, k*Cin) in the kernels' K order, cast to the compute dtype.

    K order = (128-byte channel block, tap, channel in block): the taps of one channel block are adjacent, so the
    three shifted reads of a k=3 conv happen back to back on rows that are already on chip.
    """
    w = w.detach()
    cout, cin, k = w.shape
    bk = 128 // torch.empty((), dtype=dtype).element_size()
    if k == 1:
        return w.reshape(cout, cin).to(dtype).contiguous()
    blocks = [w[:, c0:c0 + bk, :].permute(0, 2, 1).reshape(cout, -1) for c0 in range(0, cin, bk)]
    return torch.cat(blocks, dim=1).to(dtype).contiguous()


def pack_conv_weight_dgrad(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Weights of the data-gradient conv of a stride-1 Conv1d (training step).

    y[l] = sum_t W[:, :, t] x[l + t - pad]  =>  dx[u] = sum_t' W[:, :, k-1-t']^T dy[u + t' - pad]: the same gathered GEMM
    (tdx_conv_gemm) run on dy with the taps flipped and the channel roles exchanged."""
    return pack_conv_weight(w.detach().flip(2).transpose(0, 1).contiguous(), dtype)


def pack_conv_weight_dgrad_stride2(w: torch.Tensor, dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """Weights of the two phases of a stride-2, k=3, pad-1 Conv1d's data gradient (see tdx_interleave2).

    y[l] = sum_t W_t x[2l + t - 1]  =>  dx[2m] = W_1^T dy[m],  dx[2m + 1] = W_2^T dy[m] + W_0^T dy[m + 1]."""
    w = w.detach()
    wt = w.transpose(0, 1)                                   # (Cin, Cout, 3)
    even = pack_conv_weight(wt[:, :, 1:2].contiguous(), dtype)
    odd = torch.stack([torch.zeros_like(wt[:, :, 0]), wt[:, :, 2], wt[:, :, 0]], dim=2)   # taps read dy[m - 1], dy[m], dy[m + 1]
    return even, pack_conv_weight(odd.contiguous(), dtype)


def conv_dgrad_stride2(pg: "Program", dy: torch.Tensor, w_even: torch.Tensor, w_odd: torch.Tensor, dx: torch.Tensor, B: int, L: int,
                       cout: int, cin: int, accumulate: bool = False) -> None:
    """Record dx (B, L, cin) (+)= data gradient of a stride-2 conv from dy (B, ceil(L/2), cout): two tdx_conv_gemm launches on
    dy (one per phase) and tdx_interleave2."""
    lo = (L - 1) // 2 + 1
    ev = torch.empty((B, lo, cin), device=dy.device, dtype=dy.dtype)
    od = torch.empty((B, lo, cin), device=dy.device, dtype=dy.dtype)
    pg.conv([Seg(dy, cout, cout, lo, taps=1)], w_even, None, ev, cin, B, lo, cin)
    pg.conv([Seg(dy, cout, cout, lo, taps=3)], w_odd, None, od, cin, B, lo, cin)
    pg.keep += [ev, od, dx]
    pg.call("dgrad.interleave", pg.lib.tdx_interleave2, ev.data_ptr(), od.data_ptr(), dx.data_ptr(), dx.shape[-1], 0, B, L, cin,
            1 if accumulate else 0, pg.code)


class Seg:
    """One K-segment of a gathered GEMM (see tdx_kseg in include/trajdiff_hip.h)."""
    __slots__ = ("src", "ld", "ch_off", "cin", "taps", "stride", "lin", "lvirt")

    def __init__(self, src: torch.Tensor, ld: int, cin: int, lin: int, taps: int = 3, stride: int = 1, ch_off: int = 0,
                 lvirt: Optional[int] = None):
        """lvirt != lin: the conv runs on F.interpolate(src, size=lvirt, mode='nearest') without materialising it."""
        self.src, self.ld, self.ch_off, self.cin, self.taps, self.stride = src, ld, ch_off, cin, taps, stride
        self.lin = lin
        self.lvirt = lin if lvirt is None else lvirt


class Program:
    """A recorded sequence of C-ABI launches.  `run()` replays them on the current stream."""

    def __init__(self, device: torch.device, dtype: torch.dtype):
        self.device, self.dtype = device, dtype
        self.code = _lib.dtype_code(dtype)
        self.lib = _lib.lib()
        self.ops: List[Tuple] = []
        self.flops: Dict[int, float] = {}  # op index -> algorithmic FLOPs (GEMM ops only)
        self.keep: List = []  # tensors / descriptors referenced by raw pointer
        self._pool: Dict[Tuple[int, torch.dtype], List[torch.Tensor]] = {}
        # Fusing the consumer's GroupNorm into the conv epilogue is implemented and tested, but on MI355X the separate
        # GroupNorm pass (full-chip HBM stream) beats a longer, un-overlapped epilogue on every TrajUNet shape
        # (profiles/r1_gn_fusion_ablation.txt), so programs use it only on request.
        self.fuse_gn = FUSE_GN_DEFAULT
        self.fuse_norm = FUSE_NORM_DEFAULT
        # fp32 workspace that lets tdx_conv_gemm cut K into concurrent slices when a conv has too few tiles to fill the chip
        # (small batches); shared by every conv of the program (they run one after the other on one stream).  None = never.
        self.split_k_ws: Optional[torch.Tensor] = None

    # ---- workspace -------------------------------------------------------------------------
    def alloc(self, shape: Sequence[int], dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        dtype = self.dtype if dtype is None else dtype
        n = 1
        for s in shape:
            n *= int(s)
        free = self._pool.get((n, dtype))
        if free:
            return free.pop().view(*shape)
        t = torch.empty(n, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t.view(*shape)

    def release(self, t: torch.Tensor) -> None:
        """Return a workspace to the pool: later ops of the program may overwrite it."""
        self._pool.setdefault((t.numel(), t.dtype), []).append(t.reshape(-1))

    # ---- op recording ----------------------------------------------------------------------
    def _add(self, name: str, fn, *args) -> None:
        self.ops.append((name, fn, args))

    def conv(self, segs: Sequence[Seg], w: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor], out_ld: int,
             B: int, lout: int, cout: int, out_off: int = 0, act: int = ACT_NONE, res: Optional[torch.Tensor] = None,
             res_ld: int = 0, ebias: Optional[torch.Tensor] = None, tbias: Optional[torch.Tensor] = None, ebias_ld: int = 0,
             ebias_off: int = 0, dtype_code: Optional[int] = None, name: str = "conv", gn: Optional["GnSpec"] = None,
             norm: Optional["NormIn"] = None, stats_out: Optional[torch.Tensor] = None) -> None:
        """Record one tdx_conv_gemm.  `gn` = the consumer's GroupNorm(+act): fused into the conv epilogue when the library
        can (then `out` may be None = the raw tensor is not needed), else run as a separate tdx_groupnorm_act."""
        d = ConvDesc()
        ktot = 0
        for i, s in enumerate(segs):
            k = d.seg[i]
            k.src = s.src.data_ptr()
            # ATen's nearest scale: static_cast<float>(input_size) / output_size, one fp32 division
            k.up_scale = float(torch.tensor(float(s.lin), dtype=torch.float32) / torch.tensor(float(s.lvirt), dtype=torch.float32))
            k.src_ld, k.ch_off, k.cin, k.taps, k.stride, k.lin, k.lvirt = s.ld, s.ch_off, s.cin, s.taps, s.stride, s.lin, s.lvirt
            ktot += s.taps * s.cin
            self.keep.append(s.src)
        if tuple(w.shape) != (cout, ktot):
            raise TdxError(f"{name}: packed weight is {tuple(w.shape)}, segments need ({cout}, {ktot})")
        d.nseg, d.dtype = len(segs), self.code if dtype_code is None else dtype_code
        d.w, d.bias = w.data_ptr(), _lib.ptr(bias)
        d.ebias, d.tbias, d.ebias_ld, d.ebias_off = _lib.ptr(ebias), _lib.ptr(tbias), ebias_ld, ebias_off
        d.res, d.res_ld, d.act = _lib.ptr(res), res_ld, act
        d.out, d.out_ld, d.out_off = _lib.ptr(out), out_ld, out_off
        d.B, d.lout, d.cout = B, lout, cout
        if norm is not None:   # GroupNorm(+act) of segment 0, applied while it is loaded
            n0 = d.norm[0]
            n0.mr, n0.gamma, n0.beta = norm.mr.data_ptr(), norm.gamma.data_ptr() + 4 * norm.p_off, norm.beta.data_ptr() + 4 * norm.p_off
            n0.cpg, n0.act, n0.mr_ld, n0.mr_off = norm.cpg, norm.act, norm.mr.shape[1], norm.mr_off
            self.keep.extend([norm.mr, norm.gamma, norm.beta])
        if stats_out is not None:   # (B, ceil(lout / 16), granules, 2) fp32 block partial sums of the output
            d.stats_out, d.stats_out_ld, d.stats_out_off = stats_out.data_ptr(), stats_out.shape[2], 0
            self.keep.append(stats_out)
        if (norm is not None or stats_out is not None) and not self.lib.tdx_conv_gemm_norm_fusable(C.byref(d)):
            raise TdxError(f"{name}: this shape cannot take norm-on-load / stats_out (ask norm_fusable() before recording it)")
        fused, tmp = False, None
        if gn is not None:
            d.gn_gamma, d.gn_beta = gn.gamma.data_ptr() + 4 * gn.p_off, gn.beta.data_ptr() + 4 * gn.p_off
            d.gn_out, d.gn_ld, d.gn_off, d.gn_act, d.gn_cpg, d.gn_eps = gn.out.data_ptr(), gn.out_ld, gn.out_off, gn.act, gn.cpg, gn.eps
            form = int(self.lib.tdx_conv_gemm_gn_fusable(C.byref(d))) if self.fuse_gn else 0
            if self.fuse_gn == "auto":
                fused = form == 2 and out is None and (256 // lout) * lout >= FUSE_GN_MIN_FILL * 256
            else:
                fused = form > 0
            if not fused:
                d.gn_gamma = None
                if out is None:  # the separate GroupNorm kernel needs the raw tensor
                    tmp = out = self.alloc((B, lout, cout))
                    d.out, d.out_ld, d.out_off = out.data_ptr(), cout, 0
                    out_ld, out_off = cout, 0
            self.keep.extend([gn.gamma, gn.beta, gn.out])
        if self.split_k_ws is not None and not fused:
            d.ws, d.ws_bytes = self.split_k_ws.data_ptr(), self.split_k_ws.numel() * self.split_k_ws.element_size()
        self.keep.extend([d, w, bias, out, res, ebias, tbias])
        self._add(name, self.lib.tdx_conv_gemm, C.byref(d))
        self.flops[len(self.ops) - 1] = 2.0 * B * lout * cout * ktot
        if gn is not None and not fused:
            if out_off != 0:
                raise TdxError(f"{name}: unfused GroupNorm fallback needs the raw tensor at channel offset 0")
            self.groupnorm(out, out_ld, gn.out, gn.out_ld, gn.out_off, gn.gamma, gn.beta, gn.p_off, B, lout, cout, cout // gn.cpg,
                           gn.eps, gn.act, name=name + ".gn")
            if tmp is not None:
                self.release(tmp)

    def groupnorm(self, x: torch.Tensor, x_ld: int, out: torch.Tensor, out_ld: int, out_off: int, gamma: torch.Tensor,
                  beta: torch.Tensor, p_off: int, B: int, L: int, Cn: int, groups: int, eps: float, act: int,
                  name: str = "gn") -> None:
        self.keep.extend([x, out, gamma, beta])
        self._add(name, self.lib.tdx_groupnorm_act, x.data_ptr(), x_ld, out.data_ptr(), out_ld, out_off,
                  gamma.data_ptr() + 4 * p_off, beta.data_ptr() + 4 * p_off, B, L, Cn, groups, eps, act, self.code)

    def call(self, name: str, fn, *args, keep: Sequence = ()) -> None:
        self.keep.extend(keep)
        self._add(name, fn, *args)

    # ---- execution -------------------------------------------------------------------------
    def run(self) -> None:
        st = _lib.stream_ptr()
        for name, fn, args in self.ops:
            rc = fn(*args, st)
            if rc:
                _lib.check(rc, name)

    def profile(self, repeats: int = 3, inner: int = 1):
        """Per-op device time (ms, best of `repeats`) from HIP events on the launch stream.  inner > 1 launches every op that many
        times between its two events and divides: an event record is a packet of its own on the queue, worth several microseconds
        next to a 50-microsecond kernel (rocprofv3's kernel durations carry no such share)."""
        st = _lib.stream_ptr()
        best = [float("inf")] * len(self.ops)
        for _ in range(repeats):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(len(self.ops))]
            for i, (name, fn, args) in enumerate(self.ops):
                evs[i][0].record()
                for _ in range(inner):
                    _lib.check(fn(*args, st), name)
                evs[i][1].record()
            torch.cuda.synchronize()
            for i in range(len(self.ops)):
                best[i] = min(best[i], evs[i][0].elapsed_time(evs[i][1]) / inner)
        return [(self.ops[i][0], self.ops[i][1].__name__, best[i], self.flops.get(i, 0.0)) for i in range(len(self.ops))]


def fp32_param(p: torch.Tensor, name: str) -> torch.Tensor:
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise TdxError(f"parameter {name} must be contiguous fp32 (got {p.dtype}); keep master weights in fp32")
    return p.detach()


class NormIn:
    """GroupNorm(+act) that a conv applies to its input while loading it: the {rstd, -mean * rstd} table `mr` (B, groups, 2) written
    by tdx_groupnorm_finalize from the producer's block partial sums, and the affine parameters (first channel p_off)."""
    __slots__ = ("mr", "mr_off", "gamma", "beta", "p_off", "cpg", "act")

    def __init__(self, gn_module, name: str, mr: torch.Tensor, cpg: int, act: int, p_off: int = 0, mr_off: int = 0):
        self.gamma, self.beta = fp32_param(gn_module.weight, name + ".weight"), fp32_param(gn_module.bias, name + ".bias")
        self.mr, self.mr_off, self.p_off, self.cpg, self.act = mr, mr_off, p_off, cpg, act


def norm_fusable_shape(pg: "Program", B: int, lin: int, lout: int, cin: int, cout: int, norm: bool = True, stats: bool = True,
                       stride: int = 1) -> bool:
    """Can a (cin -> cout, k=3) conv of this program take norm-on-load (`norm`) and / or emit block statistics (`stats`)?  Asks the
    library with a representative descriptor (today: bf16, 128 -> 128 channels; norm-on-load on un-resampled stride-1 sources)."""
    if pg.dtype != torch.bfloat16 or cin % 8 or cout % 8 or not getattr(pg, "fuse_norm", False):
        return False
    d = ConvDesc()
    k = d.seg[0]
    probe = torch.empty(64, device=pg.device, dtype=pg.dtype)
    lvirt = lin if stride == 2 or lin == lout else lout
    k.src, k.src_ld, k.ch_off, k.cin, k.taps, k.stride, k.lin, k.lvirt = probe.data_ptr(), cin, 0, cin, 3, stride, lin, lvirt
    k.up_scale = float(torch.tensor(float(lin), dtype=torch.float32) / torch.tensor(float(lvirt), dtype=torch.float32))
    d.nseg, d.dtype, d.w, d.out, d.out_ld, d.B, d.lout, d.cout = 1, pg.code, probe.data_ptr(), probe.data_ptr(), cout, B, lout, cout
    if stats:
        d.stats_out, d.stats_out_ld = probe.data_ptr(), cout // 4
    if norm:
        cpg = max(4, cin // 32)
        d.norm[0].mr, d.norm[0].gamma, d.norm[0].beta = probe.data_ptr(), probe.data_ptr(), probe.data_ptr()
        d.norm[0].cpg, d.norm[0].act, d.norm[0].mr_ld = cpg, ACT_SWISH, cin // cpg
    return bool(pg.lib.tdx_conv_gemm_norm_fusable(C.byref(d)))


def emit_finalize(pg: "Program", stats: torch.Tensor, B: int, L: int, groups: int, cpg: int, eps: float, name: str) -> torch.Tensor:
    """tdx_groupnorm_finalize: block partial sums (B, ceil(L/16), granules, 2) -> {rstd, -mean * rstd} table (B, groups, 2)."""
    mr = pg.alloc((B, groups, 2), torch.float32)
    pg.call(name, pg.lib.tdx_groupnorm_finalize, stats.data_ptr(), stats.shape[2], 0, mr.data_ptr(), groups, 0, B, L, groups, cpg, float(eps),
            keep=[stats, mr])
    return mr


def emit_resblock_split_norm(pg: "Program", blk, packed: Dict[str, torch.Tensor], key: str, x: torch.Tensor, x_mr: torch.Tensor, B: int, L: int,
                             ebias: Optional[torch.Tensor], tbias: Optional[torch.Tensor], e_ld: int, e_off: int,
                             next_eps: Optional[float]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """ResnetBlock (Models/TrajUNetBlocks.py:92-99, in_c == out_c, no 1x1 shortcut) with both GroupNorm+Swish applied inside the
    convs that consume them: conv1 reads the raw x and normalises it on load (x_mr), leaves the raw h and its block sums;
    conv2 reads the raw h the same way, adds x, and (next_eps given) leaves the block sums of its output for the next consumer.
    Returns (raw output, its {rstd, -mean * rstd} table or None)."""
    c = blk.out_c
    if blk.in_c != c or blk.has_shortcut:
        raise TdxError(f"{key}: split-norm dataflow needs in_c == out_c")
    groups, cpg = blk.norm_groups, c // blk.norm_groups
    conv1, gn1, gn2, conv2 = blk.stage1[2], blk.stage1[0], blk.stage2[0], blk.stage2[3]
    nblk = (L + 15) // 16
    h = pg.alloc((B, L, c))
    h_stats = pg.alloc((B, nblk, c // 4, 2), torch.float32)
    pg.conv([Seg(x, c, c, L)], packed[key + "conv1.w"], fp32_param(conv1.bias, key + "stage1.2.bias"), h, c, B, L, c, ebias=ebias, tbias=tbias,
            ebias_ld=e_ld, ebias_off=e_off, name=key + "conv1", norm=NormIn(gn1, key + "stage1.0", x_mr, cpg, ACT_SWISH), stats_out=h_stats)
    h_mr = emit_finalize(pg, h_stats, B, L, groups, cpg, gn2.eps, key + "conv1.norm")
    out = pg.alloc((B, L, c))
    o_stats = pg.alloc((B, nblk, c // 4, 2), torch.float32) if next_eps is not None else None
    pg.conv([Seg(h, c, c, L)], packed[key + "conv2.w"], fp32_param(conv2.bias, key + "stage2.3.bias"), out, c, B, L, c, res=x, res_ld=c,
            name=key + "conv2", norm=NormIn(gn2, key + "stage2.0", h_mr, cpg, ACT_SWISH), stats_out=o_stats)
    o_mr = emit_finalize(pg, o_stats, B, L, groups, cpg, next_eps, key + "conv2.norm") if next_eps is not None else None
    for t in (h, h_stats, h_mr):
        pg.release(t)
    if o_stats is not None:
        pg.release(o_stats)
    return out, o_mr


class GnSpec:
    """A consumer's GroupNorm(+act) to be applied to a producer's output rows: parameters (gamma/beta fp32, first
    channel p_off), channels per group, and where the normalised rows go."""
    __slots__ = ("gamma", "beta", "p_off", "cpg", "eps", "act", "out", "out_ld", "out_off")

    def __init__(self, gn_module, name: str, p_off: int, cpg: int, act: int, out: torch.Tensor, out_ld: int, out_off: int):
        self.gamma, self.beta = fp32_param(gn_module.weight, name + ".weight"), fp32_param(gn_module.bias, name + ".bias")
        self.p_off, self.cpg, self.eps, self.act = p_off, cpg, gn_module.eps, act
        self.out, self.out_ld, self.out_off = out, out_ld, out_off


def block_input_gn(pg: Program, blk, key: str, B: int, L: int, part_off: int = 0, a1: Optional[torch.Tensor] = None) -> GnSpec:
    """GnSpec of a ResnetBlock's first GroupNorm+Swish (Models/TrajUNetBlocks.py:75-76) for the channel slice that starts at
    part_off of its (possibly concatenated) input; allocates the activated buffer (B, L, in_c) unless given."""
    if a1 is None:
        a1 = pg.alloc((B, L, blk.in_c))
    return GnSpec(blk.stage1[0], key + "stage1.0", part_off, blk.in_c // blk.norm_groups, ACT_SWISH, a1, blk.in_c, part_off)


def emit_groupnorm(pg: Program, src: torch.Tensor, c: int, B: int, L: int, gn: GnSpec, name: str) -> None:
    """Stand-alone GroupNorm kernel for a producer that is not a conv GEMM (stem) or a second consumer (decoder skip half)."""
    if c % gn.cpg:
        raise TdxError(f"{name}: {c} channels do not hold whole GroupNorm groups of {gn.cpg}")
    pg.groupnorm(src, c, gn.out, gn.out_ld, gn.out_off, gn.gamma, gn.beta, gn.p_off, B, L, c, c // gn.cpg, gn.eps, gn.act, name=name)


def emit_resblock(pg: Program, blk, packed: Dict[str, torch.Tensor], key: str, a1: torch.Tensor,
                  srcs: Sequence[Tuple[torch.Tensor, int]], B: int, L: int, ebias: Optional[torch.Tensor],
                  tbias: Optional[torch.Tensor], e_ld: int, e_off: int, want_raw: bool = True,
                  next_gn: Optional[GnSpec] = None) -> Optional[torch.Tensor]:
    """ResnetBlock (Models/TrajUNetBlocks.py:92-99) over channels-last tensors.

    a1   = Swish(GroupNorm(x)) (B, L, in_c), produced by whoever produced x (fused into that conv's epilogue);
    srcs = the raw x as [(tensor (B, L, C_k), C_k), ...]: more than one entry is the decoder's channel concat
           (Models/TrajUNet.py:90), never materialised: the 1x1 shortcut reads each source as its own K-segment.
    conv1's epilogue applies this block's second GroupNorm+Swish (its raw output is never stored); conv2's epilogue adds the
    shortcut/residual, stores the raw output when `want_raw`, and applies the NEXT consumer's GroupNorm (`next_gn`).
    """
    cin = sum(c for _, c in srcs)
    cout = blk.out_c
    if cin != blk.in_c:
        raise TdxError(f"{key}: sources carry {cin} channels, block expects {blk.in_c}")
    conv1, gn2, conv2 = blk.stage1[2], blk.stage2[0], blk.stage2[3]
    a2 = pg.alloc((B, L, cout))
    pg.conv([Seg(a1, cin, cin, L)], packed[key + "conv1.w"], fp32_param(conv1.bias, key + "stage1.2.bias"), None, cout, B, L, cout,
            ebias=ebias, tbias=tbias, ebias_ld=e_ld, ebias_off=e_off, name=key + "conv1",
            gn=GnSpec(gn2, key + "stage2.0", 0, cout // blk.norm_groups, ACT_SWISH, a2, cout, 0))
    out = pg.alloc((B, L, cout)) if (want_raw or next_gn is None) else None
    segs = [Seg(a2, cout, cout, L)]
    if blk.has_shortcut:
        segs += [Seg(src, c, c, L, taps=1) for src, c in srcs]
        pg.conv(segs, packed[key + "conv2.w"], packed[key + "conv2.b"], out, cout, B, L, cout, name=key + "conv2", gn=next_gn)
    else:
        pg.conv(segs, packed[key + "conv2.w"], fp32_param(conv2.bias, key + "stage2.3.bias"), out, cout, B, L, cout,
                res=srcs[0][0], res_ld=cin, name=key + "conv2", gn=next_gn)
    pg.release(a2)
    return out


def pack_resblock(blk, key: str, dtype: torch.dtype, out: Dict[str, torch.Tensor]) -> None:
    out[key + "conv1.w"] = pack_conv_weight(blk.stage1[2].weight, dtype)
    w2 = pack_conv_weight(blk.stage2[3].weight, dtype)
    if blk.has_shortcut:
        out[key + "conv2.w"] = torch.cat([w2, pack_conv_weight(blk.shortcut.weight, dtype)], dim=1).contiguous()
        out[key + "conv2.b"] = (blk.stage2[3].bias.detach() + blk.shortcut.bias.detach()).float().contiguous()
    else:
        out[key + "conv2.w"] = w2


def emit_embed(pg: Program, emb, packed: Dict[str, torch.Tensor], B: int, time: torch.Tensor, cat: torch.Tensor,
               num: torch.Tensor, err: torch.Tensor, emb_out: torch.Tensor, want_time: bool = True,
               want_attr: bool = True) -> None:
    """WideAndDeepEmbedBlock.forward (Models/TrajUNetBlocks.py:45-62): emb_out (B, 256) fp32 =
    swish([time_embed | attr_embed]).  All fp32 regardless of the UNet compute dtype."""
    f32 = torch.float32
    pe = pg.alloc((B, 128), f32)
    catvec = pg.alloc((B, 1280), f32)
    deep_in = pg.alloc((B, 1024), f32)
    pg.call("embed_gather", pg.lib.tdx_embed_gather, time.data_ptr(), cat.data_ptr(), num.data_ptr(), emb.pos_enc.data_ptr(),
            emb.pos_enc.shape[0], fp32_param(emb.cell_index_embedder.weight, "cell").data_ptr(),
            fp32_param(emb.day_embedder.weight, "day").data_ptr(), fp32_param(emb.wide_fc.weight, "wide_w").data_ptr(),
            fp32_param(emb.wide_fc.bias, "wide_b").data_ptr(), pe.data_ptr(), catvec.data_ptr(), deep_in.data_ptr(), B,
            err.data_ptr(), keep=[time, cat, num, emb.pos_enc, err])
    F32 = _lib.TDX_F32

    def lin(src, k, w, b, out, out_ld, n, off, act, name):
        pg.conv([Seg(src, k, k, 1, taps=1)], w, b, out, out_ld, B, 1, n, out_off=off, act=act, dtype_code=F32, name=name)

    if want_time:
        t0, t2 = emb.time_embed_layers[0], emb.time_embed_layers[2]
        h1 = pg.alloc((B, 128), f32)
        lin(pe, 128, fp32_param(t0.weight, "t0.w"), fp32_param(t0.bias, "t0.b"), h1, 128, 128, 0, ACT_LEAKY, "embed.time0")
        lin(h1, 128, fp32_param(t2.weight, "t2.w"), fp32_param(t2.bias, "t2.b"), emb_out, 256, 128, 0, ACT_SWISH, "embed.time2")
        pg.release(h1)
    if want_attr:
        cf, d1, d3 = emb.categorical_fc, emb.deep_fc[1], emb.deep_fc[3]
        lin(catvec, 1280, fp32_param(cf.weight, "cf.w"), fp32_param(cf.bias, "cf.b"), deep_in, 1024, 768, 0, ACT_LEAKY, "embed.cat_fc")
        hd = pg.alloc((B, 512), f32)
        lin(deep_in, 1024, fp32_param(d1.weight, "d1.w"), fp32_param(d1.bias, "d1.b"), hd, 512, 512, 0, ACT_LEAKY, "embed.deep1")
        lin(hd, 512, fp32_param(d3.weight, "d3.w"), fp32_param(d3.bias, "d3.b"), emb_out, 256, 128, 128, ACT_SWISH, "embed.deep3")
        pg.release(hd)
    for t in (pe, catvec, deep_in):
        pg.release(t)


def check_err_flag(err: torch.Tensor, what: str) -> None:
    code = int(err.item())
    if code:
        err.zero_()
        msgs = [m for bit, m in ((1, "time step out of [0, diffusion_steps)"), (2, "categorical attribute index out of range"),
                                 (4, "q_sample time step out of range")) if code & bit]
        raise IndexError(f"{what}: " + "; ".join(msgs))
