"""Weight cache shared by the whole-model classes (TrajUNet, SDUNet_CAR, VAE_InvTrans): GEMM-layout copies of the parameters in the
compute dtype, refreshed in place whenever a parameter changes, plus the per-shape program cache that hangs off them."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .TrajUNetBlocks import _HipBlock


class _PackedModel(_HipBlock):
    """Subclasses implement _build_packed(dtype) -> {name: tensor}."""

    def _init_cache(self) -> None:
        self._packed: Dict[torch.dtype, Tuple] = {}   # dtype -> (version sum, parameter addresses, norm digest, tensors)
        self.weights_epoch = 0                          # bumped by every (re)pack: samplers re-derive their time-step tables
        self._programs: Dict[Tuple, object] = {}
        self.low_latency = False

    def _fingerprint(self) -> Tuple[int, Tuple[int, ...]]:
        """(sum of the parameters' autograd versions, their storage addresses).  In-place updates through the parameter
        (optimizers, load_state_dict, copy_ under no_grad) bump a version; p.data = t / load_state_dict(assign=True) / .to()
        change an address.  Writes through `.data` (param.data.copy_(...), the reference's Utils.EMA.getModel, Utils.py:107) change
        NEITHER: the sampler detects them with _data_digest(); for a bare eval forward() call invalidate() after such a write."""
        ps = list(self.parameters()) + list(self.buffers())
        return sum(p._version for p in ps), tuple(p.data_ptr() for p in ps)

    def _data_digest(self) -> Optional[torch.Tensor]:
        """Three figures per parameter AND floating-point buffer (BatchNorm running statistics), on the device: L2 norm, sum, and the
        sum of the odd-indexed elements.  Compared against the figures the cached GEMM-layout copies were built from, this catches
        weight updates that no version counter sees, including ones that keep a tensor's norm (sign flips, most permutations, swaps of
        equal-norm tensors change the sums).  Not a cryptographic check: a write that preserves all three figures of every tensor goes
        unseen, and invalidate() is the explicit way.  A few passes over the weights, no host sync of its own."""
        ps = [p.detach() for p in self.parameters()] + [b.detach() for b in self.buffers() if b.is_floating_point()]
        if not ps or not all(p.is_cuda for p in ps):
            return None
        flat = [p.reshape(-1).float() for p in ps]
        return torch.cat([torch.stack(torch._foreach_norm(flat)), torch.stack([f.sum() for f in flat]), torch.stack([f[1::2].sum() for f in flat])])

    def invalidate(self):
        """Force the cached GEMM-layout weight copies to be refreshed before the next forward / sampling call.  Needed only
        after writing weights through `.data` before a bare eval-mode forward(); everything else is detected."""
        for dt, hit in list(self._packed.items()):
            self._packed[dt] = (-1,) + tuple(hit[1:])
        return self

    def _build_packed(self, dtype: torch.dtype) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def packed_weights(self, dtype: torch.dtype, check_data: bool = False) -> Dict[str, torch.Tensor]:
        """GEMM-layout weights in the compute dtype, refreshed whenever a parameter changes.  A refresh writes INTO the cached
        tensors, so programs and captured graphs (which hold their addresses) stay valid; only when a parameter's storage
        moved are the programs (which also hold raw parameter pointers: biases, GroupNorm affine, stem/head) dropped.
        check_data: also compare the parameters' norms with those at packing time (writes through `.data`)."""
        ver, ptrs = self._fingerprint()
        hit = self._packed.get(dtype)
        if hit is not None and hit[1] != ptrs:
            self._packed.clear()
            self._programs.clear()
            hit = None
        digest = None
        stale = hit is None or hit[0] != ver
        if not stale and check_data:
            digest = self._data_digest()
            stale = digest is not None and (hit[2] is None or not torch.equal(digest, hit[2]))
        if not stale:
            return hit[3]
        new = self._build_packed(dtype)
        if hit is not None:
            with torch.no_grad():
                for k, v in new.items():
                    hit[3][k].copy_(v)
            new = hit[3]
        self._packed[dtype] = (ver, ptrs, self._data_digest() if digest is None else digest, new)
        self.weights_epoch += 1
        return new

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._packed.clear()
        self._programs.clear()
        self.__dict__.pop("_train_steps", None)   # TrainStep programs hold views of the old parameter storage
        return self
