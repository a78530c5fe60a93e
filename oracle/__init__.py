"""CPU oracle for the TrajUNet denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``trajectorydiffusion_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

The oracle is a functional restatement (stock ``torch.nn.functional`` ops on CPU, fp32)
of the reference's algorithm; each function cites the reference file:line it follows.
It is pinned by ``tests/golden/*.npz``, which were produced by importing the reference
itself in the build container (``tests/golden/make_golden.py``).
"""
from .trajunet_oracle import trajunet_forward, embed_forward, resnet_block, attn_block, nearest_index_map  # noqa: F401
from .diffusion_oracle import DiffusionOracle  # noqa: F401
from .dataset_oracle import didi_item, gaussian_cell_boundaries  # noqa: F401
from . import metrics_oracle  # noqa: F401  (evaluation metrics, parity unpinned)
