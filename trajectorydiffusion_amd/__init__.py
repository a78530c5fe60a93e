"""trajectorydiffusion_amd — MI355X-native TrajUNet denoising hot path.

Drop-in for the reference's hot-path objects (same names, signatures and state_dict keys):

    from trajectorydiffusion_amd import TrajUNet, DiffusionManager, DidiTrajectoryDataset, collectFunc
    from trajectorydiffusion_amd import SDUNet_CAR, VAE_InvTrans          # the latent path of eval_sd.ipynb

All arithmetic runs in hand-written gfx950 HIP kernels (libtrajdiff_hip.so, C ABI in
include/trajdiff_hip.h).  There is no CPU / PyTorch fallback for the hot path.
"""
from .Models.TrajUNet import TrajUNet  # noqa: F401
from .Models.TrajUNetBlocks import AttnBlock, ResnetBlock, WideAndDeepEmbedBlock  # noqa: F401
from .Models.StableDiffusion import SDUNet_CAR  # noqa: F401
from .Models.StableDiffusionBlocks import ConditionMixBlock, CrossAttentionRes  # noqa: F401
from .Models.VAE import VAE_InvTrans  # noqa: F401
from .DiffusionManager import DiffusionManager  # noqa: F401
from .Dataset.DidiDataset import DidiTrajectoryDataset, collectFunc  # noqa: F401

__all__ = ["TrajUNet", "AttnBlock", "ResnetBlock", "WideAndDeepEmbedBlock", "SDUNet_CAR", "ConditionMixBlock", "CrossAttentionRes", "VAE_InvTrans",
           "DiffusionManager", "DidiTrajectoryDataset", "collectFunc"]
