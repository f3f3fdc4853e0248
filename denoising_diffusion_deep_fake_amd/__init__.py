"""MI355X-native U-Net hot path of d3f (ChainBreak/denoising_diffusion_deep_fake).

`Unet` is the drop-in for `segmentation_models_pytorch.Unet` as the reference constructs it;
all device work happens in hand-written gfx950 kernels behind include/d3f_hip.h.
"""
from ._lib import D3FError, F32, BF16  # noqa: F401
from .unet import Unet, UnetPair  # noqa: F401

__version__ = "0.1.0"
