"""MI355X-native DDPM hot path: drop-in `unet` / `diffusion` / `beta_scheduler` / `embeddings`
modules over hand-written gfx950 kernels (libddpm_mi355x.so). No CPU fallback exists."""
from . import _lib  # noqa: F401
from .schedule import BetaScheduler  # noqa: F401
from .embedding import SinusoidalTimeEmbedding  # noqa: F401
from .process import make_diffusion_class  # noqa: F401
from .unet_impl import make_unet_class  # noqa: F401
from .scorer import SimpleCNN  # noqa: F401

__all__ = ["BetaScheduler", "SinusoidalTimeEmbedding", "make_diffusion_class", "make_unet_class", "SimpleCNN"]
