"""Drop-in for third-iteration/diffusion.py."""
from _ddpm_amd import ITERATION, pkg

DiffusionModel = pkg.make_diffusion_class(ITERATION)
