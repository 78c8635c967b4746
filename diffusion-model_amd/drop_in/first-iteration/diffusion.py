"""Drop-in for first-iteration/diffusion.py."""
from _ddpm_amd import ITERATION, pkg

DiffusionModel = pkg.make_diffusion_class(ITERATION)
