"""Drop-in for third-iteration/unet.py: same class, constructor arguments and state_dict keys."""
from _ddpm_amd import ITERATION, pkg

UNet = pkg.make_unet_class(ITERATION)
