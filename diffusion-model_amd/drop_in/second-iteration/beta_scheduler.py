"""Drop-in for second-iteration/beta_scheduler.py."""
from _ddpm_amd import pkg

BetaScheduler = pkg.BetaScheduler
