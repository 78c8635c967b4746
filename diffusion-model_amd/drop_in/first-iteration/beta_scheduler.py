"""Drop-in for first-iteration/beta_scheduler.py."""
from _ddpm_amd import pkg

BetaScheduler = pkg.BetaScheduler
