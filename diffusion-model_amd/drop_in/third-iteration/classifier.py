"""Drop-in for third-iteration/classifier.py: the scorer's SimpleCNN (inference on the MI355X)."""
from _ddpm_amd import pkg

SimpleCNN = pkg.SimpleCNN
