"""Drop-in for third-iteration/embeddings.py."""
from _ddpm_amd import pkg

SinusoidalTimeEmbedding = pkg.SinusoidalTimeEmbedding
