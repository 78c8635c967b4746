"""Drop-in for second-iteration/embeddings.py."""
from _ddpm_amd import pkg

SinusoidalTimeEmbedding = pkg.SinusoidalTimeEmbedding
