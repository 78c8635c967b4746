"""Drop-in for first-iteration/embeddings.py."""
from _ddpm_amd import pkg

SinusoidalTimeEmbedding = pkg.SinusoidalTimeEmbedding
