"""Deterministic state_dict filler shared by the golden-vector generator and the tests.

The reference's trained checkpoints are not available, and the default torch
initialisation leaves BatchNorm running statistics at (0, 1) and the attention
`gamma` at 0, which would make BatchNorm-eval and attention numerically inert.
This procedure gives every entry a reproducible, non-trivial value that depends
only on (seed, key name, shape), so a 30-50 MB full-width state_dict never has
to be committed: both sides regenerate it.
"""
import zlib

import torch


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2**63 - 1))
    return g


def seeded_state_dict(template, seed: int = 0):
    """template: mapping key -> tensor (only shape/dtype are used). Returns a new dict (CPU)."""
    out = {}
    for key, ref in template.items():
        shape = tuple(ref.shape)
        g = _gen(seed, key)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros(shape, dtype=torch.int64)
        elif key.endswith("running_mean"):
            out[key] = torch.randn(shape, generator=g) * 0.1
        elif key.endswith("running_var"):
            out[key] = torch.rand(shape, generator=g) + 0.5
        elif key.endswith("gamma"):
            out[key] = torch.rand(shape, generator=g) * 0.9 + 0.1
        elif len(shape) == 1 and key.endswith("weight"):      # BatchNorm scale
            out[key] = torch.rand(shape, generator=g) + 0.5
        elif len(shape) == 1:                                  # every bias
            out[key] = torch.randn(shape, generator=g) * 0.1
        else:                                                  # conv / convT / linear / embedding
            fan = 1
            for s in shape[1:]:
                fan *= s
            out[key] = torch.randn(shape, generator=g) / float(fan) ** 0.5
    return out


def seeded_tensor(shape, seed: int, tag: str, kind: str = "randn"):
    g = _gen(seed, tag)
    if kind == "randn":
        return torch.randn(tuple(shape), generator=g)
    if kind == "uniform":                                     # U[-1, 1]
        return torch.rand(tuple(shape), generator=g) * 2 - 1
    raise ValueError(kind)


def seeded_ints(shape, high: int, seed: int, tag: str):
    return torch.randint(0, high, tuple(shape), generator=_gen(seed, tag), dtype=torch.int64)
