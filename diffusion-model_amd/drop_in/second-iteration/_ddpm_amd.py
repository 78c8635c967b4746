"""Locates the MI355X package for the bare-name drop-in modules of this directory."""
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)
import diffusion_model_amd as pkg  # noqa: E402

ITERATION = 2
