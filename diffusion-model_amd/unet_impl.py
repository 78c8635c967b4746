"""UNet modules whose state_dict matches the reference key for key; the arithmetic is the HIP
kernel schedule of engine.py. The nn layers instantiated here are parameter holders only (their
own forward is never called); they are created in the reference's construction order so that the
default initialisation under a given torch seed is identical too.

Interfaces: first-iteration/unet.py:52-115, second-iteration/unet.py:93-158,
third-iteration/unet.py:94-178.
"""
import torch
import torch.nn as nn

from .embedding import SinusoidalTimeEmbedding
from .engine import UNetEngine


class _Holder(nn.Module):
    """Names its children exactly as given (registration order = keyword order)."""

    def __init__(self, **children):
        super().__init__()
        for name, child in children.items():
            setattr(self, name, child)

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: computation happens in the UNet's kernel schedule")


def _conv3(ci, co):
    return nn.Conv2d(ci, co, kernel_size=3, padding=1)


def _double_conv(variant, ci, co):
    if variant == 1:
        seq = nn.Sequential(_conv3(ci, co), nn.BatchNorm2d(co), nn.ReLU(inplace=True),
                            _conv3(co, co), nn.BatchNorm2d(co), nn.ReLU(inplace=True))
    else:
        head = [_conv3(ci, co), nn.BatchNorm2d(co), nn.GELU()]
        res = _Holder(conv_block=nn.Sequential(_conv3(co, co), nn.BatchNorm2d(co), nn.GELU(), _conv3(co, co), nn.BatchNorm2d(co)))
        seq = nn.Sequential(*head, res)
    return _Holder(double_conv=seq)


def _attention(c):
    return _Holder(query=nn.Conv2d(c, c // 8, kernel_size=1), key=nn.Conv2d(c, c // 8, kernel_size=1),
                   value=nn.Conv2d(c, c, kernel_size=1), gamma=nn.Parameter(torch.zeros(1)))


def _down(variant, ci, co):
    parts = dict(conv=_double_conv(variant, ci, co), pool=nn.MaxPool2d(kernel_size=2, stride=2))
    if variant >= 2:
        parts["attention"] = _attention(co)
    return _Holder(**parts)


def _up(variant, ci, co):
    parts = dict(up=nn.ConvTranspose2d(ci, ci // 2, kernel_size=2, stride=2), conv=_double_conv(variant, ci, co))
    if variant >= 2:
        parts["attention"] = _attention(co)
    return _Holder(**parts)


class _UNetBase(nn.Module):
    _variant = 0

    def _build(self, in_channels, out_channels, base_channels, embedding_dim, num_classes=None):
        v, b = self._variant, base_channels
        self.time_embedding = SinusoidalTimeEmbedding(embedding_dim)
        for i, mult in enumerate((1, 2, 4, 8), start=1):
            setattr(self, f"time_fc{i}", nn.Linear(embedding_dim, b * mult))
        self.enc1 = _double_conv(v, in_channels, b)
        self.enc2 = _down(v, b, b * 2)
        self.enc3 = _down(v, b * 2, b * 4)
        if v == 1:
            self.bottleneck = _double_conv(v, b * 4, b * 8)
        else:
            self.bottleneck = nn.Sequential(_double_conv(v, b * 4, b * 8), _attention(b * 8))
        self.dec1 = _up(v, b * 8, b * 4)
        self.dec2 = _up(v, b * 4, b * 2)
        self.dec3 = _up(v, b * 2, b)
        self.final_conv = nn.Conv2d(b, out_channels, kernel_size=1)
        self.feature_maps = []
        if num_classes is not None:
            self.label_embedding = nn.Embedding(num_classes, embedding_dim)
            for i, mult in enumerate((1, 2, 4, 8), start=1):
                setattr(self, f"label_emdedding_fc{i}", nn.Linear(embedding_dim, b * mult))   # (sic) reference key
        self._engine = UNetEngine(self, v, in_channels, out_channels, b, embedding_dim, num_classes)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.invalidate()
        return out

    def _run(self, x, t, y):
        self.feature_maps.clear()
        if not x.is_cuda:
            raise RuntimeError("this UNet runs on the MI355X only: move the model and its inputs to the GPU "
                               "(there is no CPU fallback)")
        out, fms = self._engine.run(x, t, y, training=self.training)
        self.feature_maps.extend(fms)
        return out


    def denoise_step(self, x_t, t, y, step):
        """One reverse ancestral step through this network (eval mode): x_{t-1}; see UNetEngine.forward_eval."""
        self.feature_maps.clear()
        if not x_t.is_cuda:
            raise RuntimeError("this UNet runs on the MI355X only: move the model and its inputs to the GPU "
                               "(there is no CPU fallback)")
        out, fms = self._engine.denoise_step(x_t, t, y, step)
        self.feature_maps.extend(fms)
        return out


def make_unet_class(iteration):
    if iteration == 1:
        class UNet(_UNetBase):
            _variant = 1

            def __init__(self, in_channels=1, out_channels=1, base_channels=64, embedding_dim=128):
                super().__init__()
                self._build(in_channels, out_channels, base_channels, embedding_dim)

            def forward(self, x, t):
                return self._run(x, t, None)
    elif iteration == 2:
        class UNet(_UNetBase):
            _variant = 2

            def __init__(self, in_channels=3, out_channels=3, base_channels=64, embedding_dim=256):
                super().__init__()
                self._build(in_channels, out_channels, base_channels, embedding_dim)

            def forward(self, x, t):
                return self._run(x, t, None)
    else:
        class UNet(_UNetBase):
            _variant = 3

            def __init__(self, in_channels=3, out_channels=3, base_channels=64, embedding_dim=128, num_classes=None):
                super().__init__()
                self._build(in_channels, out_channels, base_channels, embedding_dim, num_classes)

            def forward(self, x, t, y=None):
                return self._run(x, t, y)
    return UNet
