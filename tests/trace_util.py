"""The sequence in which the UNet's hand-written backward (engine_train.TrainMixin.backward) asks for gradient storage and
marks its flush points: recorded on the GPU from the real engine (TraceSink), committed as tests/golden/backward_trace.json,
and replayed on CPU tensors through the real BucketedGradientAverager with two gloo ranks (tests/test_parallel_gloo.py).
A GPU test keeps the committed trace equal to what the engine does today."""
import json
import os

import torch

TRACE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "backward_trace.json")

# (name, variant, constructor arguments, input shape, labels?)
TRACE_MODELS = (
    ("v2_full_width_28", 2, (1, 1, 64, 128), (2, 1, 28, 28), False),
    ("v3_colour_labels_32", 3, (3, 3, 64, 128, 10), (2, 3, 32, 32), True),
    ("v1_tiny_30", 1, (1, 1, 8, 16), (2, 1, 30, 30), False),
)


class TraceSink:
    """Stands where a BucketedGradientAverager stands (engine.grad_sink) and records what the backward asks of it."""

    def __init__(self):
        self.events, self.flat, self.post_scale = [], None, None

    def alloc(self, numel, dev):
        self.events.append(["alloc", int(numel)])
        return torch.empty(int(numel), device=dev, dtype=torch.float32)

    def owns(self, t):
        return False

    def flush(self, engine, force=False):
        if self.events and self.events[-1] != ["flush"]:
            self.events.append(["flush"])

    def takes_over(self, grads):
        return False


def record_traces():
    """Run one training step of every model in TRACE_MODELS on the GPU with a TraceSink attached."""
    import diffusion_model_amd as pkg
    out = {}
    for name, variant, args, shape, labels in TRACE_MODELS:
        torch.manual_seed(0)
        m = pkg.make_unet_class(variant)(*args).cuda().train()
        sink = TraceSink()
        m._engine.grad_sink = sink
        x, t = torch.randn(shape, device="cuda"), torch.randint(0, 1000, (shape[0],), device="cuda")
        y = torch.randint(0, 10, (shape[0],), device="cuda") if labels else None
        pred = m(x, t, y) if variant == 3 else m(x, t)
        pred.square().mean().backward()
        torch.cuda.synchronize()
        m._engine.grad_sink = None
        out[name] = dict(events=sink.events, parameters=sum(p.numel() for p in m.parameters()))
    return out


def load_traces():
    with open(TRACE_FILE) as f:
        return json.load(f)


if __name__ == "__main__":            # on a GPU box: python tests/trace_util.py  -> writes the fixture
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    traces = record_traces()
    dst = sys.argv[1] if len(sys.argv) > 1 else TRACE_FILE
    with open(dst, "w") as f:
        json.dump(traces, f, separators=(",", ":"))
    print({k: (len(v["events"]), v["parameters"]) for k, v in traces.items()})
