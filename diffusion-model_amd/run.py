"""Launcher that runs one of the reference's UNCHANGED entry scripts on the MI355X drop-in modules.

    cd /path/to/reference/first-iteration
    PYTHONPATH=/path/to/repo python -m diffusion_model_amd.run first-iteration main.py [script args]

The reference's scripts import `unet`, `diffusion`, `beta_scheduler` and `embeddings` by bare name
(first-iteration/main.py:6-9, generate_samples.py:4-16), and Python puts the SCRIPT's directory at
sys.path[0], ahead of PYTHONPATH: started directly, a script always finds the reference's own
modules next to itself. This launcher therefore (1) puts the drop-in directory of the chosen
iteration in front of the script's directory on sys.path, (2) imports the four modules from there
and checks where they came from, so that the script's own imports resolve to them whatever the
path order, and (3) runs the script as __main__ with its own directory next on sys.path (for
`utils`, `classifier`, ...) and the working directory untouched (the scripts use relative paths
for data and checkpoints). Nothing touches the GPU before the script itself does.
"""
import importlib
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ITERATIONS = {"1": "first-iteration", "2": "second-iteration", "3": "third-iteration", "first": "first-iteration",
              "second": "second-iteration", "third": "third-iteration"}
MODULES = ("beta_scheduler", "embeddings", "diffusion", "unet")
OPTIONAL = ("classifier",)      # third-iteration only: the scorer's SimpleCNN (third-iteration/metrics2.py:8)


def drop_in_dir(iteration):
    name = ITERATIONS.get(str(iteration), str(iteration))
    path = os.path.join(HERE, "drop_in", name)
    if not os.path.isdir(path):
        raise SystemExit(f"unknown iteration {iteration!r}: expected one of first-iteration, second-iteration, third-iteration "
                         f"(or 1, 2, 3)")
    return path


def install(iteration, script_dir=None):
    """Put the drop-in modules of `iteration` in charge of the four bare module names. Returns them."""
    d = drop_in_dir(iteration)
    if script_dir is not None:
        sys.path.insert(0, script_dir)
    sys.path.insert(0, d)
    loaded = {}
    for name in MODULES + tuple(m for m in OPTIONAL if os.path.exists(os.path.join(d, m + ".py"))):
        stale = sys.modules.get(name)
        if stale is not None and os.path.dirname(os.path.abspath(getattr(stale, "__file__", "") or "")) != d:
            del sys.modules[name]                       # an earlier import of the reference's own module
        mod = importlib.import_module(name)
        origin = os.path.dirname(os.path.abspath(mod.__file__))
        if origin != d:
            raise ImportError(f"`{name}` resolved to {mod.__file__}, not to the drop-in directory {d}")
        loaded[name] = mod
    return loaded


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2 or argv[0] in ("-h", "--help"):
        raise SystemExit(__doc__)
    iteration, script, rest = argv[0], argv[1], argv[2:]
    if not os.path.isfile(script):
        raise SystemExit(f"{script}: no such script (paths are relative to the working directory, {os.getcwd()})")
    script = os.path.abspath(script)
    install(iteration, os.path.dirname(script))
    sys.argv = [script] + rest
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
