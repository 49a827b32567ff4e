import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "simdjson-go_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so) while libsjhip.so links the
    # system one (/opt/rocm).  Both can live in one process, but torch must initialise first,
    # otherwise its device enumeration fails.  Tests only use torch for device buffers.
    markexpr = getattr(config.option, "markexpr", "") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
