import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracer-rs_amd"))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    import rtamd
    ctx = rtamd.Context(0)
    # test frames are small: keep the one-launch-per-bounce kernels in play for most of each path's life and
    # leave only the last few thousand paths to the tail kernel (the default threshold would hand it everything)
    ctx.set_tail_paths(4096)
    yield ctx
    ctx.close()
