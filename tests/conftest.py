import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C oracle is test infrastructure: build it on demand (gcc only, < 1 s)."""
    so = os.path.join(ROOT, "oracle", "_build", "libwalk_oracle.so")
    src = os.path.join(ROOT, "oracle", "walk_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


# Under `-x` a failure hides everything behind it: run the contract's core first -- bit-exact walks, trees and error paths
# (test_gpu_walk), then the float kernels (steps), the bench-size workloads (scale), the end-to-end schedules (e2e) --
# and everything else in its usual order behind them.
_GPU_ORDER = ["test_gpu_walk", "test_gpu_steps", "test_gpu_scale", "test_gpu_e2e"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else len(_GPU_ORDER)
    items.sort(key=rank)  # stable: the order inside a file is kept
