import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The oracle is test infrastructure and is built on demand; the HIP library
    must already exist (it is built by __graft_entry__.build() and travels to the
    GPU box) -- if it is missing, build it here where hipcc is available."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = os.path.join(ROOT, "dddmr_navigation_amd", "csrc", "libdddmr_rollout.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(lib)])
    yield


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
