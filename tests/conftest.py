import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def crt():
    # PyTorch's wheel carries a HIP runtime of its own beside the system's one that libcrt_hip.so is linked against.
    # Both work in one process when torch's is loaded first; loaded second it finds no GPU ("No HIP GPUs are
    # available").  Tests that hand torch tensors to the library import torch inside the test, so do it here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import cudaraytracer_amd
    return cudaraytracer_amd


def scene_path(name):
    return os.path.join(ROOT, "scenes", name)
