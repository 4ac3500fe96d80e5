import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """the CPU oracle (checker) is a small C library: build it on first use"""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return so


@pytest.fixture(scope="session")
def ctx():
    import ballista_amd as ba
    return ba.Context(0)
