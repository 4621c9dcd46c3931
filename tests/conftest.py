import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (tests may use it as the checker)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def qc():
    """the product package; builds libqcx.so first if this is a fresh checkout (hipcc cross-compiles on CPU)"""
    import subprocess
    import quantumcomputer_amd
    if not os.path.exists(quantumcomputer_amd.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "quantumcomputer_amd", "csrc"), "-s"], check=True)
    quantumcomputer_amd.lib()
    return quantumcomputer_amd
