import os
import sys

import pytest

# torch first: the torch wheel bundles its own ROCm runtime (libamdhip64 etc.); a process that loads libmcmchip.so (linked
# against /opt/rocm) BEFORE torch ends up with two HIP runtimes and torch then finds no GPU.  Loaded in this order the
# dynamic linker binds both to the one copy (bench.py does the same).  The tests use torch only for device buffers.
try:
    import torch  # noqa: F401
except Exception:      # the CPU tests do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
