"""Shared fixtures.  GPU tests (`-m gpu`) call the HIP path through the C ABI;
everything else runs on the CPU (oracle vs golden vectors, host logic, symbol
export checks)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "advanced-rag-milvus_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def gpu():
    """Fails (not skips) a gpu-marked test that finds no device: a silent CPU
    fallback would void every parity claim."""
    if not _gpu_present():
        pytest.fail("gpu-marked test collected on a box without /dev/kfd")
    return 0
