import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive checks, still CPU-only")


@pytest.fixture(scope="session")
def oracle_lib():
    """The plain-C CPU restatement (oracle/libptoracle.so), compiled on demand."""
    import oracle
    if not os.path.exists(os.path.join(oracle.HERE, "libptoracle.so")):
        oracle.build()
    return oracle.Checker("oracle")


@pytest.fixture(scope="session")
def ref_lib():
    """The compiled, unmodified reference; only present where /root/reference is (the build container) or prebuilt."""
    import oracle
    path = os.path.join(oracle.HERE, "_ref", "libptref.so")
    if not os.path.exists(path):
        if os.path.exists("/root/reference/src/worker.cpp"):
            oracle.build()
        else:
            pytest.skip("oracle/_ref/libptref.so not built (no /root/reference here)")
    return oracle.Checker("ref")
