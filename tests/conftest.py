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


@pytest.fixture(scope="session")
def oracle_lib():
    """CPU oracle (test infrastructure): built on demand with gcc."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def soslam():
    """The product library; on a GPU box it must already be built in-tree (it travels with the snapshot)."""
    from stereo_orb_slam_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib
