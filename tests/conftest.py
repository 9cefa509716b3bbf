import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    import orc_py

    assert orc_py.lib().orc_self_test() == 0
    return orc_py


@pytest.fixture(scope="session")
def mi():
    """The product: HIP library through its Python mirror.  Fails loudly when the .so is missing."""
    from snarkos_amd import algorithms, lib

    lib.init()
    return algorithms
