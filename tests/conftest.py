import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


_SPAWNER = {"proc": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the process that will start the ranks of multi-process GPU tests: launched now, before anything in this pytest
    # process can have initialised the GPU (tests/spawner.py)
    import subprocess
    _SPAWNER["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "spawner.py")], stdin=subprocess.PIPE,
                                        stdout=subprocess.PIPE, text=True)


def pytest_unconfigure(config):
    p = _SPAWNER["proc"]
    if p is not None:
        try:
            p.stdin.close()
            p.wait(timeout=10)
        except Exception:  # noqa: BLE001
            p.kill()


@pytest.fixture(scope="session")
def spawn():
    """spawn(cmds, env=None, timeout=300) -> (return codes, outputs): the commands run together as children of the
    helper process started at pytest_configure"""
    import json

    def run(cmds, env=None, timeout=300):
        p = _SPAWNER["proc"]
        p.stdin.write(json.dumps({"cmds": cmds, "env": env or {}, "timeout": timeout}) + "\n")
        p.stdin.flush()
        rep = json.loads(p.stdout.readline())
        return rep["rc"], rep["out"]

    return run


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
