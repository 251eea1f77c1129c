import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def port(oracle_mod):
    return oracle_mod.Backend("port")


@pytest.fixture(scope="session")
def golden_c():
    with open(os.path.join(GOLDEN, "c_level_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_sswpy():
    with open(os.path.join(GOLDEN, "sswpy_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_checksums():
    with open(os.path.join(GOLDEN, "dataset_checksums.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library, built in-tree (hipcc cross-compiles without a GPU)."""
    from indelpost_amd import _lib
    _lib.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def emu():
    """Lock-step CPU emulation of the product kernels (tests/emu) -- test infrastructure."""
    from tests.emu_backend import EmuAligner, build_emu
    build_emu()
    return EmuAligner


@pytest.fixture(scope="session")
def gpu():
    from indelpost_amd import GpuAligner
    g = GpuAligner(0)
    yield g
    g.close()


LET = {c: i for i, c in enumerate("ACGTN")}


def codes(s):
    return np.array([LET[c] for c in s], np.int8)
