import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (TEST INFRASTRUCTURE): built on demand with gcc, loaded through the same
    ctypes binding as the product, symbol prefix oracle_."""
    from tests.oracle_harness import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def product_lib():
    """The HIP library through its C ABI; built on demand with hipcc (cross-compiles without a GPU)."""
    import torch  # noqa: F401  (runtime load order, see ffi.load_product)
    from adaptive_sph_amd import build, ffi
    build.build_hip()
    return ffi.load_product()


@pytest.fixture(scope="session")
def lab_lib():
    """The LABORATORY build of the product's sources (-DSPH_LAB: the ablation switches compiled in, adaptive_sph_amd/build.py) -- for the
    tests that put an alternative form of a kernel or a queueing policy beside the product's default.  Its defaults are the product's
    code paths (tests/test_gpu_chain.py::test_the_laboratory_build_runs_the_products_defaults)."""
    import torch  # noqa: F401
    from adaptive_sph_amd import build, ffi
    return ffi.SphLibrary(build.build_lab(), "sph_", global_symbols=False)


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()
