import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own kernels; only where oracle/_ref was built (needs /root/reference at build time)."""
    from oracle_binding import Ref, REF_LIB
    if not REF_LIB.exists():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return Ref()


@pytest.fixture(scope="session")
def golden():
    return np.load(REPO / "tests" / "golden" / "ref_kernels.npz")


@pytest.fixture(scope="session")
def cpm():
    import cpm_amd
    return cpm_amd


@pytest.fixture(scope="session")
def ctx(cpm):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests never fall back to the CPU)")
    c = cpm.binding.Context(0)
    yield c
    c.close()
