import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kgx():
    """The ctypes binding with device 0 bound.  Fails (not skips) if the HIP library is missing."""
    from kgl_gene_amd import capi

    capi.ensure_built()
    # torch (the plumbing a few tests and bench.py use for device buffers) sets its HIP context up first, once: after
    # tens of GB of library allocations and several device rebinds its lazy initialisation has been seen to find no device
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    capi.lib()
    if capi.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    capi.init(0)
    return capi
