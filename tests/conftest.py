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
    capi.WATCH_ENV = True        # the tests flip KGX_* switches between calls: hand them to the library when they changed (kgx_reload_options)
    # capi.lib() loads torch (plumbing of a few tests) BEFORE libkgx.so, so that the process holds one HIP runtime: with
    # libkgx.so first the torch wheel's bundled libamdhip64 is mapped beside the system's and torch finds no device
    # (capi._one_hip_runtime; tests/test_capi_cpu.py::test_one_hip_runtime_whatever_the_import_order).
    capi.lib()
    assert len(capi.hip_runtimes_mapped()) == 1, capi.hip_runtimes_mapped()
    if capi.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    capi.init(0)
    return capi
