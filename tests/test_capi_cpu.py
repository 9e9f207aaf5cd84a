"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/kgx.h declares, and refuses to compute without a GPU (no CPU fallback).  No compute calls."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from kgl_gene_amd import capi
from kgl_gene_amd.fws import FWS_BINS, NO_BIN, fws_bin_of_variant


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    declared = capi.declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/kgx.h but not exported by libkgx.so"
    # and the binding table covers the header exactly
    assert sorted(capi._SIGNATURES) == declared


def test_exports_are_plain_c_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", str(capi.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in capi.declared_symbols():
        assert name in exported


def test_version_and_error_strings():
    assert b"gfx950" in capi.lib().kgx_version()
    assert isinstance(capi.lib().kgx_last_error(), bytes)


def test_no_cpu_fallback_without_device():
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible here; the no-device contract is checked in the CPU container")
    with pytest.raises(capi.KgxError) as e:
        capi.init(0)
    assert e.value.code == capi.KGX_ENODEVICE
    # every compute entry point refuses, loudly
    assert not capi.lib().kgx_population_create(8, 8)
    assert b"no CPU fallback" in capi.lib().kgx_last_error()
    out = np.zeros(4, dtype=np.uint64)
    assert capi.lib().kgx_population_summary(None, capi.ptr(out)) == capi.KGX_ENODEVICE
    assert capi.lib().kgx_allele_count_by_locus(None, capi.ptr(out)) == capi.KGX_ENODEVICE
    assert capi.lib().kgx_count_by_genome(None, None, capi.ptr(out)) == capi.KGX_ENODEVICE


def test_host_generator_is_deterministic_and_sharded_consistently():
    rows, af = capi.synth_biallelic_host(1111, 0, 50, 0, 40)
    rows2, af2 = capi.synth_biallelic_host(1111, 0, 50, 0, 40)
    assert np.array_equal(rows, rows2) and np.array_equal(af, af2)
    assert af.dtype == np.float32 and af.min() >= 0.01 and af.max() <= 0.5
    # a shard starting at genome 13 sees the same genotypes as genomes 13.. of the whole
    codes = capi.unpack_dosage2(rows, 50)
    shard, _ = capi.synth_biallelic_host(1111, 13, 37, 0, 40)
    assert np.array_equal(capi.unpack_dosage2(shard, 37), codes[:, 13:])
    # a variant sub-range is the same rows
    sub, afs = capi.synth_biallelic_host(1111, 0, 50, 10, 20)
    assert np.array_equal(sub, rows[10:20]) and np.array_equal(afs, af[10:20])
    assert set(np.unique(codes)) <= {0, 1, 2}
    other, _ = capi.synth_biallelic_host(1112, 0, 50, 0, 40)
    assert not np.array_equal(other, rows)


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(1)
    for G in (1, 3, 4, 5, 64, 101):
        codes = rng.integers(0, 4, (7, G)).astype(np.uint8)
        assert np.array_equal(capi.unpack_dosage2(capi.pack_dosage2(codes), G), codes)


def test_fws_bin_edges():
    af = np.array([0.0, 0.049999, 0.05, 0.1, 0.4999, 0.5, 0.99, 1.0, np.nan, -0.1, 1.5], dtype=np.float32)
    bins = fws_bin_of_variant(af)
    assert bins.tolist() == [0, 0, 1, 2, 9, 10, 10, NO_BIN, NO_BIN, NO_BIN, NO_BIN]
    assert len(FWS_BINS) == 11
    assert fws_bin_of_variant(af, carried=np.zeros(len(af), bool)).tolist() == [NO_BIN] * len(af)
    # float32(0.1) = 0.100000001490116 >= 0.1 -> bin 2, exactly as the reference's widened compare
    assert np.float64(np.float32(0.1)) >= 0.1


def test_one_hip_runtime_whatever_the_import_order():
    """The cause behind an old workaround in conftest.py ("torch initialised first"): the torch wheel bundles its own
    libamdhip64 / libhsa-runtime64, libkgx.so links the system's.  Loaded torch-first the loader reuses the mapped copy
    (same SONAME); loaded libkgx-first, torch maps its bundled copy as well and the process holds two HIP runtimes, the
    second of which finds no device.  capi.lib() therefore loads torch first.  Checked in fresh interpreters."""
    import sys
    import textwrap

    probe = textwrap.dedent("""
        import ctypes, sys
        sys.path.insert(0, {root!r})
        from kgl_gene_amd import capi
        {body}
        print(len(capi.hip_runtimes_mapped()), *capi.hip_runtimes_mapped())
    """)
    root = str(capi.ROOT) if hasattr(capi, "ROOT") else str(capi.LIB_PATH.parent.parent.parent)
    cases = {
        "capi then torch": "capi.lib(); import torch",
        "torch then capi": "import torch; capi.lib()",
        # what the old order did: libkgx.so by itself first, torch afterwards -> two runtimes
        "raw libkgx then torch": "ctypes.CDLL(str(capi.LIB_PATH)); import torch",
    }
    counts = {}
    for name, body in cases.items():
        out = subprocess.run([sys.executable, "-c", probe.format(root=root, body=body)], capture_output=True, text=True, check=True).stdout
        counts[name] = int(out.split()[0])
    assert counts["capi then torch"] == 1 and counts["torch then capi"] == 1, counts
    assert counts["raw libkgx then torch"] == 2, counts          # the hazard is real on this image: the guard is what removes it
