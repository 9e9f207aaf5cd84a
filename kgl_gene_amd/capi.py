"""ctypes binding of include/kgx.h — the same symbols the C++ analysis packages call.

Loading fails loudly when libkgx.so is absent, and every compute call fails loudly (KgxError)
when no gfx950 device is bound: there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = ROOT / "kgl_gene_amd" / "lib" / "libkgx.so"
HEADER = ROOT / "include" / "kgx.h"

KGX_OK, KGX_EINVAL, KGX_ENODEVICE, KGX_EHIP, KGX_ENOMEM, KGX_ESTATE = 0, -1, -2, -3, -4, -5


class KgxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"kgx error {code}: {message}")
        self.code = code


_lib = None

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)

_SIGNATURES = {
    "kgx_version": (C.c_char_p, []),
    "kgx_last_error": (C.c_char_p, []),
    "kgx_device_count": (C.c_int, []),
    "kgx_init": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "kgx_bound_devices": (C.c_int, []),
    "kgx_exchange_kind": (C.c_char_p, []),
    "kgx_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), _u64p]),
    "kgx_stream": (C.c_void_p, [C.c_int]),
    "kgx_synchronize": (C.c_int, []),
    "kgx_population_create": (C.c_void_p, [C.c_uint64, C.c_uint64]),
    "kgx_population_destroy": (None, [C.c_void_p]),
    "kgx_population_shards": (C.c_uint32, [C.c_void_p]),
    "kgx_population_shard_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), _u64p, _u64p]),
    "kgx_population_genomes": (C.c_uint64, [C.c_void_p]),
    "kgx_population_variants": (C.c_uint64, [C.c_void_p]),
    "kgx_population_row_pitch": (C.c_uint64, [C.c_void_p]),
    "kgx_population_sweep_bytes": (C.c_uint64, [C.c_void_p]),
    "kgx_population_load_dosage2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kgx_population_load_dosage_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "kgx_population_read_dosage2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kgx_population_set_af": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kgx_population_set_genome_mask": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kgx_population_get_af": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kgx_population_synth_biallelic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kgx_synth_biallelic_host": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                           C.c_void_p, C.c_uint64, C.c_void_p]),
    "kgx_allele_count_by_locus": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kgx_allele_count_by_locus_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kgx_allele_frequency_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "kgx_allele_count_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "kgx_count_by_genome": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kgx_count_by_genome_binned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "kgx_count_by_genome_last_ms": (C.c_double, []),
    "kgx_population_summary": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kgx_gt8_create": (C.c_void_p, [C.c_uint64, C.c_uint64]),
    "kgx_gt8_destroy": (None, [C.c_void_p]),
    "kgx_gt8_shards": (C.c_uint32, [C.c_void_p]),
    "kgx_gt8_shard_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), _u64p, _u64p]),
    "kgx_gt8_genomes": (C.c_uint64, [C.c_void_p]),
    "kgx_gt8_loci": (C.c_uint64, [C.c_void_p]),
    "kgx_gt8_sweep_bytes": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint32]),
    "kgx_gt8_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "kgx_gt8_load_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kgx_gt8_read_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kgx_locus_class_frequencies": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p]),
    "kgx_inbreed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int,
                              C.c_int, C.c_void_p, C.c_void_p]),
    "kgx_gt8_set_wide_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kgx_inbreed_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "kgx_inbreed_reference_starts": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "kgx_release_scratch": (C.c_int, []),
    "kgx_reload_options": (C.c_int, []),
    "kgx_count_by_genome_af_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "kgx_population_resize": (C.c_int, [C.c_void_p, C.c_uint64]),
    "kgx_inbreed_last_sweep_ms": (C.c_double, []),
    "kgx_inbreed_last_kernel_ms": (C.c_double, []),
    "kgx_inbreed_last_evaluations": (C.c_int, []),
    "kgx_inbreed_last_path": (C.c_int, []),
    "kgx_inbreed_last_moments_ms": (C.c_double, []),
    "kgx_inbreed_last_search_ms": (C.c_double, []),
    "kgx_inbreed_objective": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "kgx_gt8_synth_multiallelic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "kgx_synth_multiallelic_host": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                              C.c_void_p, C.c_void_p]),
    "kgx_synth_locus_host": (C.c_int, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kgx_synth_loci_host": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kgx_gt8_synth_inbred": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "kgx_compound_offsets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "kgx_compound_offsets_listed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "kgx_genome_row_lists": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kgx_population_load_phase_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kgx_unique_phased_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "kgx_offset_filter_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                           C.c_void_p]),
}


def declared_symbols() -> list[str]:
    """Every function name include/kgx.h declares (parsed from the header itself)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kgx_[a-z0-9_]+)\s*\(", text)))


def ensure_built() -> None:
    """Compile the HIP extension in-tree if it is not there yet (hipcc must be on the box).  This only builds
    the extension; it never substitutes anything for it."""
    if not LIB_PATH.exists():
        from . import build as _build

        _build.build_kgx(verbose=True)
        _build.build_host()


def _one_hip_runtime() -> None:
    """Keep ONE HIP runtime in this process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so
    (found through an RPATH of $ORIGIN, by the file name `libamdhip64.so`), libkgx.so links the system's
    /opt/rocm/lib/libamdhip64.so.7.  Both have the SONAME libamdhip64.so.7: when torch is loaded first, libkgx's
    dependency resolves to the copy already mapped -- one runtime; when libkgx is loaded first, torch's request by file
    name matches neither the mapped library's name nor its SONAME and maps the bundled copy as well -- two HIP and two HSA
    runtimes in one process, the second of which cannot take the device the first one holds (torch then reports no GPU).
    So a Python process that uses both (bench.py, a few tests: torch is their plumbing for device buffers and RCCL) must
    load torch first; this does it whenever torch is installed and not loaded yet.  A C++ host (the reference binary with
    the GPU packages, kgx_host_driver) has no torch and runs on the system runtime alone."""
    import importlib.util
    import sys

    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def hip_runtimes_mapped() -> list[str]:
    """The libamdhip64 files mapped into this process (more than one = the state _one_hip_runtime prevents)."""
    found = set()
    with open("/proc/self/maps") as maps:
        for line in maps:
            path = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64" in path:
                found.add(path)
    return sorted(found)


# The library reads its KGX_* switches at kgx_init and at kgx_reload_options, not per call.  A process that flips them between
# calls (the tests, the comparison scripts) sets WATCH_ENV: every access to the library then first hands over the switches if they
# changed.  bench.py and production leave it off: nothing is checked per call.
WATCH_ENV = False
_switches_seen = None


def reload_options() -> None:
    """Have the library read its KGX_* switches again (kgx_reload_options)."""
    global _switches_seen
    _switches_seen = _switches_now()
    handle = _lib if _lib is not None else lib()
    check(handle.kgx_reload_options())


def _switches_now():
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("KGX_")))


def lib() -> C.CDLL:
    """Load libkgx.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        if WATCH_ENV and _switches_now() != _switches_seen:
            reload_options()
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            f"(python -m kgl_gene_amd.build, or __graft_entry__.build()). There is no CPU fallback."
        )
    _one_hip_runtime()
    handle = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != KGX_OK:
        raise KgxError(rc, lib().kgx_last_error().decode(errors="replace"))


def ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def device_count() -> int:
    return int(lib().kgx_device_count())


def init(devices=0) -> None:
    """Bind the library: `devices` = one HIP ordinal, or a list of ordinals (one genome shard per entry; [] = every
    visible device)."""
    ids = [int(devices)] if isinstance(devices, (int, np.integer)) else [int(d) for d in devices]
    if not ids:
        check(lib().kgx_init(0, None))
        return
    arr = (C.c_int * len(ids))(*ids)
    check(lib().kgx_init(len(ids), arr))


def bound_devices() -> int:
    return int(lib().kgx_bound_devices())


def exchange_kind() -> str:
    """How the shards' per-variant counts are summed: "none", "rccl" or "peer"."""
    return lib().kgx_exchange_kind().decode()


def device_info(slot: int = 0) -> dict:
    name = C.create_string_buffer(128)
    arch = C.create_string_buffer(64)
    cus = C.c_int(0)
    hbm = C.c_uint64(0)
    check(lib().kgx_device_info(int(slot), name, 128, arch, 64, C.byref(cus), C.byref(hbm)))
    return {"name": name.value.decode(), "arch": arch.value.decode(), "compute_units": cus.value,
            "hbm_bytes": hbm.value}


def stream(slot: int = 0) -> int:
    """The library's own hipStream_t of a device slot (as an integer handle)."""
    return int(lib().kgx_stream(int(slot)) or 0)


def synchronize() -> None:
    check(lib().kgx_synchronize())


def _shards(count_fn, info_fn, handle):
    out = []
    for i in range(int(count_fn(handle))):
        slot, base, n = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        check(info_fn(handle, i, C.byref(slot), C.byref(base), C.byref(n)))
        out.append({"slot": slot.value, "genome_base": base.value, "n_genomes": n.value})
    return out


def synth_biallelic_host(seed: int, genome_base: int, n_genomes: int, v0: int, v1: int):
    """Host twin of the device generator: (packed rows [v1-v0][ceil(G/4)] uint8, af float32)."""
    row_bytes = (n_genomes + 3) // 4
    rows = np.zeros((v1 - v0, row_bytes), dtype=np.uint8)
    af = np.zeros(v1 - v0, dtype=np.float32)
    check(lib().kgx_synth_biallelic_host(seed, genome_base, n_genomes, v0, v1, ptr(rows), row_bytes, ptr(af)))
    return rows, af


def unpack_dosage2(rows: np.ndarray, n_genomes: int) -> np.ndarray:
    """[V][ceil(G/4)] packed bytes -> [V][G] uint8 codes (host helper for tests and flattening)."""
    r = np.ascontiguousarray(rows, dtype=np.uint8)
    out = np.empty((r.shape[0], r.shape[1] * 4), dtype=np.uint8)
    for j in range(4):
        out[:, j::4] = (r >> (2 * j)) & 3
    return out[:, :n_genomes]


def pack_dosage2(codes: np.ndarray) -> np.ndarray:
    """[V][G] uint8 codes (0..3) -> [V][ceil(G/4)] packed bytes."""
    c = np.ascontiguousarray(codes, dtype=np.uint8)
    v, g = c.shape
    pad = (-g) % 4
    if pad:
        c = np.concatenate([c, np.zeros((v, pad), dtype=np.uint8)], axis=1)
    c = c.reshape(v, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


class Population:
    """One genome shard of a flattened PopulationDB resident in HBM (opaque kgx_pop)."""

    def __init__(self, n_genomes: int, n_variants: int):
        self._h = lib().kgx_population_create(int(n_genomes), int(n_variants))
        if not self._h:
            raise KgxError(KGX_EHIP, lib().kgx_last_error().decode(errors="replace"))
        self.n_genomes = int(n_genomes)
        self.n_variants = int(n_variants)

    def resize(self, n_variants: int) -> None:
        """Change the row count: rows held keep their content, new rows are empty; growing re-allocates by >= 1.5 x."""
        check(lib().kgx_population_resize(self._h, int(n_variants)))
        self.n_variants = int(n_variants)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().kgx_population_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def shards(self) -> list[dict]:
        """One dict per device slot: slot, genome_base, n_genomes."""
        return _shards(lib().kgx_population_shards, lib().kgx_population_shard_info, self._h)

    @property
    def row_pitch(self) -> int:
        return int(lib().kgx_population_row_pitch(self._h))

    @property
    def sweep_bytes(self) -> int:
        return int(lib().kgx_population_sweep_bytes(self._h))

    # -- loading ----------------------------------------------------------------------------
    def load_dosage2(self, rows: np.ndarray, v0: int = 0) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        check(lib().kgx_population_load_dosage2(self._h, ptr(rows), rows.shape[1], v0, v0 + rows.shape[0]))

    def load_dosage_u8(self, dosage: np.ndarray, g0: int = 0) -> None:
        """dosage: [n][n_variants] uint8, the reference's VariantDBGenomeData rows."""
        d = np.ascontiguousarray(dosage, dtype=np.uint8)
        if d.ndim != 2 or d.shape[1] != self.n_variants:
            raise ValueError("dosage must be [genomes][n_variants]")
        check(lib().kgx_population_load_dosage_u8(self._h, ptr(d), g0, g0 + d.shape[0]))

    def read_dosage2(self, v0: int = 0, v1: int | None = None) -> np.ndarray:
        v1 = self.n_variants if v1 is None else v1
        out = np.zeros((v1 - v0, (self.n_genomes + 3) // 4), dtype=np.uint8)
        check(lib().kgx_population_read_dosage2(self._h, ptr(out), out.shape[1], v0, v1))
        return out

    def set_genome_mask(self, keep: np.ndarray | None) -> None:
        """keep[g] != 0: genome g takes part in every sweep (the reference's genome-list filters); None lifts the mask."""
        if keep is None:
            check(lib().kgx_population_set_genome_mask(self._h, None))
            return
        k = np.ascontiguousarray(np.asarray(keep) != 0, dtype=np.uint8)
        if k.shape != (self.n_genomes,):
            raise ValueError("keep must be [n_genomes]")
        check(lib().kgx_population_set_genome_mask(self._h, ptr(k)))

    def set_af(self, af: np.ndarray) -> None:
        a = np.ascontiguousarray(af, dtype=np.float32)
        if a.shape != (self.n_variants,):
            raise ValueError("af must be [n_variants]")
        check(lib().kgx_population_set_af(self._h, ptr(a)))

    def get_af(self) -> np.ndarray:
        a = np.zeros(self.n_variants, dtype=np.float32)
        check(lib().kgx_population_get_af(self._h, ptr(a)))
        return a

    def synth_biallelic(self, seed: int = 1111, genome_base: int = 0, variant_base: int = 0) -> None:
        check(lib().kgx_population_synth_biallelic(self._h, seed, genome_base, variant_base))

    # -- the sweeps -------------------------------------------------------------------------
    def allele_count_by_locus(self) -> np.ndarray:
        """[n_variants][4] uint32: refHom, het, minorHom, nonDiploid (summaryByVariant)."""
        out = np.zeros((self.n_variants, 4), dtype=np.uint32)
        check(lib().kgx_allele_count_by_locus(self._h, ptr(out)))
        return out

    def allele_count_by_locus_dev(self, d_out: int, stream: int = 0) -> None:
        check(lib().kgx_allele_count_by_locus_dev(self._h, C.c_void_p(d_out), C.c_void_p(stream)))

    def allele_count_timed(self, d_out: int, stream: int, warmup: int, iters: int) -> np.ndarray:
        ms = np.zeros(iters, dtype=np.float32)
        check(lib().kgx_allele_count_timed(self._h, C.c_void_p(d_out), C.c_void_p(stream), warmup, iters, ptr(ms)))
        return ms

    def count_by_genome(self, variant_mask: np.ndarray | None = None) -> np.ndarray:
        """[n_genomes][4] uint64 (summaryByGenome), optionally over variants with mask != 0."""
        out = np.zeros((self.n_genomes, 4), dtype=np.uint64)
        if variant_mask is None:
            check(lib().kgx_count_by_genome(self._h, None, ptr(out)))
        else:
            m = np.ascontiguousarray(variant_mask, dtype=np.uint8)
            if m.shape != (self.n_variants,):
                raise ValueError("variant_mask must be [n_variants]")
            check(lib().kgx_count_by_genome(self._h, ptr(m), ptr(out)))
        return out

    def count_by_genome_binned(self, bin_of_variant: np.ndarray, n_bins: int) -> np.ndarray:
        b = np.ascontiguousarray(bin_of_variant, dtype=np.uint8)
        if b.shape != (self.n_variants,):
            raise ValueError("bin_of_variant must be [n_variants]")
        out = np.zeros((self.n_genomes, n_bins, 4), dtype=np.uint64)
        check(lib().kgx_count_by_genome_binned(self._h, ptr(b), n_bins, ptr(out)))
        return out

    def count_by_genome_af_bins(self, bin_edges) -> np.ndarray:
        """The by-genome sweep with every row's bin decided on the device from the AF column (set_af): bin b holds the rows
        with edges[b] <= af and not edges[b + 1] <= af; NaN rows are in no bin.  [n_genomes][len(edges) - 1][4] uint64."""
        e = np.ascontiguousarray(bin_edges, dtype=np.float64)
        n_bins = len(e) - 1
        out = np.zeros((self.n_genomes, n_bins, 4), dtype=np.uint64)
        check(lib().kgx_count_by_genome_af_bins(self._h, ptr(e), n_bins, ptr(out)))
        return out

    def compound_offsets(self, first_row, n_rows, bins, n_bins: int) -> np.ndarray:
        """[n_genomes][n_bins][3] uint64: het_ref_minor, hom_minor, het_minor at offsets with >= 2 distinct variants."""
        fr = np.ascontiguousarray(first_row, dtype=np.uint32)
        nr = np.ascontiguousarray(n_rows, dtype=np.uint32)
        bn = np.ascontiguousarray(bins, dtype=np.uint32)
        out = np.zeros((self.n_genomes, n_bins, 3), dtype=np.uint64)
        check(lib().kgx_compound_offsets(self._h, ptr(fr), ptr(nr), ptr(bn), len(fr), n_bins, ptr(out)))
        return out

    def compound_offsets_listed(self, member_rows, first_member, n_rows, bins, n_bins: int) -> np.ndarray:
        """compound_offsets for groups whose rows are not adjacent: group i = member_rows[first_member[i] : + n_rows[i]]."""
        mr = np.ascontiguousarray(member_rows, dtype=np.uint32)
        fm = np.ascontiguousarray(first_member, dtype=np.uint32)
        nr = np.ascontiguousarray(n_rows, dtype=np.uint32)
        bn = np.ascontiguousarray(bins, dtype=np.uint32)
        out = np.zeros((self.n_genomes, n_bins, 3), dtype=np.uint64)
        check(lib().kgx_compound_offsets_listed(self._h, ptr(mr), len(mr), ptr(fm), ptr(nr), ptr(bn), len(fm), n_bins, ptr(out)))
        return out

    def offset_filter_counts(self, single_bin, member_rows, first_member, n_rows, bins, n_bins: int) -> np.ndarray:
        """[n_genomes][n_bins][4] uint64: the Variant objects HomozygousFilter / HeterozygousFilter / DiploidFilter /
        UniqueUnphasedFilter leave each genome per bin.  single_bin[row]: the bin of a row alone at its offset (0xFF for
        the members of the listed groups and for rows not counted)."""
        sb = np.ascontiguousarray(single_bin, dtype=np.uint8)
        if sb.shape != (self.n_variants,):
            raise ValueError("single_bin must be [n_variants]")
        mr = np.ascontiguousarray(member_rows, dtype=np.uint32)
        fm = np.ascontiguousarray(first_member, dtype=np.uint32)
        nr = np.ascontiguousarray(n_rows, dtype=np.uint32)
        bn = np.ascontiguousarray(bins, dtype=np.uint32)
        out = np.zeros((self.n_genomes, n_bins, 4), dtype=np.uint64)
        check(lib().kgx_offset_filter_counts(self._h, ptr(sb), ptr(mr), len(mr), ptr(fm), ptr(nr), ptr(bn), len(fm), n_bins, ptr(out)))
        return out

    def genome_row_lists(self, g0: int = 0, g1: int | None = None, row_selected: np.ndarray | None = None):
        """(begin [g1-g0+1] uint64, rows uint32): for every genome of [g0, g1) the rows it carries, ascending."""
        g1 = self.n_genomes if g1 is None else g1
        sel = None if row_selected is None else np.ascontiguousarray(np.asarray(row_selected) != 0, dtype=np.uint8)
        if sel is not None and sel.shape != (self.n_variants,):
            raise ValueError("row_selected must be [n_variants]")
        begin = np.zeros(g1 - g0 + 1, dtype=np.uint64)
        check(lib().kgx_genome_row_lists(self._h, g0, g1, ptr(sel) if sel is not None else None, ptr(begin), None, 0))
        rows = np.zeros(int(begin[-1]), dtype=np.uint32)
        check(lib().kgx_genome_row_lists(self._h, g0, g1, ptr(sel) if sel is not None else None, ptr(begin), ptr(rows), len(rows)))
        return begin, rows

    def load_phase_plane(self, both_phases: np.ndarray, v0: int = 0) -> None:
        """both_phases: [rows][n_genomes] booleans -- the genome's copies of the row's variant sit on both phases."""
        bits = np.packbits(np.ascontiguousarray(both_phases, dtype=np.uint8), axis=1, bitorder="little")
        check(lib().kgx_population_load_phase_plane(self._h, ptr(bits), bits.shape[1], v0, v0 + bits.shape[0]))

    def unique_phased_counts(self, bin_of_variant=None, n_bins: int = 1) -> np.ndarray:
        """[n_genomes][n_bins]: the Variant objects UniquePhasedFilter leaves (one per distinct HGVS and phase)."""
        out = np.zeros((self.n_genomes, n_bins), dtype=np.uint64)
        b = None if bin_of_variant is None else np.ascontiguousarray(bin_of_variant, dtype=np.uint8)
        check(lib().kgx_unique_phased_counts(self._h, None if b is None else ptr(b), n_bins, ptr(out)))
        return out

    def population_summary(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint64)
        check(lib().kgx_population_summary(self._h, ptr(out)))
        return out


def count_by_genome_last_ms() -> float:
    """Device time of the K3 kernel of the most recent by-genome sweep (HIP events)."""
    return float(lib().kgx_count_by_genome_last_ms())


def allele_frequency_dev(d_counts: int, n_variants: int, total_genomes: int, d_af: int, stream: int = 0) -> None:
    check(lib().kgx_allele_frequency_dev(C.c_void_p(d_counts), n_variants, total_genomes, C.c_void_p(d_af),
                                         C.c_void_p(stream)))


ALGORITHMS = {"RitlandLocus": 0, "Simple": 1, "HallME": 2, "Loglikelihood": 3}

# numpy view of kgx_locus_results (field order of LocusResults, kga_analysis_inbreed_output.h:21-35)
LOCUS_RESULTS_DTYPE = np.dtype([
    ("major_hetero_count", np.uint64), ("major_hetero_freq", np.float64),
    ("minor_hetero_count", np.uint64), ("minor_hetero_freq", np.float64),
    ("minor_homo_count", np.uint64), ("minor_homo_freq", np.float64),
    ("major_homo_count", np.uint64), ("major_homo_freq", np.float64),
    ("total_allele_count", np.uint64), ("inbred_allele_sum", np.float64)])


def locus_class_frequencies(minor_af: np.ndarray, inbreeding: float = 0.0):
    """K6 alone: ([n][5] = p_major, majorHom, majorHet, minorHom, minorHet; valid[n])."""
    a = np.ascontiguousarray(minor_af, dtype=np.float64)
    n, amax = a.shape
    out = np.zeros((n, 5), dtype=np.float64)
    valid = np.zeros(n, dtype=np.uint8)
    check(lib().kgx_locus_class_frequencies(ptr(a), n, amax, float(inbreeding), ptr(out), ptr(valid)))
    return out, valid.astype(bool)


class GenotypeMatrix:
    """Locus-major allele-index bytes resident in HBM (opaque kgx_gt8)."""

    def __init__(self, n_genomes: int, n_loci: int):
        self._h = lib().kgx_gt8_create(int(n_genomes), int(n_loci))
        if not self._h:
            raise KgxError(KGX_EHIP, lib().kgx_last_error().decode(errors="replace"))
        self.n_genomes, self.n_loci = int(n_genomes), int(n_loci)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().kgx_gt8_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shards(self) -> list[dict]:
        return _shards(lib().kgx_gt8_shards, lib().kgx_gt8_shard_info, self._h)

    def load_rows(self, rows: np.ndarray, l0: int = 0) -> None:
        r = np.ascontiguousarray(rows, dtype=np.uint8)
        check(lib().kgx_gt8_load_rows(self._h, ptr(r), r.shape[1], l0, l0 + r.shape[0]))

    def load_genomes(self, by_genome: np.ndarray, g0: int = 0) -> None:
        r = np.ascontiguousarray(by_genome, dtype=np.uint8)
        check(lib().kgx_gt8_load(self._h, ptr(r), g0, g0 + r.shape[0]))

    def read_rows(self, l0: int = 0, l1: int | None = None) -> np.ndarray:
        l1 = self.n_loci if l1 is None else l1
        out = np.zeros((l1 - l0, self.n_genomes), dtype=np.uint8)
        check(lib().kgx_gt8_read_rows(self._h, ptr(out), self.n_genomes, l0, l1))
        return out

    def synth_multiallelic(self, seed: int = 1111, genome_base: int = 0, locus_base: int = 0) -> np.ndarray:
        """Fill with the synthetic multi-allelic population; returns the SNP AF table [n_loci][3] (NaN padded)."""
        table = np.zeros((self.n_loci, 3), dtype=np.float64)
        check(lib().kgx_gt8_synth_multiallelic(self._h, seed, genome_base, locus_base, ptr(table)))
        return table

    def synth_inbred(self, minor_af: np.ndarray, inbreeding: np.ndarray, seed: int = 1111) -> None:
        """Fill with the reference's synthetic-inbreeding self-check genomes (genome g has F = inbreeding[g])."""
        a = np.ascontiguousarray(minor_af, dtype=np.float64)
        f = np.ascontiguousarray(inbreeding, dtype=np.float64)
        if a.shape[0] != self.n_loci or f.shape != (self.n_genomes,):
            raise ValueError("minor_af must have one row per locus and inbreeding one value per genome")
        check(lib().kgx_gt8_synth_inbred(self._h, ptr(a), a.shape[1], ptr(f), int(seed)))

    def inbreed(self, minor_af: np.ndarray, algorithm: str, phased: bool, locus_index=None, g0: int = 0, g1: int | None = None,
                start=None):
        """start: per-genome start points of HallME / Loglikelihood (reference_starts()), None = the interval midpoints."""
        g1 = self.n_genomes if g1 is None else g1
        st = None if start is None else np.ascontiguousarray(start, dtype=np.float64)
        if st is not None and st.shape != (g1 - g0,):
            raise ValueError("start must hold one value per genome of the range")
        a = np.ascontiguousarray(minor_af, dtype=np.float64)
        n_sel, amax = a.shape if a.ndim == 2 else (0, 1)
        idx = None if locus_index is None else np.ascontiguousarray(locus_index, dtype=np.uint32)
        out = np.zeros(g1 - g0, dtype=LOCUS_RESULTS_DTYPE)
        check(lib().kgx_inbreed(self._h, g0, g1, None if idx is None else ptr(idx), n_sel, ptr(a), amax, int(bool(phased)),
                                ALGORITHMS[algorithm], None if st is None else ptr(st), ptr(out)))
        return out

    def set_wide_rows(self, locus, cells) -> None:
        """Offsets with more than 14 reference alts (kgx_gt8_set_wide_rows): locus [n_wide] rows ascending, cells uint16
        [n_wide][n_genomes] = a1 | a2 << 8.  None / empty drops them."""
        if locus is None or len(locus) == 0:
            check(lib().kgx_gt8_set_wide_rows(self._h, 0, None, None, 0))
            return
        rows = np.ascontiguousarray(locus, dtype=np.uint32)
        c = np.ascontiguousarray(cells, dtype=np.uint16)
        if c.shape != (len(rows), self.n_genomes):
            raise ValueError("cells must be [n_wide][n_genomes]")
        check(lib().kgx_gt8_set_wide_rows(self._h, len(rows), ptr(rows), ptr(c), self.n_genomes))

    def inbreed_batch(self, tasks, algorithm: str, phased: bool) -> list:
        """Many inbreed() calls in one (kgx_inbreed_batch): tasks = dicts with minor_af [n_selected][amax] and optionally
        locus_index, g0, g1, start -- all with the same amax.  Returns the tasks' result arrays in order."""
        keep, outs = [], []
        array = (InbreedTask * len(tasks))()
        amax = None
        for i, t in enumerate(tasks):
            g0 = int(t.get("g0", 0))
            g1 = int(self.n_genomes if t.get("g1") is None else t["g1"])
            a = np.ascontiguousarray(t["minor_af"], dtype=np.float64)
            if a.ndim != 2 or (amax is not None and a.shape[1] != amax):
                raise ValueError("every task's minor_af must be [n_selected][amax] with one amax")
            amax = a.shape[1]
            idx = None if t.get("locus_index") is None else np.ascontiguousarray(t["locus_index"], dtype=np.uint32)
            st = None if t.get("start") is None else np.ascontiguousarray(t["start"], dtype=np.float64)
            if st is not None and st.shape != (g1 - g0,):
                raise ValueError("start must hold one value per genome of the range")
            out = np.zeros(g1 - g0, dtype=LOCUS_RESULTS_DTYPE)
            keep += [a, idx, st]
            outs.append(out)
            array[i] = InbreedTask(g0, g1, None if idx is None else idx.ctypes.data, a.shape[0], a.ctypes.data, None if st is None else st.ctypes.data,
                                   out.ctypes.data)
        check(lib().kgx_inbreed_batch(self._h, C.cast(array, C.c_void_p), len(tasks), int(amax or 1), int(bool(phased)), ALGORITHMS[algorithm]))
        return outs

    def inbreed_objective(self, minor_af: np.ndarray, at, phased: bool, locus_index=None, g0: int = 0, g1: int | None = None,
                          by_passes: bool = False) -> np.ndarray:
        """The log-likelihood of each genome of the range at its point at[g] (kgx_inbreed_objective: a diagnostic) -- from the
        moments a large Loglikelihood call runs on, or (by_passes) from one table pass over the genotype bytes."""
        g1 = self.n_genomes if g1 is None else g1
        points = np.ascontiguousarray(at, dtype=np.float64)
        if points.shape != (g1 - g0,):
            raise ValueError("at must hold one value per genome of the range")
        a = np.ascontiguousarray(minor_af, dtype=np.float64)
        n_sel, amax = a.shape
        idx = None if locus_index is None else np.ascontiguousarray(locus_index, dtype=np.uint32)
        out = np.zeros(g1 - g0, dtype=np.float64)
        check(lib().kgx_inbreed_objective(self._h, g0, g1, None if idx is None else ptr(idx), n_sel, ptr(a), amax, int(bool(phased)),
                                          ptr(points), int(bool(by_passes)), ptr(out)))
        return out

    def inbreed_resident(self, minor_af_dev: int, n_selected: int, amax: int, algorithm: str, phased: bool, g0: int = 0, g1: int | None = None,
                         start=None):
        """inbreed() with the allele-frequency table already on this device: minor_af_dev is the device address of
        float64 [n_selected][amax] (e.g. a torch tensor's data_ptr())."""
        g1 = self.n_genomes if g1 is None else g1
        out = np.zeros(g1 - g0, dtype=LOCUS_RESULTS_DTYPE)
        st = None if start is None else np.ascontiguousarray(start, dtype=np.float64)
        check(lib().kgx_inbreed(self._h, g0, g1, None, n_selected, C.c_void_p(minor_af_dev), amax, int(bool(phased)), ALGORITHMS[algorithm],
                                None if st is None else ptr(st), ptr(out)))
        return out


def reference_starts(algorithm: str, seed: int, n: int, first_stream: int = 0) -> np.ndarray:
    """The start points the reference's HallME / Loglikelihood end up using for n genomes: genome i draws from
    std::mt19937_64(seed + first_stream + i) (seed 0: std::random_device) and its fifth draw counts (kgx.h)."""
    out = np.zeros(n, dtype=np.float64)
    check(lib().kgx_inbreed_reference_starts(ALGORITHMS[algorithm], int(seed), int(first_stream), int(n), ptr(out)))
    return out


def release_scratch() -> None:
    """Free the arena kgx_inbreed keeps its per-call device buffers in."""
    check(lib().kgx_release_scratch())


def inbreed_last_kernel_ms() -> float:
    """HIP-event time of the kernel of the last inbreed call's frequency sweep that reads the genotype bytes."""
    return float(lib().kgx_inbreed_last_kernel_ms())


def inbreed_last_sweep_ms() -> float:
    """Device time of the frequency sweep of the most recent GenotypeMatrix.inbreed call (HIP events)."""
    return float(lib().kgx_inbreed_last_sweep_ms())


class InbreedTask(C.Structure):
    """kgx_inbreed_task (kgx.h)"""
    _fields_ = [("g0", C.c_uint64), ("g1", C.c_uint64), ("locus_index", C.c_void_p), ("n_selected", C.c_uint64), ("minor_af", C.c_void_p),
                ("start", C.c_void_p), ("out", C.c_void_p)]


PATHS = {0: "none", 1: "frequency sweep", 2: "one launch", 3: "hall moments", 4: "hall passes", 5: "loglik moments",
         6: "loglik moments + passes", 7: "loglik passes"}


def inbreed_last_path() -> str:
    """What the most recent inbreed call ran on (kgx.h: KGX_PATH_*)."""
    return PATHS[int(lib().kgx_inbreed_last_path())]


def inbreed_last_moments_ms() -> float:
    """Device time of the class passes (sweeps + merges) of the last inbreed call that ran on moments; 0 otherwise."""
    return float(lib().kgx_inbreed_last_moments_ms())


def inbreed_last_search_ms() -> float:
    """Device time of the kernel that iterated / searched on the moments in the last such inbreed call; 0 otherwise."""
    return float(lib().kgx_inbreed_last_search_ms())


def inbreed_last_evaluations() -> int:
    """Objective evaluations the most recent Loglikelihood call needed."""
    return int(lib().kgx_inbreed_last_evaluations())


def synth_loci_host(seed: int, l0: int, l1: int):
    """The synthetic multi-allelic loci [l0,l1): (n_alt uint8 [n], af float32 [n][3], is_indel uint8 [n][3])."""
    n = l1 - l0
    n_alt = np.zeros(n, dtype=np.uint8)
    af = np.zeros((n, 3), dtype=np.float32)
    indel = np.zeros((n, 3), dtype=np.uint8)
    check(lib().kgx_synth_loci_host(seed, l0, l1, ptr(n_alt), ptr(af), ptr(indel)))
    return n_alt, af, indel


def synth_multiallelic_host(seed: int, genome_base: int, n_genomes: int, l0: int, l1: int):
    """Host twin: (gt8 bytes [l1-l0][G], SNP AF table [l1-l0][3], raw allele pairs [l1-l0][G][2])."""
    gt8 = np.zeros((l1 - l0, n_genomes), dtype=np.uint8)
    table = np.zeros((l1 - l0, 3), dtype=np.float64)
    alleles = np.zeros((l1 - l0, n_genomes, 2), dtype=np.uint8)
    check(lib().kgx_synth_multiallelic_host(seed, genome_base, n_genomes, l0, l1, ptr(gt8), n_genomes, ptr(table), ptr(alleles)))
    return gt8, table, alleles
