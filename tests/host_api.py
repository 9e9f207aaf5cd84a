"""ctypes window onto the host-side flatteners in kgl_gene_amd/lib/libkgx_analysis.so (pure host code)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
import os as _os

LIB = Path(_os.environ.get("KGX_SANITIZED_HOST_LIB") or ROOT / "kgl_gene_amd" / "lib" / "libkgx_analysis.so")   # see tests/tools/sanitize_host.sh
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            from kgl_gene_amd import build as kbuild

            kbuild.build_kgx()
            kbuild.build_host()
        L = C.CDLL(str(LIB))
        L.kgxh_flatten_vcf1000.restype = C.c_void_p
        L.kgxh_flatten_vcf1000.argtypes = [C.c_char_p, C.c_uint64, C.c_int]
        L.kgxh_flatten_vcf_pf.restype = C.c_void_p
        L.kgxh_flatten_vcf_pf.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int]
        L.kgxh_flatten_vcf_file.restype = C.c_void_p
        L.kgxh_flatten_vcf_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t]
        L.kgxh_flatten_vcf_file_streaming.restype = C.c_void_p
        L.kgxh_flatten_vcf_file_streaming.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, C.c_void_p]
        L.kgxh_flat_destroy.argtypes = [C.c_void_p]
        L.kgxh_flat_copy_splits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("kgxh_flat_genomes", "kgxh_flat_variants", "kgxh_flat_row_bytes", "kgxh_flat_variant_objects", "kgxh_flat_non_diploid",
                     "kgxh_flat_split_rows"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_void_p]
        L.kgxh_flat_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kgxh_flat_hgvs.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.kgxh_flat_genome_id.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.kgxh_read_vcf_text.restype = C.c_void_p
        L.kgxh_read_vcf_text.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        L.kgxh_free.argtypes = [C.c_void_p]
        L.kgxh_pfemp_location_write.restype = C.c_int
        L.kgxh_pfemp_location_write.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_char_p, C.c_char_p]
        L.kgxh_inbreed_inputs.restype = C.c_void_p
        L.kgxh_inbreed_inputs.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_uint64, C.c_int]
        L.kgxh_inbreed_inputs_file.restype = C.c_void_p
        L.kgxh_inbreed_inputs_file.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_int, C.c_uint64]
        L.kgxh_inbreed_inputs_file_streaming.restype = C.c_void_p
        L.kgxh_inbreed_inputs_file_streaming.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_char_p, C.c_size_t]
        L.kgxh_inbreed_inputs_destroy.argtypes = [C.c_void_p]
        for name in ("kgxh_inbreed_loci", "kgxh_inbreed_genomes", "kgxh_inbreed_max_alts", "kgxh_inbreed_contigs"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_void_p]
        L.kgxh_inbreed_error.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.kgxh_inbreed_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.kgxh_inbreed_genome_id.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.kgxh_variant_sort_file.restype = C.c_void_p
        L.kgxh_variant_sort_file.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint64]
        L.kgxh_variant_sort.restype = C.c_void_p
        L.kgxh_variant_sort.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        _lib = L
    return _lib


def variant_sort(text: str | None, flavour: str, what: str, names=None, genome_id: str = "Reference", threads: int = 0, path=None,
                 chunk_bytes: int = 0) -> list[tuple[str, ...]]:
    """The rsid / Ensembl indexes of kgx_variant_sort.h over VCF text (or a file read chunk_bytes at a time), as rows of strings
    in index order."""
    listed = None if names is None else "\n".join(names).encode()
    code = {"MonoGenome": 0, "Genome1000": 1}[flavour]
    if path is not None:
        ptr = lib().kgxh_variant_sort_file(str(path).encode(), code, genome_id.encode(), what.encode(), listed, threads, chunk_bytes)
        if not ptr:
            raise IOError(str(path))
    else:
        b = text.encode()
        ptr = lib().kgxh_variant_sort(b, len(b), code, genome_id.encode(), what.encode(), listed, threads)
    assert ptr, what
    try:
        out = C.string_at(ptr).decode()
    finally:
        lib().kgxh_free(ptr)
    return [tuple(line.split("\t")) for line in out.split("\n") if line]


def pfemp_location_write(sample_file, fws_file, records, statistics_csv, location_csv, radius_km: float = 0.0) -> int:
    """GpuHeteroHomoZygous' two location files from counters handed in (no device).
    records: [(genome, contig, (total, snp, indel, hom_minor, het_minor, het_ref_minor, hom_ref)), ...]"""
    import numpy as np

    n = len(records)
    genomes = (C.c_char_p * max(n, 1))(*[r[0].encode() for r in records])
    contigs = (C.c_char_p * max(n, 1))(*[r[1].encode() for r in records])
    counters = np.ascontiguousarray([r[2] for r in records], dtype=np.uint64).reshape(n, 7)
    return int(lib().kgxh_pfemp_location_write(str(sample_file).encode(), str(fws_file).encode(), n, C.cast(genomes, C.c_void_p),
                                               C.cast(contigs, C.c_void_p), C.c_void_p(counters.ctypes.data), float(radius_km),
                                               str(statistics_csv).encode(), str(location_csv).encode()))


class TwoPhaseNeeded(Exception):
    """The streaming flattener cannot take this file (the message says why): the two-phase one has to."""


class FlatVcf:
    def __init__(self, text: str | None, threads: int = 0, flavour: str = "Genome1000", quality_filter: bool = False, path=None,
                 chunk_bytes: int = 0, streaming: bool = False):
        """From text, or (path=...) from a file read chunk_bytes of text at a time (kgxh_flatten_vcf_file); streaming=True:
        through the streaming flattener and an in-memory sink, rows put back into the two-phase order."""
        assert flavour in ("Genome1000", "Falciparum")
        if streaming:
            err = C.create_string_buffer(512)
            two_phase = C.c_int(0)
            h = lib().kgxh_flatten_vcf_file_streaming(str(path).encode(), 0 if flavour == "Genome1000" else 1, threads, int(quality_filter),
                                                      chunk_bytes, err, 512, C.byref(two_phase))
            if not h:
                raise (TwoPhaseNeeded if two_phase.value else IOError)(err.value.decode())
        elif path is not None:
            err = C.create_string_buffer(512)
            h = lib().kgxh_flatten_vcf_file(str(path).encode(), 0 if flavour == "Genome1000" else 1, threads, int(quality_filter),
                                            chunk_bytes, err, 512)
            if not h:
                raise IOError(err.value.decode())
        else:
            b = text.encode()
            if flavour == "Genome1000":
                h = lib().kgxh_flatten_vcf1000(b, len(b), threads)
            else:
                h = lib().kgxh_flatten_vcf_pf(b, len(b), threads, int(quality_filter))
        assert h
        try:
            self.G, self.V = int(lib().kgxh_flat_genomes(h)), int(lib().kgxh_flat_variants(h))
            self.row_bytes = int(lib().kgxh_flat_row_bytes(h))
            self.variant_objects = int(lib().kgxh_flat_variant_objects(h))
            self.non_diploid = int(lib().kgxh_flat_non_diploid(h))
            self.packed = np.zeros((self.V, self.row_bytes), dtype=np.uint8)
            self.info_af = np.zeros(self.V, dtype=np.float32)
            self.is_snp = np.zeros(self.V, dtype=np.uint8)
            self.offsets = np.zeros(self.V, dtype=np.uint64)
            p = lambda a: C.c_void_p(a.ctypes.data)
            lib().kgxh_flat_copy(h, p(self.packed), p(self.info_af), p(self.is_snp), p(self.offsets))
            # per-bin split rows of variants whose repeated records fall in different FWS bins
            self.n_split = int(lib().kgxh_flat_split_rows(h))
            self.split_packed = np.zeros((self.n_split, self.row_bytes), dtype=np.uint8)
            self.split_info_af = np.zeros(self.n_split, dtype=np.float32)
            self.split_of = np.zeros(self.n_split, dtype=np.int64)
            self.from_splits = np.zeros(self.V, dtype=np.uint8)
            lib().kgxh_flat_copy_splits(h, p(self.split_packed), p(self.split_info_af), p(self.split_of), p(self.from_splits))
            buf = C.create_string_buffer(1024)
            self.hgvs, self.genome_ids = [], []
            for i in range(self.V):
                lib().kgxh_flat_hgvs(h, i, buf, 1024)
                self.hgvs.append(buf.value.decode())
            for i in range(self.G):
                lib().kgxh_flat_genome_id(h, i, buf, 1024)
                self.genome_ids.append(buf.value.decode())
        finally:
            lib().kgxh_flat_destroy(h)


def fws_genome_bins(flat) -> np.ndarray:
    """[G][11][3] by-genome counts per FWS bin from a FlatVcf, the way GpuAlleleAnalysis assigns rows to bins."""
    from kgl_gene_amd import capi
    from kgl_gene_amd.fws import fws_bin_of_variant

    out = np.zeros((flat.G, 11, 3), dtype=np.uint64)
    dose = capi.unpack_dosage2(flat.packed, flat.G) if flat.V else np.zeros((0, flat.G), dtype=np.uint8)
    bins = fws_bin_of_variant(np.where(np.isinf(flat.info_af), np.nan, flat.info_af)).astype(np.int64)
    bins[flat.from_splits.astype(bool)] = 255
    if flat.n_split:
        dose = np.concatenate([dose, capi.unpack_dosage2(flat.split_packed, flat.G)])
        bins = np.concatenate([bins, fws_bin_of_variant(np.where(np.isinf(flat.split_info_af), np.nan, flat.split_info_af)).astype(np.int64)])
    for b in range(11):
        sel = dose[bins == b]
        out[:, b, :] = np.stack([(sel == 0).sum(0), (sel == 1).sum(0), (sel == 2).sum(0)], 1)
    return out


class InbreedInputs:
    """The INBREED package's two inputs flattened from VCF text: reference loci (offset, alts, AF per super population)
    and the population's allele-index bytes [n_loci][genomes]."""

    def __init__(self, reference_text: str, data_source: int, diploid_text: str | None, threads: int = 0, diploid_path=None, chunk_bytes: int = 0,
                 reference_path=None, streaming: bool = False):
        rb = reference_text.encode()
        if streaming:
            why = C.create_string_buffer(512)
            two_phase = C.c_int(0)
            h = lib().kgxh_inbreed_inputs_file_streaming(rb, len(rb), data_source, str(diploid_path).encode(), threads, chunk_bytes, C.byref(two_phase), why, 512)
            if not h:
                raise (TwoPhaseNeeded if two_phase.value else IOError)(why.value.decode())
        elif diploid_path is not None:
            # reference_path: the reference site file is read in pieces too (reference_text is then ignored)
            if reference_path is not None:
                h = lib().kgxh_inbreed_inputs_file(str(reference_path).encode(), 0, data_source, str(diploid_path).encode(), threads, chunk_bytes)
            else:
                h = lib().kgxh_inbreed_inputs_file(rb, len(rb), data_source, str(diploid_path).encode(), threads, chunk_bytes)
            if not h:
                raise IOError(str(diploid_path))
        else:
            db = diploid_text.encode()
            h = lib().kgxh_inbreed_inputs(rb, len(rb), data_source, db, len(db), threads)
        assert h
        try:
            self.L, self.G = int(lib().kgxh_inbreed_loci(h)), int(lib().kgxh_inbreed_genomes(h))
            self.amax, self.contigs = int(lib().kgxh_inbreed_max_alts(h)), int(lib().kgxh_inbreed_contigs(h))
            buf = C.create_string_buffer(1024)
            lib().kgxh_inbreed_error(h, buf, 1024)
            self.error = buf.value.decode()
            self.offsets = np.zeros(self.L, dtype=np.uint64)
            self.n_alts = np.zeros(self.L, dtype=np.uint32)
            self.af = np.zeros((self.L, max(self.amax, 1), 6), dtype=np.float64)
            self.bytes = np.zeros((self.L, self.G), dtype=np.uint8)
            p = lambda a: C.c_void_p(a.ctypes.data)
            lib().kgxh_inbreed_copy(h, p(self.offsets), p(self.n_alts), p(self.af), max(self.amax, 1), p(self.bytes) if self.G and not self.error else None)
            lib().kgxh_inbreed_wide_loci.restype = C.c_uint64
            lib().kgxh_inbreed_wide_loci.argtypes = [C.c_void_p]
            lib().kgxh_inbreed_copy_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            n_wide = int(lib().kgxh_inbreed_wide_loci(h))
            self.wide_loci = np.zeros(n_wide, dtype=np.uint32)          # loci with more than 14 alts: their cells are 16-bit
            self.wide_cells = np.zeros((n_wide, self.G), dtype=np.uint16)
            if n_wide and self.G and not self.error:
                lib().kgxh_inbreed_copy_wide(h, p(self.wide_loci), p(self.wide_cells))
            self.genome_ids = []
            for i in range(self.G):
                lib().kgxh_inbreed_genome_id(h, i, buf, 1024)
                self.genome_ids.append(buf.value.decode())
        finally:
            lib().kgxh_inbreed_inputs_destroy(h)


def read_vcf_text(path, threads: int = 0) -> bytes:
    """kgx_vcf_io.h: readVcfText.  Raises ValueError with the product's message on failure."""
    n = C.c_uint64(0)
    err = C.create_string_buffer(512)
    ptr = lib().kgxh_read_vcf_text(str(path).encode(), C.byref(n), threads, err, 512)
    if not ptr:
        raise ValueError(err.value.decode())
    try:
        return C.string_at(ptr, n.value)
    finally:
        lib().kgxh_free(ptr)
