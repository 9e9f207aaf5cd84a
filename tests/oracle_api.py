"""ctypes driver of oracle/_build/libkgo.so — the CPU restatement used as the checker.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
import os as _os

# tests/tools/sanitize_host.sh points these at ASan/UBSan-instrumented builds of the same sources
LIB_PATH = Path(_os.environ.get("KGX_SANITIZED_ORACLE_LIB") or ROOT / "oracle" / "_build" / "libkgo.so")

SUPER_POPS = ["AFR", "AMR", "EAS", "EUR", "SAS", "ALL"]
ALL = 5

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle")], check=True)
    L = C.CDLL(str(LIB_PATH))
    vp, u64, i64, dbl = C.c_void_p, C.c_uint64, C.c_int64, C.c_double
    sig = {
        "kgo_banner": (C.c_char_p, []),
        "kgo_set_threads": (None, [C.c_int]),
        "kgo_default_threads": (C.c_int, []),
        "kgo_pool_threads": (C.c_int, [u64]),
        "kgo_population_create": (vp, [C.c_char_p]),
        "kgo_population_destroy": (None, [vp]),
        "kgo_population_add_genomes": (C.c_int, [vp, u64, vp, C.c_int]),
        "kgo_population_add_records": (C.c_int, [vp, C.c_int, C.c_char_p, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
        "kgo_population_add_records_coded": (C.c_int, [vp, C.c_int, C.c_char_p, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
        "kgo_population_add_vcf_1000": (C.c_long, [vp, C.c_char_p, u64]),
        "kgo_fast_count_by_variant": (C.c_int, [vp, u64, u64, u64, u64, vp, C.c_int, C.c_int, vp]),
        "kgo_population_add_vcf_pf": (C.c_long, [vp, C.c_char_p, u64]),
        "kgo_population_add_vcf_mono": (C.c_long, [vp, C.c_char_p, u64, C.c_char_p, C.c_char_p]),
        "kgo_hethom_present": (C.c_int, [vp, C.c_char_p, vp]),
        "kgo_population_filter_p7": (vp, [vp]),
        "kgo_population_filter_pf7_genomes": (vp, [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, dbl]),
        "kgo_pfemp_location_write": (C.c_int, [vp, C.c_char_p, C.c_char_p, dbl, C.c_char_p, C.c_char_p]),
        "kgo_canonical": (u64, [C.c_char_p, C.c_char_p, u64, C.c_char_p, C.c_char_p]),
        "kgo_gt_alternate_index": (C.c_int, [C.c_char_p, C.c_char_p, u64, vp]),
        "kgo_population_variant_count": (u64, [vp]),
        "kgo_population_genome_count": (u64, [vp]),
        "kgo_population_genome_order": (C.c_int, [vp, vp]),
        "kgo_population_filter_snp_pass": (vp, [vp]),
        "kgo_vdb_create": (vp, [vp, vp]),
        "kgo_vdb_destroy": (None, [vp]),
        "kgo_vdb_variants": (u64, [vp]),
        "kgo_vdb_genomes": (u64, [vp]),
        "kgo_vdb_warnings": (u64, [vp]),
        "kgo_vdb_variant_keys": (C.c_int, [vp, vp, vp]),
        "kgo_vdb_hgvs": (C.c_int, [vp, u64, C.c_char_p, C.c_size_t]),
        "kgo_vdb_genome_id": (C.c_int, [vp, u64, C.c_char_p, C.c_size_t]),
        "kgo_vdb_summary_by_variant": (C.c_int, [vp, vp, vp]),
        "kgo_vdb_summary_by_genome": (C.c_int, [vp, vp, vp]),
        "kgo_vdb_population_summary": (C.c_int, [vp, vp]),
        "kgo_vdb_dosage": (C.c_int, [vp, vp]),
        "kgo_dense_create": (vp, [vp, u64, u64]),
        "kgo_dense_destroy": (None, [vp]),
        "kgo_dense_summary_by_variant": (C.c_int, [vp, u64, u64, vp, vp]),
        "kgo_dense_summary_by_genome": (C.c_int, [vp, vp, vp, vp]),
        "kgo_fws": (C.c_int, [vp, vp, vp]),
        "kgo_hethom": (C.c_int, [vp, C.c_char_p, vp]),
        "kgo_offset_filter_counts": (C.c_int, [vp, C.c_char_p, vp]),
        "kgo_unique_phased_counts": (C.c_int, [vp, C.c_char_p, vp]),
        "kgo_wrights_fis": (dbl, [vp, vp]),
        "kgo_class_frequencies": (C.c_int, [vp, C.c_uint32, dbl, C.c_int, vp]),
        "kgo_sample_locii": (i64, [vp, C.c_int, C.c_int, u64, u64, u64, u64, dbl, dbl, vp, u64]),
        "kgo_inbreed_window": (C.c_int, [vp, vp, vp, C.c_char_p, u64, u64, u64, u64, dbl, dbl, u64, vp, vp, vp, vp]),
        "kgo_restart_draws": (C.c_int, [C.c_char_p, u64, u64, u64, vp]),
        "kgo_neldermead_path": (C.c_int, [C.c_int, dbl, dbl, vp, C.c_int, vp, vp]),
        "kgo_loglikelihood_at": (C.c_int, [vp, vp, vp, u64, u64, u64, dbl, dbl, vp, vp]),
        "kgo_inbreed_dense": (C.c_int, [vp, vp, C.c_int, u64, u64, u64, dbl, dbl, vp, u64, vp, u64, C.c_int, vp, vp, vp]),
        "kgo_population_inbreeding": (vp, [vp, vp, vp, C.c_char_p, u64, u64, u64, u64, dbl, dbl, u64]),
        "kgo_columns_destroy": (None, [vp]),
        "kgo_columns_write_ped": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_char_p, dbl, dbl, u64, u64, vp, u64]),
        "kgo_columns_count": (u64, [vp]),
        "kgo_columns_ident": (C.c_int, [vp, u64, C.c_char_p, C.c_size_t]),
        "kgo_columns_results": (C.c_int, [vp, u64, vp, vp, vp]),
        "kgo_variant_sort": (vp, [vp, C.c_char_p, C.c_char_p]),
        "kgo_free_text": (None, [vp]),
        "kgo_synthetic_check": (i64, [vp, C.c_int, C.c_char_p, u64, u64, u64, dbl, dbl, u64, vp, vp, u64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _strs(items):
    arr = (C.c_char_p * len(items))()
    arr[:] = [s.encode() if isinstance(s, str) else s for s in items]
    return arr


class Records:
    """A VCF-like block: what a parser would hand to PopulationDB."""

    def __init__(self, contig, offsets, refs, alts, af=None, passed=None):
        self.contig = contig
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.refs = list(refs)
        self.alts = [list(a) for a in alts]            # per record: list of alt strings
        self.n_alts = np.array([len(a) for a in self.alts], dtype=np.uint8)
        self.n_records = len(self.refs)
        # af: list per record of [n_alt][6] float32 (NaN = missing) or None
        self.af = af
        self.passed = None if passed is None else np.ascontiguousarray(passed, dtype=np.uint8)

    def af_flat(self):
        if self.af is None:
            return None
        return np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1, 6) for a in self.af]),
                                    dtype=np.float32)

    def alts_flat(self):
        return [x for a in self.alts for x in a]


class Population:
    PHASED, UNPHASED, REFERENCE = 0, 1, 2

    def __init__(self, name="population", handle=None):
        self._h = handle if handle is not None else lib().kgo_population_create(name.encode())
        self.genome_ids: list[str] = []

    def close(self):
        if self._h:
            lib().kgo_population_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_genomes(self, ids, precreate=True):
        self.genome_ids = list(ids)
        arr = _strs(self.genome_ids)
        assert lib().kgo_population_add_genomes(self._h, len(ids), C.cast(arr, C.c_void_p), int(precreate)) == 0

    def add_records(self, rec: Records, gt: np.ndarray | None, mode: int):
        """gt: [n_records][n_genomes][2] uint8 allele indices (0 = ref)."""
        ids = _strs(self.genome_ids)
        refs = _strs(rec.refs)
        alts = _strs(rec.alts_flat())
        af = rec.af_flat()
        if gt is not None:
            gt = np.ascontiguousarray(gt, dtype=np.uint8)
            assert gt.shape == (rec.n_records, len(self.genome_ids), 2)
        rc = lib().kgo_population_add_records(self._h, mode, rec.contig.encode(), rec.n_records, _p(rec.offsets),
                                              C.cast(refs, C.c_void_p), _p(rec.n_alts), C.cast(alts, C.c_void_p),
                                              _p(rec.passed), _p(af), len(self.genome_ids), C.cast(ids, C.c_void_p), _p(gt))
        assert rc == 0

    def add_records_coded(self, contig, offsets, ref_code, n_alts, alt_code, af_flat, gt, mode, passed=None):
        """add_records for large synthetic blocks: sequences as codes (kgo_population_add_records_coded), numpy in."""
        ids = _strs(self.genome_ids)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ref_code = np.ascontiguousarray(ref_code, dtype=np.uint8)
        n_alts = np.ascontiguousarray(n_alts, dtype=np.uint8)
        alt_code = np.ascontiguousarray(alt_code, dtype=np.uint8)
        af_flat = None if af_flat is None else np.ascontiguousarray(af_flat, dtype=np.float32)
        if gt is not None:
            gt = np.ascontiguousarray(gt, dtype=np.uint8)
            assert gt.shape == (len(offsets), len(self.genome_ids), 2)
        rc = lib().kgo_population_add_records_coded(self._h, mode, contig.encode(), len(offsets), _p(offsets), _p(ref_code), _p(n_alts),
                                                    _p(alt_code), _p(passed), _p(af_flat), len(self.genome_ids), C.cast(ids, C.c_void_p), _p(gt))
        assert rc == 0

    def add_vcf_1000(self, text: str) -> int:
        """Parse VCF text the way Genome1000VCFImpl does; genome ids come from the #CHROM line."""
        b = text.encode()
        n = lib().kgo_population_add_vcf_1000(self._h, b, len(b))
        assert n >= 0
        return int(n)

    def add_vcf_pf(self, text: str) -> int:
        """Parse VCF text the way PfVCFImpl does (unphased P. falciparum flavour): every sample becomes a genome."""
        b = text.encode()
        n = lib().kgo_population_add_vcf_pf(self._h, b, len(b))
        assert n >= 0
        return int(n)

    def add_vcf_mono(self, text: str, source: str, genome_id: str = "Reference") -> int:
        """Parse the VCF of a mono-genome frequency source (Gnomad ...) the way GrchVCFImpl does."""
        b = text.encode()
        n = lib().kgo_population_add_vcf_mono(self._h, b, len(b), source.encode(), genome_id.encode())
        assert n >= 0
        if not self.genome_ids:
            self.genome_ids = [genome_id]
        return int(n)

    def variant_sort(self, what: str, names=None) -> list[tuple[str, ...]]:
        """VariantSort / SortedVariantAnalysis on this population (oracle/kgo_sort.h): the index `what` as rows of strings."""
        listed = None if names is None else "\n".join(names).encode()
        ptr = lib().kgo_variant_sort(self._h, what.encode(), listed)
        assert ptr, what
        try:
            text = C.string_at(ptr).decode()
        finally:
            lib().kgo_free_text(ptr)
        return [tuple(line.split("\t")) for line in text.split("\n") if line]

    def filter_p7(self):
        """viewFilter(P7VariantFilter()), the per-record quality filter of FilterPf7::qualityFilter."""
        p = Population(handle=lib().kgo_population_filter_p7(self._h))
        p.genome_ids = list(self.genome_ids)
        return p

    def filter_pf7_genomes(self, sample_file, fws_file, filter_qc=True, filter_fws=True, fws_threshold=0.95):
        """FilterPf7::qualityFilter's genome part (QC pass, monoclonal FWS) + squareContigs, from the two resource files."""
        h = lib().kgo_population_filter_pf7_genomes(self._h, str(sample_file).encode(), str(fws_file).encode(), int(filter_qc), int(filter_fws),
                                                    float(fws_threshold))
        assert h, "the Pf7 resource files did not parse"
        p = Population(handle=h)
        p.genome_ids = list(self.genome_ids)
        return p

    def write_pfemp_location(self, sample_file, fws_file, statistics_csv, location_csv, radius_km=0.0):
        """HeteroHomoZygous' VariantStatistics.csv and VariantLocation.csv for this population."""
        return int(lib().kgo_pfemp_location_write(self._h, str(sample_file).encode(), str(fws_file).encode(), float(radius_km),
                                                  str(statistics_csv).encode(), str(location_csv).encode()))

    @property
    def handle(self):
        return self._h

    def variant_count(self):
        return int(lib().kgo_population_variant_count(self._h))

    def genome_count(self):
        return int(lib().kgo_population_genome_count(self._h))

    def genome_order(self):
        """Sorted-by-id position -> index in the caller's genome id list."""
        out = np.zeros(self.genome_count(), dtype=np.int64)
        assert lib().kgo_population_genome_order(self._h, _p(out)) == 0
        return out

    def filter_snp_pass(self):
        p = Population(handle=lib().kgo_population_filter_snp_pass(self._h))
        p.genome_ids = list(self.genome_ids)
        return p

    # -- allele-count analyses ----------------------------------------------------------------
    def fws(self):
        G = self.genome_count()
        vdb = VariantDB(self)
        V = vdb.n_variants
        variant_out = np.zeros((V, 3), dtype=np.uint64)
        genome_out = np.zeros((G, 11, 3), dtype=np.uint64)
        assert lib().kgo_fws(self._h, _p(variant_out), _p(genome_out)) == 0
        return variant_out, genome_out, vdb

    def hethom(self, contig):
        out = np.zeros((self.genome_count(), 7), dtype=np.uint64)
        assert lib().kgo_hethom(self._h, contig.encode(), _p(out)) == 0
        return out


    def offset_filter_counts(self, contig):
        """[genomes][4]: Variant objects HomozygousFilter / HeterozygousFilter / DiploidFilter / UniqueUnphasedFilter leave."""
        out = np.zeros((self.genome_count(), 4), dtype=np.uint64)
        assert lib().kgo_offset_filter_counts(self._h, contig.encode(), _p(out)) == 0
        return out

    def unique_phased_counts(self, contig):
        """[genomes]: the Variant objects UniquePhasedFilter (one per distinct HGVS and phase) leaves of the contig."""
        out = np.zeros(self.genome_count(), dtype=np.uint64)
        assert lib().kgo_unique_phased_counts(self._h, contig.encode(), _p(out)) == 0
        return out

    def hethom_present(self, contig):
        out = np.zeros(self.genome_count(), dtype=np.uint8)
        assert lib().kgo_hethom_present(self._h, contig.encode(), _p(out)) == 0
        return out.astype(bool)


class VariantDB:
    """VariantDBVariant (kgl_variant_db_variant.h:53-76)."""

    def __init__(self, pop: Population):
        sec = C.c_double(0)
        self._h = lib().kgo_vdb_create(pop.handle, C.byref(sec))
        self.build_seconds = sec.value
        self.n_variants = int(lib().kgo_vdb_variants(self._h))
        self.n_genomes = int(lib().kgo_vdb_genomes(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().kgo_vdb_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def variant_keys(self):
        rec = np.zeros(self.n_variants, dtype=np.uint64)
        alt = np.zeros(self.n_variants, dtype=np.uint32)
        assert lib().kgo_vdb_variant_keys(self._h, _p(rec), _p(alt)) == 0
        return rec, alt

    def hgvs(self, i):
        buf = C.create_string_buffer(512)
        assert lib().kgo_vdb_hgvs(self._h, i, buf, 512) == 0
        return buf.value.decode()

    def genome_id(self, i):
        buf = C.create_string_buffer(256)
        assert lib().kgo_vdb_genome_id(self._h, i, buf, 256) == 0
        return buf.value.decode()

    def summary_by_variant(self):
        out = np.zeros((self.n_variants, 3), dtype=np.uint64)
        sec = C.c_double(0)
        assert lib().kgo_vdb_summary_by_variant(self._h, _p(out), C.byref(sec)) == 0
        self.by_variant_seconds = sec.value
        return out

    def summary_by_genome(self):
        out = np.zeros((self.n_genomes, 3), dtype=np.uint64)
        sec = C.c_double(0)
        assert lib().kgo_vdb_summary_by_genome(self._h, _p(out), C.byref(sec)) == 0
        self.by_genome_seconds = sec.value
        return out

    def population_summary(self):
        out = np.zeros(3, dtype=np.uint64)
        assert lib().kgo_vdb_population_summary(self._h, _p(out)) == 0
        return out

    def dosage(self):
        out = np.zeros((self.n_genomes, self.n_variants), dtype=np.uint8)
        assert lib().kgo_vdb_dosage(self._h, _p(out)) == 0
        return out

    def warnings(self):
        return int(lib().kgo_vdb_warnings(self._h))


class Dense:
    """The reference's summary loops over a caller-supplied VariantDBGenomeData matrix [G][V]."""

    def __init__(self, dosage: np.ndarray):
        d = np.ascontiguousarray(dosage, dtype=np.uint8)
        self.G, self.V = d.shape
        self._h = lib().kgo_dense_create(_p(d), self.G, self.V)

    def __del__(self):
        try:
            if self._h:
                lib().kgo_dense_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def summary_by_variant(self, v0=0, v1=None):
        v1 = self.V if v1 is None else v1
        out = np.zeros((v1 - v0, 3), dtype=np.uint64)
        sec = C.c_double(0)
        assert lib().kgo_dense_summary_by_variant(self._h, v0, v1, _p(out), C.byref(sec)) == 0
        self.seconds = sec.value
        return out

    def summary_by_genome(self, mask=None):
        out = np.zeros((self.G, 3), dtype=np.uint64)
        sec = C.c_double(0)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        assert lib().kgo_dense_summary_by_genome(self._h, _p(m), _p(out), C.byref(sec)) == 0
        self.seconds = sec.value
        return out


def fast_count_by_variant(rows: np.ndarray, n_genomes: int, threads: int = 0, repeats: int = 3):
    """The tuned CPU comparator (oracle/kgo_fast.cpp; NOT the reference's algorithm): ([n][4] counts, best seconds, threads)."""
    r = np.ascontiguousarray(rows, dtype=np.uint8)
    out = np.zeros((r.shape[0], 4), dtype=np.uint32)
    sec = C.c_double(0)
    used = lib().kgo_fast_count_by_variant(_p(r), r.shape[0], r.shape[1], r.shape[1], n_genomes, _p(out), threads, repeats, C.byref(sec))
    assert used > 0
    return out, sec.value, int(used)


def class_frequencies(minor_af, inbreeding, normalize=True):
    a = np.ascontiguousarray(minor_af, dtype=np.float64)
    out = np.zeros(4, dtype=np.float64)
    assert lib().kgo_class_frequencies(_p(a), len(a), float(inbreeding), int(normalize), _p(out)) == 0
    return out  # majorHom, majorHet, minorHom, minorHet


def sample_locii(reference: Population, super_pop, by_count, lower, upper, spacing, count, min_af, max_af):
    cap = 1 << 22
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().kgo_sample_locii(reference.handle, super_pop, int(by_count), lower, upper, spacing, count, min_af, max_af,
                               _p(out), cap)
    assert n >= 0
    return out[:n].copy()


def inbreed_window(reference: Population, diploid: Population, super_pop_of_genome, algorithm, lower, upper, spacing,
                   count, min_af, max_af, seed=0):
    G = diploid.genome_count()
    sp = np.ascontiguousarray(super_pop_of_genome, dtype=np.int32)
    counts = np.zeros((G, 5), dtype=np.uint64)
    freqs = np.zeros((G, 5), dtype=np.float64)
    present = np.zeros(G, dtype=np.uint8)
    sec = C.c_double(0)
    rc = lib().kgo_inbreed_window(reference.handle, diploid.handle, _p(sp), algorithm.encode(), lower, upper, spacing,
                                  count, min_af, max_af, seed, _p(counts), _p(freqs), _p(present), C.byref(sec))
    assert rc == 0
    return counts, freqs, present.astype(bool), sec.value


def restart_draws(algorithm, seed, n, restarts=5):
    """The start points the oracle's per-genome tasks draw under inbreed_window(seed=seed): [n][restarts], task k from
    std::mt19937_64(seed + k); RetryCalcResult ends the restarts at the fifth, whose run is the result."""
    out = np.zeros((n, restarts), dtype=np.float64)
    assert lib().kgo_restart_draws(algorithm.encode(), int(seed), int(n), int(restarts), _p(out)) == 0
    return out


def neldermead_path(objective: int, a: float, x0: float):
    """The oracle's 1-D Nelder-Mead (its restatement of nlopt's LN_NELDERMEAD) on a closed-form objective over [-1, 1]:
    (points evaluated in order, result).  objective 0: -(x-a)^2, 1: -|x-a|, 2: a*x, 3: step at a."""
    path = np.zeros(600, dtype=np.float64)
    n = C.c_int(0)
    result = C.c_double(0.0)
    assert lib().kgo_neldermead_path(objective, a, x0, _p(path), len(path), C.byref(n), C.byref(result)) == 0
    return path[:n.value].copy(), result.value


def loglikelihood_at(reference: Population, diploid: Population, super_pop_of_genome, lower, upper, spacing, min_af, max_af, f):
    """logLikelihood (_calc.cpp:94-129) of every genome at its own coefficient f[g] (genome-id order), over the window's locus list."""
    sp = np.ascontiguousarray(super_pop_of_genome, dtype=np.int32)
    f = np.ascontiguousarray(f, dtype=np.float64)
    out = np.full(len(f), np.nan)
    assert lib().kgo_loglikelihood_at(reference.handle, diploid.handle, _p(sp), lower, upper, spacing, min_af, max_af, _p(f), _p(out)) == 0
    return out


def inbreed_dense(reference_all: Population, reference_snp_pass: Population, super_pop, lower, upper, spacing, min_af, max_af,
                  record_offsets, allele_pairs, phased=True):
    """The oracle's dense tier (oracle/kgo_inbreed_dense.cpp): generateFrequencies + Simple + RitlandLocus for genomes given
    as raw GT allele pairs [n_records][G][2] of a one-record-per-offset population.
    Returns (counts [G][5], freqs [G][6] = four class-frequency sums, Simple, Ritland; seconds)."""
    offsets = np.ascontiguousarray(record_offsets, dtype=np.uint64)
    pairs = np.ascontiguousarray(allele_pairs, dtype=np.uint8)
    assert pairs.ndim == 3 and pairs.shape[0] == len(offsets) and pairs.shape[2] == 2
    G = pairs.shape[1]
    counts = np.zeros((G, 5), dtype=np.uint64)
    freqs = np.zeros((G, 6), dtype=np.float64)
    sec = C.c_double(0)
    rc = lib().kgo_inbreed_dense(reference_all.handle, reference_snp_pass.handle, super_pop, lower, upper, spacing, min_af, max_af,
                                 _p(offsets), len(offsets), _p(pairs), G, int(bool(phased)), _p(counts), _p(freqs), C.byref(sec))
    assert rc == 0
    return counts, freqs, sec.value


def population_inbreeding(reference: Population, diploid: Population, super_pop_of_genome, algorithm, lower, upper,
                          spacing, count, min_af, max_af, seed=0, ped_file=None):
    """ped_file = (path, param_ident, ped_rows): also write the reference's PED result file (InbreedingOutput::writePedResults)
    from the columns; ped_rows = per genome the 9 strings genome, population, description, super population, description,
    relationship, sex, mother, father."""
    G = diploid.genome_count()
    sp = np.ascontiguousarray(super_pop_of_genome, dtype=np.int32)
    h = lib().kgo_population_inbreeding(reference.handle, diploid.handle, _p(sp), algorithm.encode(), lower, upper,
                                        spacing, count, min_af, max_af, seed)
    assert h
    cols = []
    try:
        if ped_file is not None:
            path, ident, rows = ped_file
            flat = _strs([x for row in rows for x in row])
            assert all(len(row) == 9 for row in rows)
            rc = lib().kgo_columns_write_ped(h, str(path).encode(), ident.encode(), algorithm.encode(), min_af, max_af, spacing, count,
                                             C.cast(flat, C.c_void_p), len(rows))
            assert rc == 0, rc
        for i in range(int(lib().kgo_columns_count(h))):
            buf = C.create_string_buffer(256)
            lib().kgo_columns_ident(h, i, buf, 256)
            counts = np.zeros((G, 5), dtype=np.uint64)
            freqs = np.zeros((G, 5), dtype=np.float64)
            present = np.zeros(G, dtype=np.uint8)
            lib().kgo_columns_results(h, i, _p(counts), _p(freqs), _p(present))
            cols.append((buf.value.decode(), counts, freqs, present.astype(bool)))
    finally:
        lib().kgo_columns_destroy(h)
    return cols


def synthetic_check(reference: Population, super_pop, algorithm, lower, upper, spacing, min_af, max_af, seed):
    syn = np.zeros(128, dtype=np.float64)
    calc = np.zeros(128, dtype=np.float64)
    n = lib().kgo_synthetic_check(reference.handle, super_pop, algorithm.encode(), lower, upper, spacing, min_af, max_af,
                                  seed, _p(syn), _p(calc), 128)
    assert n >= 0
    return syn[:n].copy(), calc[:n].copy()


def canonical(ref: str, alt: str, offset: int):
    """Variant::canonicalSequences -> (ref, alt, offset)."""
    r = C.create_string_buffer(len(ref) + len(alt) + 2)
    a = C.create_string_buffer(len(ref) + len(alt) + 2)
    o = lib().kgo_canonical(ref.encode(), alt.encode(), offset, r, a)
    return r.value.decode(), a.value.decode(), int(o)


def gt_alternate_index(contig: str, genotype: str, n_alt: int):
    out = np.zeros(2, dtype=np.uint64)
    lib().kgo_gt_alternate_index(contig.encode(), genotype.encode(), n_alt, _p(out))
    return int(out[0]), int(out[1])
