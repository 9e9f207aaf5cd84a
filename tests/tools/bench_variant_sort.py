"""Host-side timing of the rsid / Ensembl indexes on columns (kgx_variant_sort.h) beside the oracle's restatement of
VariantSort (node-per-entry maps over Variant objects) on the same VCF text.   python tests/tools/bench_variant_sort.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import host_api as H, oracle_api as O, test_variant_sort_cpu as T   # noqa: E402


def timed(fn):
    t = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t


for flavour, n_records, n_samples in (("MonoGenome", 200_000, 0), ("Genome1000", 20_000, 400)):
    text = T.sort_vcf(11, flavour, n_records=n_records, n_samples=max(n_samples, 1))
    population, t_load = timed(lambda: T.oracle_population(text, flavour))
    print(f"{flavour}: {n_records} records, {n_samples} samples, {len(text) / 1e6:.1f} MB of text; oracle population load {t_load:.2f} s")
    for what in ("ensembl", "id", "genome_id"):
        expect, t_oracle = timed(lambda: population.variant_sort(what))
        got, t_host = timed(lambda: H.variant_sort(text, flavour, what))
        print(f"  {what:10s} {len(expect):9d} entries  oracle (maps, after load) {t_oracle:6.2f} s   columns (text -> index -> dump) {t_host:6.2f} s   equal: {got == expect}")
    print("  columns path, build only (ms):", "  ".join(f"{row[0]} {float(row[1]):.1f}" for row in H.variant_sort(text, flavour, "timing")))
