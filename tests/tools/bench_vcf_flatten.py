"""Host-side VCF -> SoA flattener throughput vs the oracle's parse-to-objects path (CPU only)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import host_api as ha, oracle_api as oa, synth_vcf as sv, vcf_text as vt

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rec, gt = sv.multiallelic_block(G, L, rng_seed=1, dup_records=0)
ids = [f"NA{i:05d}" for i in range(G)]
text = vt.write_vcf_1000(rec, gt, ids, quirks=False)
print(f"{G} samples x {L} records: {len(text)/1e6:.1f} MB of VCF text, {G*L:.3g} genotype cells")
t0 = time.perf_counter(); flat = ha.FlatVcf(text); t1 = time.perf_counter() - t0
print(f"product flattenVcf1000 (+ copy-out): {t1:.2f} s  = {len(text)/t1/1e6:.0f} MB/s, {G*L/t1:.3g} cells/s; {flat.V} variants, {flat.variant_objects} allele copies")
t0 = time.perf_counter(); o = oa.Population('x'); o.add_vcf_1000(text); t2 = time.perf_counter() - t0
t0 = time.perf_counter(); vdb = oa.VariantDB(o); t3 = time.perf_counter() - t0
print(f"oracle (reference-style) parse to Variant objects: {t2:.2f} s; createVariantDB: {t3:.2f} s; total {t2+t3:.2f} s = {G*L/(t2+t3):.3g} cells/s")
print(f"speed-up of the direct flattener: {(t2+t3)/t1:.1f}x on {__import__('os').cpu_count()} cpus")
