"""K3 (by-genome sweep) timing at C3: python scripts/bench_by_genome.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi
from kgl_gene_amd.fws import fws_bin_of_variant
capi.init(0)
G, V = 10_000, 10_000_000
pop = capi.Population(G, V); pop.synth_biallelic(1111, 0, 0)
for rep in range(3):
    t0 = time.perf_counter(); byg = pop.count_by_genome(); dt = time.perf_counter() - t0
    print(f"all rows, 1 bin: wall {dt*1e3:.1f} ms", flush=True)
bins = fws_bin_of_variant(pop.get_af())
for rep in range(2):
    t0 = time.perf_counter(); byb = pop.count_by_genome_binned(bins, 11); dt = time.perf_counter() - t0
    print(f"11 FWS bins: wall {dt*1e3:.1f} ms (incl. 10 MB bin upload, device-side grouping, work-list upload)", flush=True)
assert np.array_equal(byb.sum(1), byg)
