"""K3 (the by-genome sweep, 11 FWS bins + the unbinned sweep) at C3 against K2 on the same box, for several work-list
shapes (KGX_K3_ROUNDS = work items per resident workgroup).   python scripts/bench_k3.py [genomes variants]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi  # noqa: E402

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls (the library reads them at kgx_init / kgx_reload_options)

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
capi.ensure_built()
capi.init(0)
pop = capi.Population(G, V)
pop.synth_biallelic(1111, 0, 0)
out = torch.empty((V, 4), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
gb = pop.sweep_bytes / 1e9
k2 = float(np.median(pop.allele_count_timed(out.data_ptr(), stream, 3, 20)))
print(f"K2 {k2:.3f} ms = {gb / k2:.2f} TB/s", flush=True)
edges = [0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.5, 1.0]
reference = None
for rounds in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "2", "3", "4", "8"]):
    os.environ["KGX_K3_ROUNDS"] = rounds
    binned, plain = [], []
    for i in range(8):
        got = pop.count_by_genome_af_bins(edges)
        if i >= 2:
            binned.append(capi.count_by_genome_last_ms())
    for i in range(5):
        whole = pop.count_by_genome()
        if i >= 1:
            plain.append(capi.count_by_genome_last_ms())
    if reference is None:
        reference = (got, whole)
    same = np.array_equal(got, reference[0]) and np.array_equal(whole, reference[1])
    b, p = float(np.median(binned)), float(np.median(plain))
    print(f"rounds {rounds}: 11 bins {b:.3f} ms = {gb / b:.2f} TB/s ({b / k2:.3f} x K2), min {min(binned):.3f}; "
          f"unbinned {p:.3f} ms ({p / k2:.3f} x K2); results equal: {same}", flush=True)
pop.close()
