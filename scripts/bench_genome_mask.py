"""K2 / K3 at C3 (10k genomes x 10M variants) with and without a genome mask (kgx_population_set_genome_mask): the masked
sweeps read the same bytes plus one cached mask row.   python scripts/bench_genome_mask.py [genomes variants]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
capi.ensure_built()
torch.cuda.init()
capi.init(0)
pop = capi.Population(G, V)
pop.synth_biallelic(1111, 0, 0)
out = torch.empty((V, 4), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
gb = pop.sweep_bytes / 1e9
for label, keep in [("no mask", None), ("mask, 70 % kept", (np.random.default_rng(1).random(G) < 0.7)), ("no mask again", None)]:
    pop.set_genome_mask(keep)
    ms = pop.allele_count_timed(out.data_ptr(), stream, 3, 10)
    med = float(np.median(ms))
    edges = [0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.5, 1.0]
    pop.count_by_genome_af_bins(edges)
    t0 = time.perf_counter()
    pop.count_by_genome_af_bins(edges)
    wall = (time.perf_counter() - t0) * 1e3
    print(f"{label}: K2 {med:.3f} ms = {gb / med:.2f} TB/s; K3 11 bins kernel {capi.count_by_genome_last_ms():.3f} ms, call {wall:.1f} ms", flush=True)
pop.close()
