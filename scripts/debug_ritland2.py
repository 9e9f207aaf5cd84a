"""A handful of loci with one odd byte each: one-pass Ritland vs the generic kernel, per genome."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi
capi.init(0)
G, L = 8, 16
rows = np.zeros((L, G), dtype=np.uint8)
rows[:, 1] = 1
rows[3, 0] = 0xFF
rows[5, 2] = 0x1F
rows[7, 3] = 0x10
rows[9, 4] = 0x41
table = np.full((L, 2), np.nan); table[:, 0] = 0.2; table[::2, 1] = 0.1
m = capi.GenotypeMatrix(G, L); m.load_rows(rows)
names = ["major_homo_count", "major_hetero_count", "minor_homo_count", "minor_hetero_count", "total_allele_count", "major_homo_freq", "major_hetero_freq", "minor_homo_freq", "minor_hetero_freq", "inbred_allele_sum"]
got = m.inbreed(table, "RitlandLocus", phased=True)
os.environ["KGX_K5_GENERIC"] = "1"
ref = m.inbreed(table, "RitlandLocus", phased=True)
for n in names:
    print(f"{n:22s} got {np.array2string(got[n][:6], precision=6)}\n{'':22s} ref {np.array2string(ref[n][:6], precision=6)}")
