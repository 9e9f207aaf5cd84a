"""K2 lanes-per-row sweep at one shape (interleaved rounds in one process): G=10000 V=10000000 by default."""
import ctypes, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
hip = ctypes.CDLL("libamdhip64.so")
G, V = int(os.environ.get("G", 10000)), int(os.environ.get("V", 10_000_000))
d_out = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(d_out), ctypes.c_size_t(V * 16))
pop = capi.Population(G, V)
pop.synth_biallelic(1111, 0, 0)
cfgs = [dict(KGX_K2_W=w, KGX_K2_U=u) for w in (64, 32, 16) for u in (4, 8)]
times = {i: [] for i in range(len(cfgs))}
ref = None
for rnd in range(4):
    for i, c in enumerate(cfgs):
        for k, v in c.items():
            os.environ[k] = str(v)
        times[i].extend(pop.allele_count_timed(d_out.value, 0, 1, 6).tolist())
for i, c in enumerate(cfgs):
    for k, v in c.items():
        os.environ[k] = str(v)
    out = pop.allele_count_by_locus()
    if ref is None:
        ref = out
    assert np.array_equal(out, ref), c
    ms = float(np.median(times[i]))
    print(c, f"median {ms:.3f} ms  {pop.sweep_bytes / ms / 1e9:.2f} TB/s", flush=True)
