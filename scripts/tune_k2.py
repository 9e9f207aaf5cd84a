"""K2 tuning sweep on one GPU (interleaved rounds in one process; guide rule 24)."""
import ctypes, itertools, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls (the library reads them at kgx_init / kgx_reload_options)

capi.init(0)
hip = ctypes.CDLL("libamdhip64.so")
G, V = int(os.environ.get("G", 10000)), int(os.environ.get("V", 10_000_000))
d_out = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(d_out), ctypes.c_size_t(V * 16))

def run(cfg, iters=8):
    for k, v in cfg.items():
        os.environ[k] = str(v)
    return pop.allele_count_timed(d_out.value, 0, 1, iters)

results = {}
for align in [int(a) for a in os.environ.get("ALIGNS", "16,128").split(",")]:
    os.environ["KGX_PITCH_ALIGN"] = str(align)
    pop = capi.Population(G, V)
    pop.synth_biallelic(1111, 0, 0)
    ref = None
    cfgs = [dict(KGX_K2_U=u, KGX_K2_NT=nt, KGX_K2_BLOCKS_PER_CU=b)
            for u, nt, b in itertools.product((2, 4, 8), (1,), (8, 32))]
    times = {i: [] for i in range(len(cfgs))}
    for rnd in range(3):
        for i, c in enumerate(cfgs):
            times[i].extend(run(c).tolist())
    # correctness of every config vs the first
    for i, c in enumerate(cfgs):
        for k, v in c.items():
            os.environ[k] = str(v)
        out = pop.allele_count_by_locus()
        if ref is None:
            ref = out
        assert np.array_equal(out, ref), c
    b = pop.sweep_bytes
    for i, c in enumerate(cfgs):
        t = np.array(times[i])
        print(f"align={align:3d} pitch={pop.row_pitch} U={c['KGX_K2_U']} NT={c['KGX_K2_NT']} bpc={c['KGX_K2_BLOCKS_PER_CU']:2d} "
              f"median={np.median(t):.3f} ms min={t.min():.3f} ms  {b/np.median(t)/1e6:7.1f} GB/s (alg)", flush=True)
    pop.close()
