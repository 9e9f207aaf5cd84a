"""HallME at C5 (10 k genomes x 5 M loci): the per-genome moments against the 50 passes over the bytes."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.WATCH_ENV = True            # this script flips KGX_* switches between calls (the library reads them at kgx_init / kgx_reload_options)

if os.environ.get("KGX_EXP_LIB"):          # another build of the library (an experiment variant)
    capi.LIB_PATH = Path(os.environ["KGX_EXP_LIB"]).resolve()
capi.init(0)
moments_only = "--moments-only" in sys.argv          # (under the counters: the 50 passes are not what is looked at)
sizes = [a for a in sys.argv[1:] if not a.startswith("--")]
G, L = (int(sizes[0]), int(sizes[1])) if len(sizes) > 1 else (10_000, 5_000_000)
m = capi.GenotypeMatrix(G, L)
table = m.synth_multiallelic(1111, 0, 0)
start = capi.reference_starts("HallME", 4242, G)
results = {}
for label, env in (("moments", None),) + ((() if moments_only else (("50 passes", "1"),))):
    if env:
        os.environ["KGX_K7_HALL_PASSES"] = env
    else:
        os.environ.pop("KGX_K7_HALL_PASSES", None)
    m.inbreed(table, "HallME", phased=True, start=start)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = m.inbreed(table, "HallME", phased=True, start=start)
        walls.append(time.perf_counter() - t0)
    results[label] = res["inbred_allele_sum"].copy()
    print(f"HallME {label}: {np.median(walls) * 1e3:.1f} ms per call ({G} genomes x {L} loci)  mean F {results[label].mean():+.6f}", flush=True)
if not moments_only:
    d = np.abs(results["moments"] - results["50 passes"])
    print(f"|dF| max {d.max():.3e}", flush=True)
