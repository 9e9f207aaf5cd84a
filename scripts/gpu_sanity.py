"""One-off GPU sanity + first timings (not a test): python scripts/gpu_sanity.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
print(capi.device_info())

def dense_counts(codes, G):
    het = (codes == 1).sum(1); hom = (codes == 2).sum(1); oth = (codes == 3).sum(1)
    return np.stack([G - het - hom - oth, het, hom, oth], 1).astype(np.uint32)

ok = True
for G, V in [(100, 500), (1000, 2000), (4097, 700), (10000, 300), (12500, 257), (3, 10), (70000, 64)]:
    pop = capi.Population(G, V)
    pop.synth_biallelic(1111, 0, 0)
    dev = pop.read_dosage2()
    host, af = capi.synth_biallelic_host(1111, 0, G, 0, V)
    same = np.array_equal(dev, host) and np.array_equal(af, pop.get_af())
    codes = capi.unpack_dosage2(host, G)
    # inject non-diploid codes + reload through the host path
    rng = np.random.default_rng(G)
    codes[rng.integers(0, V, 50), rng.integers(0, G, 50)] = 3
    pop.load_dosage2(capi.pack_dosage2(codes))
    k2 = pop.allele_count_by_locus()
    want = dense_counts(codes, G)
    k2ok = np.array_equal(k2, want)
    byg = pop.count_by_genome()
    wantg = np.stack([(codes == 0).sum(0), (codes == 1).sum(0), (codes == 2).sum(0), (codes == 3).sum(0)], 1).astype(np.uint64)
    k3ok = np.array_equal(byg, wantg)
    bins = rng.integers(0, 12, V).astype(np.uint8); bins[bins == 11] = 255
    byb = pop.count_by_genome_binned(bins, 11)
    k3b = True
    for b in range(11):
        sel = codes[bins == b]
        w = np.stack([(sel == 0).sum(0), (sel == 1).sum(0), (sel == 2).sum(0), (sel == 3).sum(0)], 1).astype(np.uint64)
        k3b &= np.array_equal(byb[:, b], w)
    summ = pop.population_summary()
    k4ok = np.array_equal(summ, want.astype(np.uint64).sum(0))
    # u8 loader
    pop2 = capi.Population(G, V)
    d8 = codes.T.copy(); d8[d8 == 3] = 7
    pop2.load_dosage_u8(d8)
    u8ok = np.array_equal(pop2.read_dosage2(), capi.pack_dosage2(codes))
    print(f"G={G} V={V} synth={same} K2={k2ok} K3={k3ok} K3bin={k3b} K4={k4ok} u8={u8ok}")
    ok &= same and k2ok and k3ok and k3b and k4ok and u8ok
    pop.close(); pop2.close()
print("ALL OK" if ok else "FAILURES")

import ctypes
lib = capi.lib()
def time_cfg(G, V, iters=20):
    pop = capi.Population(G, V)
    t0 = time.time(); pop.synth_biallelic(1111, 0, 0); capi.synchronize(); ts = time.time() - t0
    d_out = ctypes.c_void_p()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc(ctypes.byref(d_out), ctypes.c_size_t(V * 16))
    ms = pop.allele_count_timed(d_out.value, 0, 3, iters)
    b = pop.sweep_bytes
    print(f"K2 G={G} V={V}: synth {ts:.2f}s  median {np.median(ms):.3f} ms  min {ms.min():.3f} ms  "
          f"{b/np.median(ms)/1e6:.1f} GB/s  cells/s {G*V/np.median(ms)*1e3:.3e}")
    t0 = time.time(); byg = pop.count_by_genome(); tg = time.time() - t0
    print(f"   K3 (incl. host glue) {tg*1e3:.1f} ms")
    hip.hipFree(d_out); pop.close()

time_cfg(1000, 1_000_000)
time_cfg(10000, 1_000_000)
time_cfg(10000, 10_000_000, iters=10)
time_cfg(12500, 10_000_000, iters=10)
