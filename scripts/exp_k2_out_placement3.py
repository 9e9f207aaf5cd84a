"""Follow-up 2: scan one 6 GiB allocation in 160 MB windows, then fresh same-size allocations again (is a fast buffer
fast for good?), then the rows' side: the same output buffer against a second copy of the population."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

capi.init(0)
dev = torch.device("cuda:0")
G, V = 10_000, 10_000_000
pop = capi.Population(G, V)
pop.synth_biallelic(20240607, 0, 0)
capi.synchronize()
stream = torch.cuda.current_stream(dev).cuda_stream
nbytes = V * 16

def timed(p, ptr, label):
    ms = p.allele_count_timed(ptr, stream, 2, 8)
    print(f"{label:<34s} va {ptr:#016x}  median {np.median(ms):.3f}  min {ms.min():.3f}", flush=True)
    return float(np.median(ms))

big = torch.empty(6 << 30, dtype=torch.uint8, device=dev)
step = 160 << 20
res = [timed(pop, big.data_ptr() + i * step, f"6 GiB allocation, window {i}") for i in range((6 << 30) // step - 1)]
print("windows: min %.3f max %.3f" % (min(res), max(res)), flush=True)
keep = []
fast = None
for i in range(12):
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    keep.append(t)
    ms = timed(pop, t.data_ptr(), f"same size {i}")
    if fast is None or ms < fast[0]:
        fast = (ms, t)
timed(pop, fast[1].data_ptr(), "the fastest of those, again")
pop2 = capi.Population(G, V)
pop2.synth_biallelic(20240607, 0, 0)
capi.synchronize()
timed(pop2, fast[1].data_ptr(), "second population -> that buffer")
timed(pop2, keep[0].data_ptr(), "second population -> same size 0")
timed(pop, keep[0].data_ptr(), "first population -> same size 0")
