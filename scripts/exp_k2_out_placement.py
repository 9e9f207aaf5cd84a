"""K2 at C3 into differently placed output buffers: does where the [V][4] counts land move the kernel's time?
(bench.py runs showed the same kernel 5 % apart between its timed region's buffer and a scratch buffer, either way round.)"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kgl_gene_amd import capi

if len(sys.argv) > 1:                     # another build of the library (an experiment variant)
    capi.LIB_PATH = Path(sys.argv[1]).resolve()
capi.init(0)
dev = torch.device("cuda:0")
G, V = 10_000, 10_000_000
pop = capi.Population(G, V)
pop.synth_biallelic(20240607, 0, 0)
capi.synchronize()
stream = torch.cuda.current_stream(dev).cuda_stream
nbytes = V * 16
print(f"rows base: see below; output {nbytes} bytes", flush=True)

def timed(ptr, label):
    ms = pop.allele_count_timed(ptr, stream, 2, 12)
    print(f"{label:<46s} ptr%2MiB={ptr % (2 << 20):>8d}  median {np.median(ms):.3f}  min {ms.min():.3f}  max {ms.max():.3f}", flush=True)

# (a) separate allocations, kept alive so that each lands somewhere else
keep = []
for i in range(6):
    t = torch.empty(nbytes + (i * 37 << 20), dtype=torch.uint8, device=dev)        # odd sizes: different blocks of the allocator
    keep.append(t)
    timed(t.data_ptr(), f"allocation {i} ({t.numel() >> 20} MiB)")
# (b) one allocation, shifted starts
big = torch.empty(nbytes + (8 << 20), dtype=torch.uint8, device=dev)
for off in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 4 << 20):
    timed(big.data_ptr() + off, f"one allocation + {off}")
# (c) again the first, after everything else: drift over the run?
timed(keep[0].data_ptr(), "allocation 0 again")
