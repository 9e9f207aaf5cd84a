# Engine clock and socket power while K2 / K3 stream the C3 population (rocm-smi sampled beside the run).
python - <<'PY' &
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from kgl_gene_amd import capi
capi.init(0)
pop = capi.Population(10000, 10_000_000)
pop.synth_biallelic(1111, 0, 0)
out = torch.empty((10_000_000, 4), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
time.sleep(1.5)
print("K2 begins", flush=True)
ms = pop.allele_count_timed(out.data_ptr(), stream, 5, 1200)
print(f"K2 median {float(np.median(ms)):.3f} ms over {len(ms)} launches", flush=True)
print("K3 begins", flush=True)
edges = [0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.5, 1.0]
t = []
for _ in range(400):
    pop.count_by_genome_af_bins(edges); t.append(capi.count_by_genome_last_ms())
print(f"K3 median {float(np.median(t)):.3f} ms", flush=True)
PY
PID=$!
for i in $(seq 1 80); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Current Socket" | tr '\n' ' ' | sed 's/GPU\[0\]\t\t: //g'; echo
  sleep 0.25
  if ! kill -0 $PID 2>/dev/null; then break; fi
done
wait $PID
