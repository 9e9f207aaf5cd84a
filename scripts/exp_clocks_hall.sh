# Engine clock and socket power while HallME's moment passes run at C5 (rocm-smi sampled beside the run).
python - <<'PY' &
import sys, time
sys.path.insert(0, ".")
import numpy as np
from kgl_gene_amd import capi
capi.init(0)
m = capi.GenotypeMatrix(10_000, 5_000_000)
table = m.synth_multiallelic(1111, 0, 0)
start = capi.reference_starts("HallME", 4242, 10_000)
time.sleep(1.0)
print("HallME by moments begins", flush=True)
t0 = time.perf_counter()
for _ in range(120):
    m.inbreed(table, "HallME", phased=True, start=start)
print(f"HallME by moments: {(time.perf_counter() - t0) / 120 * 1e3:.1f} ms per call", flush=True)
PY
PID=$!
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Current Socket" | tr '\n' ' ' | sed 's/GPU\[0\]\t\t: //g'; echo
  sleep 0.25
  if ! kill -0 $PID 2>/dev/null; then break; fi
done
wait $PID
