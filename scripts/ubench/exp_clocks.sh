# (for the -DKGX_EXP_* builds apply scripts/ubench/power_cap_experiment.patch first: the product's kernels do not carry those branches)
# Engine clock and power while the table passes run (HallME at C5: 50 passes a call), sampled by rocm-smi beside the run.
# usage: bash scripts/exp_clocks.sh [extra hipcc flags, e.g. -DKGX_EXP_NOLOAD]
set -e
if [ -n "$1" ]; then KGX_HIPCC_FLAGS="$1" python -m kgl_gene_amd.build > /dev/null 2>&1; fi
python - <<'PY' &
import sys, time
sys.path.insert(0, ".")
from kgl_gene_amd import capi
capi.init(0)
m = capi.GenotypeMatrix(10000, 5000000)
table = m.synth_multiallelic(1111, 0, 0)
time.sleep(2.0)
print("run begins", time.time(), flush=True)
for i in range(8):
    t0 = time.perf_counter(); m.inbreed(table, "HallME", phased=True); print(f"HallME {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
print("run ends", time.time(), flush=True)
PY
PID=$!
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo " t=$(date +%s.%N)"
  sleep 0.25
  if ! kill -0 $PID 2>/dev/null; then break; fi
done
wait $PID
if [ -n "$1" ]; then python -m kgl_gene_amd.build --force > /dev/null 2>&1; fi
