#!/bin/bash
# SQ counters of the by-genome sweep (K3) inside the default bench run.   gpurun -- 'bash scripts/prof_k3_sq.sh'
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_k3_sq
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/sq -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_k3_sq"
f = glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "count_by_genome" in k or "allele_count" in k:
        print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in c.items()}, len(next(iter(c.values()))))
PY
