#!/bin/bash
# PMC counters for the by-genome sweeps at C3: gpurun -- 'bash scripts/pmc_k3.sh "<counters>" <tag>'
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_k3_$2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/scripts/bench_by_genome.py > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*counter_collection.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_count_by_genome" in r["Kernel_Name"]:
        print(r["Counter_Name"], r["Counter_Value"])
PY
