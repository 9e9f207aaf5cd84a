#!/bin/bash
# rocprofv3 --pmc <counters> --kernel-trace of one python command (on the GPU box, from the repo root): pmc_cmd.sh <tag> "<counters>" <script> [args...]
tag=$1; shift
counters=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $counters --kernel-trace -d $OUT -o t --output-format csv -- python3 "$@" > $OUT/run.log 2>&1
grep -v "rocprofv3\|amdgpu.ids" $OUT/run.log | tail -5
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
for k, v in acc.items():
    if "hall" in k or "loglik" in k or "eval" in k:
        print(k, {c: f"{x / calls[(k, c)]:.4g}" for c, x in v.items()}, "calls", max(calls[(k, c)] for c in v))
PY
