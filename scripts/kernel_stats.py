"""Per-kernel launch count and average / total duration from a rocprofv3 --kernel-trace --stats csv directory."""
import collections, csv, glob, re, sys
files = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:60s} calls {len(v):4d}  avg {sum(v) / len(v):9.3f} ms  total {sum(v):10.2f} ms")
