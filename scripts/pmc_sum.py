"""Average PMC counter values per kernel from a rocprofv3 --pmc run directory: python scripts/pmc_sum.py <dir> [needle]"""
import collections, csv, glob, re, sys
files = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
needle = sys.argv[2] if len(sys.argv) > 2 else "k_inbreed"
agg = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    if needle in r["Kernel_Name"]:
        agg[(re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:45s} {c:28s} launches {len(v):4d}  avg {sum(v) / len(v):.4g}")
