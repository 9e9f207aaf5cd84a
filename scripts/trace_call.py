"""The kernels of the LAST call of a rocprofv3 kernel trace, in time order: python scripts/trace_call.py <tag> <last kernel> [first kernel]"""
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]][-1]
first_name = sys.argv[3] if len(sys.argv) > 3 else "k_locus_tables"
first = [i for i, r in enumerate(rows[:last]) if first_name in r["Kernel_Name"]][-1]
t0 = int(rows[first]["Start_Timestamp"])
busy_until = t0
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-44:]
    if (e - s) > 30000 or "hall" in name or "class" in name:
        print(f"{(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f}  {(e - s) / 1e6:7.3f}  {name}")
    busy_until = max(busy_until, e)
print("total", (busy_until - t0) / 1e6)
