"""Condense gpurun_out/prof_<tag>/ (rocprofv3 output of scripts/profile_k2.sh) into profiles/<tag>_*.

Writes: profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table), profiles/<tag>_k2_pmc.csv
(K2 rows of the two PMC passes), profiles/<tag>_summary.md, and updates profiles/k2_traffic.json.
HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced 16-B/lane stream, so the read side is doubled; WRITE_SIZE is
exact for the 16-B/lane result stores.
"""
import csv, glob, json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)

def one(pattern):
    files = glob.glob(str(src / pattern))
    if not files:
        sys.exit(f"missing {pattern} under {src}")
    return Path(files[0])

stats = one("trace/*/*kernel_stats.csv")
(dst / f"{tag}_kernel_stats.csv").write_text(stats.read_text())
rows = list(csv.DictReader(stats.open()))
k2 = next(r for r in rows if "k_allele_count" in r["Name"])

def pmc(pattern, counter):
    vals = []
    for r in csv.DictReader(one(pattern).open()):
        if "k_allele_count" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size"], r["VGPR_Count"]))
    return vals

fetch = pmc("pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
write = pmc("pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
with (dst / f"{tag}_k2_pmc.csv").open("w") as f:
    f.write("counter,value_KiB,duration_ns,grid,vgpr\n")
    for v in fetch: f.write(f"FETCH_SIZE,{v[0]},{v[1]},{v[2]},{v[3]}\n")
    for v in write: f.write(f"WRITE_SIZE,{v[0]},{v[1]},{v[2]},{v[3]}\n")

bench = json.loads((src / "bench_trace.json").read_text().strip().splitlines()[-1])
G, V = bench["config"]["genomes_per_gpu"], bench["config"]["variants"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
fetch_kib = sum(v[0] for v in fetch) / len(fetch)
write_kib = sum(v[0] for v in write) / len(write)
read_bytes = 2.0 * fetch_kib * 1024.0
write_bytes = write_kib * 1024.0
traffic = read_bytes + write_bytes
avg_ms = float(k2["AverageNs"]) / 1e6

tj = dst / "k2_traffic.json"
db = json.loads(tj.read_text()) if tj.exists() else {}
db[f"{G}x{V}"] = {"hbm_bytes_per_launch": traffic, "read_bytes": read_bytes, "write_bytes": write_bytes,
                  "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib, "profile": f"profiles/{tag}_k2_pmc.csv",
                  "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 wide-stream correction), write = WRITE_SIZE x 1024"}
tj.write_text(json.dumps(db, indent=1) + "\n")

md = f"""# rocprofv3 summary `{tag}` — bench.py N=1, workload {bench['config']['workload']}

Command (on the GPU box, `scripts/profile_k2.sh {tag}`):
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`
plus one `--pmc FETCH_SIZE` and one `--pmc WRITE_SIZE` pass (separate runs).

| kernel | calls | avg ms (rocprofv3) | avg ms (HIP events, bench.py) |
|---|---|---|---|
| `k_allele_count<64,8>` (K2) | {k2['Calls']} | {avg_ms:.3f} | {bench['roofline']['kernel_ms']:.3f} |

| quantity | bytes per launch |
|---|---|
| algorithmic (V*ceil(G/4) + 16*V) | {alg:,} |
| HBM read  = 2 x FETCH_SIZE x 1024 | {read_bytes:,.0f} |
| HBM write = WRITE_SIZE x 1024 | {write_bytes:,.0f} |
| HBM traffic | {traffic:,.0f} ({traffic/alg:.3f} x algorithmic) |

Achieved (algorithmic bytes / rocprofv3 average duration): **{alg/avg_ms/1e6:,.0f} GB/s = {alg/avg_ms/1e6/8000:.1%} of the 8 TB/s HBM3E peak**.
Row pitch is {-(-((G + 3)//4)//128)*128 if (G+3)//4 > 512 else -(-((G+3)//4)//16)*16} B for {(G+3)//4} B of genotypes per row (128-B aligned rows), so traffic exceeds the
algorithmic figure by the padding.

Full kernel table: `profiles/{tag}_kernel_stats.csv`; raw counter rows: `profiles/{tag}_k2_pmc.csv`.
bench.py line of the traced run: value {bench['value']:.4g} {bench['unit']}, {bench['ms_per_step']:.3f} ms/step.
"""
(dst / f"{tag}_summary.md").write_text(md)
print(md)
