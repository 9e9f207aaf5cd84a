"""Condense gpurun_out/prof_<tag>_aux/ (rocprofv3 output of scripts/profile_aux.sh) into profiles/<tag>_k3_kernel_stats.csv,
profiles/<tag>_k5_kernel_stats.csv and profiles/<tag>_aux_summary.md.  Algorithmic bytes: K3 reads V*ceil(G/4) bytes of
dosage2 per pass; K5 reads L*G bytes of gt8 plus the per-locus table (kgx_gt8_sweep_bytes)."""
import csv, glob, re, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{tag}_aux"
dst = ROOT / "profiles"

def newest(files):
    """gpurun merges every run's files into the same local directory: keep the latest run's."""
    import os
    return sorted(files, key=os.path.getmtime, reverse=True)


def stats(sub):
    files = newest(glob.glob(str(src / sub / "*" / "*kernel_stats.csv")))
    if not files:
        sys.exit(f"missing kernel_stats.csv under {src / sub}")
    text = Path(files[0]).read_text()
    (dst / f"{tag}_{sub}_kernel_stats.csv").write_text(text)
    return list(csv.DictReader(text.splitlines()))

def calls(sub, needle):
    """Per-call durations (ms) of the kernels whose name contains needle, in launch order."""
    files = newest(glob.glob(str(src / sub / "*" / "*kernel_trace.csv")))
    rows = [r for r in csv.DictReader(open(files[0])) if needle in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]

def table(rows, alg_bytes, main):
    out = ["| kernel | calls | avg ms | algorithmic GB/s |", "|---|---|---|---|"]
    for r in rows:
        ms = float(r["AverageNs"]) / 1e6
        gbs = f"{alg_bytes / ms / 1e6:,.0f}" if any(m in r["Name"] for m in main) else ""
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {ms:.3f} | {gbs} |")
    return "\n".join(out)

G3, V3 = 10_000, 10_000_000
k3_bytes = V3 * ((G3 + 3) // 4)
k5_txt = (src / "k5.txt").read_text()
m = re.search(r"algorithmic bytes per frequency sweep: ([0-9.]+) GB", k5_txt)
k5_bytes = float(m.group(1)) * 1e9
md = f"""# rocprofv3 kernel stats `{tag}` — secondary sweeps (scripts/profile_aux.sh, scripts/summarize_aux.py)

## K3 by-genome sweep at C3 (10k x 10M; 3 x all rows + 2 x 11 FWS bins)

{table(stats("k3"), k3_bytes, ["k_count_by_genome"])}

Per call, in launch order (3 x all rows in one bin, then 2 x 11 FWS bins): {", ".join(f"{ms:.2f} ms = {k3_bytes / ms / 1e6:,.0f} GB/s" for ms in calls("k3", "k_count_by_genome"))}.

```
{(src / "k3.txt").read_text().strip()}
```

## K5 inbreeding sweep at C5 (10k x 5M multi-allelic; Simple then RitlandLocus, each called twice)

{table(stats("k5"), k5_bytes, ["k_inbreed_sweep", "k_inbreed_eval_lut"])}

```
{k5_txt.strip()}
```
"""
# optional: the iterative estimators (HallME, Loglikelihood) at C5, each called twice
if (src / "k7.txt").exists() and glob.glob(str(src / "k7" / "*" / "*kernel_stats.csv")):
    k7_rows = stats("k7")
    md += f"""
## K7 iterative estimators at C5 (10k x 5M multi-allelic; HallME then Loglikelihood, each called twice)

Every `k_inbreed_eval_lut<1|2, ...>` launch is one pass over the genotype bytes of the genomes still iterating (all 10k for
HallME; Loglikelihood compacts the columns of the genomes still searching, so its later launches are shorter and the
GB/s column, which prices every launch at the full {k5_bytes / 1e9:.1f} GB, understates them).

{table(k7_rows, k5_bytes, ["k_inbreed_sweep", "k_inbreed_eval_lut"])}

Loglikelihood passes in launch order (ms): {", ".join(f"{ms:.1f}" for ms in calls("k7", "k_inbreed_eval_lut<2"))}

```
{(src / "k7.txt").read_text().strip()}
```
"""

# optional: HBM fetch traffic of the K5 kernels (scripts/pmc_k5.sh "FETCH_SIZE" fetch), corrected as the K2 profile is
pmc = glob.glob(str(ROOT / "gpurun_out" / "pmc_k5_fetch" / "*" / "*counter_collection.csv"))
if pmc:
    import collections
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(pmc[0])):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_inbreed" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    (dst / f"{tag}_k5_pmc.csv").write_text("kernel,launches,fetch_size_kib_avg\n" + "".join(f"{k},{len(v)},{sum(v) / len(v)}\n" for k, v in agg.items()))
    md += "\n## K5 HBM read traffic (separate `rocprofv3 --pmc FETCH_SIZE` pass; read = 2 x FETCH_SIZE x 1024, the gfx950 wide-stream correction)\n\n"
    md += "| kernel | launches | HBM read per launch | x algorithmic |\n|---|---|---|---|\n"
    for k, v in agg.items():
        read = 2.0 * 1024.0 * sum(v) / len(v)
        md += f"| `{k}` | {len(v)} | {read / 1e9:.2f} GB | {read / k5_bytes:.3f} |\n"
(dst / f"{tag}_aux_summary.md").write_text(md)
print(md)
