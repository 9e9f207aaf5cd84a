"""Condense gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into profiles/<tag>_*: the rocprofv3 --stats table of the
default bench.py run, per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes, and the K5/K7 table passes with
their SQ counters.  Also refreshes profiles/traffic.json, which bench.py quotes ("traffic") when it does not measure.

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced >= 8-B/lane read stream, so the read side is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, re, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
dst.mkdir(exist_ok=True)


def one(pattern):
    files = glob.glob(str(src / pattern))
    if not files:
        sys.exit(f"missing {pattern} under {src}")
    return Path(max(files, key=lambda f: Path(f).stat().st_mtime))      # gpurun merges into what earlier calls left: the newest


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0].replace("kgx::", "")


stats_file = one("trace/*/*kernel_stats.csv")
(dst / f"{tag}_kernel_stats.csv").write_text(stats_file.read_text())
stats = {short(r["Name"]): r for r in csv.DictReader(stats_file.open())}


def counter(pattern, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(one(pattern).open()):
        if r["Counter_Name"] == name:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


fetch, write = counter("pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE"), counter("pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
bench = json.loads((src / "bench_trace.json").read_text().strip().splitlines()[-1])
aux = bench.get("aux", {})
kernels = [("K2", "k_allele_count", bench["roofline"], bench["config"]["workload"]),
           ("K3", "k_count_by_genome", aux.get("k3_fws_bins", {}).get("roofline"), aux.get("k3_fws_bins", {}).get("config", {}).get("workload")),
           ("K5", "k_inbreed_eval_lut<4", aux.get("c5_simple", {}).get("roofline"), aux.get("c5_simple", {}).get("config", {}).get("workload"))]
traffic_db_path = dst / "traffic.json"
traffic_db = json.loads(traffic_db_path.read_text()) if traffic_db_path.exists() else {}
lines = [f"# rocprofv3 summary `{tag}` — the default `bench.py` run (N=1)", "",
         f"Commands (on the GPU box, `scripts/profile_round.sh {tag}`):",
         "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`,",
         "then the same command (`--steps 3 --warmup 1`) once under `--pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE`.", "",
         "| sweep | kernel | calls | avg ms (rocprofv3) | ms (HIP events, bench.py) | algorithmic bytes | GB/s (rocprofv3 avg) | of 8 TB/s | HBM read = 2 x FETCH_SIZE x 1024 | HBM write = WRITE_SIZE x 1024 | traffic / algorithmic |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
pmc_rows = ["kernel,counter,launches,mean_value_KiB"]
for label, needle, roof, workload in kernels:
    names = [k for k in stats if needle in k]
    if not names or not roof:
        continue
    name = max(names, key=lambda k: float(stats[k]["TotalDurationNs"]))
    avg_ms = float(stats[name]["AverageNs"]) / 1e6
    alg = roof["algorithmic_bytes_per_launch"]
    f = [v for k, vs in fetch.items() if needle in k for v in vs]
    w = [v for k, vs in write.items() if needle in k for v in vs]
    # the dominant launches only (K3 runs small helper-sized launches of the same kernel on warm-up shapes)
    f = [v for v in f if v > 0.5 * max(f)] if f else f
    w = [v for v in w if v > 0.5 * max(w)] if w else w
    read_b = 2.0 * sum(f) / len(f) * 1024.0 if f else float("nan")
    write_b = sum(w) / len(w) * 1024.0 if w else float("nan")
    traffic = read_b + write_b
    pmc_rows += [f"{name},FETCH_SIZE,{len(f)},{sum(f) / len(f) if f else 'nan'}", f"{name},WRITE_SIZE,{len(w)},{sum(w) / len(w) if w else 'nan'}"]
    lines.append(f"| {label}: {workload} | `{name}` | {stats[name]['Calls']} | {avg_ms:.3f} | {roof['kernel_ms']:.3f} | {alg:,} | {alg / avg_ms / 1e6:,.0f} | "
                 f"{alg / avg_ms / 1e6 / 8000:.1%} | {read_b:,.0f} | {write_b:,.0f} | {traffic / alg:.3f} |")
    traffic_db[f"{label}:{workload}"] = {"kernel": name, "hbm_bytes_per_launch": traffic, "read_bytes": read_b, "write_bytes": write_b,
                                         "algorithmic_bytes_per_launch": alg, "profile": f"profiles/{tag}_pmc.csv",
                                         "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 wide-stream correction), write = WRITE_SIZE x 1024"}
# the moments' pass over the bytes in the same run (aux.c5_simple.hallme / .loglikelihood): k_class_bits reads the matrix once and
# leaves every class's hits as bit rows; (round 3's k_hall_sweep, if a run still holds it: its full classes alone)
hall_note = []
for needle, what in (("k_class_bits", "the one pass that leaves every class's hits as bit rows"), ("k_hall_sweep", "HallME's moment pass, the two classes that cover every locus")):
    hall_f = [v for k, vs in fetch.items() if needle in k for v in vs]
    hall_w = [v for k, vs in write.items() if needle in k for v in vs]
    if not (hall_f and hall_w):
        continue
    hall_f = [v for v in hall_f if v > 0.9 * max(hall_f)]
    hall_w = [v for v in hall_w if v > 0.9 * max(hall_w)]
    hall_name = max((k for k in stats if needle in k), key=lambda k: float(stats[k]["TotalDurationNs"]), default=needle)
    pmc_rows += [f"{hall_name},FETCH_SIZE,{len(hall_f)},{sum(hall_f) / len(hall_f)}", f"{hall_name},WRITE_SIZE,{len(hall_w)},{sum(hall_w) / len(hall_w)}"]
    c5 = aux.get("c5_simple", {}).get("roofline", {}).get("algorithmic_bytes_per_launch")
    if c5:
        hall_note += ["", f"`{hall_name}` ({what}): HBM read 2 x FETCH_SIZE x 1024 = {2.0 * sum(hall_f) / len(hall_f) * 1024.0:,.0f} B, "
                          f"write {sum(hall_w) / len(hall_w) * 1024.0:,.0f} B against {c5:,} algorithmic bytes of genotype rows: "
                          f"{(2.0 * sum(hall_f) / len(hall_f) + sum(hall_w) / len(hall_w)) * 1024.0 / c5:.3f} x; "
                          f"{float(stats[hall_name]['AverageNs']) / 1e6:.3f} ms a launch."]
(dst / f"{tag}_pmc.csv").write_text("\n".join(pmc_rows) + "\n")
traffic_db_path.write_text(json.dumps(traffic_db, indent=1) + "\n")
lines += hall_note
lines += ["", f"bench.py line of the traced run: value {bench['value']:.4g} {bench['unit']}, {bench['ms_per_step']:.3f} ms/step.",
          f"Full kernel table: `profiles/{tag}_kernel_stats.csv`; counter means: `profiles/{tag}_pmc.csv`.", ""]

# ---- the table passes (K5 frequency sweep, RitlandLocus, HallME, Loglikelihood) at C5
k7_stats = one("k7_trace/*/*kernel_stats.csv")
(dst / f"{tag}_k7_kernel_stats.csv").write_text(k7_stats.read_text())
k7 = {short(r["Name"]): r for r in csv.DictReader(k7_stats.open())}
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(one("k7_sq/*/*counter_collection.csv").open()):
    sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
cells = 10_000 * 5_000_000
lines += ["## The table passes at C5 (10 k genomes x 5 M loci, 50.12 GB algorithmic per pass)", "",
          "`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_inbreed.py 10000 5000000 --all`, then the same under",
          "`--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT`.", "",
          "| kernel | calls | avg ms | TB/s at 50.12 GB | VALU wave-instr / launch | VALU per cell (x64 lanes / 5e10 cells) | SALU / launch | LDS instr / launch | LDS bank-conflict cycles | wait-inst / wave cycles |",
          "|---|---|---|---|---|---|---|---|---|---|"]
for name in sorted(k7, key=lambda k: -float(k7[k]["TotalDurationNs"])):
    moment_pass = any(k in name for k in ("k_hall_sweep", "k_class_bits", "k_hall_mfma"))
    if "k_inbreed_eval_lut" not in name and "swar" not in name and not moment_pass:
        continue
    avg_ms = float(k7[name]["AverageNs"]) / 1e6
    c = {k: sum(v) / len(v) for k, v in sq.get(name, {}).items()}
    if moment_pass:
        # HallME's moment passes: the classes that cover every locus alone (the largest launches), not the average over all four
        full = {k: [x for x in v if x >= 0.9 * max(v)] for k, v in sq.get(name, {}).items() if v}
        c = {k: sum(v) / len(v) for k, v in full.items()}
        name_shown = name + " (full classes: counters; avg ms over all four classes)"
        valu = c.get("SQ_INSTS_VALU", float("nan"))
        lines.append(f"| `{name_shown}` | {k7[name]['Calls']} | {avg_ms:.3f} | - | {valu:.4g} | {valu * 64 / cells:.2f} | {c.get('SQ_INSTS_SALU', float('nan')):.4g} | "
                     f"{c.get('SQ_INSTS_LDS', float('nan')):.4g} | {c.get('SQ_LDS_BANK_CONFLICT', float('nan')):.4g} | "
                     f"{c.get('SQ_WAIT_INST_ANY', float('nan')) / c.get('SQ_WAVE_CYCLES', float('nan')):.2f} |")
        continue
    valu = c.get("SQ_INSTS_VALU", float("nan"))
    lines.append(f"| `{name}` | {k7[name]['Calls']} | {avg_ms:.3f} | {50.1208 / avg_ms:.2f} | {valu:.4g} | {valu * 64 / cells:.2f} | {c.get('SQ_INSTS_SALU', float('nan')):.4g} | "
                 f"{c.get('SQ_INSTS_LDS', float('nan')):.4g} | {c.get('SQ_LDS_BANK_CONFLICT', float('nan')):.4g} | "
                 f"{c.get('SQ_WAIT_INST_ANY', float('nan')) / c.get('SQ_WAVE_CYCLES', float('nan')):.2f} |")
lines += ["", "(Loglikelihood's passes after the first compactions sweep fewer genomes: its average is over all its launches.)", "",
          "`scripts/bench_inbreed.py` output of the traced run:", "", "```"] + (src / "k7.txt").read_text().strip().splitlines() + ["```", ""]
# ---- window-sized calls (what the INBREED package issues): the whole HallME / Loglikelihood iteration in one launch
window_files = glob.glob(str(src / "window_trace/*/*kernel_stats.csv"))
if window_files:
    window_stats = Path(max(window_files, key=lambda f: Path(f).stat().st_mtime))
    (dst / f"{tag}_window_kernel_stats.csv").write_text(window_stats.read_text())
    lines += ["## Window-sized calls: 1000 sampled loci x 512 genomes (`scripts/bench_inbreed_window.py 1000`)", "",
              "`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_inbreed_window.py 1000`: 100 timed calls per estimator after 200 warm-up calls.", "",
              "| kernel | calls | avg us | share of the traced GPU time |", "|---|---|---|---|"]
    for r in sorted(csv.DictReader(window_stats.open()), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    lines += ["", "```"] + (src / "window.txt").read_text().strip().splitlines() + ["```", ""]
# ---- HallME over a large call: per-genome moments (kgx_kernels_hall.h) against the 50 passes
hall_files = glob.glob(str(src / "hall_trace/*/*kernel_stats.csv"))
if hall_files:
    hall_stats = Path(max(hall_files, key=lambda f: Path(f).stat().st_mtime))
    (dst / f"{tag}_hall_kernel_stats.csv").write_text(hall_stats.read_text())
    lines += ["## HallME at C5 on per-genome moments (`scripts/bench_hall.py`: 4 calls by moments, then 4 by the 50 passes)", "",
              "`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_hall.py`; `k_class_bits` runs once a call (every class's hits as bit rows), "
              "`k_hall_mfma` once per class of homozygous cell (4 at C5: two over every locus, two over the loci that have a second / third alt), "
              "so its average is over unequal passes.", "",
              "| kernel | calls | avg ms |", "|---|---|---|"]
    for r in csv.DictReader(hall_stats.open()):
        if "k_hall" in r["Name"] or "k_class_bits" in r["Name"] or "rocprim" in r["Name"] or "eval_lut<1" in r["Name"]:
            lines.append(f"| `{short(r['Name'])[:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} |")
    lines += ["", "```"] + [l for l in (src / "hall.txt").read_text().strip().splitlines() if "amdgpu.ids" not in l] + ["```", ""]
# ---- Loglikelihood over a large call: the moments + the exact walk against the passes; traffic and SQ counters of its kernels
ll_files = glob.glob(str(src / "loglik_trace/*/*kernel_stats.csv"))
if ll_files:
    ll_stats = Path(max(ll_files, key=lambda f: Path(f).stat().st_mtime))
    (dst / f"{tag}_loglik_kernel_stats.csv").write_text(ll_stats.read_text())
    lines += ["## Loglikelihood at C5 on per-genome moments (`scripts/bench_loglik.py`: 4 calls by moments, then 4 by the passes)", "",
              "`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_loglik.py`; `k_class_bits` once a call, `k_hall_mfma<true, true>` once per class of "
              "homozygous cell -- the moments, and the hits' words of the bins a band can reach; `k_loglik_search` is the whole search of every genome.", "",
              "| kernel | calls | avg ms |", "|---|---|---|"]
    for r in csv.DictReader(ll_stats.open()):
        if any(k in r["Name"] for k in ("k_hall", "k_class_bits", "k_loglik", "rocprim", "eval_lut<2", "eval_lut<3", "k_eval_entries<5", "k_gather_columns")):
            lines.append(f"| `{short(r['Name'])[:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} |")
    lines += ["", "```"] + [l for l in (src / "loglik.txt").read_text().strip().splitlines() if "amdgpu.ids" not in l] + ["```", ""]
    try:
        ll_fetch, ll_write = counter("loglik_fetch/*/*counter_collection.csv", "FETCH_SIZE"), counter("loglik_write/*/*counter_collection.csv", "WRITE_SIZE")
        ll_sq = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(one("loglik_sq/*/*counter_collection.csv").open()):
            ll_sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines += ["Counters of the same calls (`--moments-only`; per launch, means; the class passes: the two that cover every locus alone):", "",
                  "| kernel | HBM read = 2 x FETCH_SIZE x 1024 | HBM write = WRITE_SIZE x 1024 | VALU wave-instr | VALU per cell (x64 / 5e10) | MFMA instr | LDS instr | wait-inst / wave cycles |",
                  "|---|---|---|---|---|---|---|---|"]
        rows_pmc = []
        for needle, full in (("k_class_bits", False), ("k_hall_mfma<true, true>", True), ("k_hall_sweep<8, true>", True), ("k_loglik_search", False), ("k_inbreed_eval_lut<3", False)):
            name = next((k for k in ll_sq if needle in k), None)
            if not name:
                continue
            pick = (lambda v: [x for x in v if x >= 0.9 * max(v)]) if full else (lambda v: v)
            f = pick([v for k, vs in ll_fetch.items() if needle in k for v in vs])
            w = pick([v for k, vs in ll_write.items() if needle in k for v in vs])
            c = {k: (lambda v: sum(v) / len(v))(pick(v)) for k, v in ll_sq[name].items() if v}
            valu = c.get("SQ_INSTS_VALU", float("nan"))
            lines.append(f"| `{name}` | {2.0 * sum(f) / len(f) * 1024.0:,.0f} | {sum(w) / len(w) * 1024.0:,.0f} | {valu:.4g} | {valu * 64 / cells:.2f} | "
                         f"{c.get('SQ_INSTS_MFMA', float('nan')):.4g} | {c.get('SQ_INSTS_LDS', float('nan')):.4g} | {c.get('SQ_WAIT_INST_ANY', float('nan')) / c.get('SQ_WAVE_CYCLES', float('nan')):.2f} |")
            rows_pmc += [f"{name},FETCH_SIZE,{len(f)},{sum(f) / len(f)}", f"{name},WRITE_SIZE,{len(w)},{sum(w) / len(w)}"]
        with (dst / f"{tag}_pmc.csv").open("a") as fh:
            fh.write("\n".join(rows_pmc) + "\n")
        lines += [""]
    except SystemExit:
        lines += ["(counter passes of the Loglikelihood calls: not collected in this run)", ""]
# ---- the batched window regime
batch_files = glob.glob(str(src / "batch_trace/*/*kernel_stats.csv"))
if batch_files:
    batch_stats = Path(max(batch_files, key=lambda f: Path(f).stat().st_mtime))
    (dst / f"{tag}_batch_kernel_stats.csv").write_text(batch_stats.read_text())
    lines += ["## Batched windows: 16 windows x 5 super populations per `kgx_inbreed_batch` (`scripts/bench_inbreed_batch.py`)", "",
              "| kernel | calls | avg us | share of the traced GPU time |", "|---|---|---|---|"]
    for r in sorted(csv.DictReader(batch_stats.open()), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        lines.append(f"| `{short(r['Name'])[:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} % |")
    lines += ["", "```"] + [l for l in (src / "batch.txt").read_text().strip().splitlines() if "amdgpu.ids" not in l] + ["```", ""]
(dst / f"{tag}_summary.md").write_text("\n".join(lines))
print("\n".join(lines))
