"""End-to-end GPU_INBREED from compressed VCF files through the package (C++ host code + the GPU), stage by stage on
the host side, then the whole package run; and the oracle's parse-to-objects + window loop on a slice for scale.
    python tests/tools/bench_inbreed_vcf.py [samples] [records]           (defaults 2504 x 40000: one 1000-Genomes-like chunk)"""
import os, subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import host_api as ha, oracle_api as oa, records_io as rio, synth_vcf as sv, vcf_text as vt

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2504
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
rec, gt = sv.multiallelic_block(G, L, rng_seed=1, dup_records=0)
for a in rec.af:
    a[:, 4] = a[:, 5]
ids = [f"NA{i:05d}" for i in range(G)]
pops = ["AFR", "AMR", "EAS", "EUR", "SAS"]
with tempfile.TemporaryDirectory() as tmp:
    tmp = Path(tmp)
    ref_text, dip_text = vt.write_vcf_mono(rec, "Gnomad2_1"), vt.write_vcf_1000(rec, gt, ids, quirks=False)
    (tmp / "gnomad.vcf.bgz").write_bytes(vt.bgzip(ref_text.encode()))
    t0 = time.perf_counter(); packed = vt.bgzip(dip_text.encode()); t_pack = time.perf_counter() - t0
    (tmp / "kg.vcf.bgz").write_bytes(packed)
    (tmp / "ped.txt").write_text("".join(f"{g}\t{pops[i % 5]}\n" for i, g in enumerate(ids)))
    print(f"{G} samples x {L} records: population VCF {len(dip_text)/1e6:.0f} MB text, {len(packed)/1e6:.0f} MB block gzip; reference {len(ref_text)/1e6:.1f} MB text", flush=True)
    t0 = time.perf_counter(); text = ha.read_vcf_text(tmp / "kg.vcf.bgz"); t_read = time.perf_counter() - t0
    assert text == dip_text.encode()
    print(f"readVcfText (parallel bgzf inflate + CRC): {t_read:.2f} s = {len(text)/t_read/1e6:.0f} MB/s of text", flush=True)
    t0 = time.perf_counter(); inputs = ha.InbreedInputs(ref_text, rio.DATA_SOURCE['Gnomad2_1'], dip_text); t_flat = time.perf_counter() - t0
    print(f"flattenReferenceVcf + flattenVcf1000Gt8 (+ copy-out): {t_flat:.2f} s = {len(dip_text)/t_flat/1e6:.0f} MB/s, {G*L/t_flat:.3g} cells/s; "
          f"{inputs.L} reference loci x {inputs.G} genomes", flush=True)
    for algorithm in ("Simple", "Loglikelihood"):
        params = dict(AnalysisType="false", OutputFile="inbreed", Algorithm=algorithm, MinAlleleFreq=0.02, MaxAlleleFreq=0.9,
                      LowerWindow=0, UpperWindow=int(rec.offsets[-1]) + 1, LociiCount=2000, SamplingDistance=int(max(1, (rec.offsets[-1] // L))))
        t0 = time.perf_counter()
        res = rio.run_driver("GPU_INBREED", tmp, [f"vcf:Gnomad2_1:{tmp / 'gnomad.vcf.bgz'}", f"vcf:Genome1000:{tmp / 'kg.vcf.bgz'}", f"ped:{tmp / 'ped.txt'}"], **params)
        t_pkg = time.perf_counter() - t0
        assert res.returncode == 0, res.stderr
        n_cols = len((tmp / "inbreed.csv").read_text().split("\n")[1].split(",")) - 2
        print(f"GPU_INBREED package, {algorithm}, from the two .bgz files to the CSVs: {t_pkg:.2f} s ({n_cols} windows)", flush=True)
    # the oracle on a slice, for scale (its parse builds one Variant object per carried allele)
    Ls = min(L, 2000)
    rec_s, gt_s = sv.multiallelic_block(G, Ls, rng_seed=1, dup_records=0)
    text_s = vt.write_vcf_1000(rec_s, gt_s, ids, quirks=False)
    t0 = time.perf_counter(); o = oa.Population("x"); o.add_vcf_1000(text_s); t_or = time.perf_counter() - t0
    print(f"oracle (reference-style) parse of a {Ls}-record slice to Variant objects: {t_or:.2f} s = {len(text_s)/t_or/1e6:.1f} MB/s "
          f"-> {t_or * L / Ls:.0f} s for the whole file", flush=True)
