"""GPU_ALLELE's file entry: a block-gzip 1000-Genomes-style VCF flattened a bounded piece at a time against read-it-all-then-flatten,
wall time and peak resident memory (each in its own process).   python scripts/bench_flatten_file.py [samples] [records]"""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from tests import host_api as ha
    mode, path, chunk = sys.argv[2], sys.argv[3], int(sys.argv[4])
    t = time.perf_counter()
    if mode == "stream":                                  # rows handed to a sink piece by piece (GpuAlleleAnalysis: to the device)
        import ctypes as C
        genomes, written = C.c_uint64(0), C.c_uint64(0)
        ha.lib().kgxh_stream_flatten_count.restype = C.c_int64
        ha.lib().kgxh_stream_flatten_count.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        rows = ha.lib().kgxh_stream_flatten_count(path.encode(), 0, 0, chunk, C.byref(genomes), C.byref(written))
        dt = time.perf_counter() - t
        hwm = next(int(line.split()[1]) for line in open("/proc/self/status") if line.startswith("VmHWM"))
        print(f"{mode:7s} chunk {chunk >> 20:5d} MiB: {dt:6.2f} s, peak RSS {hwm / 1e6:6.2f} GB, {genomes.value} genomes x {rows} variants, "
              f"{written.value / 1e6:.0f} MB of rows handed to the sink")
        sys.exit(0)
    if mode == "pieces":
        flat = ha.lib().kgxh_flatten_vcf_file(path.encode(), 0, 0, 0, chunk, None, 0)
    else:
        import ctypes as C
        n = C.c_uint64(0)
        text = ha.lib().kgxh_read_vcf_text(path.encode(), C.byref(n), 0, None, 0)
        flat = ha.lib().kgxh_flatten_vcf1000(C.cast(text, C.c_char_p), n.value, 0)
        ha.lib().kgxh_free(text)
    dt = time.perf_counter() - t
    hwm = next(int(line.split()[1]) for line in open("/proc/self/status") if line.startswith("VmHWM"))     # kB; ru_maxrss survives exec, this does not
    print(f"{mode:7s} chunk {chunk >> 20:5d} MiB: {dt:6.2f} s, peak RSS {hwm / 1e6:6.2f} GB, "
          f"{ha.lib().kgxh_flat_genomes(flat)} genomes x {ha.lib().kgxh_flat_variants(flat)} variants")
    sys.exit(0)

from tests import synth_vcf as sv, vcf_text as vt   # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2504
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
rec, gt = sv.multiallelic_block(G, L, rng_seed=1, dup_records=0)
text = vt.write_vcf_1000(rec, gt, [f"NA{i:05d}" for i in range(G)], quirks=False).encode()
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "population.vcf.bgz")
    Path(path).write_bytes(vt.bgzip(text, level=1))
    print(f"{G} samples x {L} records: {len(text) / 1e6:.0f} MB of text, {os.path.getsize(path) / 1e6:.0f} MB block gzip; packed genotypes {L * 1.4 * G / 4 / 1e6:.0f} MB")
    del text, rec, gt
    for mode, chunk in (("whole", 0), ("pieces", 256 << 20), ("pieces", 64 << 20), ("pieces", 16 << 20), ("stream", 64 << 20), ("stream", 16 << 20)):
        subprocess.run([sys.executable, __file__, "--child", mode, path, str(chunk)], check=True)
