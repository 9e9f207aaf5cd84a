"""Write the binary record file read by kgl_gene_amd/lib/kgx_host_driver (format documented in
kgl_gene_amd/csrc/host/kgx_host_driver.cpp) and run the driver."""
from __future__ import annotations

import struct
import subprocess
from pathlib import Path

import numpy as np

from . import oracle_api as oa

ROOT = Path(__file__).resolve().parent.parent
DRIVER = ROOT / "kgl_gene_amd" / "lib" / "kgx_host_driver"

# DataSourceEnum order of kgl_parser/kgl_data_file_type.h:32-44
DATA_SOURCE = {"Genome1000": 0, "GnomadGenome3_1": 1, "Falciparum": 2, "GnomadExomes3_1": 3, "GnomadExomes2_1": 4,
               "Gnomad3_1": 5, "Gnomad3_0": 6, "Gnomad2_1": 7}
# INFO field per super population (AFR, AMR, EAS, EUR, SAS, ALL), kgl_variant_db_freq.h:84-96
AF_FIELDS = {
    "Gnomad2_1": ["AF_afr", "AF_amr", "AF_eas", "AF_nfe", None, "AF"],       # SAS shares "AF" with ALL in the reference table
    "Genome1000": ["AFR_AF", "AMR_AF", "EAS_AF", "EUR_AF", "SAS_AF", "AF"],
    "Falciparum": [None, None, None, None, None, "AF"],
}


def _s(b: bytes | str) -> bytes:
    if isinstance(b, str):
        b = b.encode()
    return struct.pack("<I", len(b)) + b


def write_records(path, rec: oa.Records, gt, genome_ids, mode, data_source, population_id="population", ped=None):
    out = bytearray()
    out += b"KGXR" + struct.pack("<III", 1, mode, DATA_SOURCE[data_source])
    out += _s(population_id) + _s(rec.contig)
    out += struct.pack("<Q", len(genome_ids))
    for g in genome_ids:
        out += _s(g)
    out += struct.pack("<Q", rec.n_records)
    fields = AF_FIELDS[data_source]
    for r in range(rec.n_records):
        out += struct.pack("<Q", int(rec.offsets[r])) + _s(rec.refs[r]) + struct.pack("<B", len(rec.alts[r]))
        for a in rec.alts[r]:
            out += _s(a)
        out += struct.pack("<B", 1 if rec.passed is None else int(rec.passed[r]))
        if rec.af is None:
            out += struct.pack("<B", 0)
        else:
            af = np.asarray(rec.af[r], dtype=np.float32).reshape(-1, 6)      # [n_alt][6]
            present = [(name, sp) for sp, name in enumerate(fields) if name is not None]
            out += struct.pack("<B", len(present))
            for name, sp in present:
                out += _s(name) + struct.pack("<I", af.shape[0]) + af[:, sp].astype("<f4").tobytes()
    if mode != oa.Population.REFERENCE:
        g = np.ascontiguousarray(gt, dtype=np.uint8)
        assert g.shape == (rec.n_records, len(genome_ids), 2)
        out += g.tobytes()
    if ped is not None:
        out += struct.pack("<Q", len(ped))
        for genome, sp in ped:
            out += _s(genome) + _s(sp)
    Path(path).write_bytes(bytes(out))


def run_driver(ident, work_dir, files, **params):
    if not DRIVER.exists():
        from kgl_gene_amd import build as kbuild

        kbuild.build_host()
    args = [str(DRIVER), ident, str(work_dir)] + [f"{k}={v}" for k, v in params.items()] + ["quiet=1", "--"] + [str(f) for f in files]
    return subprocess.run(args, capture_output=True, text=True)


def read_csv(path):
    lines = Path(path).read_text().strip().split("\n")
    return lines[0].split(","), [ln.split(",") for ln in lines[1:]]
