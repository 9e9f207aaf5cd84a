"""Build the in-tree native libraries.

`libkgx.so` (kgl_gene_amd/lib/) is the product: hand-written HIP for gfx950 behind the C ABI of
include/kgx.h.  hipcc cross-compiles without a GPU, so this runs in the CPU-only build container
and the resulting .so travels to the GPU box with the tree.

The oracle (oracle/, test infrastructure only) is built by its own Makefile; `build_oracle()` just
drives it.  Building the checker is not using it: nothing in this package links or loads it.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "kgl_gene_amd" / "csrc"
LIBDIR = ROOT / "kgl_gene_amd" / "lib"
LIBKGX = LIBDIR / "libkgx.so"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++20",
    "-ffp-contract=off",  # host and device fp64 must agree bit for bit (synthetic generator, K6)
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    # k_hall_mfma: accumulators in VGPRs, updated in place.  Left to choose, the register allocator puts them in AGPRs and copies
    # every one through a[0:3] around its MFMA (213 instead of 121 instructions per block, 156 instead of 118 registers).
    "-mllvm", "-amdgpu-mfma-vgpr-form",
]
OBJDIR = LIBDIR / "obj"


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _headers() -> list[Path]:
    return sorted(CSRC.glob("*.h")) + [ROOT / "include" / "kgx.h"]


def _includes(src: Path) -> list[Path]:
    """The in-tree headers one translation unit pulls in (transitively), so that editing the inbreeding kernels does
    not recompile the dosage side."""
    seen, todo = {}, [src]
    while todo:
        f = todo.pop()
        for line in f.read_text().splitlines():
            line = line.strip()
            if not line.startswith('#include "'):
                continue
            name = line.split('"')[1]
            h = (f.parent / name).resolve()
            if h.exists() and h not in seen:
                seen[h] = True
                todo.append(h)
    return list(seen)


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build_kgx(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP translation unit (in parallel, one hipcc per unit) and link kgl_gene_amd/lib/libkgx.so."""
    LIBDIR.mkdir(parents=True, exist_ok=True)
    OBJDIR.mkdir(parents=True, exist_ok=True)
    extra = os.environ.get("KGX_HIPCC_FLAGS", "").split()   # experiments only (e.g. scripts/ubench/power_cap_experiment.patch + -DKGX_EXP_NOLOAD); the default build has none
    jobs, objects = [], []
    for src in _sources():
        obj = OBJDIR / (src.stem + ".o")
        objects.append(obj)
        if force or extra or _stale(obj, [src] + _includes(src)):
            cmd = [HIPCC, *HIP_FLAGS, *extra, "-I", str(ROOT / "include"), "-c", "-o", str(obj), str(src)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((cmd, subprocess.Popen(cmd, cwd=str(ROOT))))
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if jobs or not LIBKGX.exists() or _stale(LIBKGX, objects):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIBKGX), *map(str, objects), "-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=str(ROOT))
    return LIBKGX


def build_host(force: bool = False, verbose: bool = False) -> None:
    """Build the C++ host layer (VirtualAnalysis mirror + analysis packages), if present."""
    mk = CSRC / "host" / "Makefile"
    if mk.exists():
        subprocess.run(["make", "-s", "-C", str(mk.parent)] + (["-B"] if force else []), check=True)


def build_oracle(force: bool = False) -> None:
    mk = ROOT / "oracle" / "Makefile"
    if mk.exists():
        subprocess.run(["make", "-s", "-C", str(mk.parent)] + (["-B"] if force else []), check=True)


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_kgx(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)
    build_oracle(force=force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print(LIBKGX)
