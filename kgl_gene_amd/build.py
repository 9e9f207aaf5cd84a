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
    "-std=c++17",
    "-ffp-contract=off",  # host and device fp64 must agree bit for bit (synthetic generator, K6)
    "-fPIC",
    "-shared",
    "-Wall",
    "-Wno-unused-function",
]


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _deps() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [ROOT / "include" / "kgx.h"])


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build_kgx(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP translation unit into kgl_gene_amd/lib/libkgx.so."""
    LIBDIR.mkdir(parents=True, exist_ok=True)
    if not force and not _stale(LIBKGX, _deps()):
        return LIBKGX
    extra = os.environ.get("KGX_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DKGX_EXP_...); the default build has none
    cmd = [HIPCC, *HIP_FLAGS, *extra, "-I", str(ROOT / "include"), "-o", str(LIBKGX), *map(str, _sources())]
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
