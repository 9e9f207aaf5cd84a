"""kgl_gene_amd — MI355X-native population allele-count / inbreeding sweep (KGL_Gene hot path).

The product is the HIP library behind include/kgx.h (kgl_gene_amd/lib/libkgx.so) and the C++
analysis packages in kgl_gene_amd/csrc/host/ that mirror the reference's VirtualAnalysis surface.
This Python package is the thin binding used by tests and bench.py; PyTorch appears only as
plumbing (torch.distributed / RCCL) in bench.py.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
