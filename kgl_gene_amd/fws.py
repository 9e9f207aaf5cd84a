"""Host-side logic of the FWS allele-frequency bins (CalcFWS, kga_analytic/kga_PfEMP/kga_analysis_PfEMP_FWS.cpp).

The C++ analysis package (csrc/host) carries the same logic; this is the Python mirror used by tests
and bench.py.  No counting happens here: the sweeps run on the GPU through the C ABI.
"""
from __future__ import annotations

import numpy as np

# AlleleFrequencyBins / CalcFWS::getFrequency (kga_analysis_PfEMP_FWS.cpp:104-145): 11 half-open ranges.
FWS_BINS = [(0.0, 0.05), (0.05, 0.10), (0.10, 0.15), (0.15, 0.20), (0.20, 0.25), (0.25, 0.30),
            (0.30, 0.35), (0.35, 0.40), (0.40, 0.45), (0.45, 0.5), (0.5, 1.0)]
NO_BIN = 0xFF


def fws_bin_of_variant(af32: np.ndarray, carried: np.ndarray | None = None) -> np.ndarray:
    """Bin index per variant row, NO_BIN if the variant is in no bin.

    A variant is in bin [lo,hi) iff P7FrequencyFilter(lo) and not P7FrequencyFilter(hi)
    (kga_analysis_PfEMP_FWS.cpp:27-29): the float32 INFO value widened to double is compared with >=
    (kgl_variant_filter/kgl_variant_filter_Pf7.cpp:48-58); a missing AF passes BOTH filters (:61-64) and
    is therefore excluded from every bin by the NOT.  `carried` (bool per row) removes variants no genome
    carries: the reference's sparse store never holds them (kgl_variant_db_variant.cpp:13-30).
    """
    af = np.asarray(af32, dtype=np.float32).astype(np.float64)
    out = np.full(af.shape, NO_BIN, dtype=np.uint8)
    missing = np.isnan(af)
    for b, (lo, hi) in enumerate(FWS_BINS):
        sel = (~missing) & (af >= lo) & ~(af >= hi)
        out[sel] = b
    if carried is not None:
        out[~np.asarray(carried, dtype=bool)] = NO_BIN
    return out
