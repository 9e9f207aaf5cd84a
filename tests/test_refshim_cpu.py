"""The stand-in headers the GPU packages build against here (kgx_refshim.h, kgx_pf7_resources.h) against the reference
headers they mirror: every public member function, enumerator and constant the shim declares must exist in the mirrored
reference class with the same signature (scripts/check_refshim.py).  Runs only where /root/reference exists -- the build
container; the GPU box has no reference and runs no CPU tests."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path("/root/reference")


@pytest.mark.skipif(not REFERENCE.exists(), reason="the reference tree is only present in the build container")
def test_shim_declarations_match_the_reference_headers():
    sys.path.insert(0, str(ROOT / "scripts"))
    import check_refshim

    rows, problems = check_refshim.check(REFERENCE)
    assert len(rows) >= 100, len(rows)                           # the classes were found and parsed
    assert not problems, "\n".join(f"{c}::{n}: {what}" for c, n, what in problems)
    # a member the packages lean on is really among the checked ones
    checked = {(c, n) for c, n, *_ in rows}
    for needed in [("VirtualAnalysis", "fileReadAnalysis"), ("AnalysisResources", "getSingleResource"), ("HsGenomeGenealogyData", "getGenomeGenealogyRecord"),
                   ("InfoEvidenceAnalysis", "getTypedInfoData"), ("FrequencyDatabaseRead", "superPopFrequency"), ("GenomeDB", "getContig"),
                   ("Pf7SampleLocation", "sampleRadius"), ("Pf7FwsResource", "filterFWS")]:
        assert needed in checked, needed
