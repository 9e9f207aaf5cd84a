#!/usr/bin/env python3
"""Check kgl_gene_amd/csrc/host/kgx_refshim.h (+ kgx_pf7_resources.h) against the reference headers they mirror.

The GPU packages compile against the reference's own headers inside its tree (-DKGX_WITH_REFERENCE_HEADERS) and against
the shim here, where the reference cannot be built.  Nothing compiles both; this script is the next best thing: for
every public member function a shim class declares it looks for a member of the same name in the mirrored reference
class and compares the normalised signatures -- return type, parameter types, const -- so that a package written
against the shim meets the same declarations in the reference tree.  Runs only where /root/reference exists (the build
container); tests/test_refshim_cpu.py runs it and fails on anything not listed, with its reason, in ACCEPTED.

  python scripts/check_refshim.py [--reference /root/reference] [--report docs/refshim_check.txt]
"""
from __future__ import annotations

import argparse
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SHIMS = [ROOT / "kgl_gene_amd/csrc/host/kgx_refshim.h", ROOT / "kgl_gene_amd/csrc/host/kgx_pf7_resources.h"]

# shim class -> (reference header, reference class)
MIRRORS = {
    "DataDB": ("kgl_genomics/kgl_parser/kgl_data_file_type.h", "DataDB"),
    "VariantEvidence": ("kgl_genomics/kgl_evidence/kgl_variant_evidence.h", "VariantEvidence"),
    "Variant": ("kgl_genomics/kgl_variant_db/kgl_variant_db.h", "Variant"),
    "OffsetDB": ("kgl_genomics/kgl_variant_db/kgl_variant_db_offset.h", "OffsetDB"),
    "ContigDB": ("kgl_genomics/kgl_variant_db/kgl_variant_db_contig.h", "ContigDB"),
    "GenomeDB": ("kgl_genomics/kgl_variant_db/kgl_variant_db_genome.h", "GenomeDB"),
    "PopulationDB": ("kgl_genomics/kgl_variant_db/kgl_variant_db_population.h", "PopulationDB"),
    "FrequencyDatabaseRead": ("kgl_genomics/kgl_variant_db/kgl_variant_db_freq.h", "FrequencyDatabaseRead"),
    "InfoEvidenceAnalysis": ("kgl_genomics/kgl_evidence/kgl_variant_factory_vcf_evidence_analysis.h", "InfoEvidenceAnalysis"),
    "ParameterMap": ("kgl_app/kgl_runtime.h", "ParameterMap"),
    "ActiveParameterList": ("kgl_app/kgl_runtime.h", "ActiveParameterList"),
    "ResourceBase": ("kgl_app/kgl_runtime_resource.h", "ResourceBase"),
    "AnalysisResources": ("kgl_app/kgl_runtime_resource.h", "AnalysisResources"),
    "HsGenealogyRecord": ("kgl_genomics/kgl_parser/kgl_hsgenealogy_parser.h", "HsGenealogyRecord"),
    "HsGenomeGenealogyData": ("kgl_genomics/kgl_parser/kgl_hsgenealogy_parser.h", "HsGenomeGenealogyData"),
    "VirtualAnalysis": ("kgl_app/kgl_package_analysis_virtual.h", "VirtualAnalysis"),
    "Pf7SampleRecord": ("kgl_genomics/kgl_parser/kgl_pf7_sample_parser.h", "Pf7SampleRecord"),
    "Pf7SampleResource": ("kgl_genomics/kgl_parser/kgl_pf7_sample_parser.h", "Pf7SampleResource"),
    "Pf7FwsResource": ("kgl_genomics/kgl_parser/kgl_pf7_fws_parser.h", "Pf7FwsResource"),
    "Pf7SampleLocation": ("kgl_genomics/kgl_parser/kgl_Pf7_physical_distance.h", "Pf7SampleLocation"),
}

# enums and constant holders: every enumerator / constant of the shim must exist, spelled the same, in the reference's
ENUMS = {
    "DataSourceEnum": "kgl_genomics/kgl_parser/kgl_data_file_type.h",
    "DataStructureEnum": "kgl_genomics/kgl_parser/kgl_data_file_type.h",
    "VariantPhase": "kgl_genomics/kgl_variant_db/kgl_variant_db.h",
}
CONSTANTS = {   # shim class -> (reference header, constants the packages use)
    "ResourceProperties": ("kgl_app/kgl_properties_resource.h", ["GENEALOGY_RESOURCE_ID_", "PF7SAMPLE_RESOURCE_ID_", "PF7FWS_RESOURCE_ID_"]),
    "FrequencyDatabaseRead": ("kgl_genomics/kgl_variant_db/kgl_variant_db_freq.h",
                              ["SUPER_POP_AFR_", "SUPER_POP_AMR_", "SUPER_POP_EAS_", "SUPER_POP_EUR_", "SUPER_POP_SAS_", "SUPER_POP_ALL_"]),
    "ParameterMap": ("kgl_app/kgl_runtime.h", ["ANY_SIZE"]),
}

# (class, member) -> why the difference is accepted.  Everything else must match.
ACCEPTED: dict[tuple[str, str], str] = {}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def class_body(text: str, name: str) -> str | None:
    """The text between the braces of `class name ... {` (the definition, not a forward declaration)."""
    for m in re.finditer(r"\b(?:class|struct)\s+" + re.escape(name) + r"\b[^;{]*\{", text):
        depth, i = 1, m.end()
        while i < len(text) and depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        return text[m.end():i - 1]
    return None


def top_level(body: str) -> str:
    """The class body with the bodies of its inline functions and nested classes blanked out."""
    out, depth = [], 0
    for ch in body:
        if ch == "{":
            depth += 1
            out.append(";" if depth == 1 else " ")       # an inline body ends the declaration like a semicolon
        elif ch == "}":
            depth -= 1
            out.append(" ")
        else:
            out.append(ch if depth == 0 else " ")
    return "".join(out)


def public_part(body: str, is_struct: bool) -> str:
    parts, public = [], is_struct
    for piece in re.split(r"\b(public|private|protected)\s*:", body):
        if piece in ("public", "private", "protected"):
            public = piece == "public"
        elif public:
            parts.append(piece)
    return ";".join(parts)


DECLARATION = re.compile(r"(?P<ret>[\w:<>,\s\*&]+?)\s*\b(?P<name>\w+)\s*\((?P<params>[^()]*(?:\([^()]*\)[^()]*)*)\)\s*(?P<const>const)?")


def split_params(params: str) -> list[str]:
    out, depth, cur = [], 0, ""
    for ch in params:
        depth += {"<": 1, "(": 1, ">": -1, ")": -1}.get(ch, 0)
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def normalise_type(t: str) -> str:
    t = re.sub(r"\b(kellerberrin::genome::|kellerberrin::|kgl::|kel::|std::)", "", t)
    t = re.sub(r"\b(virtual|static|inline|constexpr|explicit|friend|typename|class)\b", " ", t)
    t = re.sub(r"\[\[[^\]]*\]\]", " ", t)
    t = re.sub(r"\s+", " ", t).strip()
    t = re.sub(r"\s*([<>,&\*])\s*", r"\1", t)
    t = re.sub(r"\bconst (\w[\w:<>,]*)", r"\1 const", t)           # west const -> east const
    return t


def param_type(p: str) -> str:
    p = p.split("=")[0].strip()                                    # default arguments do not change the type
    m = re.match(r"(.*?[\s&\*>])(\w+)$", p)                        # a trailing identifier is the parameter's name
    if m and m.group(2) not in ("int", "bool", "double", "float", "size_t", "char", "long", "unsigned", "uint32_t", "uint64_t", "string"):
        p = m.group(1)
    return normalise_type(p)


def members(body: str, is_struct: bool) -> dict[str, set[str]]:
    found: dict[str, set[str]] = {}
    for statement in public_part(top_level(body), is_struct).split(";"):
        statement = " ".join(statement.split())
        if "(" not in statement or statement.startswith(("using ", "typedef ", "template")) and "(" not in statement:
            continue
        statement = re.sub(r"^template\s*<[^>]*>\s*", "", statement)
        statement = re.sub(r"\s*(\boverride\b|\bfinal\b|\bnoexcept\b|= 0|= default|= delete)\s*", " ", statement)
        statement = re.sub(r"\brequires\s+[\w:]+\s*<[^>]*>\s*", "", statement)         # a constraint does not change what a caller writes
        statement = re.sub(r":\s*\w+[({].*$", "", statement)       # a constructor's initialiser list
        m = DECLARATION.search(statement)
        if not m or m.group("name") in ("if", "for", "while", "switch", "return", "operator"):
            continue
        ret = normalise_type(m.group("ret"))
        signature = f"{ret} ({','.join(param_type(p) for p in split_params(m.group('params')))}){' const' if m.group('const') else ''}"
        found.setdefault(m.group("name"), set()).add(signature)
    return found


def check(reference: Path):
    shim_text = "\n".join(strip_comments(p.read_text()) for p in SHIMS if p.exists())
    rows, problems = [], []
    for shim_class, (header, ref_class) in MIRRORS.items():
        body = class_body(shim_text, shim_class)
        if body is None:
            continue                                               # the shim does not (or no longer) mirror this class
        ref_path = reference / header
        if not ref_path.exists():
            problems.append((shim_class, "*", f"reference header {header} not found"))
            continue
        ref_body = class_body(strip_comments(ref_path.read_text()), ref_class)
        if ref_body is None:
            problems.append((shim_class, "*", f"class {ref_class} not found in {header}"))
            continue
        is_struct = re.search(r"\bstruct\s+" + shim_class + r"\b", shim_text) is not None
        ref_is_struct = re.search(r"\bstruct\s+" + ref_class + r"\b", strip_comments(ref_path.read_text())) is not None
        mine, theirs = members(body, is_struct), members(ref_body, ref_is_struct)
        for name, signatures in sorted(mine.items()):
            if name == shim_class:                                 # constructors: the packages never construct these
                continue
            if name not in theirs:
                status = "accepted" if (shim_class, name) in ACCEPTED else "MISSING in the reference"
            elif signatures & theirs[name]:
                status = "same"
            else:
                status = "accepted" if (shim_class, name) in ACCEPTED else "DIFFERENT"
            rows.append((shim_class, name, status, sorted(signatures), sorted(theirs.get(name, [])), header))
            if status in ("MISSING in the reference", "DIFFERENT"):
                problems.append((shim_class, name, f"{status}: shim {sorted(signatures)} vs {header} {sorted(theirs.get(name, []))}"))
    for enum, header in ENUMS.items():
        def enumerators(text):
            m = re.search(r"\benum\s+class\s+" + enum + r"\b[^{]*\{([^}]*)\}", text)
            return None if not m else [e.split("=")[0].strip() for e in m.group(1).split(",") if e.strip()]
        mine, theirs = enumerators(shim_text), enumerators(strip_comments((reference / header).read_text()))
        if mine is None or theirs is None:
            problems.append((enum, "*", f"enum not found (shim: {mine is not None}, {header}: {theirs is not None})"))
            continue
        for position, e in enumerate(mine):
            same_place = position < len(theirs) and theirs[position] == e
            status = "same" if same_place else ("DIFFERENT" if e in theirs else "MISSING in the reference")
            rows.append((enum, e, status, [f"enumerator {position}"], [f"enumerator {theirs.index(e)}"] if e in theirs else [], header))
            if status != "same":                                   # the record files of the tests carry the enum's integer value
                problems.append((enum, e, f"{status} ({header})"))
    for holder, (header, names) in CONSTANTS.items():
        text = strip_comments((reference / header).read_text())
        for name in names:
            present = re.search(r"\b" + name + r"\b", text) is not None and re.search(r"\b" + name + r"\b", shim_text) is not None
            rows.append((holder, name, "same" if present else "MISSING in the reference", ["constant"], ["constant"] if present else [], header))
            if not present:
                problems.append((holder, name, f"constant not found in {header} or in the shim"))
    return rows, problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--report", default="")
    args = ap.parse_args()
    reference = Path(args.reference)
    if not reference.exists():
        sys.exit(f"{reference} does not exist: this check runs in the build container only")
    rows, problems = check(reference)
    lines = [f"{c}::{n}: {s}" + ("" if s == "same" else f"\n    shim      {a}\n    reference {b} ({h})" + (f"\n    accepted: {ACCEPTED[(c, n)]}" if (c, n) in ACCEPTED else ""))
             for c, n, s, a, b, h in rows]
    summary = f"{len(rows)} shim members checked, {sum(1 for r in rows if r[2] == 'same')} identical in signature, " \
              f"{sum(1 for r in rows if r[2] == 'accepted')} accepted differences, {len(problems)} problems"
    text = "\n".join(lines + ["", summary]) + "\n"
    if args.report:
        Path(args.report).write_text(text)
    print(text)
    for shim_class, name, what in problems:
        print(f"PROBLEM {shim_class}::{name}: {what}")
    sys.exit(1 if problems else 0)


if __name__ == "__main__":
    main()
