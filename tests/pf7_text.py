"""Synthetic Pf7 sample / FWS resource files (the layout of Pf7_samples.txt and Pf7_fws.txt: tab separated, one header
row) for the genome-filter and location tests."""
from __future__ import annotations

import numpy as np

SAMPLE_HEADER = ["Sample", "Study", "Country", "Admin level 1", "Country latitude", "Country longitude", "Admin level 1 latitude",
                 "Admin level 1 longitude", "Year", "ENA", "All samples same case", "Population", "% callable", "QC pass",
                 "Exclusion reason", "Sample type", "Sample was in Pf6"]

# site, country, region, (country lat, lon), (site lat, lon); a blank coordinate reads as 0
SITES = [
    ("Kassena", "Ghana", "AF-W", ("7.95", "-1.03"), ("10.83", "-1.17")),
    ("Navrongo", "Ghana", "AF-W", ("7.95", "-1.03"), ("10.89", "-1.09")),       # 11 km from Kassena
    ("Kilifi", "Kenya", "AF-E", ("0.18", "37.91"), ("-3.51", "39.91")),
    ("Pailin", "Cambodia", "AS-SE-E", ("12.57", "104.99"), ("12.85", "102.61")),
    ("Pursat", "Cambodia", "AS-SE-E", ("12.57", "104.99"), ("12.53", "103.92")),
    ("Peru", "Peru", "SA", ("-9.19", "-75.02"), ("", "")),                      # a site named as its country, no coordinates
    ("", "Mali", "AF-W", ("17.57", "-4.0"), ("", "")),                          # no site at all
]


def write_resources(tmp_path, ids, rng_seed=5, big_site=0, extra_samples=7):
    """Sample records for `ids` (plus a few samples that are in no VCF) and FWS values for most of them.
    About half of the genomes sit at SITES[big_site] so that one site clears the 20-sample bar.  Returns (sample_path, fws_path,
    records) with records[id] = dict(site, country, qc, fws or None)."""
    rng = np.random.default_rng(rng_seed)
    records = {}
    lines = ["\t".join(SAMPLE_HEADER), "# a comment line between the header and the data"]
    fws_lines = ["Sample\tFws"]
    everyone = list(ids) + [f"XTRA{i:03d}-C" for i in range(extra_samples)]
    for k, sample in enumerate(everyone):
        s = big_site if rng.random() < 0.5 else int(rng.integers(0, len(SITES)))
        site, country, region, (clat, clon), (slat, slon) = SITES[s]
        qc = rng.random() < 0.8
        study = f"10{int(rng.integers(1, 5)):02d}-PF-XX"
        year = str(int(rng.integers(2005, 2019)))
        qc_text = ("True" if k % 3 else "TRUE") if qc else "False"
        pad = "  " if k % 5 == 0 else ""                       # fields are trimmed at both ends
        fields = [pad + sample + pad, study, country, site, clat, clon, slat, slon, year, f"ERR{k:06d}", sample.upper(), region,
                  f"{rng.uniform(40, 95):.2f}", qc_text, "Analysis_set" if qc else "Low_coverage", "gDNA", "True" if k % 2 else "False"]
        lines.append("\t".join(fields))
        fws = None
        if rng.random() < 0.93:                                # some samples have no published FWS
            fws = float(np.round(rng.uniform(0.90, 1.0) if rng.random() < 0.8 else rng.uniform(0.3, 0.9), 4))
            fws_lines.append(f"{sample}\t{fws}")
        records[sample] = dict(site=site, country=country, qc=qc, fws=fws)
    fws_lines.append("BADVALUE-C\tnot_a_number")                # a value that does not parse: the line is dropped
    sample_path, fws_path = tmp_path / "Pf7_samples.txt", tmp_path / "Pf7_fws.txt"
    sample_path.write_text("\n".join(lines) + "\n")
    fws_path.write_text("\n".join(fws_lines) + "\n")
    return sample_path, fws_path, records
