// Synthetic multi-allelic SNP+indel population (BASELINE config 4, SURVEY.md §8d "C5"), host + device.
//
// Per locus: 1/2/3 alts with probability 0.7/0.2/0.1; each alt is an indel with probability 0.15; raw minor
// AFs U[0.01,0.5], rescaled so that their sum is <= 0.6, stored as float32 (the reference's INFO type).
// Per (locus, genome): allele class drawn from AlleleFreqVector::alleleClassFrequencies(F_g) over ALL alts and
// alleles drawn as InbreedSynthetic::generateSyntheticPopulation does (kga_analysis_inbreed_syngen.cpp:83-180,
// selectors kga_analysis_inbreed_freq.cpp:221-420), phases A/B.  The inbreeding view (gt8 byte) keeps only SNP
// alleles, indexed into the locus's SNP alt list, as the SNP filters of INBREED would.
#ifndef KGX_SYNTH_MULTI_H
#define KGX_SYNTH_MULTI_H

#include "kgx_synth.h"

constexpr int KGX_SYNTH_MAX_ALTS = 3;

struct kgx_synth_locus {
  int n_alt;
  float af[KGX_SYNTH_MAX_ALTS];       // INFO AF per alt (all super populations alike)
  int is_indel[KGX_SYNTH_MAX_ALTS];
  int snp_index[KGX_SYNTH_MAX_ALTS];  // 1-based index in the locus's SNP alt list, 0 for indels
  int n_snp;
};

KGX_HD kgx_synth_locus kgx_synth_make_locus(uint64_t seed, uint64_t l) {
  kgx_synth_locus loc;
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const kgx_u32x4 r = kgx_philox4x32_10(static_cast<uint32_t>(l), static_cast<uint32_t>(l >> 32), 0u, KGX_STREAM_LOCUS, k0, k1);
  const double u = kgx_u01(r.v[0]);
  loc.n_alt = u < 0.7 ? 1 : (u < 0.9 ? 2 : 3);
  double raw[KGX_SYNTH_MAX_ALTS];
  double sum = 0.0;
  loc.n_snp = 0;
  for (int a = 0; a < KGX_SYNTH_MAX_ALTS; ++a) {
    loc.af[a] = 0.0f;
    loc.is_indel[a] = 0;
    loc.snp_index[a] = 0;
    raw[a] = 0.0;
    if (a < loc.n_alt) {
      const kgx_u32x4 q = kgx_philox4x32_10(static_cast<uint32_t>(l), static_cast<uint32_t>(l >> 32), static_cast<uint32_t>(a), KGX_STREAM_ALT, k0, k1);
      raw[a] = 0.01 + 0.49 * kgx_u01(q.v[0]);
      loc.is_indel[a] = kgx_u01(q.v[1]) < 0.15 ? 1 : 0;
      sum += raw[a];
      if (!loc.is_indel[a]) loc.snp_index[a] = ++loc.n_snp;
    }
  }
  const double scale = sum > 0.6 ? 0.6 / sum : 1.0;
  for (int a = 0; a < loc.n_alt; ++a) loc.af[a] = static_cast<float>(raw[a] * scale);
  return loc;
}

// Allele pair (1-based alt indices over ALL alts, 0 = reference) of (global) genome g at locus l.
KGX_HD void kgx_synth_multi_genotype(uint64_t seed, uint64_t l, uint64_t g, const kgx_synth_locus& loc, int& a1, int& a2) {
  a1 = a2 = 0;
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const kgx_u32x4 r = kgx_philox4x32_10(static_cast<uint32_t>(l), static_cast<uint32_t>(l >> 32), static_cast<uint32_t>(g), KGX_STREAM_ALLELE, k0, k1);
  const double u_class = kgx_u01(r.v[0]), u_allele = kgx_u01(r.v[1]);
  const double F = kgx_synth_inbreeding(g);
  double f[KGX_SYNTH_MAX_ALTS];
  double sum_minor = 0.0;
  for (int a = 0; a < loc.n_alt; ++a) { f[a] = static_cast<double>(loc.af[a]); sum_minor += f[a]; }
  const double major = kgx_fmax0(1.0 - sum_minor);
  double minor_hom = 0.0, minor_het = 0.0, major_het = 0.0;
  for (int a = 0; a < loc.n_alt; ++a) minor_hom += (F * f[a]) + ((1.0 - F) * f[a] * f[a]);
  for (int a = 0; a < loc.n_alt; ++a)
    for (int b = a + 1; b < loc.n_alt; ++b) minor_het += (1.0 - F) * 2.0 * f[a] * f[b];
  double major_hom = (F * major) + ((1.0 - F) * major * major);
  for (int a = 0; a < loc.n_alt; ++a) major_het += (1.0 - F) * 2.0 * major * f[a];
  minor_hom = kgx_fmax0(minor_hom); minor_het = kgx_fmax0(minor_het); major_hom = kgx_fmax0(major_hom); major_het = kgx_fmax0(major_het);
  const double total = major_hom + major_het + minor_hom + minor_het;
  minor_hom /= total; minor_het /= total; major_hom /= total; major_het /= total;

  double cum = minor_hom;
  if (u_class <= cum) {                                       // MINOR_HOMOZYGOUS: selectMinorHomozygous
    if (minor_hom == 0.0) return;
    if (loc.n_alt == 1) { a1 = a2 = 1; return; }
    double s = 0.0;
    for (int a = 0; a < loc.n_alt; ++a) {
      s += ((f[a] * F) + (1.0 - F) * f[a] * f[a]) / minor_hom;
      if (u_allele <= s) { a1 = a2 = a + 1; return; }
    }
    return;
  }
  cum += minor_het;
  if (u_class <= cum) {                                       // MINOR_HETEROZYGOUS: selectMinorHeterozygous
    if (loc.n_alt < 2 || minor_het == 0.0) return;
    if (loc.n_alt == 2) { a1 = 1; a2 = 2; return; }
    double s = 0.0;
    for (int a = 0; a < loc.n_alt; ++a)
      for (int b = a + 1; b < loc.n_alt; ++b) {
        s += ((1.0 - F) * 2.0 * f[a] * f[b]) / minor_het;
        if (u_allele <= s) { a1 = a + 1; a2 = b + 1; return; }
      }
    return;
  }
  cum += major_hom;
  if (u_class <= cum) return;                                 // MAJOR_HOMOZYGOUS
  cum += major_het;
  if (u_class <= cum) {                                       // MAJOR_HETEROZYGOUS: selectMajorHeterozygous
    if (major_het == 0.0) return;
    if (loc.n_alt == 1) { a1 = 1; return; }
    const double major_freq = kgx_fmax0(1.0 - (sum_minor > 1.0 ? 1.0 : sum_minor));
    double s = 0.0;
    for (int a = 0; a < loc.n_alt; ++a) {
      s += ((1.0 - F) * 2.0 * major_freq * f[a]) / major_het;
      if (u_allele <= s) { a1 = a + 1; return; }
    }
  }
}

// The INBREED view of the genotype: SNP alleles only, indexed into the locus's SNP alt list.
KGX_HD uint32_t kgx_synth_gt8_byte(const kgx_synth_locus& loc, int a1, int a2) {
  uint32_t c1 = a1 ? static_cast<uint32_t>(loc.snp_index[a1 - 1]) : 0u;
  uint32_t c2 = a2 ? static_cast<uint32_t>(loc.snp_index[a2 - 1]) : 0u;
  if (c1 == 0) { c1 = c2; c2 = 0; }
  return c1 | (c2 << 4);
}

#endif  // KGX_SYNTH_MULTI_H
