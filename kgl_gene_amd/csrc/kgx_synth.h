// Synthetic population definition, shared by the HIP generator kernel and its
// host twin.  All fp64 arithmetic uses only + - * / max on IEEE doubles and the library is built
// with -ffp-contract=off on both sides, so host and device agree bit for bit.
//
// Genotype model = the reference's own: allele-class probabilities under Hardy-Weinberg with an
// inbreeding coefficient F (AlleleFreqVector::unadjustedAlleleClassFrequencies,
// kga_analytic/kga_inbreed/kga_analysis_inbreed_freq.cpp:127-205), clamped and normalised as
// AlleleClassFrequencies::normalize (kga_analysis_inbreed_freq.h:45-63), class drawn in the order
// minor-hom, minor-het, major-hom, major-het (selectAlleleClass, _freq.cpp:221-261).  Genome g has
// F on the reference's synthetic grid -0.5 ... 0.5 step 0.01 (kga_analysis_inbreed_synthetic.h:40-42),
// cycled over genomes.
#ifndef KGX_SYNTH_H
#define KGX_SYNTH_H

#include "kgx_philox.h"

// Philox counter word 3 selects the stream.
enum : uint32_t {
  KGX_STREAM_GENOTYPE = 0,   // ctr = (variant_lo, variant_hi, genome/4, 0) -> 4 genotype draws
  KGX_STREAM_AF       = 1,   // ctr = (variant_lo, variant_hi, 0, 1)        -> v[0] = AF draw
  KGX_STREAM_LOCUS    = 2,   // ctr = (variant_lo, variant_hi, 0, 2)        -> gap, ref, alt, n_alt draws
  KGX_STREAM_ALLELE   = 3,   // ctr = (locus_lo, locus_hi, genome/4, 3)     -> multiallelic allele picks
  KGX_STREAM_ALT      = 4    // ctr = (locus_lo, locus_hi, alt, 4)          -> per-alt AF / indel draws
};

KGX_HD double kgx_fmax0(double x) { return x > 0.0 ? x : 0.0; }

// Inbreeding coefficient of (global) genome index g.
KGX_HD double kgx_synth_inbreeding(uint64_t g) {
  return static_cast<double>(static_cast<int>(g % 101u) - 50) / 100.0;
}

// AF of variant row v: float32(U[0.01, 0.5]).
KGX_HD float kgx_synth_af(uint64_t seed, uint64_t v) {
  const kgx_u32x4 r = kgx_philox4x32_10(static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32), 0u,
                                        KGX_STREAM_AF, static_cast<uint32_t>(seed),
                                        static_cast<uint32_t>(seed >> 32));
  const double u = kgx_u01(r.v[0]);
  return static_cast<float>(0.01 + 0.49 * u);
}

// Normalised class thresholds for one minor allele of frequency p at inbreeding F.
// t_hom: P(minor homozygous); t_ref: t_hom + P(major homozygous).  u <= t_hom -> dosage 2,
// else u <= t_ref -> dosage 0, else dosage 1 (the minor-het class is empty for one alt).
KGX_HD void kgx_synth_thresholds(double p, double F, double& t_hom, double& t_ref, double& t_all) {
  const double major = kgx_fmax0(1.0 - p);
  double minor_hom = (F * p) + ((1.0 - F) * p * p);
  double major_hom = (F * major) + ((1.0 - F) * major * major);
  double major_het = (1.0 - F) * 2.0 * major * p;
  minor_hom = kgx_fmax0(minor_hom);
  major_hom = kgx_fmax0(major_hom);
  major_het = kgx_fmax0(major_het);
  const double sum = major_hom + major_het + minor_hom + 0.0;
  minor_hom = minor_hom / sum;
  major_hom = major_hom / sum;
  major_het = major_het / sum;
  t_hom = minor_hom;
  t_ref = (minor_hom + 0.0) + major_hom;
  t_all = t_ref + major_het;
}

KGX_HD uint32_t kgx_synth_dosage_from_u(double u, double t_hom, double t_ref, double t_all) {
  if (u <= t_hom) return 2u;
  if (u <= t_ref) return 0u;
  if (u <= t_all) return 1u;
  return 0u;  // selectAlleleClass falls back to MAJOR_HOMOZYGOUS (_freq.cpp:256-260)
}

// Four consecutive genomes (4*gq .. 4*gq+3, global indices) of variant row v -> one packed byte.
KGX_HD uint32_t kgx_synth_quad(uint64_t seed, uint64_t v, uint64_t gq, double p) {
  const kgx_u32x4 r = kgx_philox4x32_10(static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32),
                                        static_cast<uint32_t>(gq), KGX_STREAM_GENOTYPE,
                                        static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  uint32_t byte = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 0; j < 4; ++j) {
    const double F = kgx_synth_inbreeding(gq * 4u + static_cast<uint64_t>(j));
    double t_hom, t_ref, t_all;
    kgx_synth_thresholds(p, F, t_hom, t_ref, t_all);
    byte |= kgx_synth_dosage_from_u(kgx_u01(r.v[j]), t_hom, t_ref, t_all) << (2 * j);
  }
  return byte;
}

// Single (global) genome g of variant row v; same bits as the matching lane of kgx_synth_quad.
KGX_HD uint32_t kgx_synth_dosage(uint64_t seed, uint64_t v, uint64_t g, double p) {
  const kgx_u32x4 r = kgx_philox4x32_10(static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32),
                                        static_cast<uint32_t>(g >> 2), KGX_STREAM_GENOTYPE,
                                        static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  double t_hom, t_ref, t_all;
  kgx_synth_thresholds(p, kgx_synth_inbreeding(g), t_hom, t_ref, t_all);
  return kgx_synth_dosage_from_u(kgx_u01(r.v[g & 3u]), t_hom, t_ref, t_all);
}

#endif  // KGX_SYNTH_H
