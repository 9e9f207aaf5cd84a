// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h, kgo_inbreed.h).
//
// Dense tier of the inbreeding oracle, for slices too large for the pointer-chasing store (64 genomes x 5M loci is
// 10^8 OffsetDB nodes): generateFrequencies (kga_analysis_inbreed_freq.cpp:425-583) + processSimple /
// processRitlandLocus (_calc.cpp:318-431) restated for a population in which EVERY reference offset holds exactly one
// VCF record and the genomes are given as that record's raw GT allele pairs instead of Variant objects.  What stays the
// oracle's own code: the locus list (getLocusList), AlleleFreqVector and its validity / class frequencies, Variant::
// isSNP / analogous / homozygous on the reference's Variant objects, and the summation order (ascending offset, one
// add per classified locus, exactly the second loop of generateFrequencies).  What is restated: "the genome's SNP
// variants at this offset" = the record's alt Variants its two allele indices name, phase A before phase B.
// tests/test_oracle_pins.py holds it bit for bit against generateFrequencies on populations both can take.
#include <chrono>

#include "kgo_inbreed.h"

namespace kgo {

struct DenseLocus {
  uint64_t record = 0;                               // row of the allele-pair matrix
  AlleleFreqVector alleles;                          // of the locus list's variants, as generateFrequencies builds it
  AlleleClassFrequencies class_frequencies;          // alleleClassFrequencies(0.0): one value per locus, whoever asks
  std::vector<VariantPtr> variant_of_alt;            // the record's alt Variants in alt order (unfiltered reference)
  std::vector<uint8_t> alt_is_snp;
  DenseLocus(uint64_t r, AlleleFreqVector a) : record(r), alleles(std::move(a)), class_frequencies(alleles.alleleClassFrequencies(0.0)) {}
};

struct DenseResults {
  LocusResults locus_results;
  double simple = 0.0, ritland = 0.0;
};

// One genome: allele_pairs[record][2] = 1-based indices over ALL alts of the record (0 = reference), phase A then B.
static DenseResults denseGenome(const std::vector<DenseLocus>& loci, const uint8_t* allele_pairs, size_t pair_stride, bool phased) {
  DenseResults out;
  LocusResults& r = out.locus_results;
  constexpr double minimum_frequency = 0.001;          // processRitlandLocus (_calc.cpp:380)
  size_t ritland_count = 0;
  double ritland_sum = 0.0;
  for (const DenseLocus& locus : loci) {
    const uint8_t* pair = allele_pairs + locus.record * pair_stride;
    // contig.viewFilter(SNPFilter) then findOffsetArray(offset): the genome's SNP variants here, in OffsetDB order
    const Variant* carried[2];
    VariantPhase phase[2];
    size_t n = 0;
    for (int copy = 0; copy < 2; ++copy) {
      const uint32_t a = pair[copy];
      if (a == 0 || a > locus.variant_of_alt.size() || !locus.alt_is_snp[a - 1]) continue;
      carried[n] = locus.variant_of_alt[a - 1].get();
      phase[n] = !phased ? VariantPhase::UNPHASED : (copy == 0 ? VariantPhase::DIPLOID_PHASE_A : VariantPhase::DIPLOID_PHASE_B);
      ++n;
    }
    enum { NONE, MAJOR_HOM, MAJOR_HET, MINOR_HOM, MINOR_HET } cls = NONE;
    double first_frequency = 0.0;
    if (n > 0) {
      for (const auto& allele_freq : locus.alleles.alleleFrequencies()) {
        if (!carried[0]->analogous(*allele_freq.allele())) continue;
        if (n == 1) {
          cls = MAJOR_HET; first_frequency = allele_freq.frequency();
          break;
        }
        // n == 2 (a record names at most two alleles; >= 3 variants need repeated records, which this tier excludes)
        if (carried[0]->analogous(*carried[1]) && phase[0] != phase[1]) {        // homozygous()
          cls = MINOR_HOM; first_frequency = allele_freq.frequency();
          break;
        }
        bool found_second_minor = false;
        for (const auto& second : locus.alleles.alleleFrequencies())
          if (carried[1]->analogous(*second.allele())) { found_second_minor = true; break; }
        if (found_second_minor) { cls = MINOR_HET; first_frequency = allele_freq.frequency(); break; }
      }
    } else if (!locus.alleles.alleleFrequencies().empty()) {
      const double major_allele_frequency = locus.alleles.majorAlleleFrequency();
      constexpr double minimum_major_frequency = 0.01;
      if (major_allele_frequency > minimum_major_frequency) { cls = MAJOR_HOM; first_frequency = major_allele_frequency; }
    }
    if (cls == NONE) continue;
    ++r.total_allele_count;
    r.major_homo_freq += locus.class_frequencies.majorHomozygous();
    r.minor_homo_freq += locus.class_frequencies.minorHomozygous();
    r.major_hetero_freq += locus.class_frequencies.majorHeterozygous();
    r.minor_hetero_freq += locus.class_frequencies.minorHeterozygous();
    switch (cls) {
      case MINOR_HOM: ++r.minor_homo_count; break;
      case MAJOR_HET: ++r.major_hetero_count; break;
      case MINOR_HET: ++r.minor_hetero_count; break;
      default: ++r.major_homo_count; break;
    }
    if (cls == MAJOR_HOM || cls == MINOR_HOM) {
      if (first_frequency > minimum_frequency) {
        const double ratio = (1.0 / first_frequency);
        ritland_sum += ratio;
        ritland_sum -= 1.0;
        ++ritland_count;
      }
    } else {
      ritland_sum -= 1.0;
      ++ritland_count;
    }
  }
  if (r.total_allele_count > 0) {                      // processSimple (_calc.cpp:335-344)
    const auto observed_homozygous = static_cast<double>(r.minor_homo_count + r.major_homo_count);
    const auto expected_homozygous = r.minor_homo_freq + r.major_homo_freq;
    out.simple = (observed_homozygous - expected_homozygous) / (static_cast<double>(r.total_allele_count) - expected_homozygous);
  }
  out.ritland = ritland_count > 0 ? ritland_sum / static_cast<double>(ritland_count) : 0.0;
  return out;
}

// reference_all: the mono-genome frequency source as parsed; reference_snp_pass: its SNP & PASS view (what INBREED
// keeps, kga_analysis_inbreed.cpp:79).  record_offsets[n_records]: the offset of each row of allele_pairs
// [n_records][n_genomes][2].  counts_out [G][5], freqs_out [G][6] = the four class-frequency sums (order of
// kgo_inbreed_window), Simple, RitlandLocus.  Returns -1 if an offset of the locus list holds more than one record.
int inbreedDense(const ContigDB& reference_all, const ContigDB& reference_snp_pass, int super_pop, const LociiVectorArguments& args,
                 const uint64_t* record_offsets, uint64_t n_records, const uint8_t* allele_pairs, uint64_t n_genomes, bool phased,
                 uint64_t* counts_out, double* freqs_out, double* seconds) {
  std::map<uint64_t, uint64_t> record_of_offset;
  for (uint64_t r = 0; r < n_records; ++r)
    if (!record_of_offset.emplace(record_offsets[r], r).second) return -1;
  std::shared_ptr<const ContigDB> locus_list = getLocusList(reference_snp_pass, super_pop, args);
  std::vector<DenseLocus> loci;
  for (const auto& [offset, offset_ptr] : locus_list->getMap()) {
    AlleleFreqVector allele_freq_vector(offset_ptr->getVariantArray(), super_pop);
    if (!allele_freq_vector.checkValidAlleleVector()) continue;
    auto record = record_of_offset.find(offset);
    if (record == record_of_offset.end()) return -1;
    DenseLocus locus(record->second, std::move(allele_freq_vector));
    auto all_alts = reference_all.findOffsetArray(offset);
    if (!all_alts) return -1;
    for (const auto& variant : all_alts.value()) {
      if (variant->altVariantIndex() != locus.variant_of_alt.size()) return -1;      // a second record at this offset
      locus.variant_of_alt.push_back(variant);
      locus.alt_is_snp.push_back(variant->isSNP() ? 1 : 0);
    }
    loci.push_back(std::move(locus));
  }
  const auto t0 = std::chrono::steady_clock::now();
  WorkflowThreads thread_pool(poolThreads(n_genomes));
  std::vector<std::future<DenseResults>> futures;
  for (uint64_t g = 0; g < n_genomes; ++g)
    futures.push_back(thread_pool.enqueueFuture([&loci, allele_pairs, g, n_genomes, phased]() {
      return denseGenome(loci, allele_pairs + g * 2, n_genomes * 2, phased);
    }));
  for (uint64_t g = 0; g < n_genomes; ++g) {
    const DenseResults res = futures[g].get();
    const LocusResults& r = res.locus_results;
    uint64_t* c = counts_out + g * 5;
    double* f = freqs_out + g * 6;
    c[0] = r.major_hetero_count; c[1] = r.minor_hetero_count; c[2] = r.minor_homo_count; c[3] = r.major_homo_count; c[4] = r.total_allele_count;
    f[0] = r.major_hetero_freq; f[1] = r.minor_hetero_freq; f[2] = r.minor_homo_freq; f[3] = r.major_homo_freq;
    f[4] = res.simple; f[5] = res.ritland;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

}  // namespace kgo
