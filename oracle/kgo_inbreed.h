// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
// Inbreeding analysis restated from kga_analytic/kga_inbreed/:
//   AlleleClassFrequencies, AlleleFreqVector, AlleleFreqInfo   kga_analysis_inbreed_freq.{h,cpp}
//   LociiVectorArguments                                        kga_analysis_inbreed_args.h:69-108
//   RetrieveLociiVector, InbreedSampling                        kga_analysis_inbreed_locus.{h,cpp}
//   LocusResults                                                kga_analysis_inbreed_output.h:21-35
//   RetryCalcResult, InbreedingCalculation                      kga_analysis_inbreed_calc.{h,cpp}
//   InbreedingAnalysis (window loop, per-genome fan-out)        kga_analysis_inbreed_diploid.cpp
//   InbreedSynthetic::generateSyntheticPopulation               kga_analysis_inbreed_syngen.cpp:20-196
//
// Third-party arithmetic: the reference maximises the log-likelihood with nlopt LN_NELDERMEAD
// (kel_math/kel_optimize.cpp:329-449; nlopt is un-vendored and its version is not pinned anywhere in
// the reference — CMakeLists.txt:665 only names the library).  neldermead1D() restates the published
// Nelder-Mead simplex method (Nelder & Mead 1965; nlopt's variant after Box: reflection 1, expansion 2,
// contraction 1/2, shrink 1/2, points clamped to the bounds) for one parameter.  No reference test
// pins its results: parity for "Loglikelihood" is unpinned and checked to the reference's own 1e-4
// convergence band only.
#ifndef KGO_INBREED_H
#define KGO_INBREED_H

#include <list>
#include <random>

#include "kgo_core.h"

namespace kgo {

class AlleleClassFrequencies {
 public:
  AlleleClassFrequencies(double major_hom, double major_het, double minor_hom, double minor_het, double inbreeding)
      : major_homozygous_(major_hom), major_heterozygous_(major_het), minor_homozygous_(minor_hom),
        minor_heterozygous_(minor_het), inbreeding_(inbreeding) {}
  bool validFrequencies() const { return std::fabs(sumFrequencies() - 1.0) < 1.0e-04; }
  double sumFrequencies() const { return major_homozygous_ + major_heterozygous_ + minor_homozygous_ + minor_heterozygous_; }
  double majorHomozygous() const { return major_homozygous_; }
  double majorHeterozygous() const { return major_heterozygous_; }
  double minorHomozygous() const { return minor_homozygous_; }
  double minorHeterozygous() const { return minor_heterozygous_; }
  double inbreeding() const { return inbreeding_; }
  void nonNegative() {
    major_homozygous_ = std::max(0.0, major_homozygous_);
    major_heterozygous_ = std::max(0.0, major_heterozygous_);
    minor_homozygous_ = std::max(0.0, minor_homozygous_);
    minor_heterozygous_ = std::max(0.0, minor_heterozygous_);
  }
  void normalize() {
    nonNegative();
    const double sum_freqs = sumFrequencies();
    major_homozygous_ = major_homozygous_ / sum_freqs;
    major_heterozygous_ = major_heterozygous_ / sum_freqs;
    minor_homozygous_ = minor_homozygous_ / sum_freqs;
    minor_heterozygous_ = minor_heterozygous_ / sum_freqs;
  }
 private:
  double major_homozygous_, major_heterozygous_, minor_homozygous_, minor_heterozygous_, inbreeding_;
};

enum class AlleleClassType { MAJOR_HOMOZYGOUS, MAJOR_HETEROZYGOUS, MINOR_HETEROZYGOUS, MINOR_HOMOZYGOUS };

class AlleleFreqRecord {
 public:
  AlleleFreqRecord(VariantPtr allele, double frequency) : allele_(std::move(allele)), frequency_(frequency) {}
  const VariantPtr& allele() const { return allele_; }
  double frequency() const { return frequency_; }
 private:
  VariantPtr allele_;
  double frequency_;
};

class AlleleFreqVector {
 public:
  AlleleFreqVector(const OffsetDBArray& variant_vector, int super_pop);            // _freq.cpp:18-57
  const std::vector<AlleleFreqRecord>& alleleFrequencies() const { return allele_frequencies_; }
  double minorAlleleFrequencies() const;                                           // :113-117
  double majorAlleleFrequency() const;                                             // :119-123
  bool checkValidAlleleVector() const;                                             // :61-75
  AlleleClassFrequencies unadjustedAlleleClassFrequencies(double inbreeding) const;   // :127-205
  AlleleClassFrequencies alleleClassFrequencies(double inbreeding) const;          // :208-217
  AlleleClassType selectAlleleClass(double unit_rand, const AlleleClassFrequencies& cf) const;   // :221-261
  std::optional<AlleleFreqRecord> selectMinorHomozygous(double unit_rand, const AlleleClassFrequencies& cf) const;
  std::optional<AlleleFreqRecord> selectMajorHeterozygous(double unit_rand, const AlleleClassFrequencies& cf) const;
  std::optional<std::pair<AlleleFreqRecord, AlleleFreqRecord>> selectMinorHeterozygous(
      double unit_rand, const AlleleClassFrequencies& cf) const;
  size_t classSumErrors() const { return class_sum_errors_; }
 private:
  double sumAlleleFrequencies() const;
  std::vector<AlleleFreqRecord> allele_frequencies_;
  mutable size_t class_sum_errors_ = 0;
};

class AlleleFreqInfo {
 public:
  AlleleFreqInfo(AlleleClassType type, const AlleleFreqRecord& first, const AlleleFreqRecord& second,
                 const AlleleFreqVector& all)
      : allele_type_(type), first_allele_freq_(first), second_allele_freq_(second), allele_frequencies_(all) {}
  AlleleClassType alleleType() const { return allele_type_; }
  const AlleleFreqRecord& firstAllele() const { return first_allele_freq_; }
  const AlleleFreqRecord& secondAllele() const { return second_allele_freq_; }
  const AlleleFreqVector& alleleFrequencies() const { return allele_frequencies_; }
 private:
  AlleleClassType allele_type_;
  AlleleFreqRecord first_allele_freq_, second_allele_freq_;
  AlleleFreqVector allele_frequencies_;
};

// kga_analysis_inbreed_args.h:69-108 (defaults :98-104)
struct LociiVectorArguments {
  uint64_t lower_offset = 0;
  uint64_t upper_offset = 1000000000;
  size_t spacing = 1000;
  size_t locii_count = 1000;
  double allele_frequency_min = 0.0;
  double allele_frequency_max = 1.0;
};

// kga_analysis_inbreed_locus.cpp:105-156 / 21-72
std::vector<AlleleFreqVector> getAllelesCount(const ContigDB& reference_contig, int super_pop, const LociiVectorArguments& a);
std::vector<AlleleFreqVector> getAllelesFromTo(const ContigDB& reference_contig, int super_pop, const LociiVectorArguments& a);
std::vector<uint64_t> getLociiCount(const ContigDB& reference_contig, int super_pop, const LociiVectorArguments& a);
std::vector<uint64_t> getLociiFromTo(const ContigDB& reference_contig, int super_pop, const LociiVectorArguments& a);
// InbreedSampling::getLocusList (:263-326): the sampled loci as a ContigDB carrying every reference variant there.
std::shared_ptr<const ContigDB> getLocusList(const ContigDB& reference_contig, int super_pop, const LociiVectorArguments& a);

// kga_analysis_inbreed_output.h:21-35
struct LocusResults {
  std::string genome;
  size_t major_hetero_count{0};
  double major_hetero_freq{0.0};
  size_t minor_hetero_count{0};
  double minor_hetero_freq{0.0};
  size_t minor_homo_count{0};
  double minor_homo_freq{0.0};
  size_t major_homo_count{0};
  double major_homo_freq{0.0};
  size_t total_allele_count{0};
  double inbred_allele_sum{0.0};
};

// kga_analysis_inbreed_calc.{h,cpp}:17-68 — including the iterator slip in checkTolerance that makes
// it compare every entry with itself (so it always passes once min_retry values are held).
class RetryCalcResult {
 public:
  RetryCalcResult(double tolerance, size_t min_retry, size_t max_retry)
      : tolerance_(tolerance), min_retry_(min_retry), max_retry_(max_retry) {}
  bool checkRetry(double retry);
  size_t retries() const { return retry_count_; }
 private:
  bool checkTolerance() const;
  const double tolerance_;
  const size_t min_retry_, max_retry_;
  size_t retry_count_{0};
  std::list<double> current_retries_;
};

enum class InbreedAlgorithm { RitlandLocus, Simple, HallME, Loglikelihood };
std::optional<InbreedAlgorithm> namedAlgorithm(const std::string& name);   // calc.h:103-106

// The per-genome algorithms.  `phased` only documents the data: phase lives in the Variants.
// start_seed drives the random restarts of HallME/Loglikelihood (the reference uses std::random_device).
std::pair<std::vector<AlleleFreqInfo>, LocusResults> generateFrequencies(const std::string& genome_id,
                                                                         const ContigDB& contig, int super_pop,
                                                                         const ContigDB& locus_list);   // _freq.cpp:425-583
LocusResults processSimple(const std::string& genome_id, const ContigDB& contig, int super_pop, const ContigDB& locus_list);
LocusResults processRitlandLocus(const std::string& genome_id, const ContigDB& contig, int super_pop, const ContigDB& locus_list);
LocusResults processHallME(const std::string& genome_id, const ContigDB& contig, int super_pop, const ContigDB& locus_list,
                           uint64_t start_seed);
// The start points one task draws, restart by restart, exactly as processHallME (_calc.cpp:235-246) / processLogLikelihood
// (:163-180) construct and use their entropy source and distribution: draws[r] = the start of restart r.
std::vector<double> restartDraws(InbreedAlgorithm algorithm, uint64_t start_seed, size_t restarts);
LocusResults processLogLikelihood(const std::string& genome_id, const ContigDB& contig, int super_pop,
                                  const ContigDB& locus_list, uint64_t start_seed);
double logLikelihood(double f, const std::vector<AlleleFreqInfo>& data);   // _calc.cpp:94-129
// 1-D Nelder-Mead maximiser on [lb,ub] (see header comment).  Returns argmax; evals out.
double neldermead1D(const std::function<double(double)>& objective, double x0, double lb, double ub, double xtol_abs,
                    int maxeval, int* evals);

// InbreedingAnalysis::processResults (_diploid.cpp:98-166): one pool task per genome; results by genome id.
// super_pop_of_genome replaces the PED lookup (:125-139): genome id -> super-population index.
struct InbreedingParameters {
  LociiVectorArguments locii;
  std::string algorithm = "Loglikelihood";
  uint64_t start_seed = 0;
};
using ResultsMap = std::map<std::string, LocusResults>;
ResultsMap processResults(const PopulationDB& diploid_population, const std::string& contig_id,
                          const std::map<int, std::shared_ptr<const ContigDB>>& locus_map,
                          const std::map<std::string, int>& super_pop_of_genome, const InbreedingParameters& params);
// InbreedingAnalysis::populationInbreeding (_diploid.cpp:18-79): the window loop.  One (ident, results)
// column per window; ident = InbreedingResultColumn::generateIdent "contig_lower_upper".
std::vector<std::pair<std::string, ResultsMap>> populationInbreeding(const PopulationDB& reference_population,
                                                                     const PopulationDB& diploid_population,
                                                                     const std::map<std::string, int>& super_pop_of_genome,
                                                                     const InbreedingParameters& params);

// Dense tier (kgo_inbreed_dense.cpp): generateFrequencies + processSimple / processRitlandLocus for genomes given as the
// raw GT allele pairs of a population with one VCF record per reference offset.
int inbreedDense(const ContigDB& reference_all, const ContigDB& reference_snp_pass, int super_pop, const LociiVectorArguments& args,
                 const uint64_t* record_offsets, uint64_t n_records, const uint8_t* allele_pairs, uint64_t n_genomes, bool phased,
                 uint64_t* counts_out, double* freqs_out, double* seconds);

// InbreedSynthetic::generateSyntheticPopulation (_syngen.cpp:20-196) with a seeded mt19937_64.
std::shared_ptr<PopulationDB> generateSyntheticPopulation(double lower_inbreeding, double upper_inbreeding,
                                                          double step_inbreeding, int super_pop,
                                                          const ContigDB& locus_list, uint64_t seed);
std::string generateSyntheticGenomeId(double inbreeding, const std::string& super_population, size_t counter);   // :202-223
std::pair<bool, double> generateInbreeding(const std::string& genome_id);                                        // :226-275

}  // namespace kgo

#endif  // KGO_INBREED_H
