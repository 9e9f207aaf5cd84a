// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h, kgo_inbreed.h).
#include "kgo_inbreed.h"

#include <iomanip>
#include <sstream>

namespace kgo {

static bool snpFilter(const Variant& v) { return v.isSNP(); }

// ---- AlleleFreqVector (kga_analysis_inbreed_freq.cpp) ------------------------------------------

AlleleFreqVector::AlleleFreqVector(const OffsetDBArray& variant_vector, int super_pop) {
  for (const auto& variant : variant_vector) {
    auto opt_value = variant->superPopFrequency(super_pop);
    if (!opt_value) continue;   // allele not defined for this super population
    bool found_duplicate = false;
    for (const auto& allele : allele_frequencies_) {
      if (allele.allele()->analogous(*variant)) {
        found_duplicate = true;
        break;
      }
    }
    if (!found_duplicate) {
      const double variant_freq = std::clamp(opt_value.value(), 0.0, 1.0);
      allele_frequencies_.emplace_back(variant, variant_freq);
    }
  }
}

bool AlleleFreqVector::checkValidAlleleVector() const {
  const double check_allele_sum = sumAlleleFrequencies() - 1.0;
  if (check_allele_sum > 1.0e-5) return false;
  return !allele_frequencies_.empty();
}

double AlleleFreqVector::sumAlleleFrequencies() const {
  double sum_allele_freq = 0.0;
  for (const auto& a : allele_frequencies_) sum_allele_freq += a.frequency();
  return sum_allele_freq;
}

double AlleleFreqVector::minorAlleleFrequencies() const { return std::clamp(sumAlleleFrequencies(), 0.0, 1.0); }

double AlleleFreqVector::majorAlleleFrequency() const { return std::clamp((1.0 - minorAlleleFrequencies()), 0.0, 1.0); }

AlleleClassFrequencies AlleleFreqVector::unadjustedAlleleClassFrequencies(double inbreeding) const {
  std::vector<double> minor_allele_frequencies;
  double sum_minor_freq = 0.0;
  for (const auto& minor_allele : allele_frequencies_) sum_minor_freq += minor_allele.frequency();

  const double major_frequency = std::max(0.0, (1.0 - sum_minor_freq));
  for (const auto& minor_allele : allele_frequencies_) {
    if (sum_minor_freq > 1.0) minor_allele_frequencies.push_back(minor_allele.frequency() / sum_minor_freq);
    else minor_allele_frequencies.push_back(minor_allele.frequency());
  }

  double minor_homozygous = 0.0;
  for (const auto& minor_frequency : minor_allele_frequencies)
    minor_homozygous += (inbreeding * minor_frequency) + ((1.0 - inbreeding) * minor_frequency * minor_frequency);

  double minor_heterozygous = 0.0;
  const size_t minor_allele_count = minor_allele_frequencies.size();
  for (size_t idx1 = 0; idx1 < minor_allele_count; ++idx1)
    for (size_t idx2 = (idx1 + 1); idx2 < minor_allele_count; ++idx2)
      minor_heterozygous += (1.0 - inbreeding) * 2.0 * minor_allele_frequencies[idx1] * minor_allele_frequencies[idx2];

  const double major_homozygous = (inbreeding * major_frequency) + ((1.0 - inbreeding) * major_frequency * major_frequency);

  double major_heterozygous = 0.0;
  for (const auto& minor_frequency : minor_allele_frequencies)
    major_heterozygous += (1.0 - inbreeding) * 2.0 * major_frequency * minor_frequency;

  const double sum_freq_classes = major_homozygous + major_heterozygous + minor_homozygous + minor_heterozygous;
  if (std::fabs(sum_freq_classes - 1.0) > 1.0e-5) ++class_sum_errors_;   // the reference logs an error (:186-201)

  return AlleleClassFrequencies(major_homozygous, major_heterozygous, minor_homozygous, minor_heterozygous, inbreeding);
}

AlleleClassFrequencies AlleleFreqVector::alleleClassFrequencies(double inbreeding) const {
  AlleleClassFrequencies class_freqs = unadjustedAlleleClassFrequencies(inbreeding);
  class_freqs.normalize();
  return class_freqs;
}

AlleleClassType AlleleFreqVector::selectAlleleClass(double unit_rand, const AlleleClassFrequencies& cf) const {
  double sum_freqs = cf.minorHomozygous();
  if (unit_rand <= sum_freqs) return AlleleClassType::MINOR_HOMOZYGOUS;
  sum_freqs += cf.minorHeterozygous();
  if (unit_rand <= sum_freqs) return AlleleClassType::MINOR_HETEROZYGOUS;
  sum_freqs += cf.majorHomozygous();
  if (unit_rand <= sum_freqs) return AlleleClassType::MAJOR_HOMOZYGOUS;
  sum_freqs += cf.majorHeterozygous();
  if (unit_rand <= sum_freqs) return AlleleClassType::MAJOR_HETEROZYGOUS;
  return AlleleClassType::MAJOR_HOMOZYGOUS;
}

std::optional<AlleleFreqRecord> AlleleFreqVector::selectMinorHomozygous(double unit_rand, const AlleleClassFrequencies& cf) const {
  if (allele_frequencies_.empty()) return std::nullopt;
  if (cf.minorHomozygous() == 0.0) return std::nullopt;
  if (allele_frequencies_.size() == 1) return allele_frequencies_.front();
  double allele_freq_sum = 0.0;
  for (const auto& allele : allele_frequencies_) {
    const double allele_freq = allele.frequency();
    const double hom_prob = (allele_freq * cf.inbreeding()) + (1.0 - cf.inbreeding()) * allele_freq * allele_freq;
    allele_freq_sum += hom_prob / cf.minorHomozygous();
    if (unit_rand <= allele_freq_sum) return allele;
  }
  return std::nullopt;
}

std::optional<AlleleFreqRecord> AlleleFreqVector::selectMajorHeterozygous(double unit_rand, const AlleleClassFrequencies& cf) const {
  if (allele_frequencies_.empty()) return std::nullopt;
  if (cf.majorHeterozygous() == 0.0) return std::nullopt;
  if (allele_frequencies_.size() == 1) return allele_frequencies_.front();
  double allele_freq_sum = 0.0;
  const double major_freq = majorAlleleFrequency();
  for (const auto& allele : allele_frequencies_) {
    const double allele_freq = allele.frequency();
    const double het_prob = (1.0 - cf.inbreeding()) * 2.0 * major_freq * allele_freq;
    allele_freq_sum += het_prob / cf.majorHeterozygous();
    if (unit_rand <= allele_freq_sum) return allele;
  }
  return std::nullopt;
}

std::optional<std::pair<AlleleFreqRecord, AlleleFreqRecord>> AlleleFreqVector::selectMinorHeterozygous(
    double unit_rand, const AlleleClassFrequencies& cf) const {
  if (allele_frequencies_.size() < 2) return std::nullopt;
  if (cf.minorHeterozygous() == 0.0) return std::nullopt;
  if (allele_frequencies_.size() == 2)
    return std::pair<AlleleFreqRecord, AlleleFreqRecord>{allele_frequencies_.front(), allele_frequencies_.back()};
  double allele_freq_sum = 0.0;
  const size_t allele_count = allele_frequencies_.size();
  for (size_t idx1 = 0; idx1 < allele_count; ++idx1) {
    const double allele1_freq = allele_frequencies_[idx1].frequency();
    for (size_t idx2 = (idx1 + 1); idx2 < allele_count; ++idx2) {
      const double allele2_freq = allele_frequencies_[idx2].frequency();
      const double het_prob = (1.0 - cf.inbreeding()) * 2.0 * allele1_freq * allele2_freq;
      allele_freq_sum += het_prob / cf.minorHeterozygous();
      if (unit_rand <= allele_freq_sum)
        return std::pair<AlleleFreqRecord, AlleleFreqRecord>{allele_frequencies_[idx1], allele_frequencies_[idx2]};
    }
  }
  return std::nullopt;
}

// ---- locus sampling (kga_analysis_inbreed_locus.cpp) -------------------------------------------

static std::vector<AlleleFreqVector> getAlleles(const ContigDB& contig, int super_pop, const LociiVectorArguments& a,
                                                bool by_count) {
  std::vector<AlleleFreqVector> locii_vector;
  auto current_offset = contig.getMap().lower_bound(a.lower_offset);
  uint64_t previous_offset = 0;
  while (current_offset != contig.getMap().end()) {
    const auto& [offset, offset_ptr] = *current_offset;
    const bool stop = by_count ? (locii_vector.size() >= a.locii_count) : (offset > a.upper_offset);
    if (stop) break;
    if ((offset >= previous_offset + a.spacing) || previous_offset == 0) {
      AlleleFreqVector allele_freq_vector(offset_ptr->getVariantArray(), super_pop);
      if (!allele_freq_vector.checkValidAlleleVector()) {
        ++current_offset;
        continue;
      }
      const double sum_frequencies = allele_freq_vector.minorAlleleFrequencies();
      const size_t minor_allele_count = allele_freq_vector.alleleFrequencies().size();
      if (minor_allele_count == 0 || sum_frequencies == 0.0 || sum_frequencies < a.allele_frequency_min ||
          sum_frequencies > a.allele_frequency_max) {
        ++current_offset;
        continue;
      }
      previous_offset = offset;
      locii_vector.push_back(allele_freq_vector);
    }
    ++current_offset;
  }
  return locii_vector;
}

std::vector<AlleleFreqVector> getAllelesCount(const ContigDB& c, int sp, const LociiVectorArguments& a) { return getAlleles(c, sp, a, true); }
std::vector<AlleleFreqVector> getAllelesFromTo(const ContigDB& c, int sp, const LociiVectorArguments& a) { return getAlleles(c, sp, a, false); }

static std::vector<uint64_t> toOffsets(const std::vector<AlleleFreqVector>& freq_vector) {
  std::vector<uint64_t> locii_vector;
  for (const auto& alleles : freq_vector)
    if (!alleles.alleleFrequencies().empty()) locii_vector.push_back(alleles.alleleFrequencies().front().allele()->offset());
  return locii_vector;
}

std::vector<uint64_t> getLociiCount(const ContigDB& c, int sp, const LociiVectorArguments& a) { return toOffsets(getAllelesCount(c, sp, a)); }
std::vector<uint64_t> getLociiFromTo(const ContigDB& c, int sp, const LociiVectorArguments& a) { return toOffsets(getAllelesFromTo(c, sp, a)); }

std::shared_ptr<const ContigDB> getLocusList(const ContigDB& reference_contig, int super_pop, const LociiVectorArguments& a) {
  auto locus_list = std::make_shared<ContigDB>(superPopName(super_pop));
  for (auto locus : getLociiFromTo(reference_contig, super_pop, a)) {
    auto variant_array_opt = reference_contig.findOffsetArray(locus);
    if (variant_array_opt)
      for (const auto& variant : variant_array_opt.value()) locus_list->addVariant(variant);
  }
  return locus_list;
}

// ---- generateFrequencies (kga_analysis_inbreed_freq.cpp:425-583) -------------------------------

std::pair<std::vector<AlleleFreqInfo>, LocusResults> generateFrequencies(const std::string& genome_id, const ContigDB& contig,
                                                                         int super_pop, const ContigDB& locus_list) {
  std::vector<AlleleFreqInfo> frequency_vector;
  LocusResults locus_results;
  locus_results.genome = genome_id;

  auto snp_contig_ptr = contig.viewFilter(snpFilter);   // deep copy per call, as in the reference (:436)

  for (const auto& [offset, offset_ptr] : locus_list.getMap()) {
    const OffsetDBArray& locus_variant_array = offset_ptr->getVariantArray();
    AlleleFreqVector allele_freq_vector(locus_variant_array, super_pop);
    if (!allele_freq_vector.checkValidAlleleVector()) continue;

    auto diploid_variant_opt = snp_contig_ptr->findOffsetArray(offset);
    if (diploid_variant_opt) {
      const auto& diploid_offset = diploid_variant_opt.value();
      for (const auto& allele_freq : allele_freq_vector.alleleFrequencies()) {
        if (diploid_offset.front()->analogous(*allele_freq.allele())) {
          if (diploid_offset.size() == 1) {
            const double major_allele_frequency = allele_freq_vector.majorAlleleFrequency();
            AlleleFreqRecord major_allele(nullptr, major_allele_frequency);   // cloneNullVariant() placeholder
            frequency_vector.emplace_back(AlleleClassType::MAJOR_HETEROZYGOUS, allele_freq, major_allele, allele_freq_vector);
            break;
          } else if (diploid_offset.size() == 2) {
            if (diploid_offset.front()->homozygous(*diploid_offset.back())) {
              frequency_vector.emplace_back(AlleleClassType::MINOR_HOMOZYGOUS, allele_freq, allele_freq, allele_freq_vector);
              break;
            } else {
              bool found_second_minor = false;
              size_t second_allele_index = 0;
              for (const auto& second_allele_freq : allele_freq_vector.alleleFrequencies()) {
                if (diploid_offset.back()->analogous(*second_allele_freq.allele())) {
                  found_second_minor = true;
                  break;
                }
                ++second_allele_index;
              }
              if (found_second_minor) {
                const AlleleFreqRecord& second_allele = allele_freq_vector.alleleFrequencies().at(second_allele_index);
                frequency_vector.emplace_back(AlleleClassType::MINOR_HETEROZYGOUS, allele_freq, second_allele, allele_freq_vector);
                break;
              }
              // not found: the reference warns and keeps scanning; no other allele can match front()
            }
          }
          // size >= 3: nothing recorded
        }
      }
    } else {
      if (!allele_freq_vector.alleleFrequencies().empty()) {
        const double major_allele_frequency = allele_freq_vector.majorAlleleFrequency();
        constexpr double minimum_major_frequency = 0.01;
        if (major_allele_frequency > minimum_major_frequency) {
          AlleleFreqRecord major_allele(nullptr, major_allele_frequency);
          frequency_vector.emplace_back(AlleleClassType::MAJOR_HOMOZYGOUS, major_allele, major_allele, allele_freq_vector);
        }
      }
    }
  }

  locus_results.total_allele_count = frequency_vector.size();
  for (const auto& allele_freq : frequency_vector) {
    AlleleClassFrequencies class_frequencies = allele_freq.alleleFrequencies().alleleClassFrequencies(0.0);
    locus_results.major_homo_freq += class_frequencies.majorHomozygous();
    locus_results.minor_homo_freq += class_frequencies.minorHomozygous();
    locus_results.major_hetero_freq += class_frequencies.majorHeterozygous();
    locus_results.minor_hetero_freq += class_frequencies.minorHeterozygous();
    switch (allele_freq.alleleType()) {
      case AlleleClassType::MINOR_HOMOZYGOUS: ++locus_results.minor_homo_count; break;
      case AlleleClassType::MAJOR_HETEROZYGOUS: ++locus_results.major_hetero_count; break;
      case AlleleClassType::MINOR_HETEROZYGOUS: ++locus_results.minor_hetero_count; break;
      case AlleleClassType::MAJOR_HOMOZYGOUS: ++locus_results.major_homo_count; break;
    }
  }
  return {frequency_vector, locus_results};
}

// ---- RetryCalcResult (kga_analysis_inbreed_calc.cpp:17-68) -------------------------------------

bool RetryCalcResult::checkRetry(double retry) {
  ++retry_count_;
  if (retry_count_ > max_retry_) return true;
  current_retries_.push_back(retry);
  if (current_retries_.size() > min_retry_) {
    current_retries_.pop_front();
    return checkTolerance();
  } else if (current_retries_.size() == min_retry_) {
    return checkTolerance();
  }
  return false;
}

bool RetryCalcResult::checkTolerance() const {
  auto current_entry = current_retries_.begin();
  while (current_entry != current_retries_.end()) {
    auto next_entry = ++current_entry;   // as written in the reference: advances current_entry as well
    if (next_entry == current_retries_.end()) break;
    if (std::fabs(*current_entry - *next_entry) > tolerance_) return false;
    current_entry = next_entry;
  }
  return true;
}

std::optional<InbreedAlgorithm> namedAlgorithm(const std::string& name) {
  if (name == "RitlandLocus") return InbreedAlgorithm::RitlandLocus;
  if (name == "Simple") return InbreedAlgorithm::Simple;
  if (name == "HallME") return InbreedAlgorithm::HallME;
  if (name == "Loglikelihood") return InbreedAlgorithm::Loglikelihood;
  return std::nullopt;
}

// ---- estimators (kga_analysis_inbreed_calc.cpp) ------------------------------------------------

static constexpr double FINAL_ACCURACY_ = 1E-04;
static constexpr double INIT_UPPER_ = 0.5;
static constexpr double INIT_LOWER_ = -0.5;
static constexpr size_t MINIMUM_ITERATIONS_ = 50;
static constexpr size_t MAXIMUM_ITERATIONS_ = 1000;
static constexpr size_t MAX_RETRIES_ = 50;
static constexpr size_t MIN_RETRIES_ = 5;

// The entropy of one per-genome task.  The reference builds a RandomEntropySource (kel_math/kel_distribution.h:25-43: a
// std::mt19937_64 seeded from std::random_device) inside every process* call: seed 0.  A non-zero seed is the reference's
// DeterministicEntropySource (kel_distribution.h:53-70) instead -- same generator, known start -- so that a run can be
// repeated; processResults gives the k-th task it enqueues the seed start_seed + k.  Nothing else differs.
static std::mt19937_64 makeEntropy(uint64_t seed) {
  if (seed == 0) {
    std::random_device rd;
    return std::mt19937_64(rd());
  }
  return std::mt19937_64(seed);
}

std::vector<double> restartDraws(InbreedAlgorithm algorithm, uint64_t start_seed, size_t restarts) {
  std::mt19937_64 entropy_mt = makeEntropy(start_seed);
  std::uniform_real_distribution<> initialize_distribution(INIT_UPPER_, algorithm == InbreedAlgorithm::HallME ? 0.0 : INIT_LOWER_);
  std::vector<double> draws;
  for (size_t r = 0; r < restarts; ++r) draws.push_back(initialize_distribution(entropy_mt));
  return draws;
}

double logLikelihood(double f, const std::vector<AlleleFreqInfo>& data) {
  double log_prob_sum = 0.0;
  static const double small_prob = 1e-10;
  for (const auto& allele_freq : data) {
    switch (allele_freq.alleleType()) {
      case AlleleClassType::MAJOR_HOMOZYGOUS:
      case AlleleClassType::MINOR_HOMOZYGOUS: {
        const double freq_sqd = allele_freq.firstAllele().frequency() * allele_freq.firstAllele().frequency();
        double prob = (f * allele_freq.firstAllele().frequency()) + ((1.0 - f) * freq_sqd);
        prob = std::clamp<double>(prob, small_prob, 1.0);
        log_prob_sum += std::log(prob);
      } break;
      case AlleleClassType::MINOR_HETEROZYGOUS:
      case AlleleClassType::MAJOR_HETEROZYGOUS: {
        double prob = 2 * (1.0 - f) * allele_freq.firstAllele().frequency() * allele_freq.secondAllele().frequency();
        prob = std::clamp<double>(prob, small_prob, 1.0);
        log_prob_sum += std::log(prob);
      } break;
    }
  }
  return log_prob_sum;
}

double neldermead1D(const std::function<double(double)>& objective, double x0, double lb, double ub, double xtol_abs,
                    int maxeval, int* evals) {
  // Simplex of two points; maximise.  Initial step: a quarter of the box, turned inward at a bound.
  int n = 0;
  auto f = [&](double x) { ++n; return objective(x); };
  auto clampx = [&](double x) { return std::min(ub, std::max(lb, x)); };
  double step = (ub - lb) * 0.25;
  double xa = clampx(x0);
  double xb = xa + step;
  if (xb > ub) xb = xa - step;
  xb = clampx(xb);
  double fa = f(xa), fb = f(xb);
  while (n < maxeval) {
    if (fb > fa) { std::swap(xa, xb); std::swap(fa, fb); }   // xa best, xb worst
    if (std::fabs(xa - xb) < xtol_abs) break;
    const double centroid = xa;                 // centroid of all but the worst
    const double xr = clampx(centroid + (centroid - xb));
    const double fr = f(xr);
    if (fr > fa) {
      const double xe = clampx(centroid + 2.0 * (centroid - xb));
      const double fe = f(xe);
      if (fe > fr) { xb = xe; fb = fe; } else { xb = xr; fb = fr; }
    } else if (fr > fb) {
      // better than the worst but not the best: outside contraction
      const double xc = clampx(centroid + 0.5 * (xr - centroid));
      const double fc = f(xc);
      if (fc >= fr) { xb = xc; fb = fc; } else { xb = xr; fb = fr; }
    } else {
      const double xc = centroid + 0.5 * (xb - centroid);   // inside contraction == shrink in one dimension
      const double fc = f(xc);
      xb = xc; fb = fc;
    }
  }
  if (evals) *evals = n;
  return fa >= fb ? xa : xb;
}

LocusResults processLogLikelihood(const std::string& genome_id, const ContigDB& contig, int super_pop,
                                  const ContigDB& locus_list, uint64_t start_seed) {
  auto snp_contig_ptr = contig.viewFilter(snpFilter);   // computed and unused, as in the reference (:160)
  (void)snp_contig_ptr;
  std::mt19937_64 entropy_mt = makeEntropy(start_seed);
  std::uniform_real_distribution<> initialize_distribution(INIT_UPPER_, INIT_LOWER_);   // (0.5, -0.5) as written
  auto [frequency_vector, locus_results] = generateFrequencies(genome_id, contig, super_pop, locus_list);

  double updated_coefficient = 0.0;
  RetryCalcResult retry_results(FINAL_ACCURACY_, MIN_RETRIES_, MAX_RETRIES_);
  do {
    const double initial_f = initialize_distribution(entropy_mt);
    const auto& data = frequency_vector;
    updated_coefficient = neldermead1D([&data](double f) { return logLikelihood(f, data); }, initial_f, -1.0, 1.0, 1e-06, 500, nullptr);
  } while (!retry_results.checkRetry(updated_coefficient));
  if (retry_results.retries() >= MAX_RETRIES_) updated_coefficient = 0.0;
  locus_results.inbred_allele_sum = updated_coefficient;
  return locus_results;
}

LocusResults processHallME(const std::string& genome_id, const ContigDB& contig, int super_pop, const ContigDB& locus_list,
                           uint64_t start_seed) {
  auto snp_contig_ptr = contig.viewFilter(snpFilter);   // unused, as in the reference (:232)
  (void)snp_contig_ptr;
  std::mt19937_64 entropy_mt = makeEntropy(start_seed);
  std::uniform_real_distribution<> initialize_distribution(INIT_UPPER_, 0);   // (0.5, 0) as written
  auto [frequency_vector, locus_results] = generateFrequencies(genome_id, contig, super_pop, locus_list);

  double updated_coefficient = 0.0;
  double inbreed_coefficient;
  RetryCalcResult retry_results(FINAL_ACCURACY_, MIN_RETRIES_, MAX_RETRIES_);
  do {
    updated_coefficient = initialize_distribution(entropy_mt);
    RetryCalcResult converge_retry(FINAL_ACCURACY_, MINIMUM_ITERATIONS_, MAXIMUM_ITERATIONS_);
    do {
      inbreed_coefficient = updated_coefficient;
      double expectation_sum = 0.0;
      for (const auto& allele_freq : frequency_vector) {
        switch (allele_freq.alleleType()) {
          case AlleleClassType::MAJOR_HOMOZYGOUS:
          case AlleleClassType::MINOR_HOMOZYGOUS: {
            const double denominator = (inbreed_coefficient + ((1.0 - inbreed_coefficient) * allele_freq.firstAllele().frequency()));
            if (denominator != 0) expectation_sum += inbreed_coefficient / denominator;
          } break;
          case AlleleClassType::MAJOR_HETEROZYGOUS:
          case AlleleClassType::MINOR_HETEROZYGOUS:
            break;
        }
      }
      updated_coefficient = expectation_sum / static_cast<double>(frequency_vector.size());
    } while (!converge_retry.checkRetry(updated_coefficient));
  } while (!retry_results.checkRetry(updated_coefficient));
  if (retry_results.retries() >= MAX_RETRIES_) updated_coefficient = 0.0;
  locus_results.inbred_allele_sum = updated_coefficient;
  return locus_results;
}

LocusResults processSimple(const std::string& genome_id, const ContigDB& contig, int super_pop, const ContigDB& locus_list) {
  auto [frequency_vector, locus_results] = generateFrequencies(genome_id, contig, super_pop, locus_list);
  const bool calc_hetero = false;
  double heterozygous_inbreeding = 0.0;
  double homozygous_inbreeding = 0.0;
  if (locus_results.total_allele_count > 0) {
    const auto observed_heterozygous = static_cast<double>(locus_results.major_hetero_count + locus_results.minor_hetero_count);
    const auto observed_homozygous = static_cast<double>(locus_results.minor_homo_count + locus_results.major_homo_count);
    const auto expected_heterozygous = locus_results.minor_hetero_freq + locus_results.major_hetero_freq;
    const auto expected_homozygous = locus_results.minor_homo_freq + locus_results.major_homo_freq;
    heterozygous_inbreeding = 1.0 - (observed_heterozygous / expected_heterozygous);
    homozygous_inbreeding = (observed_homozygous - expected_homozygous) /
                            (static_cast<double>(locus_results.total_allele_count) - expected_homozygous);
  }
  locus_results.inbred_allele_sum = calc_hetero ? heterozygous_inbreeding : homozygous_inbreeding;
  return locus_results;
}

LocusResults processRitlandLocus(const std::string& genome_id, const ContigDB& contig, int super_pop, const ContigDB& locus_list) {
  constexpr double minimum_frequency = 0.001;
  size_t sum_allele = 0;
  double locus_allele_sum = 0.0;
  auto [frequency_vector, locus_results] = generateFrequencies(genome_id, contig, super_pop, locus_list);
  for (const auto& allele_freq : frequency_vector) {
    switch (allele_freq.alleleType()) {
      case AlleleClassType::MAJOR_HOMOZYGOUS:
      case AlleleClassType::MINOR_HOMOZYGOUS:
        if (allele_freq.firstAllele().frequency() > minimum_frequency) {
          const double ratio = (1.0 / allele_freq.firstAllele().frequency());
          locus_allele_sum += ratio;
          locus_allele_sum -= 1.0;
          ++sum_allele;
        }
        break;
      case AlleleClassType::MAJOR_HETEROZYGOUS:
      case AlleleClassType::MINOR_HETEROZYGOUS:
        locus_allele_sum -= 1.0;
        ++sum_allele;
        break;
    }
  }
  locus_results.inbred_allele_sum = (sum_allele > 0 ? locus_allele_sum / static_cast<double>(sum_allele) : 0.0);
  return locus_results;
}

// ---- drivers (kga_analysis_inbreed_diploid.cpp) ------------------------------------------------

ResultsMap processResults(const PopulationDB& diploid_population, const std::string& contig_id,
                          const std::map<int, std::shared_ptr<const ContigDB>>& locus_map,
                          const std::map<std::string, int>& super_pop_of_genome, const InbreedingParameters& params) {
  ResultsMap results_map;
  auto algorithm_opt = namedAlgorithm(params.algorithm);
  if (!algorithm_opt) return results_map;
  const InbreedAlgorithm algo = algorithm_opt.value();

  WorkflowThreads thread_pool(poolThreads(diploid_population.getMap().size()));
  std::vector<std::future<LocusResults>> future_vector;
  for (const auto& [genome_id, genome_ptr] : diploid_population.getMap()) {
    auto contig_opt = genome_ptr->getContig(contig_id);
    if (!contig_opt) continue;
    auto sp_it = super_pop_of_genome.find(genome_id);
    if (sp_it == super_pop_of_genome.end()) continue;    // no PED record (:127-131)
    auto locus_result = locus_map.find(sp_it->second);
    if (locus_result == locus_map.end()) continue;
    const int super_pop = sp_it->second;
    std::shared_ptr<const ContigDB> contig = contig_opt.value();
    std::shared_ptr<const ContigDB> locus_list = locus_result->second;
    const std::string gid = genome_id;
    const uint64_t seed = params.start_seed ? params.start_seed + future_vector.size() : 0;
    future_vector.push_back(thread_pool.enqueueFuture([=]() -> LocusResults {
      switch (algo) {
        case InbreedAlgorithm::RitlandLocus: return processRitlandLocus(gid, *contig, super_pop, *locus_list);
        case InbreedAlgorithm::Simple: return processSimple(gid, *contig, super_pop, *locus_list);
        case InbreedAlgorithm::HallME: return processHallME(gid, *contig, super_pop, *locus_list, seed);
        case InbreedAlgorithm::Loglikelihood: return processLogLikelihood(gid, *contig, super_pop, *locus_list, seed);
      }
      return LocusResults();
    }));
  }
  for (auto& future : future_vector) {
    auto locus_results = future.get();
    results_map[locus_results.genome] = locus_results;
  }
  return results_map;
}

std::vector<std::pair<std::string, ResultsMap>> populationInbreeding(const PopulationDB& reference_population,
                                                                     const PopulationDB& diploid_population,
                                                                     const std::map<std::string, int>& super_pop_of_genome,
                                                                     const InbreedingParameters& params) {
  std::vector<std::pair<std::string, ResultsMap>> columns;
  if (reference_population.getMap().size() != 1) return columns;
  const auto& genome_ptr = reference_population.getMap().begin()->second;
  if (genome_ptr->getMap().size() != 1) return columns;
  const auto& [contig_id, contig_ptr] = *genome_ptr->getMap().begin();

  InbreedingParameters local_params = params;
  std::vector<uint64_t> locii_vector = getLociiCount(*contig_ptr, ALL, local_params.locii);
  if (locii_vector.empty()) return columns;   // the reference calls back() on an empty vector here (UB)
  local_params.locii.upper_offset = locii_vector.back();

  while (local_params.locii.upper_offset < params.locii.upper_offset && locii_vector.size() >= 100) {
    // populationInbreedingSample: one locus list per super population (getPopulationLocus, _locus.cpp:225-256)
    std::map<int, std::shared_ptr<const ContigDB>> locus_map;
    for (int sp = 0; sp < SUPER_POP_COUNT; ++sp) locus_map[sp] = getLocusList(*contig_ptr, sp, local_params.locii);
    ResultsMap results_map = processResults(diploid_population, contig_id, locus_map, super_pop_of_genome, local_params);

    std::stringstream ss;
    ss << contig_id << "_" << local_params.locii.lower_offset << "_" << local_params.locii.upper_offset;
    columns.emplace_back(ss.str(), std::move(results_map));

    local_params.locii.lower_offset = local_params.locii.upper_offset;
    locii_vector = getLociiCount(*contig_ptr, ALL, local_params.locii);
    if (locii_vector.empty()) break;
    local_params.locii.upper_offset = locii_vector.back();
  }
  return columns;
}

// ---- synthetic population (kga_analysis_inbreed_syngen.cpp) ------------------------------------

std::string generateSyntheticGenomeId(double inbreeding, const std::string& super_population, size_t counter) {
  std::stringstream s;
  s << std::fixed;
  s << std::setprecision(0);
  if (inbreeding >= 0) s << super_population << "_" << (inbreeding * 1000000.0) << "_" << counter;
  else s << super_population << "_N" << (-inbreeding * 1000000.0) << "_" << counter;
  return s.str();
}

std::pair<bool, double> generateInbreeding(const std::string& genome_id) {
  bool valid_value = false;
  bool negative = false;
  double inbreed_coefficient = -1000.0;
  auto first_pos = genome_id.find_first_of("N");
  if (first_pos == std::string::npos) first_pos = genome_id.find_first_of("_");
  else negative = true;
  if (first_pos != std::string::npos) {
    ++first_pos;
    auto second_pos = genome_id.find_first_of("_", first_pos);
    if (second_pos != std::string::npos) {
      const std::string coefficient_string = genome_id.substr(first_pos, second_pos - first_pos);
      try {
        const size_t coefficent = std::stod(coefficient_string);
        inbreed_coefficient = static_cast<double>(coefficent) / 1000000.0;
        valid_value = true;
        if (negative) inbreed_coefficient = -1.0 * inbreed_coefficient;
      } catch (std::exception&) {
      }
    }
  }
  return {valid_value, inbreed_coefficient};
}

std::shared_ptr<PopulationDB> generateSyntheticPopulation(double lower_inbreeding, double upper_inbreeding,
                                                          double step_inbreeding, int super_pop, const ContigDB& locus_list,
                                                          uint64_t seed) {
  auto synthetic_pop_ptr = std::make_shared<PopulationDB>("SyntheticInbreedingPopulation");
  std::mt19937_64 entropy_mt = makeEntropy(seed);
  std::uniform_real_distribution<> unit_distribution(0.0, 1.0);
  std::uniform_int_distribution<int> random_boolean(0, 1);

  size_t counter = 0;
  std::vector<std::pair<std::string, double>> inbreeding_vector;
  double inbreeding = lower_inbreeding;
  while (inbreeding <= (upper_inbreeding + 0.000001)) {
    inbreeding_vector.emplace_back(generateSyntheticGenomeId(inbreeding, superPopName(super_pop), counter), inbreeding);
    inbreeding += step_inbreeding;
    ++counter;
  }

  for (const auto& [genome_id, inbreeding_coefficient] : inbreeding_vector) {
    std::vector<std::string> genome_vector{genome_id};
    for (const auto& [offset, offset_ptr] : locus_list.getMap()) {
      const OffsetDBArray variant_vec = offset_ptr->getVariantArray();
      AlleleFreqVector freq_vector(variant_vec, super_pop);
      const double class_selection = unit_distribution(entropy_mt);
      AlleleClassFrequencies class_freqs = freq_vector.alleleClassFrequencies(inbreeding_coefficient);
      switch (freq_vector.selectAlleleClass(class_selection, class_freqs)) {
        case AlleleClassType::MINOR_HOMOZYGOUS: {
          const double allele_selection = unit_distribution(entropy_mt);
          auto selected_allele = freq_vector.selectMinorHomozygous(allele_selection, class_freqs);
          if (selected_allele) {
            synthetic_pop_ptr->addVariant(selected_allele->allele()->clonePhase(VariantPhase::DIPLOID_PHASE_A), genome_vector);
            synthetic_pop_ptr->addVariant(selected_allele->allele()->clonePhase(VariantPhase::DIPLOID_PHASE_B), genome_vector);
          }
        } break;
        case AlleleClassType::MAJOR_HETEROZYGOUS: {
          const double allele_selection = unit_distribution(entropy_mt);
          auto selected_allele = freq_vector.selectMajorHeterozygous(allele_selection, class_freqs);
          if (selected_allele) {
            const VariantPhase phase = random_boolean(entropy_mt) ? VariantPhase::DIPLOID_PHASE_A : VariantPhase::DIPLOID_PHASE_B;
            synthetic_pop_ptr->addVariant(selected_allele->allele()->clonePhase(phase), genome_vector);
          }
        } break;
        case AlleleClassType::MINOR_HETEROZYGOUS: {
          const double allele_selection = unit_distribution(entropy_mt);
          auto selected_alleles = freq_vector.selectMinorHeterozygous(allele_selection, class_freqs);
          if (selected_alleles) {
            synthetic_pop_ptr->addVariant(selected_alleles->first.allele()->clonePhase(VariantPhase::DIPLOID_PHASE_A), genome_vector);
            synthetic_pop_ptr->addVariant(selected_alleles->second.allele()->clonePhase(VariantPhase::DIPLOID_PHASE_B), genome_vector);
          }
        } break;
        case AlleleClassType::MAJOR_HOMOZYGOUS:
          break;
      }
    }
  }
  return synthetic_pop_ptr;
}

}  // namespace kgo
