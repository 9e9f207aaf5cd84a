// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
// Allele-count sweeps restated from
//   kgl_genomics/kgl_variant_db/kgl_variant_db_variant.{h,cpp}      VariantDBVariant
//   kga_analytic/kga_PfEMP/kga_analysis_PfEMP_FWS.{h,cpp}            CalcFWS
//   kga_analytic/kga_PfEMP/kga_analysis_PfEMP_heterozygous.{h,cpp}   HeteroHomoZygous
//   kgl_genomics/kgl_variant_filter/kgl_variant_filter_Pf7.cpp:20-66 P7FrequencyFilter
#ifndef KGO_ANALYSIS_H
#define KGO_ANALYSIS_H

#include <array>

#include "kgo_core.h"

namespace kgo {

// kgl_variant_db_variant.h:26-41
struct AlleleSummmary {
  size_t referenceHomozygous_{0};
  size_t minorHeterozygous_{0};
  size_t minorHomozygous_{0};
  void operator+=(const AlleleSummmary& rhs) {
    minorHomozygous_ += rhs.minorHomozygous_;
    referenceHomozygous_ += rhs.referenceHomozygous_;
    minorHeterozygous_ += rhs.minorHeterozygous_;
  }
};

using VariantDBVariantIndex = std::map<std::string, std::pair<VariantPtr, size_t>>;
using VariantDBGenomeIndex = std::map<std::string, size_t>;
using VariantDBGenomeData = std::vector<std::pair<std::string, std::vector<uint8_t>>>;

class VariantDBVariant {
 public:
  explicit VariantDBVariant(const std::shared_ptr<const PopulationDB>& population) { createVariantDB(population); }
  const VariantDBGenomeIndex& genomeMap() const { return genome_index_; }
  const VariantDBVariantIndex& variantMap() const { return variant_index_; }
  const VariantDBGenomeData& genomeData() const { return genome_data_; }
  AlleleSummmary summaryByVariant(const VariantPtr& variant) const;   // kgl_variant_db_variant.cpp:126-178
  AlleleSummmary summaryByGenome(const std::string& genome) const;    // :180-231
  AlleleSummmary populationSummary() const;                           // :234-279
  size_t warnings() const { return warnings_; }                       // conservation-identity violations seen
 private:
  void createVariantDB(const std::shared_ptr<const PopulationDB>& population);   // :11-123
  VariantDBGenomeIndex genome_index_;
  VariantDBVariantIndex variant_index_;
  VariantDBGenomeData genome_data_;
  mutable size_t warnings_ = 0;
};

// P7FrequencyFilter on the "AF" INFO field: missing AF passes (kgl_variant_filter_Pf7.cpp:20-66).
bool p7FrequencyFilter(const Variant& v, double freq_cutoff);

constexpr size_t FWS_FREQUENCY_ARRAY_SIZE = 11;
using FwsFrequencyArray = std::array<AlleleSummmary, FWS_FREQUENCY_ARRAY_SIZE>;
using GenomeFWSMap = std::map<std::string, FwsFrequencyArray>;
using VariantFWSMap = std::map<std::string, AlleleSummmary>;

class CalcFWS {
 public:
  void calcFwsStatistics(const std::shared_ptr<const PopulationDB>& population);   // _FWS.cpp:15-38
  const GenomeFWSMap& getGenomeMap() const { return genome_fws_map_; }
  const VariantFWSMap& getVariantMap() const { return variant_fws_map_; }
  static std::pair<double, double> getFrequency(size_t bin);                        // :104-145
 private:
  void updateGenomeFWSMap(const std::shared_ptr<const PopulationDB>& freq_population, size_t freq_bin);   // :72-101
  void updateVariantFWSMap(const std::shared_ptr<const PopulationDB>& population);                         // :41-70
  GenomeFWSMap genome_fws_map_;
  VariantFWSMap variant_fws_map_;
};

// kga_analysis_PfEMP_heterozygous.h:22-32
struct VariantAnalysisType {
  size_t total_variants_{0};
  size_t snp_count_{0};
  size_t indel_count_{0};
  size_t homozygous_minor_alleles_{0};
  size_t heterozygous_minor_alleles_{0};
  size_t heterozygous_reference_minor_alleles_{0};
  size_t homozygous_reference_alleles_{0};
};

// kga_analysis_PfEMP_heterozygous.cpp:61-105
void updateVariantAnalysisType(const OffsetDB& offset, VariantAnalysisType& record);
// :16-57 without the Pf7 sample-metadata join: genome -> contig -> counters.
std::map<std::string, std::map<std::string, VariantAnalysisType>> analyzeVariantPopulation(const PopulationDB& population);
// Wright's F_IS of UpdateSampleLocation (:400-406) for one genome against a location aggregate.
double wrightsFIS(const VariantAnalysisType& location, const VariantAnalysisType& genome);

}  // namespace kgo

#endif  // KGO_ANALYSIS_H
