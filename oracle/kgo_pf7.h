// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
// The Pf7 sample resources and what the PfEMP package does with them, restated from
//   kgl_genomics/kgl_parser/kgl_square_parser.cpp:157-197              SquareTextParser::parseFlatFile
//   kgl_genomics/kgl_parser/kgl_pf7_sample_parser.{h,cpp}              Pf7SampleRecord / Pf7SampleResource / ParsePf7Sample
//   kgl_genomics/kgl_parser/kgl_pf7_fws_parser.{h,cpp}                 Pf7FwsResource / ParsePf7Fws
//   kgl_genomics/kgl_parser/kgl_Pf7_physical_distance.{h,cpp}          LocationCoordinates / Pf7SampleLocation
//   kga_analytic/kga_analysis_library/kga_analysis_lib_PfFilter.cpp    FilterPf7::qualityFilter, filterPassQCGenomes
//   kga_analytic/kga_PfEMP/kga_analysis_PfEMP_heterozygous.{h,cpp}     HeteroHomoZygous (location summary, F_IS, writers)
#ifndef KGO_PF7_H
#define KGO_PF7_H

#include <set>

#include "kgo_analysis.h"

namespace kgo {

// kgl_pf7_sample_parser.h:22-45
struct Pf7SampleRecord {
  std::string Pf7Sample_id, study_, country_, location1_, country_latitude_, country_longitude_, location1_latitude_, location1_longitude_,
      year_, ena_, all_samples_, population_, callable_, qc_pass_, qc_fail_reason_, sample_type_, sample_in_pf6_;
  bool pass() const;   // toupper(qc_pass_) == "TRUE"
};
using Pf7SampleMap = std::map<std::string, Pf7SampleRecord>;

// kgl_pf7_sample_parser.cpp:59-119 + :15-36; false when the file does not have 17 columns on every row (or no row).
bool parsePf7SampleFile(const std::string& file_name, Pf7SampleMap& out);

struct Pf7FwsRecord { std::string Pf7Sample_id; double FWS_value{0.0}; };
using Pf7FwsMap = std::map<std::string, Pf7FwsRecord>;
bool parsePf7FwsFile(const std::string& file_name, Pf7FwsMap& out);                       // kgl_pf7_fws_parser.cpp:107-166, :13-33
double getFWS(const Pf7FwsMap& fws, const std::string& genome);                           // :35-49
std::vector<std::string> filterFWS(const Pf7FwsMap& fws, bool greater_equal, double threshold, const std::vector<std::string>& samples);   // :73-104
constexpr double MONOCLONAL_FWS_THRESHOLD = 0.95;

enum class LocationType { City, Country };
// kgl_Pf7_physical_distance.h:28-80
struct LocationCoordinates {
  LocationCoordinates(std::string location, LocationType type, const Pf7SampleRecord& sample_record);
  double distance_km(const LocationCoordinates& other) const;   // kgl_Pf7_physical_distance.cpp:86-105
  void addSample(const Pf7SampleRecord& sample_record);         // :117-123
  double latitude_{0.0}, longitude_{0.0};
  std::pair<std::string, LocationType> location_;
  std::string city_, country_, region_;
  std::vector<std::string> sample_id_vec_;
  std::map<std::string, size_t> studies_;
};
using SampleLocationMap = std::map<std::string, LocationCoordinates>;

class Pf7SampleLocation {   // :133-375
 public:
  explicit Pf7SampleLocation(const Pf7SampleMap& samples);
  const SampleLocationMap& locationMap() const { return location_map_; }
  std::vector<std::string> locationRadius(const std::string& location, double radius, bool all = false) const;
  std::vector<std::string> sampleRadius(const std::string& location, double radius, bool all = false) const;
 private:
  double calculateDistance(const std::string& a, const std::string& b) const;
  SampleLocationMap location_map_;
  std::map<std::string, std::map<std::string, double>> distance_cache_;
};

// FilterPf7::qualityFilter's genome part (kga_analysis_lib_PfFilter.cpp:26-58) followed by squareContigs (:107-110,
// kgl_variant_db_population.cpp:258-295): the genomes that pass QC (filter_qc) and are monoclonal (filter_fws), each
// holding every contig any of them holds.  The variant-level filter of :63-67 is PopulationDB::viewFilter(p7VariantFilter),
// applied by the caller first where wanted.
std::shared_ptr<PopulationDB> pf7GenomeFilter(const PopulationDB& population, const Pf7SampleMap& samples, const Pf7FwsMap& fws, bool filter_qc,
                                               bool filter_fws, double fws_threshold);

// kga_analysis_PfEMP_heterozygous.h:85-106
struct LocationSummary {
  std::string location_;
  LocationType location_type_{LocationType::City};
  std::string city_, country_, region_;
  double radius_km_{0.0};
  size_t radii_samples_{0}, radii_samples_OK_{0};
  std::map<std::string, size_t> studies_;
  double monoclonal_Fst_{0.0}, hom_het_ratio_{0.0};
  size_t total_variants_{0};
  double variant_rate_{0.0};
  size_t homozygous_reference_alleles_{0}, heterozygous_reference_minor_alleles_{0}, homozygous_minor_alleles_{0}, heterozygous_minor_alleles_{0},
      snp_count_{0}, indel_count_{0};
};
using LocationSummaryMap = std::map<std::string, LocationSummary>;

class HeteroHomoZygous {
 public:
  // kga_analysis_PfEMP_heterozygous.cpp:16-57 (genomes without a sample record are skipped)
  void analyzeVariantPopulation(const PopulationDB& population, const Pf7FwsMap& fws, const Pf7SampleMap& samples);
  VariantAnalysisType aggregateResults(const std::vector<std::string>& sample_vector) const;                      // :229-263
  LocationSummaryMap location_summary(const Pf7SampleMap& samples, const Pf7SampleLocation& distance, double radius_km, const Pf7FwsMap& fws) const;   // :266-360
  void UpdateSampleLocation(const LocationSummaryMap& summary);                                                   // :363-414
  void write_variant_results(const std::string& file_name, const LocationSummaryMap& summary);                    // :108-226
  void write_location_results(const std::string& file_name, const LocationSummaryMap& summary) const;             // :418-510
 private:
  struct Obj {
    Pf7SampleRecord sample_record_;
    double fws_value_{0.0}, fis_value_{0.0};
    std::map<std::string, VariantAnalysisType> analysis_map_;
  };
  std::map<std::string, Obj> variant_analysis_map_;
  static constexpr size_t MINIMUM_LOCATION_SAMPLES_ = 20;
};

constexpr double SAMPLE_LOCATION_RADIUS = 0.0;   // kga_analysis_PfEMP.h:64

}  // namespace kgo

#endif  // KGO_PF7_H
