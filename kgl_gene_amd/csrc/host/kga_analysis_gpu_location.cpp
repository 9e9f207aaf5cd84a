// GpuHeteroHomoZygous: the per-genome x contig counters joined with the Pf7 sample resources -- the per-site summaries,
// location F_IS and the files HeteroHomoZygous writes (kga_analytic/kga_PfEMP/kga_analysis_PfEMP_heterozygous.cpp:108-510)
// -- and GpuAlleleAnalysis' genome-level filter.  The counters themselves come off the device (kga_analysis_gpu_allele.cpp);
// everything here is per genome or per site.
#include <fstream>
#include <set>

#include "kga_analysis_gpu_allele.h"

namespace kga = kellerberrin::genome::analysis;
namespace kgl = kellerberrin::genome;

namespace {

void accumulate(kga::VariantAnalysisType& sum, const kga::VariantAnalysisType& r) {
  sum.total_variants_ += r.total_variants_;
  sum.snp_count_ += r.snp_count_;
  sum.indel_count_ += r.indel_count_;
  sum.homozygous_minor_alleles_ += r.homozygous_minor_alleles_;
  sum.heterozygous_minor_alleles_ += r.heterozygous_minor_alleles_;
  sum.heterozygous_reference_minor_alleles_ += r.heterozygous_reference_minor_alleles_;
  sum.homozygous_reference_alleles_ += r.homozygous_reference_alleles_;
}

// (a;a) offsets per heterozygous offset, 0 without any
double homHetRatio(const kga::VariantAnalysisType& r) {
  const size_t heterozygous = r.heterozygous_reference_minor_alleles_ + r.heterozygous_minor_alleles_;
  return heterozygous > 0 ? static_cast<double>(r.homozygous_minor_alleles_) / static_cast<double>(heterozygous) : 0.0;
}

// the seven counters of one "Contig" block
void writeCounters(std::ofstream& out, char delimiter, const kga::VariantAnalysisType& r) {
  out << delimiter << r.total_variants_ << delimiter << r.homozygous_reference_alleles_ << delimiter << r.heterozygous_reference_minor_alleles_
      << delimiter << r.homozygous_minor_alleles_ << delimiter << r.heterozygous_minor_alleles_ << delimiter << r.snp_count_ << delimiter
      << r.indel_count_;
}

}  // namespace

bool kga::GpuAlleleAnalysis::keepGenome(const GenomeId_t& genome_id) const {
  if (!pf7_sample_ptr_) return true;
  if (filter_qc_) {                               // a genome without a sample record does not pass
    const auto found = pf7_sample_ptr_->getMap().find(genome_id);
    if (found == pf7_sample_ptr_->getMap().end() || !found->second.pass()) return false;
  }
  if (filter_fws_) {                              // nor does one without a published FWS value
    const auto found = pf7_fws_ptr_->getMap().find(genome_id);
    if (found == pf7_fws_ptr_->getMap().end() || !(found->second.FWS_value >= fws_monoclonal_threshold_)) return false;
  }
  return true;
}

void kga::GpuHeteroHomoZygous::setResources(std::shared_ptr<const Pf7SampleResource> sample_ptr, std::shared_ptr<const Pf7FwsResource> fws_ptr,
                                            std::shared_ptr<const Pf7SampleLocation> physical_distance_ptr) {
  pf7_sample_ptr_ = std::move(sample_ptr);
  pf7_fws_ptr_ = std::move(fws_ptr);
  pf7_physical_distance_ptr_ = std::move(physical_distance_ptr);
}

double kga::GpuHeteroHomoZygous::wrightsInbreeding(const VariantAnalysisType& location, const VariantAnalysisType& genome) {
  if (location.total_variants_ == 0 || genome.total_variants_ == 0) return 0.0;
  const double expected = static_cast<double>(location.heterozygous_minor_alleles_ + location.heterozygous_reference_minor_alleles_) /
                          static_cast<double>(location.total_variants_);
  const double observed = static_cast<double>(genome.heterozygous_minor_alleles_ + genome.heterozygous_reference_minor_alleles_) /
                          static_cast<double>(genome.total_variants_);
  return (expected - observed) / expected;
}

kga::VariantAnalysisType kga::GpuHeteroHomoZygous::aggregateResults(const std::vector<GenomeId_t>& sample_vector) const {
  VariantAnalysisType summary;
  const std::set<GenomeId_t> once(sample_vector.begin(), sample_vector.end());
  for (const auto& genome_id : once) {
    const auto found = variant_analysis_map_.find(genome_id);
    if (found == variant_analysis_map_.end()) continue;
    if (pf7_sample_ptr_ && !pf7_sample_ptr_->getMap().contains(genome_id)) continue;   // no sample record, no analysis record (:22-28)
    for (const auto& [contig_id, record] : found->second) accumulate(summary, record);
  }
  return summary;
}

kga::GpuLocationSummaryMap kga::GpuHeteroHomoZygous::locationSummary(double radius_km) const {
  GpuLocationSummaryMap summary_map;
  if (!pf7_sample_ptr_ || !pf7_fws_ptr_ || !pf7_physical_distance_ptr_) return summary_map;
  for (const auto& [location, coordinates] : pf7_physical_distance_ptr_->locationMap()) {
    const std::vector<GenomeId_t> radii_samples = pf7_physical_distance_ptr_->sampleRadius(location, radius_km);
    std::vector<GenomeId_t> radii_passed;
    for (const auto& sample : radii_samples) {
      const auto record = pf7_sample_ptr_->getMap().find(sample);
      if (record != pf7_sample_ptr_->getMap().end() && record->second.pass()) radii_passed.push_back(sample);
    }
    const VariantAnalysisType aggregated = aggregateResults(radii_samples);
    GpuLocationSummary s;
    s.location_ = location;
    s.location_type_ = coordinates.location().second;
    s.city_ = coordinates.city();
    s.country_ = coordinates.city();            // as the reference fills it (:336): the "Country" column repeats the site
    s.region_ = coordinates.region();
    s.radius_km_ = radius_km;
    s.radii_samples_ = radii_samples.size();
    s.radii_samples_OK_ = radii_passed.size();
    s.studies_ = coordinates.locationStudies();
    if (!radii_passed.empty())
      s.monoclonal_Fst_ = static_cast<double>(pf7_fws_ptr_->filterFWS(FwsFilterType::GREATER_EQUAL, Pf7FwsResource::MONOCLONAL_FWS_THRESHOLD, radii_passed).size()) /
                          static_cast<double>(radii_passed.size());
    s.hom_het_ratio_ = homHetRatio(aggregated);
    s.total_variants_ = aggregated.total_variants_;
    if (!radii_samples.empty()) s.variant_rate_ = static_cast<double>(aggregated.total_variants_) / static_cast<double>(radii_samples.size());
    s.homozygous_reference_alleles_ = aggregated.homozygous_reference_alleles_;
    s.heterozygous_reference_minor_alleles_ = aggregated.heterozygous_reference_minor_alleles_;
    s.homozygous_minor_alleles_ = aggregated.homozygous_minor_alleles_;
    s.heterozygous_minor_alleles_ = aggregated.heterozygous_minor_alleles_;
    s.snp_count_ = aggregated.snp_count_;
    s.indel_count_ = aggregated.indel_count_;
    summary_map.emplace(location, std::move(s));
  }
  return summary_map;
}

std::map<kgl::GenomeId_t, double> kga::GpuHeteroHomoZygous::locationInbreeding(const GpuLocationSummaryMap& location_summary) const {
  std::map<GenomeId_t, double> inbreeding;
  if (!pf7_sample_ptr_) return inbreeding;
  for (const auto& [genome_id, contig_map] : variant_analysis_map_) {
    const auto sample = pf7_sample_ptr_->getMap().find(genome_id);
    if (sample == pf7_sample_ptr_->getMap().end()) continue;
    const std::string& city = sample->second.location1_;
    const std::string& country = sample->second.country_;
    auto where = location_summary.find(city);
    if (where == location_summary.end()) {
      ExecEnv::log().error("GpuHeteroHomoZygous::locationInbreeding; Unable to find the location record for sample/genome city: {}", city);
      continue;
    }
    if (where->second.radii_samples_OK_ < MINIMUM_LOCATION_SAMPLES_) {     // too few samples at the site: the country stands in
      where = location_summary.find(country);
      if (where == location_summary.end()) {
        ExecEnv::log().error("GpuHeteroHomoZygous::locationInbreeding; Unable to find the location record for sample/genome country: {}", country);
        continue;
      }
    }
    VariantAnalysisType place;
    place.total_variants_ = where->second.total_variants_;
    place.heterozygous_minor_alleles_ = where->second.heterozygous_minor_alleles_;
    place.heterozygous_reference_minor_alleles_ = where->second.heterozygous_reference_minor_alleles_;
    inbreeding[genome_id] = wrightsInbreeding(place, aggregateResults({genome_id}));
  }
  return inbreeding;
}

bool kga::GpuHeteroHomoZygous::writeSampleResults(const std::string& file_name, const GpuLocationSummaryMap& location_summary) const {
  std::ofstream out(file_name);
  if (!out.good()) {
    ExecEnv::log().error("GpuHeteroHomoZygous::writeSampleResults; Unable to open results file: {}", file_name);
    return false;
  }
  // Every genome holds every contig (PopulationDB::squareContigs after the filters, kga_analysis_lib_PfFilter.cpp:107-110).
  std::set<std::string> contigs;
  std::vector<GenomeId_t> genomes;
  for (const auto& [genome_id, contig_map] : variant_analysis_map_) {
    for (const auto& [contig_id, record] : contig_map) contigs.insert(contig_id);
    if (pf7_sample_ptr_->getMap().contains(genome_id)) genomes.push_back(genome_id);
    else ExecEnv::log().error("GpuHeteroHomoZygous::writeSampleResults; Unexpected, could not find sample record for genome:{}", genome_id);
  }
  if (genomes.empty()) return out.good();
  const std::map<GenomeId_t, double> inbreeding = locationInbreeding(location_summary);

  out << "Genome" << CSV_DELIMITER_ << "FWS" << CSV_DELIMITER_ << "FIS (inbreed)" << CSV_DELIMITER_ << "City" << CSV_DELIMITER_ << "Country"
      << CSV_DELIMITER_ << "Region" << CSV_DELIMITER_ << "Study" << CSV_DELIMITER_ << "Year" << CSV_DELIMITER_ << "Hom/Het";
  for (size_t block = 0; block <= contigs.size(); ++block)
    out << CSV_DELIMITER_ << "Contig" << CSV_DELIMITER_ << "Variant Count" << CSV_DELIMITER_ << "Hom Ref (A;A)" << CSV_DELIMITER_
        << "Het Ref Minor (A;a)" << CSV_DELIMITER_ << "Hom Minor (a;a)" << CSV_DELIMITER_ << "Het Diff Minor (a;b)" << CSV_DELIMITER_ << "SNP"
        << CSV_DELIMITER_ << "Indel";
  out << '\n';

  const VariantAnalysisType nothing;
  for (const auto& genome_id : genomes) {
    const Pf7SampleRecord& sample = pf7_sample_ptr_->getMap().at(genome_id);
    const auto& contig_map = variant_analysis_map_.at(genome_id);
    const VariantAnalysisType combined = aggregateResults({genome_id});
    const auto fis = inbreeding.find(genome_id);
    const auto site = location_summary.find(sample.location1_);
    out << genome_id << CSV_DELIMITER_ << pf7_fws_ptr_->getFWS(genome_id) << CSV_DELIMITER_ << (fis == inbreeding.end() ? 0.0 : fis->second)
        << CSV_DELIMITER_ << sample.location1_ << CSV_DELIMITER_ << sample.country_ << CSV_DELIMITER_
        << (site == location_summary.end() ? std::string() : site->second.region_) << CSV_DELIMITER_ << sample.study_ << CSV_DELIMITER_
        << sample.year_ << CSV_DELIMITER_ << homHetRatio(combined);
    out << CSV_DELIMITER_ << "Combined";
    writeCounters(out, CSV_DELIMITER_, combined);
    for (const auto& contig_id : contigs) {
      const auto found = contig_map.find(contig_id);
      out << CSV_DELIMITER_ << contig_id;
      writeCounters(out, CSV_DELIMITER_, found == contig_map.end() ? nothing : found->second);
    }
    out << '\n';
  }
  return out.good();
}

bool kga::GpuHeteroHomoZygous::writeLocationResults(const std::string& file_name, const GpuLocationSummaryMap& location_summary) const {
  std::ofstream out(file_name);
  if (!out.good()) {
    ExecEnv::log().error("GpuHeteroHomoZygous::writeLocationResults; Unable to open results file: {}", file_name);
    return false;
  }
  out << "Location" << CSV_DELIMITER_ << "Type" << CSV_DELIMITER_ << "City" << CSV_DELIMITER_ << "Country" << CSV_DELIMITER_ << "Region"
      << CSV_DELIMITER_ << "Radius KM" << CSV_DELIMITER_ << "Genomes (samples)" << CSV_DELIMITER_ << "Passed QC" << CSV_DELIMITER_ << "Studies"
      << CSV_DELIMITER_ << "QC Monoclonal" << CSV_DELIMITER_ << "Hom/Het" << CSV_DELIMITER_ << "Variant Count" << CSV_DELIMITER_ << "Variant Rate"
      << CSV_DELIMITER_ << "Hom Ref (A;A)" << CSV_DELIMITER_ << "Het Ref Minor (A;a)" << CSV_DELIMITER_ << "Hom Minor (a;a)" << CSV_DELIMITER_
      << "Het Diff Minor (a;b)" << CSV_DELIMITER_ << "SNP" << CSV_DELIMITER_ << "Indel" << '\n';
  for (const auto& [location, s] : location_summary)
    out << location << CSV_DELIMITER_ << (s.location_type_ == LocationType::City ? "City" : "Country") << CSV_DELIMITER_ << s.city_ << CSV_DELIMITER_
        << s.country_ << CSV_DELIMITER_ << s.region_ << CSV_DELIMITER_ << s.radius_km_ << CSV_DELIMITER_ << s.radii_samples_ << CSV_DELIMITER_
        << s.radii_samples_OK_ << CSV_DELIMITER_ << s.studies_.size() << CSV_DELIMITER_ << s.monoclonal_Fst_ << CSV_DELIMITER_ << s.hom_het_ratio_
        << CSV_DELIMITER_ << s.total_variants_ << CSV_DELIMITER_ << s.variant_rate_ << CSV_DELIMITER_ << s.homozygous_reference_alleles_
        << CSV_DELIMITER_ << s.heterozygous_reference_minor_alleles_ << CSV_DELIMITER_ << s.homozygous_minor_alleles_ << CSV_DELIMITER_
        << s.heterozygous_minor_alleles_ << CSV_DELIMITER_ << s.snp_count_ << CSV_DELIMITER_ << s.indel_count_ << '\n';
  return out.good();
}

// One line per genome x contig with the VariantAnalysisType counters and Wright's F_IS against the
// whole-population aggregate of the contig.
bool kga::GpuHeteroHomoZygous::writeContigResults(const std::string& file_name) const {
  std::ofstream out(file_name);
  if (!out.good()) {
    ExecEnv::log().error("GpuHeteroHomoZygous::writeContigResults; Unable to open results file: {}", file_name);
    return false;
  }
  std::map<std::string, VariantAnalysisType> aggregate;
  for (const auto& [genome_id, contig_map] : variant_analysis_map_)
    for (const auto& [contig_id, r] : contig_map) {
      VariantAnalysisType& a = aggregate[contig_id];
      a.total_variants_ += r.total_variants_;
      a.heterozygous_reference_minor_alleles_ += r.heterozygous_reference_minor_alleles_;
      a.homozygous_minor_alleles_ += r.homozygous_minor_alleles_;
      a.heterozygous_minor_alleles_ += r.heterozygous_minor_alleles_;
      a.snp_count_ += r.snp_count_;
      a.indel_count_ += r.indel_count_;
    }
  out << "Genome" << CSV_DELIMITER_ << "Contig" << CSV_DELIMITER_ << "Variant Count" << CSV_DELIMITER_ << "SNP" << CSV_DELIMITER_ << "Indel"
      << CSV_DELIMITER_ << "Hom Ref (A;A)" << CSV_DELIMITER_ << "Het Ref Minor (A;a)" << CSV_DELIMITER_ << "Hom Minor (a;a)"
      << CSV_DELIMITER_ << "Het Diff Minor (a;b)" << CSV_DELIMITER_ << "FIS" << '\n';
  for (const auto& [genome_id, contig_map] : variant_analysis_map_)
    for (const auto& [contig_id, r] : contig_map)
      out << genome_id << CSV_DELIMITER_ << contig_id << CSV_DELIMITER_ << r.total_variants_ << CSV_DELIMITER_ << r.snp_count_
          << CSV_DELIMITER_ << r.indel_count_ << CSV_DELIMITER_ << r.homozygous_reference_alleles_ << CSV_DELIMITER_
          << r.heterozygous_reference_minor_alleles_ << CSV_DELIMITER_ << r.homozygous_minor_alleles_ << CSV_DELIMITER_
          << r.heterozygous_minor_alleles_ << CSV_DELIMITER_ << wrightsInbreeding(aggregate.at(contig_id), r) << '\n';
  return out.good();
}
