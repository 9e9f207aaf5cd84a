// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).  See kgo_pf7.h for the reference files restated.
#include "kgo_pf7.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <iostream>

namespace kgo {

namespace {

// Utility::trimEndWhiteSpace (kel_utility/kel_utility.cpp:190-236): both ends.
std::string trimEndWhiteSpace(const std::string& s) {
  size_t b = 0, e = s.size();
  while (b < e && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
  while (e > b && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
  return s.substr(b, e - b);
}

// Utility::charTokenizer (:266-311)
std::vector<std::string> charTokenizer(const std::string& str, char delim) {
  std::vector<std::string> tokens;
  size_t token_index = 0, index = 0;
  for (; index < str.size(); ++index)
    if (str[index] == delim) {
      tokens.emplace_back(str.substr(token_index, index - token_index));
      token_index = index + 1;
    }
  if (token_index > index) tokens.emplace_back();
  else tokens.emplace_back(str.substr(token_index, index - token_index));
  return tokens;
}

// SquareTextParser::parseFlatFile (kgl_square_parser.cpp:157-197): every line not starting with '#', cut on tabs.
bool parseFlatFile(const std::string& file_name, std::vector<std::vector<std::string>>& rows) {
  std::ifstream in(file_name);
  if (!in.good()) return false;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '#') continue;
    rows.push_back(charTokenizer(line, '\t'));
  }
  return true;
}

bool checkRowSize(const std::vector<std::vector<std::string>>& rows, size_t size) {
  for (const auto& row : rows)
    if (row.size() != size) return false;
  return true;
}

}  // namespace

bool Pf7SampleRecord::pass() const {
  std::string upper = qc_pass_;
  for (auto& c : upper) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  return upper == "TRUE";
}

bool parsePf7SampleFile(const std::string& file_name, Pf7SampleMap& out) {
  std::vector<std::vector<std::string>> rows;
  if (!parseFlatFile(file_name, rows)) return false;
  if (rows.size() < 1) return false;
  if (!checkRowSize(rows, 17)) return false;
  size_t record_count = 0;
  for (const auto& row : rows) {
    ++record_count;
    if (record_count == 1) continue;   // the header
    Pf7SampleRecord r;
    r.Pf7Sample_id = trimEndWhiteSpace(row[0]);
    r.study_ = trimEndWhiteSpace(row[1]);
    r.country_ = trimEndWhiteSpace(row[2]);
    r.location1_ = trimEndWhiteSpace(row[3]);
    r.country_latitude_ = trimEndWhiteSpace(row[4]);
    r.country_longitude_ = trimEndWhiteSpace(row[5]);
    r.location1_latitude_ = trimEndWhiteSpace(row[6]);
    r.location1_longitude_ = trimEndWhiteSpace(row[7]);
    r.year_ = trimEndWhiteSpace(row[8]);
    r.ena_ = trimEndWhiteSpace(row[9]);
    r.all_samples_ = trimEndWhiteSpace(row[10]);
    r.population_ = trimEndWhiteSpace(row[11]);
    r.callable_ = trimEndWhiteSpace(row[12]);
    r.qc_pass_ = trimEndWhiteSpace(row[13]);
    r.qc_fail_reason_ = trimEndWhiteSpace(row[14]);
    r.sample_type_ = trimEndWhiteSpace(row[15]);
    r.sample_in_pf6_ = trimEndWhiteSpace(row[16]);
    if (r.Pf7Sample_id.empty()) continue;            // indexPf7SampleData: blank ids skipped, the first record of an id wins
    out.try_emplace(r.Pf7Sample_id, r);
  }
  return true;
}

bool parsePf7FwsFile(const std::string& file_name, Pf7FwsMap& out) {
  std::vector<std::vector<std::string>> rows;
  if (!parseFlatFile(file_name, rows)) return false;
  if (rows.size() < 1) return false;
  if (!checkRowSize(rows, 2)) return false;
  size_t record_count = 0;
  for (const auto& row : rows) {
    ++record_count;
    if (record_count == 1) continue;
    Pf7FwsRecord r;
    r.Pf7Sample_id = trimEndWhiteSpace(row[0]);
    const std::string text = trimEndWhiteSpace(row[1]);
    try {
      r.FWS_value = std::stod(text);
    } catch (std::exception&) {
      continue;
    }
    if (r.Pf7Sample_id.empty()) continue;
    out.try_emplace(r.Pf7Sample_id, r);
  }
  return true;
}

double getFWS(const Pf7FwsMap& fws, const std::string& genome) {
  auto it = fws.find(genome);
  if (it != fws.end()) return it->second.FWS_value;
  return std::nan("n/a");
}

std::vector<std::string> filterFWS(const Pf7FwsMap& fws, bool greater_equal, double threshold, const std::vector<std::string>& samples) {
  std::vector<std::string> filtered;
  for (const auto& genome_id : samples) {
    auto it = fws.find(genome_id);
    if (it != fws.end()) {
      const bool accept = greater_equal ? it->second.FWS_value >= threshold : it->second.FWS_value <= threshold;
      if (accept) filtered.push_back(genome_id);
    }
  }
  return filtered;
}

// ---- LocationCoordinates / Pf7SampleLocation ------------------------------------------------------

static void convertLatLong(const std::string& latitude_text, const std::string& longitude_text, double& latitude, double& longitude) {
  const double pi = 3.141592653589793238462643383279502884;   // std::numbers::pi
  try {
    latitude = std::stod(latitude_text);
    latitude = (latitude / 360.0) * 2 * pi;
  } catch (std::exception&) {
    latitude = 0.0;
  }
  try {
    longitude = std::stod(longitude_text);
    longitude = (longitude / 360.0) * 2 * pi;
  } catch (std::exception&) {
    if (!longitude_text.empty()) longitude = 0.0;   // blank: left at its initial 0.0 (kgl_Pf7_physical_distance.cpp:62-78)
  }
}

LocationCoordinates::LocationCoordinates(std::string location, LocationType type, const Pf7SampleRecord& sample_record)
    : location_({std::move(location), type}) {
  if (type == LocationType::City) {
    city_ = sample_record.location1_;
    convertLatLong(sample_record.location1_latitude_, sample_record.location1_longitude_, latitude_, longitude_);
  } else {
    city_.clear();
    convertLatLong(sample_record.country_latitude_, sample_record.country_longitude_, latitude_, longitude_);
  }
  country_ = sample_record.country_;
  region_ = sample_record.population_;
}

double LocationCoordinates::distance_km(const LocationCoordinates& other) const {
  if (location_.first == other.location_.first) return 0.0;
  double spherical_offset = std::sin(latitude_) * std::sin(other.latitude_);
  spherical_offset += std::cos(latitude_) * std::cos(other.latitude_) * std::cos((other.longitude_ - longitude_));
  return std::acos(spherical_offset) * 6371.0;
}

void LocationCoordinates::addSample(const Pf7SampleRecord& sample_record) {
  sample_id_vec_.push_back(sample_record.Pf7Sample_id);
  const size_t year = std::stoll(sample_record.year_);   // throws on a blank year, as the reference does
  studies_[sample_record.study_] = year;
}

Pf7SampleLocation::Pf7SampleLocation(const Pf7SampleMap& samples) {
  for (const auto& [sample_id, sample_record] : samples) {
    if (!sample_record.location1_.empty()) {
      auto it = location_map_.find(sample_record.location1_);
      if (it == location_map_.end())
        it = location_map_.try_emplace(sample_record.location1_, sample_record.location1_, LocationType::City, sample_record).first;
      it->second.addSample(sample_record);
    }
    if (!sample_record.country_.empty()) {
      auto it = location_map_.find(sample_record.country_);
      if (it == location_map_.end())
        it = location_map_.try_emplace(sample_record.country_, sample_record.country_, LocationType::Country, sample_record).first;
      it->second.addSample(sample_record);
    }
  }
  for (const auto& [location1, record1] : location_map_)
    for (const auto& [location2, record2] : location_map_) distance_cache_[location1][location2] = calculateDistance(location1, location2);
}

double Pf7SampleLocation::calculateDistance(const std::string& a, const std::string& b) const {
  if (a == b) return 0.0;
  auto ia = location_map_.find(a), ib = location_map_.find(b);
  if (ia == location_map_.end() || ib == location_map_.end()) return 0.0;
  return ia->second.distance_km(ib->second);
}

std::vector<std::string> Pf7SampleLocation::locationRadius(const std::string& location, double radius, bool all) const {
  std::vector<std::string> locations;
  auto cache_iter = distance_cache_.find(location);
  if (cache_iter == distance_cache_.end()) return locations;
  auto location_iter = location_map_.find(location);
  if (location_iter == location_map_.end()) return locations;
  const LocationType type = location_iter->second.location_.second;
  for (const auto& [proximity_location, distance] : cache_iter->second) {
    if (distance <= radius) {
      if (all) {
        locations.push_back(proximity_location);
      } else {
        auto proximity_iter = location_map_.find(proximity_location);
        if (proximity_iter == location_map_.end()) { locations.clear(); return locations; }
        if (type == proximity_iter->second.location_.second) locations.push_back(proximity_location);
      }
    }
  }
  return locations;
}

std::vector<std::string> Pf7SampleLocation::sampleRadius(const std::string& location, double radius, bool all) const {
  std::vector<std::string> sample_vec;
  for (const auto& proximity_location : locationRadius(location, radius, all)) {
    auto it = location_map_.find(proximity_location);
    if (it == location_map_.end()) { sample_vec.clear(); return sample_vec; }
    for (const auto& genome_sample : it->second.sample_id_vec_) sample_vec.push_back(genome_sample);
  }
  return sample_vec;
}

// ---- FilterPf7 (genome part) ----------------------------------------------------------------------

std::shared_ptr<PopulationDB> pf7GenomeFilter(const PopulationDB& population, const Pf7SampleMap& samples, const Pf7FwsMap& fws, bool filter_qc,
                                               bool filter_fws, double fws_threshold) {
  // filterPassQCGenomes (kga_analysis_lib_PfFilter.cpp:124-158)
  std::vector<std::string> genomes;
  for (const auto& [genome_id, genome_ptr] : population.getMap()) {
    if (filter_qc) {
      auto it = samples.find(genome_id);
      if (it == samples.end() || !it->second.pass()) continue;
    }
    genomes.push_back(genome_id);
  }
  // viewFilterFWS (kgl_pf7_fws_parser.cpp:55-70)
  if (filter_fws) genomes = filterFWS(fws, true, fws_threshold, genomes);
  const std::set<std::string> keep(genomes.begin(), genomes.end());
  // deepCopy of the filtered view (an all-pass viewFilter copies the containers), then squareContigs
  auto out = std::make_shared<PopulationDB>(population.populationId());
  auto all = population.viewFilter([](const Variant&) { return true; });
  std::set<std::string> contig_set;
  for (const auto& [genome_id, genome_ptr] : all->getMap()) {
    if (!keep.count(genome_id)) continue;
    out->addGenome(genome_ptr);
    for (const auto& [contig_id, contig_ptr] : genome_ptr->getMap()) contig_set.insert(contig_id);
  }
  for (const auto& [genome_id, genome_ptr] : out->getMap())
    for (const auto& contig_id : contig_set) genome_ptr->getCreateContig(contig_id);
  return out;
}

// ---- HeteroHomoZygous -----------------------------------------------------------------------------

void HeteroHomoZygous::analyzeVariantPopulation(const PopulationDB& population, const Pf7FwsMap& fws, const Pf7SampleMap& samples) {
  for (const auto& [genome_id, genome_ptr] : population.getMap()) {
    auto record_iter = samples.find(genome_id);
    if (record_iter == samples.end()) continue;
    auto [genome_iter, result] = variant_analysis_map_.try_emplace(genome_id);
    Obj& obj = genome_iter->second;
    if (result) {
      obj.sample_record_ = record_iter->second;
      obj.fws_value_ = getFWS(fws, genome_id);
    }
    for (const auto& [contig_id, contig_ptr] : genome_ptr->getMap()) {
      auto& contig_count = obj.analysis_map_[contig_id];
      if (contig_ptr->variantCount() == 0) continue;
      for (const auto& [offset, offset_ptr] : contig_ptr->getMap()) updateVariantAnalysisType(*offset_ptr, contig_count);
    }
  }
}

VariantAnalysisType HeteroHomoZygous::aggregateResults(const std::vector<std::string>& sample_vector) const {
  VariantAnalysisType summary;
  const std::set<std::string> sample_set(sample_vector.begin(), sample_vector.end());
  for (const auto& genome_id : sample_set) {
    auto it = variant_analysis_map_.find(genome_id);
    if (it == variant_analysis_map_.end()) continue;
    for (const auto& [contig_id, r] : it->second.analysis_map_) {
      summary.total_variants_ += r.total_variants_;
      summary.heterozygous_reference_minor_alleles_ += r.heterozygous_reference_minor_alleles_;
      summary.homozygous_minor_alleles_ += r.homozygous_minor_alleles_;
      summary.heterozygous_minor_alleles_ += r.heterozygous_minor_alleles_;
      summary.snp_count_ += r.snp_count_;
      summary.indel_count_ += r.indel_count_;
      summary.homozygous_reference_alleles_ += r.homozygous_reference_alleles_;
    }
  }
  return summary;
}

LocationSummaryMap HeteroHomoZygous::location_summary(const Pf7SampleMap& samples, const Pf7SampleLocation& distance, double radius_km,
                                                      const Pf7FwsMap& fws) const {
  std::set<std::string> pass_genomes;
  for (const auto& [genome_id, sample_record] : samples)
    if (sample_record.pass()) pass_genomes.insert(genome_id);
  LocationSummaryMap summary_map;
  for (const auto& [location, location_record] : distance.locationMap()) {
    auto radii_samples = distance.sampleRadius(location, radius_km);
    std::vector<std::string> radii_passed;
    for (const auto& sample : radii_samples)
      if (pass_genomes.count(sample)) radii_passed.push_back(sample);
    auto aggregated = aggregateResults(radii_samples);
    double hom_het_ratio = 0.0;
    const size_t total_heterozygous = aggregated.heterozygous_reference_minor_alleles_ + aggregated.heterozygous_minor_alleles_;
    if (total_heterozygous > 0) hom_het_ratio = static_cast<double>(aggregated.homozygous_minor_alleles_) / static_cast<double>(total_heterozygous);
    double variant_rate = 0.0;
    if (!radii_samples.empty()) variant_rate = static_cast<double>(aggregated.total_variants_) / static_cast<double>(radii_samples.size());
    double monoclonal = 0.0;
    if (!radii_passed.empty()) {
      auto mono_samples = filterFWS(fws, true, MONOCLONAL_FWS_THRESHOLD, radii_passed);
      monoclonal = static_cast<double>(mono_samples.size()) / static_cast<double>(radii_passed.size());
    }
    LocationSummary s;
    s.location_ = location;
    s.location_type_ = location_record.location_.second;
    s.city_ = location_record.city_;
    s.country_ = location_record.city_;   // sic (kga_analysis_PfEMP_heterozygous.cpp:336)
    s.region_ = location_record.region_;
    s.radius_km_ = radius_km;
    s.radii_samples_ = radii_samples.size();
    s.radii_samples_OK_ = radii_passed.size();
    s.studies_ = location_record.studies_;
    s.monoclonal_Fst_ = monoclonal;
    s.hom_het_ratio_ = hom_het_ratio;
    s.total_variants_ = aggregated.total_variants_;
    s.variant_rate_ = variant_rate;
    s.homozygous_reference_alleles_ = aggregated.homozygous_reference_alleles_;
    s.heterozygous_reference_minor_alleles_ = aggregated.heterozygous_reference_minor_alleles_;
    s.homozygous_minor_alleles_ = aggregated.homozygous_minor_alleles_;
    s.heterozygous_minor_alleles_ = aggregated.heterozygous_minor_alleles_;
    s.snp_count_ = aggregated.snp_count_;
    s.indel_count_ = aggregated.indel_count_;
    summary_map[location] = s;
  }
  return summary_map;
}

void HeteroHomoZygous::UpdateSampleLocation(const LocationSummaryMap& summary_map) {
  for (auto& [genome_id, obj] : variant_analysis_map_) {
    auto record_iter = summary_map.find(obj.sample_record_.location1_);
    if (record_iter != summary_map.end()) {
      if (record_iter->second.radii_samples_OK_ < MINIMUM_LOCATION_SAMPLES_) {
        if (summary_map.count(obj.sample_record_.country_)) record_iter = summary_map.find(obj.sample_record_.country_);
        else continue;
      }
    } else {
      continue;
    }
    const LocationSummary& location = record_iter->second;
    auto aggregated = aggregateResults({genome_id});
    double wrights_inbreeding = 0.0;
    if (location.total_variants_ > 0 && aggregated.total_variants_ > 0) {
      const double expected = static_cast<double>(location.heterozygous_minor_alleles_ + location.heterozygous_reference_minor_alleles_) /
                              static_cast<double>(location.total_variants_);
      const double observed = static_cast<double>(aggregated.heterozygous_minor_alleles_ + aggregated.heterozygous_reference_minor_alleles_) /
                              static_cast<double>(aggregated.total_variants_);
      wrights_inbreeding = (expected - observed) / expected;
    }
    obj.fis_value_ = wrights_inbreeding;
  }
}

void HeteroHomoZygous::write_variant_results(const std::string& file_name, const LocationSummaryMap& location_summary) {
  std::ofstream out(file_name);
  if (!out.good() || variant_analysis_map_.empty()) return;
  const char d = ',';
  const size_t contig_count = variant_analysis_map_.begin()->second.analysis_map_.size();
  out << "Genome" << d << "FWS" << d << "FIS (inbreed)" << d << "City" << d << "Country" << d << "Region" << d << "Study" << d << "Year" << d
      << "Hom/Het";
  for (size_t i = 0; i <= contig_count; ++i)
    out << d << "Contig" << d << "Variant Count" << d << "Hom Ref (A;A)" << d << "Het Ref Minor (A;a)" << d << "Hom Minor (a;a)" << d
        << "Het Diff Minor (a;b)" << d << "SNP" << d << "Indel";
  out << '\n';
  for (auto& [genome_id, obj] : variant_analysis_map_) {
    auto aggregated = aggregateResults({genome_id});
    double hom_het_ratio = 0.0;
    const size_t total_heterozygous = aggregated.heterozygous_reference_minor_alleles_ + aggregated.heterozygous_minor_alleles_;
    if (total_heterozygous > 0) hom_het_ratio = static_cast<double>(aggregated.homozygous_minor_alleles_) / static_cast<double>(total_heterozygous);
    std::string region;
    auto iter = location_summary.find(obj.sample_record_.location1_);
    if (iter != location_summary.end()) region = iter->second.region_;
    out << genome_id << d << obj.fws_value_ << d << obj.fis_value_ << d << obj.sample_record_.location1_ << d << obj.sample_record_.country_ << d
        << region << d << obj.sample_record_.study_ << d << obj.sample_record_.year_ << d << hom_het_ratio;
    out << d << "Combined" << d << aggregated.total_variants_ << d << aggregated.homozygous_reference_alleles_ << d
        << aggregated.heterozygous_reference_minor_alleles_ << d << aggregated.homozygous_minor_alleles_ << d
        << aggregated.heterozygous_minor_alleles_ << d << aggregated.snp_count_ << d << aggregated.indel_count_;
    for (auto& [contig_id, c] : obj.analysis_map_)
      out << d << contig_id << d << c.total_variants_ << d << c.homozygous_reference_alleles_ << d << c.heterozygous_reference_minor_alleles_
          << d << c.homozygous_minor_alleles_ << d << c.heterozygous_minor_alleles_ << d << c.snp_count_ << d << c.indel_count_;
    out << '\n';
  }
}

void HeteroHomoZygous::write_location_results(const std::string& file_name, const LocationSummaryMap& summary_map) const {
  std::ofstream out(file_name);
  if (!out.good()) return;
  const char d = ',';
  out << "Location" << d << "Type" << d << "City" << d << "Country" << d << "Region" << d << "Radius KM" << d << "Genomes (samples)" << d
      << "Passed QC" << d << "Studies" << d << "QC Monoclonal" << d << "Hom/Het" << d << "Variant Count" << d << "Variant Rate" << d
      << "Hom Ref (A;A)" << d << "Het Ref Minor (A;a)" << d << "Hom Minor (a;a)" << d << "Het Diff Minor (a;b)" << d << "SNP" << d << "Indel"
      << '\n';
  for (const auto& [location, s] : summary_map)
    out << location << d << (s.location_type_ == LocationType::City ? "City" : "Country") << d << s.city_ << d << s.country_ << d << s.region_ << d
        << s.radius_km_ << d << s.radii_samples_ << d << s.radii_samples_OK_ << d << s.studies_.size() << d << s.monoclonal_Fst_ << d
        << s.hom_het_ratio_ << d << s.total_variants_ << d << s.variant_rate_ << d << s.homozygous_reference_alleles_ << d
        << s.heterozygous_reference_minor_alleles_ << d << s.homozygous_minor_alleles_ << d << s.heterozygous_minor_alleles_ << d << s.snp_count_
        << d << s.indel_count_ << '\n';
}

}  // namespace kgo
