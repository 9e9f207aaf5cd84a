// Pf7 sample / FWS resource files and the sampling-site geometry (kgx_pf7_resources.h).
#include "kgx_pf7_resources.h"

#ifndef KGX_WITH_REFERENCE_HEADERS

#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <numbers>

namespace kellerberrin::genome {

namespace {

std::string trimmed(std::string_view text) {
  while (!text.empty() && std::isspace(static_cast<unsigned char>(text.front()))) text.remove_prefix(1);
  while (!text.empty() && std::isspace(static_cast<unsigned char>(text.back()))) text.remove_suffix(1);
  return std::string(text);
}

// The rows of a tab-separated file, comment lines left out; every row must hold `columns` fields (a tab after the last
// field opens one more, empty, field).  The header row is the caller's to skip.
bool readSquareText(const std::string& file_name, size_t columns, std::vector<std::vector<std::string>>& rows, const char* who) {
  std::ifstream in(file_name);
  if (!in.good()) {
    ExecEnv::log().error("{}; I/O error; could not open file: {}", who, file_name);
    return false;
  }
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.front() == '#') continue;
    std::vector<std::string> fields;
    size_t begin = 0;
    while (true) {
      const size_t tab = line.find('\t', begin);
      fields.push_back(trimmed(std::string_view(line).substr(begin, tab == std::string::npos ? std::string::npos : tab - begin)));
      if (tab == std::string::npos) break;
      begin = tab + 1;
    }
    rows.push_back(std::move(fields));
  }
  if (rows.empty()) {
    ExecEnv::log().error("{}; Row count: 0 for file: {} is below minimum", who, file_name);
    return false;
  }
  for (const auto& row : rows)
    if (row.size() != columns) {
      ExecEnv::log().error("{}; Not all rows have expected column count: {} for file: {}", who, columns, file_name);
      return false;
    }
  return true;
}

// Degrees as text -> radians; false when the text is not a number.
bool radiansOf(const std::string& degrees_text, double& radians) {
  try {
    radians = (std::stod(degrees_text) / 360.0) * 2 * std::numbers::pi;
    return true;
  } catch (const std::exception&) {
    return false;
  }
}

}  // namespace

bool Pf7SampleRecord::pass() const {
  static const char kPass[] = "TRUE";
  if (qc_pass_.size() != sizeof(kPass) - 1) return false;
  for (size_t i = 0; i < qc_pass_.size(); ++i)
    if (std::toupper(static_cast<unsigned char>(qc_pass_[i])) != kPass[i]) return false;
  return true;
}

Pf7SampleResource::Pf7SampleResource(std::string identifier, Pf7SampleVector sample_vector)
    : ResourceBase(ResourceProperties::PF7SAMPLE_RESOURCE_ID_, std::move(identifier)) {
  for (auto& record : sample_vector) {
    if (record.Pf7Sample_id.empty()) continue;
    const std::string id = record.Pf7Sample_id;
    if (!sample_map_.try_emplace(id, std::move(record)).second) ExecEnv::log().warn("Pf7SampleResource; duplicate Pf7Sample record ({})", id);
  }
  ExecEnv::log().info("Pf7SampleResource loaded {}, (Pf7Sample_id, record), lookup pairs", sample_map_.size());
}

bool ParsePf7Sample::parsePf7SampleFile(const std::string& file_name) {
  std::vector<std::vector<std::string>> rows;
  if (!readSquareText(file_name, 17, rows, "ParsePf7Sample::parsePf7SampleFile")) return false;
  for (size_t r = 1; r < rows.size(); ++r) {
    auto& f = rows[r];
    Pf7SampleRecord record;
    std::string* const fields[17] = {&record.Pf7Sample_id, &record.study_, &record.country_, &record.location1_, &record.country_latitude_,
                                     &record.country_longitude_, &record.location1_latitude_, &record.location1_longitude_, &record.year_,
                                     &record.ena_, &record.all_samples_, &record.population_, &record.callable_, &record.qc_pass_,
                                     &record.qc_fail_reason_, &record.sample_type_, &record.sample_in_pf6_};
    for (size_t c = 0; c < 17; ++c) *fields[c] = std::move(f[c]);
    sample_vector_.push_back(std::move(record));
  }
  ExecEnv::log().info("ParsePf7Sample::parsePf7SampleFile; Parsed: {} Pf7Sample data records from file: {}", sample_vector_.size(), file_name);
  return true;
}

Pf7FwsResource::Pf7FwsResource(std::string identifier, Pf7FwsVector fws_vector)
    : ResourceBase(ResourceProperties::PF7FWS_RESOURCE_ID_, std::move(identifier)) {
  for (auto& record : fws_vector) {
    if (record.Pf7Sample_id.empty()) continue;
    const std::string id = record.Pf7Sample_id;
    if (!fws_map_.try_emplace(id, std::move(record)).second) ExecEnv::log().warn("Pf7FwsResource; duplicate Pf7Sample record ({})", id);
  }
  ExecEnv::log().info("Pf7FwsResource loaded {}, (Pf7Sample_id, record), lookup pairs", fws_map_.size());
}

double Pf7FwsResource::getFWS(const GenomeId_t& genome_id) const {
  const auto found = fws_map_.find(genome_id);
  if (found != fws_map_.end()) return found->second.FWS_value;
  ExecEnv::log().warn("Pf7FwsResource::getFWS; Unable to find FWS statistic for genome: {}", genome_id);
  return std::nan("n/a");
}

std::vector<GenomeId_t> Pf7FwsResource::filterFWS(FwsFilterType filter_type, double fws_threshold, const std::vector<GenomeId_t>& sample_vector) const {
  std::vector<GenomeId_t> kept;
  for (const auto& genome_id : sample_vector) {
    const auto found = fws_map_.find(genome_id);
    if (found == fws_map_.end()) {
      ExecEnv::log().warn("Pf7FwsResource::filterFWS; Genome: {} not found in FWS data", genome_id);
      continue;
    }
    const double fws = found->second.FWS_value;
    if (filter_type == FwsFilterType::GREATER_EQUAL ? fws >= fws_threshold : fws <= fws_threshold) kept.push_back(genome_id);
  }
  return kept;
}

bool ParsePf7Fws::parsePf7FwsFile(const std::string& file_name) {
  std::vector<std::vector<std::string>> rows;
  if (!readSquareText(file_name, 2, rows, "ParsePf7Fws::parsePf7FwsFile")) return false;
  for (size_t r = 1; r < rows.size(); ++r) {
    Pf7FwsRecord record;
    record.Pf7Sample_id = rows[r][0];
    try {
      record.FWS_value = std::stod(rows[r][1]);
    } catch (const std::exception& e) {
      ExecEnv::log().info("ParsePf7Fws::parsePf7FwsFile; FWS text: {} not valid float text, reason: {}, line: {}, file: {}", rows[r][1], e.what(), r + 1,
                          file_name);
      continue;
    }
    fws_vector_.push_back(std::move(record));
  }
  ExecEnv::log().info("ParsePf7Fws::parsePf7FwsFile; Parsed: {} Pf7 FWS data records from file: {}", fws_vector_.size(), file_name);
  return true;
}

LocationCoordinates::LocationCoordinates(std::string location, LocationType location_type, const Pf7SampleRecord& sample_record)
    : location_(std::move(location), location_type) {
  const bool site = location_type == LocationType::City;
  const std::string& latitude_text = site ? sample_record.location1_latitude_ : sample_record.country_latitude_;
  const std::string& longitude_text = site ? sample_record.location1_longitude_ : sample_record.country_longitude_;
  if (site) city_ = sample_record.location1_;
  country_ = sample_record.country_;
  region_ = sample_record.population_;
  // blank coordinates are allowed and read as 0
  if (!radiansOf(latitude_text, latitude_)) {
    latitude_ = 0.0;
    if (!latitude_text.empty()) ExecEnv::log().error("LocationCoordinates; Unable to convert text: {} to a latitude", latitude_text);
  }
  if (!radiansOf(longitude_text, longitude_)) {
    longitude_ = 0.0;
    if (!longitude_text.empty()) ExecEnv::log().error("LocationCoordinates; Unable to convert text: {} to a longitude", longitude_text);
  }
}

double LocationCoordinates::distance_km(const LocationCoordinates& other) const {
  if (location_.first == other.location_.first) return 0.0;
  // spherical law of cosines
  double central = std::sin(latitude_) * std::sin(other.latitude_);
  central += std::cos(latitude_) * std::cos(other.latitude_) * std::cos(other.longitude_ - longitude_);
  return std::acos(central) * 6371.0;
}

void LocationCoordinates::addSample(const Pf7SampleRecord& sample_record) {
  sample_id_vec_.push_back(sample_record.Pf7Sample_id);
  size_t year = 0;
  try {
    year = static_cast<size_t>(std::stoll(sample_record.year_));
  } catch (const std::exception&) {
    // the reference's std::stoll is unguarded here (kgl_Pf7_physical_distance.cpp:120): such a file ends its run
    ExecEnv::log().error("LocationCoordinates::addSample; sample: {} year: '{}' is not a number, recorded as 0", sample_record.Pf7Sample_id,
                         sample_record.year_);
  }
  studies_[sample_record.study_] = year;
}

Pf7SampleLocation::Pf7SampleLocation(const Pf7SampleResource& sample_resource) {
  auto place = [this](const std::string& name, LocationType type, const Pf7SampleRecord& record) {
    if (name.empty()) return;
    // a name is a site or a country, whichever a sample made it first; the samples of both uses gather under it
    location_map_.try_emplace(name, name, type, record).first->second.addSample(record);
  };
  for (const auto& [sample_id, record] : sample_resource.getMap()) {
    place(record.location1_, LocationType::City, record);
    place(record.country_, LocationType::Country, record);
  }
  const size_t n = location_map_.size();
  by_index_.reserve(n);
  for (const auto& [name, coordinates] : location_map_) {
    index_of_.emplace(name, by_index_.size());
    by_index_.push_back(&coordinates);
  }
  distance_km_.resize(n * n);
  for (size_t a = 0; a < n; ++a)
    for (size_t b = 0; b < n; ++b) distance_km_[a * n + b] = a == b ? 0.0 : by_index_[a]->distance_km(*by_index_[b]);
}

double Pf7SampleLocation::distance(const std::string& location1, const std::string& location2) const {
  const auto a = index_of_.find(location1), b = index_of_.find(location2);
  if (a == index_of_.end() || b == index_of_.end()) {
    ExecEnv::log().warn("Pf7SampleLocation::distance; Location: {} not found", a == index_of_.end() ? location1 : location2);
    return 0.0;
  }
  return distance_km_[a->second * by_index_.size() + b->second];
}

std::vector<std::string> Pf7SampleLocation::locationRadius(const std::string& location, double radius, bool all) const {
  std::vector<std::string> locations;
  const auto found = index_of_.find(location);
  if (found == index_of_.end()) {
    ExecEnv::log().warn("Pf7SampleLocation::locationRadius; Location: {} not found", location);
    return locations;
  }
  const size_t n = by_index_.size(), a = found->second;
  const LocationType kind = by_index_[a]->location().second;
  for (size_t b = 0; b < n; ++b)                                     // index order = name order
    if (distance_km_[a * n + b] <= radius && (all || by_index_[b]->location().second == kind)) locations.push_back(by_index_[b]->location().first);
  return locations;
}

std::vector<std::string> Pf7SampleLocation::sampleRadius(const std::string& location, double radius, bool all) const {
  std::vector<std::string> samples;
  for (const auto& near : locationRadius(location, radius, all)) {
    const auto& here = by_index_[index_of_.at(near)]->locationSamples();
    samples.insert(samples.end(), here.begin(), here.end());
  }
  return samples;
}

}  // namespace kellerberrin::genome

#endif  // KGX_WITH_REFERENCE_HEADERS
