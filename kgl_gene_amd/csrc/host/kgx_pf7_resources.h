// The Pf7 sample resources the PfEMP package joins its per-genome counters with: sample records (site, country, study,
// year, QC verdict), the published within-host FWS value per sample, and the great-circle geometry of the sampling
// sites.  Stand-alone mirrors of
//   kgl_genomics/kgl_parser/kgl_pf7_sample_parser.h:22-125          Pf7SampleRecord, Pf7SampleResource, ParsePf7Sample
//   kgl_genomics/kgl_parser/kgl_pf7_fws_parser.h:20-103             Pf7FwsRecord, Pf7FwsResource, ParsePf7Fws
//   kgl_genomics/kgl_parser/kgl_Pf7_physical_distance.h:15-112      LocationCoordinates, Pf7SampleLocation
// with the members GpuAlleleAnalysis uses, same names and meaning, so that a build against the reference's headers
// (KGX_WITH_REFERENCE_HEADERS) takes the reference's classes instead.  Both files are tab-separated text with one header
// row; lines starting with '#' are skipped (SquareTextParser::parseFlatFile, kgl_square_parser.cpp:157-197).
#ifndef KGX_PF7_RESOURCES_H
#define KGX_PF7_RESOURCES_H

#ifdef KGX_WITH_REFERENCE_HEADERS
#include "kgl_pf7_sample_parser.h"
#include "kgl_pf7_fws_parser.h"
#include "kgl_Pf7_physical_distance.h"
#else

#include <map>
#include <string>
#include <vector>

#include "kgx_refshim.h"

namespace kellerberrin::genome {

struct Pf7SampleRecord {
  std::string Pf7Sample_id;
  std::string study_;
  std::string country_;
  std::string location1_;            // the sampling site ("city")
  std::string country_latitude_;
  std::string country_longitude_;
  std::string location1_latitude_;
  std::string location1_longitude_;
  std::string year_;
  std::string ena_;
  std::string all_samples_;
  std::string population_;           // the region
  std::string callable_;
  std::string qc_pass_;
  std::string qc_fail_reason_;
  std::string sample_type_;
  std::string sample_in_pf6_;
  [[nodiscard]] bool pass() const;   // "True" in any case
};
using Pf7SampleVector = std::vector<Pf7SampleRecord>;
using Pf7SampleMap = std::map<std::string, Pf7SampleRecord>;

class Pf7SampleResource : public ResourceBase {
 public:
  Pf7SampleResource(std::string identifier, Pf7SampleVector sample_vector);
  ~Pf7SampleResource() override = default;
  [[nodiscard]] const Pf7SampleMap& getMap() const { return sample_map_; }
 private:
  Pf7SampleMap sample_map_;          // the first record of an id wins, blank ids are left out
};

class ParsePf7Sample {
 public:
  [[nodiscard]] bool parsePf7SampleFile(const std::string& file_name);     // 17 columns on every row, else false
  [[nodiscard]] const Pf7SampleVector& getPf7SampleVector() const { return sample_vector_; }
 private:
  Pf7SampleVector sample_vector_;
};

struct Pf7FwsRecord {
  std::string Pf7Sample_id;
  double FWS_value{0.0};
};
using Pf7FwsVector = std::vector<Pf7FwsRecord>;
using Pf7FwsMap = std::map<std::string, Pf7FwsRecord>;
enum class FwsFilterType { GREATER_EQUAL, LESS_EQUAL };

class Pf7FwsResource : public ResourceBase {
 public:
  Pf7FwsResource(std::string identifier, Pf7FwsVector fws_vector);
  ~Pf7FwsResource() override = default;
  [[nodiscard]] const Pf7FwsMap& getMap() const { return fws_map_; }
  [[nodiscard]] double getFWS(const GenomeId_t& genome) const;             // NaN (and a warning) for an unknown sample
  // The samples of the list whose FWS is on the wanted side of the threshold; samples without a value drop out.
  [[nodiscard]] std::vector<GenomeId_t> filterFWS(FwsFilterType filter_type, double fws_threshold, const std::vector<GenomeId_t>& sample_vector) const;
  constexpr static const double MONOCLONAL_FWS_THRESHOLD{0.95};
 private:
  Pf7FwsMap fws_map_;
};

class ParsePf7Fws {
 public:
  [[nodiscard]] bool parsePf7FwsFile(const std::string& file_name);        // 2 columns on every row, else false
  [[nodiscard]] const Pf7FwsVector& getPf7FwsVector() const { return fws_vector_; }
 private:
  Pf7FwsVector fws_vector_;
};

enum class LocationType { City, Country };

// One sampling site or one country: its coordinates (radians) and the samples taken there.
class LocationCoordinates {
 public:
  LocationCoordinates(std::string location, LocationType location_type, const Pf7SampleRecord& sample_record);
  [[nodiscard]] double latitudeRadians() const { return latitude_; }
  [[nodiscard]] double longitudeRadians() const { return longitude_; }
  [[nodiscard]] const std::pair<std::string, LocationType>& location() const { return location_; }
  [[nodiscard]] const std::vector<std::string>& locationSamples() const { return sample_id_vec_; }
  [[nodiscard]] const std::string& city() const { return city_; }          // blank for a country
  [[nodiscard]] const std::string& country() const { return country_; }
  [[nodiscard]] const std::string& region() const { return region_; }
  [[nodiscard]] const std::map<std::string, size_t>& locationStudies() const { return studies_; }   // study -> year
  [[nodiscard]] double distance_km(const LocationCoordinates& other_location) const;                  // great circle, R = 6371 km
  void addSample(const Pf7SampleRecord& sample_record);
 private:
  double latitude_{0.0}, longitude_{0.0};
  std::pair<std::string, LocationType> location_;
  std::string city_, country_, region_;
  std::vector<std::string> sample_id_vec_;
  std::map<std::string, size_t> studies_;
};
using SampleLocationMap = std::map<std::string, LocationCoordinates>;

class Pf7SampleLocation {
 public:
  explicit Pf7SampleLocation(const Pf7SampleResource& sample_resource);
  [[nodiscard]] const SampleLocationMap& locationMap() const { return location_map_; }
  [[nodiscard]] double distance(const std::string& location1, const std::string& location2) const;
  // Locations (names ascending) / their samples within the radius of a location; unless `all`, only locations of the
  // location's own kind (sites around a site, countries around a country).
  [[nodiscard]] std::vector<std::string> locationRadius(const std::string& location, double radius, bool all = false) const;
  [[nodiscard]] std::vector<std::string> sampleRadius(const std::string& location, double radius, bool all = false) const;
 private:
  SampleLocationMap location_map_;
  std::map<std::string, size_t> index_of_;     // location -> row of the distance table
  std::vector<const LocationCoordinates*> by_index_;
  std::vector<double> distance_km_;            // [locations][locations]
};

}  // namespace kellerberrin::genome

#endif  // KGX_WITH_REFERENCE_HEADERS
#endif  // KGX_PF7_RESOURCES_H
