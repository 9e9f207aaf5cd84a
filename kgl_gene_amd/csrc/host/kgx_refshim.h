// Minimal stand-in for the slice of the reference's headers that an analysis package touches.
//
// The GPU analysis packages in this directory (GpuAlleleAnalysis, GpuInbreedAnalysis) are written
// against the reference's own interface: VirtualAnalysis (kgl_app/kgl_package_analysis_virtual.h:20-55),
// the read-side of PopulationDB -> GenomeDB -> ContigDB -> OffsetDB -> Variant
// (kgl_genomics/kgl_variant_db/*.h), ActiveParameterList / ParameterMap (kgl_app/kgl_runtime.h:251-321),
// AnalysisResources (kgl_app/kgl_runtime_resource.h:25-100), DataDB (kgl_genomics/kgl_parser/kgl_data_file_type.h:92-136),
// FrequencyDatabaseRead (kgl_variant_db/kgl_variant_db_freq.h:26-132), ExecEnv::log() (kel_app/kel_logging.h).
// Inside the reference tree they include the real headers (define KGX_WITH_REFERENCE_HEADERS, see
// INTEGRATION.md); here, where the reference cannot be compiled (no <format>, Boost, nlopt), this header
// supplies same-named types with the same member signatures and nothing more, so the packages build and
// are testable stand-alone.  Only accessors the packages call are present; the write side (addVariant, ...)
// exists so tests can populate a store the way a parser would.
#ifndef KGX_REFSHIM_H
#define KGX_REFSHIM_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <string_view>
#include <vector>

namespace kellerberrin {

// kel_app/kel_logging.h:74-229 — four levels, "{}" placeholders, critical() exits.
class ExecEnvLogger {
 public:
  template <typename... Args> void info(const std::string& fmt, Args&&... args) { emit("INFO", fmt, std::forward<Args>(args)...); }
  template <typename... Args> void warn(const std::string& fmt, Args&&... args) { ++warn_count_; if (warn_count_ <= kMaxMessages) emit("WARN", fmt, std::forward<Args>(args)...); }
  template <typename... Args> void error(const std::string& fmt, Args&&... args) {
    ++error_count_;
    emit("ERROR", fmt, std::forward<Args>(args)...);
    if (error_count_ >= kMaxMessages) { std::fprintf(stderr, "too many errors, aborting\n"); std::exit(EXIT_FAILURE); }   // kel_logging.cpp:70-77
  }
  template <typename... Args> [[noreturn]] void critical(const std::string& fmt, Args&&... args) {
    emit("CRITICAL", fmt, std::forward<Args>(args)...);
    std::exit(EXIT_FAILURE);
  }
  size_t errorCount() const { return error_count_; }
  size_t warnCount() const { return warn_count_; }
  void quiet(bool q) { quiet_ = q; }
 private:
  static constexpr size_t kMaxMessages = 100;
  template <typename T> static void append(std::ostringstream& os, const T& v) { os << v; }
  static void substitute(std::ostringstream& os, std::string_view fmt) { os << fmt; }
  template <typename T, typename... Rest>
  static void substitute(std::ostringstream& os, std::string_view fmt, const T& v, const Rest&... rest) {
    const size_t open = fmt.find('{');
    if (open == std::string_view::npos) { os << fmt; return; }
    const size_t close = fmt.find('}', open);
    os << fmt.substr(0, open);
    append(os, v);
    substitute(os, close == std::string_view::npos ? std::string_view() : fmt.substr(close + 1), rest...);
  }
  template <typename... Args> void emit(const char* level, const std::string& fmt, const Args&... args) {
    if (quiet_ && level[0] == 'I') return;
    std::ostringstream os;
    substitute(os, fmt, args...);
    std::fprintf(stderr, "[%s] %s\n", level, os.str().c_str());
  }
  size_t error_count_ = 0, warn_count_ = 0;
  bool quiet_ = false;
};

class ExecEnv {
 public:
  static ExecEnvLogger& log() { static ExecEnvLogger logger; return logger; }   // kel_app/kel_exec_env.h:34
};

namespace genome {

using ContigId_t = std::string;       // kgl_genome/kgl_genome_types.h:21-32
using GenomeId_t = std::string;
using ContigOffset_t = uint64_t;

// kgl_parser/kgl_data_file_type.h:32-62
enum class DataSourceEnum { Genome1000, GnomadGenome3_1, Falciparum, GnomadExomes3_1, GnomadExomes2_1, Gnomad3_1,
                            Gnomad3_0, Gnomad2_1, Clinvar, dbSNP, JSONdbSNP, BioPMID, NotImplemented };
enum class DataStructureEnum { DiploidPhased, DiploidUnphased, UnphasedMonoGenome, CitationMap, BioPMIDMap, NoStructure };

struct DataCharacteristic {
  std::string source_text;
  DataSourceEnum data_source;
  DataStructureEnum data_structure;
};

class DataDB {
 public:
  explicit DataDB(DataSourceEnum data_source) : data_source_(data_source) {}
  virtual ~DataDB() = default;
  [[nodiscard]] virtual const std::string& fileId() const = 0;
  [[nodiscard]] DataSourceEnum dataSource() const { return data_source_; }
  [[nodiscard]] DataCharacteristic dataCharacteristic() const {
    switch (data_source_) {   // the rows of DataDB::data_characteristics_ (kgl_data_file_type.h:118-134) that carry populations
      case DataSourceEnum::Genome1000: return {"Genome1000", data_source_, DataStructureEnum::DiploidPhased};
      case DataSourceEnum::GnomadGenome3_1: return {"GnomadGenome3_1", data_source_, DataStructureEnum::DiploidUnphased};
      case DataSourceEnum::Falciparum: return {"Falciparum", data_source_, DataStructureEnum::DiploidUnphased};
      case DataSourceEnum::GnomadExomes3_1: case DataSourceEnum::GnomadExomes2_1: case DataSourceEnum::Gnomad3_1:
      case DataSourceEnum::Gnomad3_0: case DataSourceEnum::Gnomad2_1: case DataSourceEnum::Clinvar: case DataSourceEnum::dbSNP:
        return {"MonoGenome", data_source_, DataStructureEnum::UnphasedMonoGenome};
      default: return {"FileNameOnly", data_source_, DataStructureEnum::NoStructure};
    }
  }
 private:
  DataSourceEnum data_source_;
};

// kgl_parser/kgl_variant_factory_parsers.cpp:65-66: a "FileNameOnly" data file reaches the package as its name only.
class FilenameDataDB : public DataDB {
 public:
  FilenameDataDB(DataSourceEnum data_source, std::string file_name) : DataDB(data_source), file_name_(std::move(file_name)) {}
  [[nodiscard]] const std::string& fileId() const override { return file_name_; }
 private:
  std::string file_name_;
};

// kgl_variant_db/kgl_variant_db.h:25-28
enum class VariantPhase : std::uint8_t { HAPLOID_PHASED = 0, DIPLOID_PHASE_A = 1, DIPLOID_PHASE_B = 2, UNPHASED = 255 };

// DNA5SequenceLinear: only getStringView()/length() are used on the path.
class DNA5SequenceLinear {
 public:
  explicit DNA5SequenceLinear(std::string s) : seq_(std::move(s)) {}
  [[nodiscard]] std::string_view getStringView() const { return seq_; }
  [[nodiscard]] size_t length() const { return seq_.size(); }
  [[nodiscard]] char operator[](size_t i) const { return seq_[i]; }
 private:
  std::string seq_;
};

// The INFO payload of one VCF record, shared by the Variants cut from it (DataMemoryBlock in the reference).
// Float vector fields are stored as float32 (kgl_parser/kgl_variant_factory_vcf_parse_info.h:27-37).
struct InfoRecord {
  std::map<std::string, std::vector<float>> float_fields;
};

// kgl_evidence/kgl_variant_evidence.h:73-152
class VariantEvidence {
 public:
  VariantEvidence(size_t vcf_record_count, DataSourceEnum data_source, bool pass_filter,
                  std::shared_ptr<const InfoRecord> info, uint32_t alt_variant_index, uint32_t alt_variant_count)
      : vcf_record_count_(vcf_record_count), data_source_(data_source), pass_filter_(pass_filter), info_(std::move(info)),
        alt_variant_index_(alt_variant_index), alt_variant_count_(alt_variant_count) {}
  [[nodiscard]] uint32_t altVariantIndex() const { return alt_variant_index_; }
  [[nodiscard]] uint32_t altVariantCount() const { return alt_variant_count_; }
  [[nodiscard]] bool passFilter() const { return pass_filter_; }
  [[nodiscard]] DataSourceEnum dataSource() const { return data_source_; }
  [[nodiscard]] size_t vcfRecordCount() const { return vcf_record_count_; }
 private:
  // the INFO payload behind FrequencyDatabaseRead::infoFloatField / InfoEvidenceAnalysis::getTypedInfoData (the
  // reference keeps a DataMemoryBlock here, kgl_variant_evidence.h:140-152): theirs alone, no package reads it
  friend class FrequencyDatabaseRead;
  friend class InfoEvidenceAnalysis;
  [[nodiscard]] const std::shared_ptr<const InfoRecord>& infoRecord() const { return info_; }
  size_t vcf_record_count_;
  DataSourceEnum data_source_;
  bool pass_filter_;
  std::shared_ptr<const InfoRecord> info_;
  uint32_t alt_variant_index_, alt_variant_count_;
};

// kgl_variant_db/kgl_variant_db.h:46-176 (read side)
class Variant {
 public:
  Variant(ContigId_t contig_id, ContigOffset_t offset, VariantPhase phase_id, std::string identifier,
          DNA5SequenceLinear&& reference, DNA5SequenceLinear&& alternate, const VariantEvidence& evidence)
      : contig_id_(std::move(contig_id)), contig_reference_offset_(offset), phase_id_(phase_id), identifier_(std::move(identifier)),
        reference_(std::move(reference)), alternate_(std::move(alternate)), evidence_(evidence) {}
  [[nodiscard]] const ContigId_t& contigId() const { return contig_id_; }
  [[nodiscard]] ContigOffset_t offset() const { return contig_reference_offset_; }
  [[nodiscard]] VariantPhase phaseId() const { return phase_id_; }
  [[nodiscard]] const std::string& identifier() const { return identifier_; }
  [[nodiscard]] const DNA5SequenceLinear& reference() const { return reference_; }
  [[nodiscard]] const DNA5SequenceLinear& alternate() const { return alternate_; }
  [[nodiscard]] size_t referenceSize() const { return reference_.length(); }
  [[nodiscard]] size_t alternateSize() const { return alternate_.length(); }
  [[nodiscard]] const VariantEvidence& evidence() const { return evidence_; }
  [[nodiscard]] std::string HGVS() const {   // kgl_variant_db.cpp:287-291
    std::string s(contig_id_);
    s += ":g.";
    s += std::to_string(contig_reference_offset_);
    s += reference_.getStringView();
    s += '>';
    s += alternate_.getStringView();
    return s;
  }
  [[nodiscard]] bool isSNP() const {          // kgl_variant_db.cpp:121-158
    if (reference_.length() == 1 && alternate_.length() == 1) return true;
    if (reference_.length() != alternate_.length()) return false;
    bool diff = false;
    for (size_t i = 0; i < reference_.length(); ++i)
      if (reference_[i] != alternate_[i]) { if (diff) return false; diff = true; }
    return true;
  }
 private:
  ContigId_t contig_id_;
  ContigOffset_t contig_reference_offset_;
  VariantPhase phase_id_;
  std::string identifier_;
  DNA5SequenceLinear reference_, alternate_;
  VariantEvidence evidence_;
};

using OffsetDBArray = std::vector<std::shared_ptr<const Variant>>;

class OffsetDB {   // kgl_variant_db_offset.h:24-55
 public:
  [[nodiscard]] const OffsetDBArray& getVariantArray() const { return variant_vector_; }
  void addVariant(const std::shared_ptr<const Variant>& v) { variant_vector_.push_back(v); }
 private:
  OffsetDBArray variant_vector_;
};

using OffsetDBMap = std::map<ContigOffset_t, std::unique_ptr<OffsetDB>>;   // kgl_variant_db_contig.h:21
class ContigDB {   // kgl_variant_db_contig.h:24-90
 public:
  explicit ContigDB(ContigId_t id) : contig_id_(std::move(id)) {}
  [[nodiscard]] const ContigId_t& contigId() const { return contig_id_; }
  [[nodiscard]] const OffsetDBMap& getMap() const { return contig_offset_map_; }
  bool addVariant(const std::shared_ptr<const Variant>& v) {
    auto it = contig_offset_map_.find(v->offset());
    if (it == contig_offset_map_.end()) it = contig_offset_map_.try_emplace(v->offset(), std::make_unique<OffsetDB>()).first;
    it->second->addVariant(v);
    return true;
  }
 private:
  ContigId_t contig_id_;
  OffsetDBMap contig_offset_map_;
};

using ContigDBMap = std::map<ContigId_t, std::shared_ptr<ContigDB>>;
class GenomeDB {   // kgl_variant_db_genome.h:24-95
 public:
  explicit GenomeDB(GenomeId_t id) : genome_id_(std::move(id)) {}
  [[nodiscard]] const GenomeId_t& genomeId() const { return genome_id_; }
  [[nodiscard]] const ContigDBMap& getMap() const { return contig_map_; }
  [[nodiscard]] std::optional<std::shared_ptr<const ContigDB>> getContig(const ContigId_t& id) const {
    auto it = contig_map_.find(id);
    if (it == contig_map_.end()) return std::nullopt;
    return std::shared_ptr<const ContigDB>(it->second);
  }
  bool addVariant(const std::shared_ptr<const Variant>& v) {
    auto it = contig_map_.find(v->contigId());
    if (it == contig_map_.end()) it = contig_map_.emplace(v->contigId(), std::make_shared<ContigDB>(v->contigId())).first;
    return it->second->addVariant(v);
  }
 private:
  GenomeId_t genome_id_;
  ContigDBMap contig_map_;
};

using GenomeDBMap = std::map<GenomeId_t, std::shared_ptr<GenomeDB>>;
class PopulationDB : public DataDB {   // kgl_variant_db_population.h:33-155
 public:
  PopulationDB(std::string population_id, DataSourceEnum data_source) : DataDB(data_source), population_id_(std::move(population_id)) {}
  [[nodiscard]] const std::string& fileId() const override { return population_id_; }
  [[nodiscard]] const std::string& populationId() const { return population_id_; }
  [[nodiscard]] const GenomeDBMap& getMap() const { return genome_map_; }
  [[nodiscard]] std::optional<std::shared_ptr<GenomeDB>> getCreateGenome(const GenomeId_t& id) {
    auto it = genome_map_.find(id);
    if (it == genome_map_.end()) it = genome_map_.emplace(id, std::make_shared<GenomeDB>(id)).first;
    return it->second;
  }
  bool addVariant(const std::shared_ptr<const Variant>& v, const std::vector<GenomeId_t>& genomes) {   // population.cpp:298-325
    bool ok = true;
    for (const auto& g : genomes) ok = getCreateGenome(g).value()->addVariant(v) && ok;
    return ok;
  }
 private:
  std::string population_id_;
  GenomeDBMap genome_map_;
};

// kgl_variant_db/kgl_variant_db_freq.{h,cpp}: super-population AF lookup through the INFO field named for the data source.
class FrequencyDatabaseRead {
 public:
  constexpr static const char* SUPER_POP_AFR_{"AFR"};
  constexpr static const char* SUPER_POP_AMR_{"AMR"};
  constexpr static const char* SUPER_POP_EAS_{"EAS"};
  constexpr static const char* SUPER_POP_EUR_{"EUR"};
  constexpr static const char* SUPER_POP_SAS_{"SAS"};
  constexpr static const char* SUPER_POP_ALL_{"ALL"};
  [[nodiscard]] static const std::vector<std::string>& superPopulations() {
    static const std::vector<std::string> pops{SUPER_POP_AFR_, SUPER_POP_AMR_, SUPER_POP_EAS_, SUPER_POP_EUR_, SUPER_POP_SAS_, SUPER_POP_ALL_};
    return pops;
  }
  [[nodiscard]] static std::optional<double> infoFloatField(const Variant& variant, const std::string& field);   // freq.cpp:72-122
  [[nodiscard]] static std::optional<double> superPopFrequency(const Variant& variant, const std::string& super_population);
 private:
  // Field-name table of kgl_variant_db_freq.h:84-96 and its lookup (:126-128): private in the reference as well.
  [[nodiscard]] static std::optional<std::string> lookupVariantSuperPopField(DataSourceEnum src, const std::string& sp) {
    static const std::map<std::string, std::vector<std::string>> table{
        {"AFR", {"AF_afr", "AF_afr", "AF_afr", "gnomad_AF_afr", "AFR_AF"}}, {"AMR", {"AF_amr", "AF_amr", "AF_amr", "gnomad_AF_amr", "AMR_AF"}},
        {"EAS", {"AF_eas", "AF_eas", "AF_eas", "gnomad_AF_eas", "EAS_AF"}}, {"EUR", {"AF_nfe", "AF_nfe", "AF_nfe", "gnomad_AF_nfe", "EUR_AF"}},
        {"SAS", {"AF", "AF_sas", "AF_sas", "gnomad_AF_sas", "SAS_AF"}},     {"ALL", {"AF", "AF", "AF", "gnomad_AF", "AF"}}};
    auto it = table.find(sp);
    if (it == table.end()) return std::nullopt;
    switch (src) {
      case DataSourceEnum::Gnomad2_1: return it->second[0];
      case DataSourceEnum::GnomadExomes2_1: return it->second[1];
      case DataSourceEnum::Gnomad3_1: case DataSourceEnum::GnomadExomes3_1: case DataSourceEnum::Gnomad3_0: return it->second[2];
      case DataSourceEnum::GnomadGenome3_1: return it->second[3];
      case DataSourceEnum::Genome1000: return it->second[4];
      default: return std::nullopt;
    }
  }
};
inline std::optional<double> FrequencyDatabaseRead::infoFloatField(const Variant& variant, const std::string& field) {   // freq.cpp:72-122
  const auto& info = variant.evidence().infoRecord();
  if (!info) return std::nullopt;
  auto it = info->float_fields.find(field);
  if (it == info->float_fields.end()) return std::nullopt;
  const std::vector<float>& vec = it->second;
  float f;
  if (vec.size() == 1) f = vec.front();
  else if (vec.empty()) return std::nullopt;
  else if (variant.evidence().altVariantCount() == vec.size() && variant.evidence().altVariantIndex() < vec.size())
    f = vec[variant.evidence().altVariantIndex()];
  else return std::nullopt;
  if (std::isnan(f)) return std::nullopt;
  return static_cast<double>(f);
}
inline std::optional<double> FrequencyDatabaseRead::superPopFrequency(const Variant& variant, const std::string& super_population) {
  auto field = lookupVariantSuperPopField(variant.evidence().dataSource(), super_population);
  if (!field) return 0.0;   // freq.cpp:18-23: warn and return 0.0
  return infoFloatField(variant, field.value());
}

// kgl_evidence/kgl_variant_factory_vcf_evidence_analysis.h: typed INFO read, as used by P7FrequencyFilter.
class InfoEvidenceAnalysis {
 public:
  template <typename T>
  [[nodiscard]] static std::optional<T> getTypedInfoData(const Variant& variant, const std::string& field);
};
template <>
inline std::optional<std::vector<double>> InfoEvidenceAnalysis::getTypedInfoData<std::vector<double>>(const Variant& variant,
                                                                                                    const std::string& field) {
  const auto& info = variant.evidence().infoRecord();
  if (!info) return std::nullopt;
  auto it = info->float_fields.find(field);
  if (it == info->float_fields.end()) return std::nullopt;
  return std::vector<double>(it->second.begin(), it->second.end());
}

// kgl_app/kgl_runtime.h:251-321
class ParameterMap {
 public:
  void insert(const std::string& ident, const std::string& value) { parameter_map_.emplace(ident, value); }
  [[nodiscard]] std::vector<std::string> retrieve(const std::string& ident) const {
    std::vector<std::string> out;
    auto range = parameter_map_.equal_range(ident);
    for (auto it = range.first; it != range.second; ++it) out.push_back(it->second);
    return out;
  }
  [[nodiscard]] std::optional<std::vector<double>> getFloat(const std::string& ident, size_t n = 1) const {
    auto v = retrieve(ident);
    if (n != ANY_SIZE && v.size() != n) return std::nullopt;
    std::vector<double> out;
    try { for (auto& s : v) out.push_back(std::stod(s)); } catch (...) { return std::nullopt; }
    return out;
  }
  [[nodiscard]] std::optional<std::vector<std::string>> getString(const std::string& ident, size_t n = 1) const {
    auto v = retrieve(ident);
    if (n != ANY_SIZE && v.size() != n) return std::nullopt;
    return v;
  }
  [[nodiscard]] std::optional<std::vector<size_t>> getSize(const std::string& ident, size_t n = 1) const {
    auto v = retrieve(ident);
    if (n != ANY_SIZE && v.size() != n) return std::nullopt;
    std::vector<size_t> out;
    try { for (auto& s : v) out.push_back(static_cast<size_t>(std::stoull(s))); } catch (...) { return std::nullopt; }
    return out;
  }
  [[nodiscard]] std::optional<std::vector<double>> getFloat(const std::pair<std::string, size_t>& f) const { return getFloat(f.first, f.second); }
  [[nodiscard]] std::optional<std::vector<std::string>> getString(const std::pair<std::string, size_t>& f) const { return getString(f.first, f.second); }
  [[nodiscard]] std::optional<std::vector<size_t>> getSize(const std::pair<std::string, size_t>& f) const { return getSize(f.first, f.second); }
  [[nodiscard]] std::optional<bool> getBool(const std::string& ident) const {
    auto v = retrieve(ident);
    if (v.size() != 1) return std::nullopt;
    std::string s;
    for (char c : v.front()) s += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    if (s == "TRUE") return true;      // kgl_app/kgl_runtime.cpp:295-324
    if (s == "FALSE") return false;
    return std::nullopt;
  }
  constexpr static const size_t ANY_SIZE{99999999999};
 private:
  std::multimap<std::string, std::string> parameter_map_;
};
using ParameterVector = std::vector<ParameterMap>;
using NamedParameterVector = std::pair<std::string, ParameterVector>;
using ParameterListMap = std::map<const std::string, const NamedParameterVector>;
class ActiveParameterList {
 public:
  [[nodiscard]] const ParameterListMap& getMap() const { return active_parameter_vectors_; }
  bool addNamedParameterVector(const NamedParameterVector& nv) { return active_parameter_vectors_.try_emplace(nv.first, nv).second; }
 private:
  ParameterListMap active_parameter_vectors_;
};

// kgl_app/kgl_runtime_resource.h:25-100
class ResourceBase {
 public:
  ResourceBase(std::string type, std::string ident) : resource_type_(std::move(type)), resource_ident_(std::move(ident)) {}
  virtual ~ResourceBase() = default;
  [[nodiscard]] const std::string& resourceType() const { return resource_type_; }
  [[nodiscard]] const std::string& resourceIdent() const { return resource_ident_; }
 private:
  std::string resource_type_, resource_ident_;
};
using ResourceMap = std::multimap<std::string, std::shared_ptr<const ResourceBase>>;
class AnalysisResources {
 public:
  void addResource(const std::shared_ptr<const ResourceBase>& r) { resource_map_.emplace(r->resourceType(), r); }
  [[nodiscard]] std::vector<std::shared_ptr<const ResourceBase>> getResources(const std::string& type, const std::string& ident = "") const {
    std::vector<std::shared_ptr<const ResourceBase>> out;
    auto range = resource_map_.equal_range(type);
    for (auto it = range.first; it != range.second; ++it)
      if (ident.empty() || it->second->resourceIdent() == ident) out.push_back(it->second);
    return out;
  }
  template <class ResourceClass>
  [[nodiscard]] std::shared_ptr<const ResourceClass> getSingleResource(std::string type, std::string ident = "") const {
    auto v = getResources(type, ident);
    if (v.size() != 1) ExecEnv::log().critical("Request Resource Type: {} Ident: {} expected 1 resource, found: {} resources - unrecoverable error", type, ident, v.size());
    auto p = std::dynamic_pointer_cast<const ResourceClass>(v.front());
    if (!p) ExecEnv::log().critical("Request Resource Ident: {} invalid resource type found - unrecoverable error", ident);
    return p;
  }
  [[nodiscard]] const ResourceMap& getMap() const { return resource_map_; }
 private:
  ResourceMap resource_map_;
};
struct ResourceProperties {   // kgl_app/kgl_properties_resource.h:70
  constexpr static const char GENEALOGY_RESOURCE_ID_[] = "genomeGenealogy";
  constexpr static const char PF7SAMPLE_RESOURCE_ID_[] = "Pf7Sample";      // :86
  constexpr static const char PF7FWS_RESOURCE_ID_[] = "Pf7Fws";            // :90
};

// kgl_parser/kgl_hsgenealogy_parser.h:22-135 — the PED table: genome -> super population.
// kgl_parser/kgl_hsgenealogy_parser.h:23-110 -- the PED record, every field and accessor of the reference's class.
class HsGenealogyRecord {
 public:
  HsGenealogyRecord(std::string family_id, std::string individual_id, std::string paternal_id, std::string maternal_id, std::string sex,
                    std::string pheno_type, std::string population, std::string population_description, std::string super_population,
                    std::string super_description, std::string relationship, std::string siblings, std::string second_order,
                    std::string third_order, std::string comments)
      : family_id_(std::move(family_id)), individual_id_(std::move(individual_id)), paternal_id_(std::move(paternal_id)),
        maternal_id_(std::move(maternal_id)), sex_(std::move(sex)), pheno_type_(std::move(pheno_type)), population_(std::move(population)),
        population_description_(std::move(population_description)), super_population_(std::move(super_population)),
        super_description_(std::move(super_description)), relationship_(std::move(relationship)), siblings_(std::move(siblings)),
        second_order_(std::move(second_order)), third_order_(std::move(third_order)), comments_(std::move(comments)) {}
  // the two fields the sweep itself needs (tests that carry no PED file)
  HsGenealogyRecord(std::string individual, std::string super_population)
      : individual_id_(std::move(individual)), super_population_(std::move(super_population)) {}
  [[nodiscard]] const std::string& familyId() const { return family_id_; }
  [[nodiscard]] const std::string& individualId() const { return individual_id_; }
  [[nodiscard]] const std::string& paternalId() const { return paternal_id_; }
  [[nodiscard]] const std::string& maternalId() const { return maternal_id_; }
  [[nodiscard]] const std::string& sex() const { return sex_; }
  [[nodiscard]] const std::string& phenoType() const { return pheno_type_; }
  [[nodiscard]] const std::string& population() const { return population_; }
  [[nodiscard]] const std::string& populationDescription() const { return population_description_; }
  [[nodiscard]] const std::string& superPopulation() const { return super_population_; }
  [[nodiscard]] const std::string& superDescription() const { return super_description_; }
  [[nodiscard]] const std::string& relationship() const { return relationship_; }
  [[nodiscard]] const std::string& siblings() const { return siblings_; }
  [[nodiscard]] const std::string& secondOrder() const { return second_order_; }
  [[nodiscard]] const std::string& thirdOrder() const { return third_order_; }
  [[nodiscard]] const std::string& comments() const { return comments_; }
  [[nodiscard]] static size_t genealogyFieldCount() { return 15; }
 private:
  std::string family_id_, individual_id_, paternal_id_, maternal_id_, sex_, pheno_type_, population_, population_description_,
      super_population_, super_description_, relationship_, siblings_, second_order_, third_order_, comments_;
};
class HsGenomeGenealogyData : public ResourceBase {
 public:
  explicit HsGenomeGenealogyData(std::string ident) : ResourceBase(ResourceProperties::GENEALOGY_RESOURCE_ID_, std::move(ident)) {}
  bool addGenealogyRecord(const HsGenealogyRecord& r) { return map_.try_emplace(r.individualId(), r).second; }   // kgl_hsgenealogy_parser.cpp: false on a repeated individual
  [[nodiscard]] std::optional<HsGenealogyRecord> getGenomeGenealogyRecord(const std::string& genome) const {
    auto it = map_.find(genome);
    if (it == map_.end()) return std::nullopt;
    return it->second;
  }
 private:
  std::map<std::string, HsGenealogyRecord> map_;
};

// kgl_app/kgl_package_analysis_virtual.h:20-55 — the plugin boundary.
class VirtualAnalysis {
 public:
  VirtualAnalysis() = default;
  virtual ~VirtualAnalysis() = default;
  [[nodiscard]] virtual std::string ident() const = 0;
  [[nodiscard]] virtual bool initializeAnalysis(const std::string& work_directory, const ActiveParameterList& named_parameters,
                                                const std::shared_ptr<const AnalysisResources>& resource_ptr) = 0;
  [[nodiscard]] virtual bool fileReadAnalysis(std::shared_ptr<const DataDB> data_object_ptr) = 0;
  [[nodiscard]] virtual bool iterationAnalysis() = 0;
  [[nodiscard]] virtual bool finalizeAnalysis() = 0;
  using AnalysisFactoryMap = std::map<std::string, std::function<std::unique_ptr<VirtualAnalysis>(void)>>;
};

}  // namespace genome
}  // namespace kellerberrin

#endif  // KGX_REFSHIM_H
