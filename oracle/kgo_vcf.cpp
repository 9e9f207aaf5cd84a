// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
// VCF text -> PopulationDB.  The unphased P. falciparum (Pf7) flavour, at the end of the file, restates
//   PfVCFImpl::ParseRecord / setupPopulationStructure / createAddVariant   kgl_parser/kgl_variant_factory_pf_impl.cpp:73-467
//   Variant::canonicalSequences / isCanonical                              kgl_variant_db/kgl_variant_db.cpp:165-220
//   AlphabetString::commonPrefix / commonSuffix / removePrefixSuffix       kgl_sequence/kgl_alphabet_string.h:249-314
//   P7VariantFilter::applyFilter                                           kgl_variant_filter/kgl_variant_filter_Pf7.cpp:131-318
// The phased-diploid (1000 Genomes) flavour is restated from
//   ParseVCF::moveToVcfRecord                 kgl_genomics/kgl_parser/kgl_variant_vcf_impl.cpp:96-170
//   Genome1000VCFImpl::ParseRecord            kgl_parser/kgl_variant_factory_1000_impl.cpp:63-145
//   Genome1000VCFImpl::alternateIndex         :148-272   (constants kgl_variant_factory_1000_impl.h:55-63)
//   Genome1000VCFImpl::addVariants            :274-318
//   VCFInfoParser float conversion (std::stof) kgl_parser/kgl_variant_factory_vcf_parse_info.cpp:205-260
//   Utility::viewTokenizer / trimEndWhiteSpace kel_utility/kel_utility.cpp:190-295
#include <cctype>
#include <cstring>
#include <limits>
#include <string_view>

#include "kgo_core.h"

namespace kgo {

static std::vector<std::string_view> viewTokenizer(std::string_view str_view, char delim) {
  std::vector<std::string_view> token_vector;
  size_t token_index = 0, index = 0;
  for (; index < str_view.size(); ++index) {
    if (str_view[index] == delim) {
      token_vector.emplace_back(str_view.data() + token_index, index - token_index);
      token_index = index + 1;
    }
  }
  if (token_index > index) token_vector.emplace_back();
  else token_vector.emplace_back(str_view.data() + token_index, index - token_index);
  return token_vector;
}

static std::string trimWhiteSpace(const std::string& s) {
  size_t b = 0, e = s.size();
  while (b < e && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
  while (e > b && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
  return s.substr(b, e - b);
}

enum class ChromosomeType { AUTOSOME, ALLOSOME_X, ALLOSOME_Y };

static ChromosomeType lookupType(const std::string& contig) {
  if (contig == "X" || contig == "chrX") return ChromosomeType::ALLOSOME_X;
  if (contig == "Y" || contig == "chrY") return ChromosomeType::ALLOSOME_Y;
  return ChromosomeType::AUTOSOME;
}

// Genome1000VCFImpl::alternateIndex: { phase A alt, phase B alt }, 0 = reference.
std::pair<size_t, size_t> alternateIndex1000(const std::string& contig, const std::string& genotype, size_t n_alt) {
  constexpr size_t REFERENCE_VARIANT_INDEX_ = 0;
  const std::string trim_genotype = trimWhiteSpace(genotype);
  if (trim_genotype.empty()) return {REFERENCE_VARIANT_INDEX_, REFERENCE_VARIANT_INDEX_};
  size_t GT_size = trim_genotype.find(':');
  if (GT_size == std::string::npos) GT_size = trim_genotype.size();
  std::string_view unphased_view(trim_genotype.c_str(), GT_size);
  std::vector<std::string_view> phase_vector = viewTokenizer(unphased_view, '|');
  size_t phase_A_alt = REFERENCE_VARIANT_INDEX_, phase_B_alt = REFERENCE_VARIANT_INDEX_;
  try {
    if (phase_vector.size() == 1) {
      if (unphased_view != "." && unphased_view != "-") {
        switch (lookupType(contig)) {
          case ChromosomeType::ALLOSOME_X: phase_A_alt = std::stoul(std::string(unphased_view)); break;
          case ChromosomeType::ALLOSOME_Y: phase_B_alt = std::stoul(std::string(unphased_view)); break;
          default: break;   // autosome with a single phase: warned, stays reference
        }
      }
      if (phase_A_alt > n_alt || phase_B_alt > n_alt) return {REFERENCE_VARIANT_INDEX_, REFERENCE_VARIANT_INDEX_};
      return {phase_A_alt, phase_B_alt};
    }
    if (phase_vector[0].find('<') == std::string_view::npos) {
      if (phase_vector[0] != "." && phase_vector[0] != "-") phase_A_alt = std::stoul(std::string(phase_vector[0]));
    }
    if (phase_vector[1].find('<') == std::string_view::npos) {
      // as written in the reference: the second test looks at phase_vector[0]
      if (phase_vector[1] != "." && phase_vector[0] != "-") phase_B_alt = std::stoul(std::string(phase_vector[1]));
    }
  } catch (...) {
    return {REFERENCE_VARIANT_INDEX_, REFERENCE_VARIANT_INDEX_};
  }
  if (phase_A_alt > n_alt || phase_B_alt > n_alt) return {REFERENCE_VARIANT_INDEX_, REFERENCE_VARIANT_INDEX_};
  return {phase_A_alt, phase_B_alt};
}

static float convertToFloat(const std::string& value) {
  if (value == ".") return std::numeric_limits<float>::quiet_NaN();
  std::string uc;
  for (char c : value) uc += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  if (uc == "NAN") return std::numeric_limits<float>::quiet_NaN();
  try {
    return std::stof(value);
  } catch (std::out_of_range&) {
    if (uc.find("E-") != std::string::npos) return std::numeric_limits<float>::min();
    if (uc.find("E") != std::string::npos) return std::numeric_limits<float>::max();
    return std::numeric_limits<float>::quiet_NaN();
  } catch (...) {
    return std::numeric_limits<float>::quiet_NaN();
  }
}

// INFO "k=v;k2=a,b" -> per super-population AF vectors using the 1000-Genomes field names (kgl_variant_db_freq.h:84-96).
static const char* const kFields1000[SUPER_POP_COUNT] = {"AFR_AF", "AMR_AF", "EAS_AF", "EUR_AF", "SAS_AF", "AF"};

// Field names per data source (kgl_variant_db_freq.h:84-96); nullptr for a source the table does not hold.
static const char* const* superPopFields(const std::string& source) {
  static const char* const gnomad2_1[SUPER_POP_COUNT] = {"AF_afr", "AF_amr", "AF_eas", "AF_nfe", "AF", "AF"};
  static const char* const gnomad_ex2_1[SUPER_POP_COUNT] = {"AF_afr", "AF_amr", "AF_eas", "AF_nfe", "AF_sas", "AF"};
  static const char* const gnomad_genome3_1[SUPER_POP_COUNT] = {"gnomad_AF_afr", "gnomad_AF_amr", "gnomad_AF_eas", "gnomad_AF_nfe", "gnomad_AF_sas", "gnomad_AF"};
  if (source == "Gnomad2_1") return gnomad2_1;
  if (source == "GnomadExomes2_1" || source == "Gnomad3_1" || source == "GnomadExomes3_1" || source == "Gnomad3_0") return gnomad_ex2_1;
  if (source == "GnomadGenome3_1") return gnomad_genome3_1;
  if (source == "Genome1000") return kFields1000;
  return nullptr;
}

static void parseInfoAF(std::string_view info, uint32_t n_alt, std::vector<float>& af, int& info_af_size,
                        const char* const* fields = kFields1000) {
  info_af_size = -1;
  af.assign(static_cast<size_t>(SUPER_POP_COUNT) * n_alt, std::numeric_limits<float>::quiet_NaN());
  for (auto item : viewTokenizer(info, ';')) {
    const size_t eq = item.find('=');
    if (eq == std::string_view::npos) continue;
    const std::string_view key = item.substr(0, eq), value = item.substr(eq + 1);
    for (int sp = 0; sp < SUPER_POP_COUNT; ++sp) {
      if (key != fields[sp]) continue;
      auto values = viewTokenizer(value, ',');
      if (sp == ALL && key == "AF") info_af_size = static_cast<int>(values.size());
      if (values.size() == n_alt) {
        for (uint32_t a = 0; a < n_alt; ++a) af[static_cast<size_t>(sp) * n_alt + a] = convertToFloat(std::string(values[a]));
      } else if (values.size() == 1) {       // scalar field: the same value for every alt (infoFloatField, kgl_variant_db_freq.cpp:78-81)
        for (uint32_t a = 0; a < n_alt; ++a) af[static_cast<size_t>(sp) * n_alt + a] = convertToFloat(std::string(values[0]));
      }
    }
  }
}

// VCFParseHeader::parseHeader / parseVcfHeader / tokenizeVcfHeaderKeyValues (kgl_parser/kgl_variant_factory_vcf_parse_header.cpp:17-265)
// for the one header line VariantSort needs: ##INFO=<ID=vep,...,Description="... Format: a|b|c">.  '<' and '>' are erased
// everywhere in the value; items split on ',' outside double quotes ('\\' escapes, quotes dropped:
// boost::escaped_list_separator); an item splits on '=' with empty tokens dropped, so a value ends at its first '='; a
// record without Type or Number is ignored; the sub-field names follow "Format: " and a repeated name voids the header
// (VEPSubFieldHeader::parseHeader, kgl_evidence/kgl_variant_factory_vcf_evidence.cpp:24-58).
std::shared_ptr<const std::vector<std::string>> parseVepHeader(std::string_view text) {
  std::shared_ptr<const std::vector<std::string>> result;
  for (auto line : viewTokenizer(text, '\n')) {
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (line.rfind("#CHROM", 0) == 0) break;
    const size_t eq = line.find('=');
    if (eq == std::string_view::npos) continue;
    std::string key(line.substr(0, eq));
    if (const size_t hash = key.find('#'); hash != std::string::npos) key.erase(hash, hash + 2);
    for (char& c : key) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    if (key != "INFO") continue;
    std::string value;
    for (char c : line.substr(eq + 1)) if (c != '<' && c != '>') value += c;
    std::vector<std::string> items(1);
    bool in_quote = false;
    for (size_t i = 0; i < value.size(); ++i) {
      const char c = value[i];
      if (c == '\\' && i + 1 < value.size()) { items.back() += value[++i] == 'n' ? '\n' : value[i]; continue; }
      if (c == '"') { in_quote = !in_quote; continue; }
      if (c == ',' && !in_quote) { items.emplace_back(); continue; }
      items.back() += c;
    }
    std::map<std::string, std::string> item_map;
    for (const auto& item : items) {
      std::vector<std::string> item_vec;
      for (auto token : viewTokenizer(item, '=')) if (!token.empty()) item_vec.emplace_back(token);
      if (item_vec.empty()) continue;
      for (char& c : item_vec[0]) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
      item_map[item_vec[0]] = item_vec.size() >= 2 ? item_vec[1] : std::string();
    }
    const auto id = item_map.find("ID");
    if (id == item_map.end() || id->second != "vep") continue;
    if (item_map.find("TYPE") == item_map.end() || item_map.find("NUMBER") == item_map.end()) continue;
    const auto description = item_map.find("DESCRIPTION");
    auto headers = std::make_shared<std::vector<std::string>>();
    if (description != item_map.end()) {
      const size_t at = description->second.find("Format: ");
      if (at != std::string::npos) {
        const std::string unparsed_header = description->second.substr(at + 8);   // lives past the views cut from it
        for (auto sub_field : viewTokenizer(unparsed_header, '|')) headers->emplace_back(sub_field);
        std::map<std::string, size_t> index_map;
        for (size_t i = 0; i < headers->size(); ++i)
          if (!index_map.try_emplace((*headers)[i], i).second) { headers->clear(); break; }
      }
    }
    result = headers;    // vcf_info_map[ID] = record: a later header line of the same ID replaces an earlier one
  }
  return result;
}

// What a record hands every Variant cut from it beyond PASS and AF: the ID column and the "vep" INFO vector
// (VCFInfoParser::infoTokenParser, kgl_parser/kgl_variant_factory_vcf_parse_info.cpp:14-146: fields split on ';', key and
// value on the FIRST '=', the first of two equal keys is kept; a String vector splits on ',').
static void recordAnnotation(RecordEvidence& ev, std::string_view id, std::string_view info,
                             const std::shared_ptr<const std::vector<std::string>>& vep_header) {
  // moveToVcfRecord (kgl_variant_vcf_impl.cpp:119-130): "." is no identifier; otherwise the text less its trailing white space
  if (id != ".") {
    ev.identifier = std::string(id);
    while (!ev.identifier.empty() && std::isspace(static_cast<unsigned char>(ev.identifier.back()))) ev.identifier.pop_back();
  }
  ev.vep_header = vep_header;
  for (auto item : viewTokenizer(info, ';')) {
    const size_t eq = item.find('=');
    if ((eq == std::string_view::npos ? item : item.substr(0, eq)) != "vep") continue;
    if (eq != std::string_view::npos)
      for (auto value : viewTokenizer(item.substr(eq + 1), ',')) ev.vep.emplace_back(value);
    break;
  }
}

// The whole path: text -> records -> Variants added to genomes.  Returns the number of records parsed, -1 on a
// malformed file.  sample names come from the #CHROM line.
long addVcf1000(PopulationDB& population, std::string_view text, std::vector<std::string>* genome_names_out) {
  std::vector<std::string> genome_names;
  const auto vep_header = parseVepHeader(text);
  long n_records = 0;
  size_t line_number = 0;
  for (auto line : viewTokenizer(text, '\n')) {
    ++line_number;
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (line.empty()) continue;
    if (line[0] == '#') {
      if (line.rfind("#CHROM", 0) == 0) {
        auto f = viewTokenizer(line, '\t');
        for (size_t i = 9; i < f.size(); ++i) genome_names.emplace_back(f[i]);
      }
      continue;
    }
    auto field_views = viewTokenizer(line, '\t');
    if (field_views.size() < 8) continue;                                   // error + record dropped
    const std::string contig(field_views[0]);
    const uint64_t offset = std::stoull(std::string(field_views[1])) - 1;   // VCF is 1-based
    const std::string ref(field_views[3]);
    const std::string alt = field_views[4] == "." ? std::string() : std::string(field_views[4]);
    std::string filter_uc;
    for (char c : field_views[6]) filter_uc += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    const bool passed_filter = filter_uc == "PASS";
    std::vector<std::string> alt_vector;
    for (auto a : viewTokenizer(alt, ',')) alt_vector.emplace_back(a);
    auto ev = std::make_shared<RecordEvidence>();
    ev->record_index = static_cast<uint64_t>(n_records);
    ev->pass = passed_filter;
    ev->alt_count = static_cast<uint32_t>(alt_vector.size());
    parseInfoAF(field_views[7], ev->alt_count, ev->af, ev->info_af_size);
    recordAnnotation(*ev, field_views[2], field_views[7], vep_header);

    std::map<size_t, std::vector<std::string>> phase_A_map, phase_B_map;
    for (size_t idx = 9; idx < field_views.size(); ++idx) {
      const size_t genotype_count = idx - 9;
      if (genotype_count >= genome_names.size()) break;
      const auto [A_index, B_index] = alternateIndex1000(contig, std::string(field_views[idx]), alt_vector.size());
      if (A_index != 0) phase_A_map[A_index - 1].push_back(genome_names[genotype_count]);
      if (B_index != 0) phase_B_map[B_index - 1].push_back(genome_names[genotype_count]);
    }
    for (int phase = 0; phase < 2; ++phase) {
      for (const auto& [alt_allele, genome_vector] : (phase == 0 ? phase_A_map : phase_B_map)) {
        auto v = std::make_shared<const Variant>(contig, offset, phase == 0 ? VariantPhase::DIPLOID_PHASE_A : VariantPhase::DIPLOID_PHASE_B,
                                                 ref, alt_vector[alt_allele], ev, static_cast<uint32_t>(alt_allele));
        population.addVariant(v, genome_vector);
      }
    }
    ++n_records;
  }
  if (genome_names_out) *genome_names_out = genome_names;
  return n_records;
}

// ---- the unphased mono-genome frequency sources (Gnomad ...) -------------------------------------------------------

// GrchVCFImpl::ProcessVCFRecord (kgl_parser/kgl_variant_factory_grch_impl.cpp:53-156): one UNPHASED Variant per alt, as
// written (no canonicalisation), all in the one reference genome; an ALT without ',' (or empty) is a single alt.
long addVcfMonoGenome(PopulationDB& population, std::string_view text, const std::string& source, const std::string& genome_id) {
  const char* const* fields = superPopFields(source);
  if (!fields) return -1;
  long n_records = 0;
  const auto vep_header = parseVepHeader(text);
  const std::vector<std::string> genome_vector{genome_id};
  for (auto line : viewTokenizer(text, '\n')) {
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (line.empty() || line[0] == '#') continue;
    auto field_views = viewTokenizer(line, '\t');
    if (field_views.size() < 8) continue;
    const std::string contig(field_views[0]);
    const uint64_t offset = std::stoull(std::string(field_views[1])) - 1;
    const std::string ref(field_views[3]);
    const std::string alt = field_views[4] == "." ? std::string() : std::string(field_views[4]);   // moveToVcfRecord: "." is a missing alt (kgl_variant_vcf_impl.cpp:133-141)
    std::string filter_uc;
    for (char c : field_views[6]) filter_uc += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    std::vector<std::string> alt_vector;
    if (alt.find(',') == std::string::npos || alt.empty()) alt_vector.push_back(alt);
    else for (auto a : viewTokenizer(alt, ',')) alt_vector.emplace_back(a);
    auto ev = std::make_shared<RecordEvidence>();
    ev->record_index = static_cast<uint64_t>(n_records);
    ev->pass = filter_uc == "PASS";
    ev->alt_count = static_cast<uint32_t>(alt_vector.size());
    parseInfoAF(field_views[7], ev->alt_count, ev->af, ev->info_af_size, fields);
    recordAnnotation(*ev, field_views[2], field_views[7], vep_header);
    for (uint32_t a = 0; a < alt_vector.size(); ++a)
      population.addVariant(std::make_shared<const Variant>(contig, offset, VariantPhase::UNPHASED, ref, alt_vector[a], ev, a), genome_vector);
    ++n_records;
  }
  return n_records;
}

// ---- the unphased P. falciparum flavour ---------------------------------------------------------------------------

// Variant::canonicalSequences: SNPs as '1X', deletes as '1MnD', inserts as '1MnI'; everything else loses its common
// prefix (all but one base) and suffix, with the reference's unsigned arithmetic.
void canonicalSequences(const std::string& ref, const std::string& alt, uint64_t offset, std::string& c_ref, std::string& c_alt,
                        uint64_t& c_offset) {
  const bool canonical = (ref.size() == 1 && alt.size() == 1) || (alt.size() == 1 && ref.size() > alt.size()) ||
                         (ref.size() == 1 && ref.size() < alt.size());
  if (canonical) {
    c_ref = ref; c_alt = alt; c_offset = offset;
    return;
  }
  const size_t common_size = std::min(ref.size(), alt.size());
  size_t prefix_size = 0;
  while (prefix_size < common_size && ref[prefix_size] == alt[prefix_size]) ++prefix_size;
  prefix_size = prefix_size > 0 ? (prefix_size - 1) : 0;
  size_t suffix_size = 0;
  while (suffix_size < common_size && ref[ref.size() - 1 - suffix_size] == alt[alt.size() - 1 - suffix_size]) ++suffix_size;
  const size_t min_size = common_size;
  int64_t adj_suffix_size = static_cast<int64_t>(std::min(min_size - prefix_size - 1, suffix_size));   // size_t arithmetic, as written
  adj_suffix_size = adj_suffix_size < 0 ? 0 : adj_suffix_size;
  auto remove = [&](const std::string& s) {
    // std::ranges::next(begin, n, end) / prev(end, n, begin): bounded steps; empty if nothing is left
    const size_t from = std::min(prefix_size, s.size());
    const size_t drop = std::min(static_cast<size_t>(adj_suffix_size), s.size());
    const size_t to = s.size() - drop;
    return to > from ? s.substr(from, to - from) : std::string();
  };
  c_ref = remove(ref);
  c_alt = remove(alt);
  c_offset = offset + prefix_size;
}

static bool allDigits(const std::string& s) { return s.find_first_not_of("0123456789") == std::string::npos; }

// VCFInfoParser::convertToFloat with "missing" kept apart from a value (kgl_variant_factory_vcf_parse_info.cpp:206-279).
static std::optional<float> convertToFloatValue(const std::string& value) {
  if (value.size() == 3) {
    std::string uc;
    for (char c : value) uc += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    if (uc == "NAN") return std::nullopt;
  }
  try {
    return std::stof(value);
  } catch (std::out_of_range&) {
    std::string uc;
    for (char c : value) uc += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    if (uc.find("E-") != std::string::npos) return std::numeric_limits<float>::min();
    if (uc.find("E") != std::string::npos) return std::numeric_limits<float>::max();
    return std::nullopt;
  } catch (...) {
    return std::nullopt;
  }
}

// P7VariantFilter::applyFilter: VQSLOD alone when the record has it, else every other threshold the record has a value for.
bool p7VariantFilter(const Variant& v) {
  const RecordEvidence& ev = v.evidence();
  if (auto x = ev.infoScalar("VQSLOD")) return x.value() >= 0.0;
  if (auto x = ev.infoScalar("QD"); x && !(x.value() >= 2.0)) return false;
  if (auto x = ev.infoScalar("MQ")) {
    const double mq_level = v.contigId() == "Pf3D7_MIT_v3" ? 5.0 : 30.0;
    if (!(x.value() >= mq_level)) return false;
  }
  if (auto x = ev.infoScalar("SOR"); x && !(x.value() <= 3.0)) return false;
  if (auto x = ev.infoScalar("MQRankSum"); x && !(x.value() >= -12.5)) return false;
  if (auto x = ev.infoScalar("ReadPosRankSum"); x && !(x.value() >= -8.0)) return false;
  return true;
}

// text -> records -> UNPHASED canonical Variants, one per called alt copy.  Every sample of the #CHROM line becomes a
// genome holding every contig of the ##contig header lines (setupPopulationStructure), carrier or not.
long addVcfPf(PopulationDB& population, std::string_view text, std::vector<std::string>* genome_names_out) {
  std::vector<std::string> genome_names, header_contigs;
  const auto vep_header = parseVepHeader(text);
  bool structure_done = false;
  auto setupPopulationStructure = [&]() {
    if (structure_done) return;
    structure_done = true;
    for (const auto& genome_id : genome_names) {
      auto genome = population.getCreateGenome(genome_id);
      for (const auto& contig : header_contigs) genome->getCreateContig(contig);
    }
  };
  long n_records = 0;
  for (auto line : viewTokenizer(text, '\n')) {
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (line.empty()) continue;
    if (line[0] == '#') {
      if (line.rfind("##contig=<", 0) == 0) {
        const size_t id = line.find("ID=");
        if (id != std::string_view::npos) {
          size_t end = line.find_first_of(",>", id);
          if (end == std::string_view::npos) end = line.size();
          header_contigs.emplace_back(line.substr(id + 3, end - id - 3));
        }
      } else if (line.rfind("#CHROM", 0) == 0) {
        auto f = viewTokenizer(line, '\t');
        for (size_t i = 9; i < f.size(); ++i) genome_names.emplace_back(f[i]);
      }
      continue;
    }
    setupPopulationStructure();
    auto field_views = viewTokenizer(line, '\t');
    if (field_views.size() < 9) continue;                                    // no FORMAT column: nothing to read
    const std::string contig(field_views[0]);
    const uint64_t offset = std::stoull(std::string(field_views[1])) - 1;
    const std::string reference(field_views[3]);
    std::vector<std::string> alleles;
    for (auto a : viewTokenizer(field_views[4] == "." ? std::string_view() : field_views[4], ',')) alleles.emplace_back(a);   // "." = missing alt (kgl_variant_vcf_impl.cpp:133-141)
    std::string filter_uc;
    for (char c : field_views[6]) filter_uc += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    std::vector<std::string> format_fields;
    for (auto a : viewTokenizer(field_views[8], ':')) format_fields.emplace_back(a);
    auto formatIndex = [&](const char* code) -> std::optional<size_t> {
      for (size_t i = 0; i < format_fields.size(); ++i)
        if (format_fields[i] == code) return i;
      return std::nullopt;
    };
    const auto GT_offset_opt = formatIndex("GT"), AD_offset_opt = formatIndex("AD");
    ++n_records;
    if (!GT_offset_opt || !AD_offset_opt) continue;                          // error: record contributes nothing

    auto ev_base = std::make_shared<RecordEvidence>();
    ev_base->record_index = static_cast<uint64_t>(n_records - 1);
    ev_base->pass = filter_uc == "PASS";
    ev_base->alt_count = static_cast<uint32_t>(alleles.size());
    parseInfoAF(field_views[7], ev_base->alt_count, ev_base->af, ev_base->info_af_size);
    recordAnnotation(*ev_base, field_views[2], field_views[7], vep_header);
    for (auto item : viewTokenizer(field_views[7], ';')) {
      const size_t eq = item.find('=');
      if (eq == std::string_view::npos) continue;
      const std::string key(item.substr(0, eq));
      if (key != "VQSLOD" && key != "QD" && key != "MQ" && key != "SOR" && key != "MQRankSum" && key != "ReadPosRankSum") continue;
      if (auto value = convertToFloatValue(std::string(item.substr(eq + 1)))) ev_base->info_scalar.emplace_back(key, value.value());
    }

    try {   // ProcessVCFRecord catches whatever ParseRecord throws: the rest of the record is lost, what was added stays
      for (size_t genotype_count = 0; genotype_count + 9 < field_views.size() && genotype_count < genome_names.size(); ++genotype_count) {
        const std::string genotype(field_views[9 + genotype_count]);
        const std::vector<std::string> genome_vector{genome_names[genotype_count]};
        auto genotype_formats = viewTokenizer(genotype, ':');
        if (genotype_formats.size() <= GT_offset_opt.value()) continue;
        const std::string GT_format(genotype_formats[GT_offset_opt.value()]);
        std::vector<std::string> gt_vector;
        for (auto a : viewTokenizer(GT_format, '/')) gt_vector.emplace_back(a);
        if (gt_vector.size() != 2) {
          gt_vector.clear();
          for (auto a : viewTokenizer(GT_format, '|')) gt_vector.emplace_back(a);
          if (gt_vector.size() != 2) continue;                               // missing ('.') or not diploid
        }
        size_t A_allele = 0, B_allele = 0;
        if (allDigits(gt_vector[0])) A_allele = static_cast<size_t>(std::stoll(gt_vector[0]));
        if (allDigits(gt_vector[0])) B_allele = static_cast<size_t>(std::stoll(gt_vector[1]));   // [0] tested, [1] converted: as written
        if (A_allele == 0 && B_allele == 0) continue;
        if (genotype_formats.size() <= AD_offset_opt.value()) continue;
        const std::string AD_text(genotype_formats[AD_offset_opt.value()]);
        auto ad_vector = viewTokenizer(AD_text, ',');
        if (ad_vector.size() != alleles.size() + 1) continue;
        std::vector<size_t> ad_count_vector;
        for (auto depth_count_text : ad_vector) {
          const std::string t(depth_count_text);
          if (!allDigits(t)) continue;
          ad_count_vector.push_back(static_cast<size_t>(std::stoll(t)));
        }
        for (const size_t allele_number : {A_allele, B_allele}) {
          if (allele_number == 0) continue;
          // the reference indexes alleles()[n-1] and ad_count_vector[n] unchecked (undefined past the end): skipped here
          if (allele_number > alleles.size() || allele_number >= ad_count_vector.size()) continue;
          const std::string& allele = alleles[allele_number - 1];
          const size_t ref_count = ad_count_vector[0], alt_count = ad_count_vector[allele_number];
          const bool downstream_variant = ref_count == 0 && alt_count == 0;   // spanning upstream deletion
          if (allele == "*" || downstream_variant) continue;
          std::string c_ref, c_alt;
          uint64_t c_offset = 0;
          canonicalSequences(reference, allele, offset, c_ref, c_alt, c_offset);
          auto v = std::make_shared<const Variant>(contig, c_offset, VariantPhase::UNPHASED, c_ref, c_alt, ev_base,
                                                   static_cast<uint32_t>(allele_number - 1));
          population.addVariant(v, genome_vector);
        }
      }
    } catch (const std::exception&) {
    }
  }
  setupPopulationStructure();
  if (genome_names_out) *genome_names_out = genome_names;
  return n_records;
}

}  // namespace kgo
