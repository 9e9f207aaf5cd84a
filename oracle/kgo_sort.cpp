// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
#include "kgo_sort.h"

namespace kgo {

static std::vector<std::string_view> vepSubFields(std::string_view vep_field) {   // Utility::viewTokenizer(.., '|')
  std::vector<std::string_view> token_vector;
  size_t token_index = 0, index = 0;
  for (; index < vep_field.size(); ++index) {
    if (vep_field[index] == '|') {
      token_vector.emplace_back(vep_field.data() + token_index, index - token_index);
      token_index = index + 1;
    }
  }
  if (token_index > index) token_vector.emplace_back();
  else token_vector.emplace_back(vep_field.data() + token_index, index - token_index);
  return token_vector;
}

std::optional<std::vector<std::string>> vepCheckedFields(const Variant& variant) {
  const RecordEvidence& ev = variant.evidence();
  if (!ev.vep_header) return std::nullopt;                      // the field is not in the header: not subscribed (:29-35)
  if (ev.vep_header->empty()) return std::nullopt;              // "found empty 'vep' header" (:51-56)
  if (ev.vep.empty()) return std::nullopt;                      // header but no data (:71-76)
  std::vector<std::string> checked_field_vector;
  for (const auto& vep_field : ev.vep)                          // the Gnomad 3 work-around: entries of another size are dropped (:86-114)
    if (vepSubFields(vep_field).size() == ev.vep_header->size()) checked_field_vector.push_back(vep_field);
  if (checked_field_vector.empty()) return std::nullopt;
  return checked_field_vector;
}

using VepIndexVector = std::vector<std::pair<std::string, size_t>>;

static VepIndexVector getVepIndexes(const Variant& variant, const std::vector<std::string>& vep_field_list) {
  if (!vepCheckedFields(variant)) return {};
  const auto& headers = *variant.evidence().vep_header;
  VepIndexVector field_index;
  for (const auto& field_header : vep_field_list) {             // (an empty list, every field, is not used by VariantSort)
    auto it = std::find(headers.begin(), headers.end(), field_header);
    if (it == headers.end()) return {};                         // error: sub field not found (:154-159)
    field_index.emplace_back(field_header, static_cast<size_t>(it - headers.begin()));
  }
  return field_index;
}

static std::vector<std::map<std::string, std::string>> getVepData(const Variant& variant, const VepIndexVector& vep_field_list) {
  if (vep_field_list.empty()) return {};
  auto fields = vepCheckedFields(variant);
  if (!fields) return {};
  std::vector<std::map<std::string, std::string>> vep_value_map;
  for (const auto& vep_field : fields.value()) {
    const auto sub_field_vector = vepSubFields(vep_field);
    std::map<std::string, std::string> sub_field_map;
    for (const auto& [field_ident, index] : vep_field_list) sub_field_map[field_ident] = std::string(sub_field_vector[index]);
    vep_value_map.push_back(std::move(sub_field_map));
  }
  return vep_value_map;
}

std::shared_ptr<EnsemblIndexMap> VariantSort::ensemblIndex(const std::shared_ptr<const PopulationDB>& population) {
  auto index = std::make_shared<EnsemblIndexMap>();
  ensemblAddIndex(population, {}, index);
  return index;
}

void VariantSort::ensemblAddIndex(const std::shared_ptr<const PopulationDB>& population, const std::vector<std::string>& ensembl_gene_list,
                                  std::shared_ptr<EnsemblIndexMap>& index_map) {
  const std::set<std::string> ensembl_gene_set(ensembl_gene_list.begin(), ensembl_gene_list.end());
  std::set<std::string> unique_ident;
  VepIndexVector field_index;
  bool initialized = false;
  population->processAll([&](const VariantPtr& variant) {
    if (!initialized) {
      // The "Gene" column is looked up ONCE, on the first variant visited; if that variant has no usable vep data the
      // index vector stays empty and nothing is ever indexed (kgl_variant_sort.cpp:56-63).
      field_index = getVepIndexes(*variant, {"Gene"});
      initialized = true;
    }
    for (const auto& field : getVepData(*variant, field_index))
      if (!field.empty() && !field.begin()->second.empty()) unique_ident.insert(field.begin()->second);
    for (const auto& ident : unique_ident)
      if (!ident.empty() && (ensembl_gene_set.empty() || ensembl_gene_set.count(ident))) index_map->emplace(ident, variant);
    unique_ident.clear();
    return true;
  });
}

size_t VariantSort::nonEnsemblIdentifiers(const EnsemblIndexMap& index_map) {
  size_t non_ensembl_identifiers = 0;
  for (const auto& [ident, variant] : index_map)
    if (ident.find("ENSG") == std::string::npos) ++non_ensembl_identifiers;
  return non_ensembl_identifiers;
}

std::shared_ptr<VariantIdIndexMap> VariantSort::variantIdIndex(const std::shared_ptr<const PopulationDB>& population) {
  auto index = std::make_shared<VariantIdIndexMap>();
  population->processAll([&](const VariantPtr& variant) {
    if (!variant->identifier().empty()) index->emplace(variant->identifier(), variant);   // the first visit keeps the key
    return true;
  });
  return index;
}

static std::shared_ptr<VariantIdIndexMap> indexGenome(const GenomeDB& genome) {
  auto index = std::make_shared<VariantIdIndexMap>();
  genome.processAll([&](const VariantPtr& variant) {
    if (!variant->identifier().empty()) index->emplace(variant->identifier(), variant);
    return true;
  });
  return index;
}

std::shared_ptr<VariantGenomeIndexMap> VariantSort::variantGenomeIndex(const std::shared_ptr<const PopulationDB>& population) {
  auto genome_index_map = std::make_shared<VariantGenomeIndexMap>();
  for (const auto& [genome_id, genome] : population->getMap()) genome_index_map->try_emplace(genome_id, indexGenome(*genome));
  return genome_index_map;
}

std::shared_ptr<VariantGenomeIndexMap> VariantSort::variantGenomeIndexMT(const std::shared_ptr<const PopulationDB>& population) {
  WorkflowThreads thread_pool(poolThreads(population->getMap().size()));
  std::vector<std::pair<std::string, std::future<std::shared_ptr<VariantIdIndexMap>>>> future_vector;
  for (const auto& [genome_id, genome] : population->getMap()) {
    std::shared_ptr<const GenomeDB> genome_ptr = genome;
    future_vector.emplace_back(genome_id, thread_pool.enqueueFuture([genome_ptr]() { return indexGenome(*genome_ptr); }));
  }
  auto genome_index_map = std::make_shared<VariantGenomeIndexMap>();
  for (auto& [genome_id, future] : future_vector) genome_index_map->try_emplace(genome_id, future.get());
  return genome_index_map;
}

EnsemblIndexMap SortedVariantAnalysis::filterEnsembl(const std::vector<std::string>& ensembl_list) const {
  EnsemblIndexMap filtered_map;
  for (const auto& ensembl_code : ensembl_list) {
    auto lower_bound = ensembl_index_map_->lower_bound(ensembl_code);
    const auto upper_bound = ensembl_index_map_->upper_bound(ensembl_code);
    for (; lower_bound != upper_bound; ++lower_bound) filtered_map.insert(*lower_bound);
  }
  return filtered_map;
}

const std::shared_ptr<const VariantEnsemblIndexMap>& SortedVariantAnalysis::alleleEnsemblMap() const {
  if (variant_ensembl_index_map_) return variant_ensembl_index_map_;
  VariantEnsemblIndexMap variant_ensembl_map;
  for (const auto& [ensembl_code, variant] : *ensembl_index_map_)
    if (!variant->identifier().empty() && !ensembl_code.empty()) variant_ensembl_map[variant->identifier()].insert(ensembl_code);
  variant_ensembl_index_map_ = std::make_shared<const VariantEnsemblIndexMap>(std::move(variant_ensembl_map));
  return variant_ensembl_index_map_;
}

}  // namespace kgo
