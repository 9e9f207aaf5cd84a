// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
//
// VariantSort and SortedVariantAnalysis restated on the oracle's PopulationDB:
//   VariantSort::{ensemblIndex, ensemblAddIndex, nonEnsemblIdentifiers, variantIdIndex, variantGenomeIndex(MT)}
//                                       kgl_genomics/kgl_variant_analysis/kgl_variant_sort.{h,cpp}
//   SortedVariantAnalysis::{filterEnsembl, alleleEnsemblMap}
//                                       kgl_genomics/kgl_variant_analysis/kgl_variant_sort_analysis.{h,cpp}
//   InfoEvidenceAnalysis::{getVepSubFields, getVepIndexes, getVepData}
//                                       kgl_genomics/kgl_evidence/kgl_variant_factory_vcf_evidence_vep.cpp:25-236
#ifndef KGO_SORT_H
#define KGO_SORT_H

#include <set>

#include "kgo_core.h"

namespace kgo {

using EnsemblIndexMap = std::multimap<std::string, VariantPtr>;                       // kgl_variant_sort.h:27
using VariantIdIndexMap = std::map<std::string, VariantPtr>;                          // :30
using VariantGenomeIndexMap = std::map<std::string, std::shared_ptr<VariantIdIndexMap>>;   // :33
using VariantEnsemblIndexMap = std::map<std::string, std::set<std::string>>;          // :36

// The vep entries of a variant that hold exactly as many '|' sub-fields as the header names; nullopt when there is
// no header, no vep data, or no entry of the right size (getVepSubFields, _vep.cpp:25-124).
std::optional<std::vector<std::string>> vepCheckedFields(const Variant& variant);

class VariantSort {
 public:
  static std::shared_ptr<EnsemblIndexMap> ensemblIndex(const std::shared_ptr<const PopulationDB>& population);
  static void ensemblAddIndex(const std::shared_ptr<const PopulationDB>& population, const std::vector<std::string>& ensembl_gene_list,
                              std::shared_ptr<EnsemblIndexMap>& index_map);
  static size_t nonEnsemblIdentifiers(const EnsemblIndexMap& index_map);
  static std::shared_ptr<VariantIdIndexMap> variantIdIndex(const std::shared_ptr<const PopulationDB>& population);
  static std::shared_ptr<VariantGenomeIndexMap> variantGenomeIndex(const std::shared_ptr<const PopulationDB>& population);
  static std::shared_ptr<VariantGenomeIndexMap> variantGenomeIndexMT(const std::shared_ptr<const PopulationDB>& population);
};

class SortedVariantAnalysis {
 public:
  explicit SortedVariantAnalysis(const std::shared_ptr<const PopulationDB>& population) : ensembl_index_map_(VariantSort::ensemblIndex(population)) {}
  const std::shared_ptr<const EnsemblIndexMap>& ensemblMap() const { return ensembl_index_map_; }
  EnsemblIndexMap filterEnsembl(const std::vector<std::string>& ensembl_list) const;
  const std::shared_ptr<const VariantEnsemblIndexMap>& alleleEnsemblMap() const;
 private:
  const std::shared_ptr<const EnsemblIndexMap> ensembl_index_map_;
  mutable std::shared_ptr<const VariantEnsemblIndexMap> variant_ensembl_index_map_;
};

}  // namespace kgo

#endif  // KGO_SORT_H
