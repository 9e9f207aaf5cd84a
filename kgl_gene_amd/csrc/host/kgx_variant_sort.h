// VariantSort / SortedVariantAnalysis on columns: the rsid and Ensembl-gene indexes of
//   kgl_genomics/kgl_variant_analysis/kgl_variant_sort.{h,cpp}            (VariantSort)
//   kgl_genomics/kgl_variant_analysis/kgl_variant_sort_analysis.{h,cpp}   (SortedVariantAnalysis)
// without Variant objects or node-per-entry maps.  The reference builds a std::multimap<gene, shared_ptr<Variant>>, a
// std::map<id, shared_ptr<Variant>> and one such map per genome by visiting every Variant of every genome
// (PopulationDB::processAll); here a VCF is cut once into record columns plus the carriers of every record, and each
// index is a sorted key column with the variants beside it, searched by bisection.  Host C++ only: these indexes feed
// the MUTATION / LITERATURE packages, not the counting sweeps, and hold strings.
#ifndef KGX_VARIANT_SORT_H
#define KGX_VARIANT_SORT_H

#include <map>
#include <set>

#include "kgx_flatten.h"

namespace kellerberrin::genome::analysis::gpu {

// One VCF record: what every Variant cut from it shares.
struct SortRecord {
  ContigId_t contig;
  ContigOffset_t offset{0};
  std::string identifier;                  // Variant::identifier(): the ID column, "" for "." (kgl_variant_vcf_impl.cpp:119-130)
  std::string ref;
  std::vector<std::string> alts;
  // Distinct non-empty "Gene" values of the record's usable vep entries, ascending (the std::set of
  // kgl_variant_sort.cpp:64-83); usable = as many '|' sub-fields as the ##INFO=<ID=vep ... Format: ...> header names.
  std::vector<std::string> genes;
  bool vep_usable{false};                  // at least one usable entry (and a header naming "Gene")
};

// A Variant object of the population: record + alt + phase (VariantPhase's integer: 1 = A, 2 = B, 255 = unphased).
struct SortVariant {
  uint32_t record{0};
  uint16_t alt{0};
  uint8_t phase{255};
  bool operator==(const SortVariant& o) const { return record == o.record && alt == o.alt && phase == o.phase; }
};

struct SortColumns {
  std::vector<GenomeId_t> genome_ids;      // genomes holding a variant, id order (the PopulationDB map's order)
  std::vector<SortRecord> records;         // in a genome's visiting order: contig id, offset, then file order
  // Variant objects per genome in PopulationDB::processAll order: genome g owns visits[genome_begin[g] .. genome_begin[g+1]).
  std::vector<uint64_t> genome_begin;
  std::vector<SortVariant> visits;
  std::vector<std::string> vep_header;     // sub-field names; empty = no usable header

  [[nodiscard]] std::string hgvs(const SortVariant& v) const;        // Variant::HGVS()
  [[nodiscard]] std::string hgvsPhase(const SortVariant& v) const;   // Variant::HGVS_Phase()
};

enum class SortVcfFlavour {
  MonoGenome,     // GrchVCFImpl: no sample columns read, every alt one UNPHASED Variant of the one genome
  Phased1000      // Genome1000VCFImpl: phase A / phase B alleles of every sample column
};
// genome_id names the one genome of the MonoGenome flavour (ignored otherwise).  Implemented in kgx_vcf_flatten.cpp.
[[nodiscard]] SortColumns sortColumnsFromVcf(std::string_view text, SortVcfFlavour flavour, const GenomeId_t& genome_id = "Reference",
                                             size_t threads = 0);

// The same from a file (plain / gzip / block gzip) read a bounded piece at a time (VcfChunkReader): only the record
// columns and the carriers stay between pieces.  false + error on an I/O or format error.
[[nodiscard]] bool sortColumnsFromVcfFile(const std::string& file_name, SortVcfFlavour flavour, SortColumns& columns, std::string& error,
                                          const GenomeId_t& genome_id = "Reference", size_t threads = 0, size_t chunk_bytes = size_t{64} << 20);

// EnsemblIndexMap (kgl_variant_sort.h:27): gene code -> Variants, equal codes in visiting order.
class EnsemblIndex {
 public:
  [[nodiscard]] size_t size() const { return variants_.size(); }
  [[nodiscard]] const std::vector<std::string>& genes() const { return genes_; }          // ascending, distinct
  // The Variants of one code: [first, last) into variants(); an unknown code gives an empty range.
  [[nodiscard]] std::pair<size_t, size_t> equalRange(const std::string& gene) const;
  [[nodiscard]] const std::vector<SortVariant>& variants() const { return variants_; }
  [[nodiscard]] const std::string& geneOf(size_t entry) const;
  // VariantSort::nonEnsemblIdentifiers (kgl_variant_sort.cpp:116-132): ENTRIES whose code does not contain "ENSG".
  [[nodiscard]] size_t nonEnsemblIdentifiers() const;
  // SortedVariantAnalysis::filterEnsembl (kgl_variant_sort_analysis.cpp:12-34); a code listed twice doubles its entries.
  [[nodiscard]] EnsemblIndex filterEnsembl(const std::vector<std::string>& ensembl_list) const;
  // SortedVariantAnalysis::alleleEnsemblMap (:38-72): variant identifier -> the codes it is indexed under.
  [[nodiscard]] std::map<std::string, std::set<std::string>> alleleEnsemblMap(const SortColumns& columns) const;

 private:
  friend class VariantSortIndex;
  std::vector<std::string> genes_;
  std::vector<uint64_t> begin_;            // genes_.size() + 1
  std::vector<SortVariant> variants_;
};

// VariantIdIndexMap (kgl_variant_sort.h:30): identifier -> the first Variant visited that bears it.
class IdIndex {
 public:
  [[nodiscard]] size_t size() const { return ids_.size(); }
  [[nodiscard]] const std::vector<std::string>& ids() const { return ids_; }              // ascending, distinct
  [[nodiscard]] const std::vector<SortVariant>& variants() const { return variants_; }
  [[nodiscard]] const SortVariant* find(const std::string& id) const;

 private:
  friend class VariantSortIndex;
  std::vector<std::string> ids_;
  std::vector<SortVariant> variants_;
};

// VariantGenomeIndexMap (kgl_variant_sort.h:33): per genome, identifier -> its first Variant bearing it.
class GenomeIdIndex {
 public:
  [[nodiscard]] const std::vector<std::string>& ids() const { return ids_; }              // every identifier of the population, ascending
  [[nodiscard]] size_t genomes() const { return genome_begin_.empty() ? 0 : genome_begin_.size() - 1; }
  [[nodiscard]] size_t size(size_t genome) const { return genome_begin_[genome + 1] - genome_begin_[genome]; }
  // Entry e of a genome, in identifier order.
  [[nodiscard]] const std::string& id(size_t genome, size_t e) const { return ids_[id_rank_[genome_begin_[genome] + e]]; }
  [[nodiscard]] const SortVariant& variant(size_t genome, size_t e) const { return variants_[genome_begin_[genome] + e]; }
  [[nodiscard]] const SortVariant* find(size_t genome, const std::string& id) const;

 private:
  friend class VariantSortIndex;
  std::vector<std::string> ids_;
  std::vector<uint64_t> genome_begin_;
  std::vector<uint32_t> id_rank_;          // ascending within a genome
  std::vector<SortVariant> variants_;
};

class VariantSortIndex {
 public:
  VariantSortIndex() = delete;
  // VariantSort::ensemblIndex / ensemblAddIndex (kgl_variant_sort.cpp:16-111); an empty gene list keeps every code.
  // As there, the "Gene" column is looked up on the first Variant visited: if that one has no usable vep entry,
  // nothing is indexed.
  [[nodiscard]] static EnsemblIndex ensemblIndex(const SortColumns& columns, const std::vector<std::string>& ensembl_gene_list = {});
  // VariantSort::variantIdIndex (:136-172)
  [[nodiscard]] static IdIndex variantIdIndex(const SortColumns& columns);
  // VariantSort::variantGenomeIndex / variantGenomeIndexMT (:176-306): one task per genome there, one slice per thread here.
  [[nodiscard]] static GenomeIdIndex variantGenomeIndex(const SortColumns& columns, size_t threads = 0);
};

}  // namespace kellerberrin::genome::analysis::gpu

#endif  // KGX_VARIANT_SORT_H
