// The rsid / Ensembl indexes as sorted columns (kgx_variant_sort.h).  Every index is built by walking
// SortColumns::visits -- the Variant objects in PopulationDB::processAll order -- once, tagging each entry with its key's
// rank and its visit number, and sorting; "the first visit keeps the key" (std::map::emplace) and "equal keys stay in
// insertion order" (std::multimap::emplace) both fall out of the (rank, visit) order.
#include "kgx_variant_sort.h"

#include <algorithm>
#include <atomic>
#include <thread>

namespace kellerberrin::genome::analysis::gpu {

std::string SortColumns::hgvs(const SortVariant& v) const {
  const SortRecord& rec = records[v.record];
  return rec.contig + ":g." + std::to_string(rec.offset) + rec.ref + ">" + rec.alts[v.alt];
}

std::string SortColumns::hgvsPhase(const SortVariant& v) const { return hgvs(v) + ":" + std::to_string(static_cast<unsigned>(v.phase)); }

namespace {

// The distinct values of one string attribute over the records, ascending, and each record's ranks into them.
std::vector<std::string> distinctSorted(std::vector<std::string> values) {
  std::sort(values.begin(), values.end());
  values.erase(std::unique(values.begin(), values.end()), values.end());
  return values;
}

uint32_t rankOf(const std::vector<std::string>& sorted, const std::string& key) {
  return static_cast<uint32_t>(std::lower_bound(sorted.begin(), sorted.end(), key) - sorted.begin());
}

}  // namespace

// ---- Ensembl ---------------------------------------------------------------------------------------------------------

EnsemblIndex VariantSortIndex::ensemblIndex(const SortColumns& columns, const std::vector<std::string>& ensembl_gene_list) {
  EnsemblIndex index;
  index.begin_.assign(1, 0);
  if (columns.visits.empty() || !columns.records[columns.visits.front().record].vep_usable) return index;
  const std::vector<std::string> wanted = distinctSorted(ensembl_gene_list);
  // the codes that will be keys, then a counting sort of the visits by code (stable = insertion order kept)
  std::vector<std::string> codes;
  for (const auto& rec : columns.records)
    for (const auto& gene : rec.genes)
      if (wanted.empty() || std::binary_search(wanted.begin(), wanted.end(), gene)) codes.push_back(gene);
  index.genes_ = distinctSorted(std::move(codes));
  std::vector<std::vector<uint32_t>> record_codes(columns.records.size());
  for (size_t r = 0; r < columns.records.size(); ++r)
    for (const auto& gene : columns.records[r].genes) {
      const uint32_t k = rankOf(index.genes_, gene);
      if (k < index.genes_.size() && index.genes_[k] == gene) record_codes[r].push_back(k);
    }
  index.begin_.assign(index.genes_.size() + 1, 0);
  for (const auto& v : columns.visits) for (const uint32_t k : record_codes[v.record]) ++index.begin_[k + 1];
  for (size_t k = 0; k < index.genes_.size(); ++k) index.begin_[k + 1] += index.begin_[k];
  index.variants_.resize(index.begin_.back());
  std::vector<uint64_t> cursor(index.begin_.begin(), index.begin_.end() - 1);
  for (const auto& v : columns.visits) for (const uint32_t k : record_codes[v.record]) index.variants_[cursor[k]++] = v;
  // unused codes (a listed gene no visited record bears) cannot occur: codes come from records, and a record is in
  // columns.records only if the file holds it; a record no genome carries adds a key with an empty range -- drop those
  std::vector<std::string> kept_genes;
  std::vector<uint64_t> kept_begin{0};
  for (size_t k = 0; k < index.genes_.size(); ++k)
    if (index.begin_[k + 1] > index.begin_[k]) { kept_genes.push_back(std::move(index.genes_[k])); kept_begin.push_back(index.begin_[k + 1]); }
  index.genes_ = std::move(kept_genes);
  index.begin_ = std::move(kept_begin);
  return index;
}

std::pair<size_t, size_t> EnsemblIndex::equalRange(const std::string& gene) const {
  const uint32_t k = rankOf(genes_, gene);
  if (k >= genes_.size() || genes_[k] != gene) return {0, 0};
  return {begin_[k], begin_[k + 1]};
}

const std::string& EnsemblIndex::geneOf(size_t entry) const {
  const size_t k = static_cast<size_t>(std::upper_bound(begin_.begin(), begin_.end(), static_cast<uint64_t>(entry)) - begin_.begin()) - 1;
  return genes_[k];
}

size_t EnsemblIndex::nonEnsemblIdentifiers() const {
  size_t count = 0;
  for (size_t k = 0; k < genes_.size(); ++k)
    if (genes_[k].find("ENSG") == std::string::npos) count += begin_[k + 1] - begin_[k];
  return count;
}

EnsemblIndex EnsemblIndex::filterEnsembl(const std::vector<std::string>& ensembl_list) const {
  // a multimap insert per listed code and entry: a code listed n times holds n copies of its range, copy after copy
  std::map<std::string, size_t> times;
  for (const auto& code : ensembl_list) ++times[code];
  EnsemblIndex out;
  out.begin_.assign(1, 0);
  for (const auto& [code, n] : times) {
    const auto [first, last] = equalRange(code);
    if (first == last) continue;
    out.genes_.push_back(code);
    for (size_t copy = 0; copy < n; ++copy) out.variants_.insert(out.variants_.end(), variants_.begin() + first, variants_.begin() + last);
    out.begin_.push_back(out.variants_.size());
  }
  return out;
}

std::map<std::string, std::set<std::string>> EnsemblIndex::alleleEnsemblMap(const SortColumns& columns) const {
  std::map<std::string, std::set<std::string>> out;
  for (size_t k = 0; k < genes_.size(); ++k) {
    if (genes_[k].empty()) continue;
    for (uint64_t e = begin_[k]; e < begin_[k + 1]; ++e) {
      const std::string& id = columns.records[variants_[e].record].identifier;
      if (!id.empty()) out[id].insert(genes_[k]);
    }
  }
  return out;
}

// ---- identifiers -----------------------------------------------------------------------------------------------------

IdIndex VariantSortIndex::variantIdIndex(const SortColumns& columns) {
  IdIndex index;
  std::vector<std::string> ids;
  for (const auto& rec : columns.records) if (!rec.identifier.empty()) ids.push_back(rec.identifier);
  const std::vector<std::string> all_ids = distinctSorted(std::move(ids));
  std::vector<uint32_t> record_rank(columns.records.size(), UINT32_MAX);
  for (size_t r = 0; r < columns.records.size(); ++r)
    if (!columns.records[r].identifier.empty()) record_rank[r] = rankOf(all_ids, columns.records[r].identifier);
  constexpr uint64_t kNone = UINT64_MAX;
  std::vector<uint64_t> first_visit(all_ids.size(), kNone);
  for (uint64_t i = 0; i < columns.visits.size(); ++i) {
    const uint32_t k = record_rank[columns.visits[i].record];
    if (k != UINT32_MAX && first_visit[k] == kNone) first_visit[k] = i;
  }
  for (size_t k = 0; k < all_ids.size(); ++k)
    if (first_visit[k] != kNone) { index.ids_.push_back(all_ids[k]); index.variants_.push_back(columns.visits[first_visit[k]]); }
  return index;
}

const SortVariant* IdIndex::find(const std::string& id) const {
  const uint32_t k = rankOf(ids_, id);
  return (k < ids_.size() && ids_[k] == id) ? &variants_[k] : nullptr;
}

GenomeIdIndex VariantSortIndex::variantGenomeIndex(const SortColumns& columns, size_t threads) {
  GenomeIdIndex index;
  std::vector<std::string> ids;
  for (const auto& rec : columns.records) if (!rec.identifier.empty()) ids.push_back(rec.identifier);
  index.ids_ = distinctSorted(std::move(ids));
  std::vector<uint32_t> record_rank(columns.records.size(), UINT32_MAX);
  for (size_t r = 0; r < columns.records.size(); ++r)
    if (!columns.records[r].identifier.empty()) record_rank[r] = rankOf(index.ids_, columns.records[r].identifier);
  const size_t G = columns.genome_ids.size();
  // per genome: its visits tagged (id rank, visit), sorted, first of each rank kept
  std::vector<std::vector<std::pair<uint32_t, SortVariant>>> per_genome(G);
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    std::vector<std::pair<uint64_t, uint64_t>> tagged;      // (rank, visit)
    for (size_t g = next.fetch_add(1); g < G; g = next.fetch_add(1)) {
      tagged.clear();
      for (uint64_t i = columns.genome_begin[g]; i < columns.genome_begin[g + 1]; ++i) {
        const uint32_t k = record_rank[columns.visits[i].record];
        if (k != UINT32_MAX) tagged.emplace_back(k, i);
      }
      std::sort(tagged.begin(), tagged.end());
      auto& kept = per_genome[g];
      for (size_t t = 0; t < tagged.size(); ++t)
        if (t == 0 || tagged[t].first != tagged[t - 1].first) kept.emplace_back(static_cast<uint32_t>(tagged[t].first), columns.visits[tagged[t].second]);
    }
  };
  const size_t workers = std::max<size_t>(1, std::min(threads, G));
  std::vector<std::thread> pool;
  for (size_t t = 1; t < workers; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  index.genome_begin_.assign(G + 1, 0);
  for (size_t g = 0; g < G; ++g) index.genome_begin_[g + 1] = index.genome_begin_[g] + per_genome[g].size();
  index.id_rank_.reserve(index.genome_begin_[G]);
  index.variants_.reserve(index.genome_begin_[G]);
  for (size_t g = 0; g < G; ++g)
    for (const auto& [k, v] : per_genome[g]) { index.id_rank_.push_back(k); index.variants_.push_back(v); }
  return index;
}

const SortVariant* GenomeIdIndex::find(size_t genome, const std::string& id) const {
  const uint32_t k = rankOf(ids_, id);
  if (k >= ids_.size() || ids_[k] != id) return nullptr;
  const auto first = id_rank_.begin() + static_cast<ptrdiff_t>(genome_begin_[genome]);
  const auto last = id_rank_.begin() + static_cast<ptrdiff_t>(genome_begin_[genome + 1]);
  const auto at = std::lower_bound(first, last, k);
  return (at != last && *at == k) ? &variants_[static_cast<size_t>(at - id_rank_.begin())] : nullptr;
}

}  // namespace kellerberrin::genome::analysis::gpu
