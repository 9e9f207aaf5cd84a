// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
// Sparse variant store restated from kgl_genomics/kgl_variant_db/*.
#include "kgo_core.h"

#include <atomic>
#include <unordered_set>

namespace kgo {

static const char* kSuperPopNames[SUPER_POP_COUNT] = {"AFR", "AMR", "EAS", "EUR", "SAS", "ALL"};

const char* superPopName(int sp) { return (sp >= 0 && sp < SUPER_POP_COUNT) ? kSuperPopNames[sp] : "?"; }

int superPopIndex(const std::string& name) {
  for (int i = 0; i < SUPER_POP_COUNT; ++i)
    if (name == kSuperPopNames[i]) return i;
  return -1;
}

static std::atomic<size_t> g_thread_override{0};
void setThreadOverride(size_t threads) { g_thread_override = threads; }
size_t poolThreads(size_t job_size) {
  const size_t base = g_thread_override.load() ? g_thread_override.load() : WorkflowThreads::defaultThreads();
  return job_size > 0 ? std::min(base, job_size) : 1;
}

// ---- Variant -----------------------------------------------------------------------------------

std::string Variant::HGVS() const {
  // "{}:g.{}{}>{}"  (kgl_variant_db.cpp:287-291)
  std::string s;
  s.reserve(contig_.size() + ref_.size() + alt_.size() + 26);
  s += contig_;
  s += ":g.";
  s += std::to_string(offset_);
  s += ref_;
  s += '>';
  s += alt_;
  return s;
}

std::string Variant::HGVS_Phase() const {
  // "{}:g.{}{}>{}:{}" with the phase as an integer (kgl_variant_db.cpp:294-298)
  std::string s = HGVS();
  s += ':';
  s += std::to_string(static_cast<unsigned>(phase_));
  return s;
}

bool Variant::isSNP() const {
  if (ref_.size() == 1 && alt_.size() == 1) return true;
  if (ref_.size() != alt_.size()) return false;
  bool diff_found = false;
  for (size_t i = 0; i < ref_.size(); ++i) {
    if (ref_[i] != alt_[i]) {
      if (diff_found) return false;
      diff_found = true;
    }
  }
  return true;
}

std::shared_ptr<Variant> Variant::clonePhase(VariantPhase phase) const {
  return std::make_shared<Variant>(contig_, offset_, phase, ref_, alt_, evidence_, alt_index_);
}

std::optional<double> Variant::superPopFrequency(int super_pop) const {
  // infoFloatField (kgl_variant_db_freq.cpp:72-122): a vector field is indexed by altVariantIndex
  // when its size equals altVariantCount; a missing value is nullopt.
  if (super_pop < 0 || super_pop >= SUPER_POP_COUNT) return 0.0;   // unknown field -> warn, 0.0 (:18-23)
  const RecordEvidence& ev = *evidence_;
  if (ev.af.size() != static_cast<size_t>(SUPER_POP_COUNT) * ev.alt_count) return std::nullopt;
  if (alt_index_ >= ev.alt_count) return std::nullopt;
  const float f = ev.af[static_cast<size_t>(super_pop) * ev.alt_count + alt_index_];
  if (std::isnan(f)) return std::nullopt;
  return static_cast<double>(f);
}

// ---- OffsetDB ----------------------------------------------------------------------------------

std::unique_ptr<OffsetDB> OffsetDB::viewFilter(const VariantFilter& f) const {
  auto out = std::make_unique<OffsetDB>();
  for (const auto& v : variant_vector_)
    if (f(*v)) out->addVariant(v);
  return out;
}

static std::map<std::string, std::vector<VariantPtr>> groupByHGVS(const OffsetDB& offset) {
  std::map<std::string, std::vector<VariantPtr>> variant_map;
  for (const auto& v : offset.getVariantArray()) variant_map[v->HGVS()].push_back(v);
  return variant_map;
}

std::unique_ptr<OffsetDB> homozygousFilter(const OffsetDB& offset) {
  auto out = std::make_unique<OffsetDB>();
  if (offset.getVariantArray().size() != 2) return out;
  for (const auto& [hash, vec] : groupByHGVS(offset))
    if (vec.size() >= 2)
      for (const auto& v : vec) out->addVariant(v);
  return out;
}

std::unique_ptr<OffsetDB> heterozygousFilter(const OffsetDB& offset) {
  auto out = std::make_unique<OffsetDB>();
  for (const auto& [hash, vec] : groupByHGVS(offset))
    if (vec.size() == 1) out->addVariant(vec.front());
  return out;
}

std::unique_ptr<OffsetDB> uniqueUnphasedFilter(const OffsetDB& offset) {
  std::unordered_set<std::string> hashed;
  auto out = std::make_unique<OffsetDB>();
  for (const auto& v : offset.getVariantArray()) {
    auto h = v->HGVS();
    if (!hashed.count(h)) {
      hashed.insert(h);
      out->addVariant(v);
    }
  }
  return out;
}

// kgl_variant_filter_db_offset.cpp:160-181: unique variants including phase
std::unique_ptr<OffsetDB> uniquePhasedFilter(const OffsetDB& offset) {
  std::unordered_set<std::string> hashed;
  auto out = std::make_unique<OffsetDB>();
  for (const auto& v : offset.getVariantArray()) {
    auto h = v->HGVS_Phase();
    if (!hashed.count(h)) {
      hashed.insert(h);
      out->addVariant(v);
    }
  }
  return out;
}

std::unique_ptr<OffsetDB> diploidFilter(const OffsetDB& offset) {
  auto out = std::make_unique<OffsetDB>();
  if (offset.getVariantArray().size() <= 2)
    for (const auto& v : offset.getVariantArray()) out->addVariant(v);
  return out;
}

// ---- ContigDB ----------------------------------------------------------------------------------

bool ContigDB::addVariant(const VariantPtr& v) {
  std::scoped_lock lock(lock_contig_mutex_);
  auto it = contig_offset_map_.find(v->offset());
  if (it != contig_offset_map_.end()) {
    it->second->addVariant(v);
  } else {
    auto offset_ptr = std::make_unique<OffsetDB>();
    offset_ptr->addVariant(v);
    if (!contig_offset_map_.try_emplace(v->offset(), std::move(offset_ptr)).second) return false;
  }
  return true;
}

size_t ContigDB::variantCount() const {
  size_t n = 0;
  for (const auto& [offset, ptr] : contig_offset_map_) n += ptr->getVariantArray().size();
  return n;
}

std::optional<OffsetDBArray> ContigDB::findOffsetArray(uint64_t offset) const {
  auto it = contig_offset_map_.find(offset);
  if (it == contig_offset_map_.end()) return std::nullopt;
  OffsetDBArray copy = it->second->getVariantArray();
  return copy;
}

std::unique_ptr<ContigDB> ContigDB::viewFilter(const VariantFilter& f) const {
  auto out = std::make_unique<ContigDB>(contig_id_);
  for (const auto& [offset, ptr] : contig_offset_map_) {
    auto filtered = ptr->viewFilter(f);
    if (!filtered->getVariantArray().empty())   // addOffset + trimEmpty (kgl_variant_db_contig.cpp:135-148)
      out->contig_offset_map_.try_emplace(offset, std::move(filtered));
  }
  return out;
}

// ---- GenomeDB ----------------------------------------------------------------------------------

std::shared_ptr<ContigDB> GenomeDB::getCreateContig(const std::string& contig_id) {
  std::scoped_lock lock(add_variant_mutex_);
  auto it = contig_map_.find(contig_id);
  if (it != contig_map_.end()) return it->second;
  auto ptr = std::make_shared<ContigDB>(contig_id);
  contig_map_.insert({contig_id, ptr});
  return ptr;
}

std::optional<std::shared_ptr<const ContigDB>> GenomeDB::getContig(const std::string& contig_id) const {
  auto it = contig_map_.find(contig_id);
  if (it == contig_map_.end()) return std::nullopt;
  return std::shared_ptr<const ContigDB>(it->second);
}

bool GenomeDB::addVariant(const VariantPtr& v) { return getCreateContig(v->contigId())->addVariant(v); }

size_t GenomeDB::variantCount() const {
  size_t n = 0;
  for (const auto& [id, c] : contig_map_) n += c->variantCount();
  return n;
}

std::shared_ptr<GenomeDB> GenomeDB::viewFilter(const VariantFilter& f) const {
  auto out = std::make_shared<GenomeDB>(genome_id_);
  for (const auto& [id, c] : contig_map_) out->contig_map_.insert({id, std::shared_ptr<ContigDB>(c->viewFilter(f))});
  return out;
}

bool GenomeDB::processAll(const std::function<bool(const VariantPtr&)>& f) const {
  for (const auto& [contig_id, contig] : contig_map_)
    for (const auto& [offset, offset_ptr] : contig->getMap())
      for (const auto& v : offset_ptr->getVariantArray())
        if (!f(v)) return false;
  return true;
}

// ---- WorkflowThreads ---------------------------------------------------------------------------

WorkflowThreads::WorkflowThreads(size_t threads) {
  threads = std::max<size_t>(threads, 1);
  for (size_t i = 0; i < threads; ++i) threads_.emplace_back(&WorkflowThreads::worker, this);
}

WorkflowThreads::~WorkflowThreads() {
  {
    std::lock_guard<std::mutex> lk(mutex_);
    stop_ = true;
  }
  cv_.notify_all();
  for (auto& t : threads_) t.join();
}

void WorkflowThreads::worker() {
  while (true) {
    std::function<void()> job;
    {
      std::unique_lock<std::mutex> lk(mutex_);
      cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) return;   // stop_ and drained
      job = std::move(queue_.front());
      queue_.pop();
    }
    job();
  }
}

// ---- PopulationDB ------------------------------------------------------------------------------

std::shared_ptr<GenomeDB> PopulationDB::getCreateGenome(const std::string& genome_id) {
  std::scoped_lock lock(add_variant_mutex_);
  auto it = genome_map_.find(genome_id);
  if (it != genome_map_.end()) return it->second;
  auto ptr = std::make_shared<GenomeDB>(genome_id);
  genome_map_.insert({genome_id, ptr});
  return ptr;
}

bool PopulationDB::addGenome(const std::shared_ptr<GenomeDB>& genome) {
  std::scoped_lock lock(add_variant_mutex_);
  return genome_map_.try_emplace(genome->genomeId(), genome).second;
}

bool PopulationDB::addVariant(const VariantPtr& v, const std::vector<std::string>& genome_vector) {
  bool result = true;
  for (const auto& genome : genome_vector)
    if (!getCreateGenome(genome)->addVariant(v)) result = false;
  return result;
}

size_t PopulationDB::variantCount() const {
  if (genome_map_.empty()) return 0;
  WorkflowThreads pool(poolThreads(genome_map_.size()));
  std::vector<std::future<size_t>> futures;
  for (const auto& [id, g] : genome_map_) {
    std::shared_ptr<const GenomeDB> gp = g;
    futures.push_back(pool.enqueueFuture([gp]() { return gp->variantCount(); }));
  }
  size_t n = 0;
  for (auto& f : futures) n += f.get();
  return n;
}

std::map<std::string, VariantPtr> PopulationDB::uniqueVariants() const {
  std::map<std::string, VariantPtr> unique_map;
  processAll([&](const VariantPtr& v) {
    auto hgvs = v->HGVS();
    if (!unique_map.count(hgvs)) unique_map[hgvs] = v;
    return true;
  });
  return unique_map;
}

std::unique_ptr<PopulationDB> PopulationDB::viewFilter(const VariantFilter& f) const {
  auto out = std::make_unique<PopulationDB>(population_id_);
  if (genome_map_.empty()) return out;
  WorkflowThreads pool(poolThreads(genome_map_.size()));
  std::vector<std::future<std::shared_ptr<GenomeDB>>> futures;
  for (const auto& [id, g] : genome_map_) {
    std::shared_ptr<const GenomeDB> gp = g;
    futures.push_back(pool.enqueueFuture([gp, &f]() { return gp->viewFilter(f); }));
  }
  for (auto& fut : futures) out->addGenome(fut.get());
  return out;
}

bool PopulationDB::processAll(const std::function<bool(const VariantPtr&)>& f) const {
  for (const auto& [id, g] : genome_map_)
    if (!g->processAll(f)) return false;
  return true;
}

bool PopulationDB::processAll_MT(
    const std::function<bool(const std::shared_ptr<const GenomeDB>&, const VariantPtr&)>& f, size_t threads) const {
  const size_t n = threads ? std::min(threads, std::max<size_t>(genome_map_.size(), 1)) : poolThreads(genome_map_.size());
  WorkflowThreads pool(n);
  std::vector<std::future<bool>> futures;
  for (const auto& [id, g] : genome_map_) {
    std::shared_ptr<const GenomeDB> gp = g;
    futures.push_back(pool.enqueueFuture([gp, &f]() {
      return gp->processAll([&](const VariantPtr& v) { return f(gp, v); });
    }));
  }
  bool ok = true;
  for (auto& fut : futures) ok = fut.get() && ok;
  return ok;
}

}  // namespace kgo
