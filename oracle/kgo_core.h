// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of the KGL_Gene population variant-analysis hot path, written to
// follow the reference's data structures and loops as closely as this image's g++ 11 allows.  It is
// the checker for the HIP path and the timed CPU baseline ("port"); it is NOT the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// "Parity unpinned": the reference holds no tests, golden vectors or fixtures for this path
//  and cannot be built here (<format>, std::move_only_function, Boost, nlopt),
// so this restatement is pinned only by the reference's own conservation identities, hand-derived
// closed forms and its synthetic-inbreeding self-check (tests/test_oracle_*.py).
//
// This header: the sparse variant store.
//   Variant                     kgl_genomics/kgl_variant_db/kgl_variant_db.h:46-176
//   OffsetDB / ContigDB         kgl_variant_db_offset.h:24-55, kgl_variant_db_contig.{h,cpp}
//   GenomeDB / PopulationDB     kgl_variant_db_genome.{h,cpp}, kgl_variant_db_population.{h,cpp}
//   WorkflowThreads             kel_thread/kel_workflow_threads.h:27-149
#ifndef KGO_CORE_H
#define KGO_CORE_H

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <string>
#include <thread>
#include <vector>

namespace kgo {

// Super-population slots of FrequencyDatabaseRead (kgl_variant_db_freq.h:53-71).
enum SuperPop : int { AFR = 0, AMR = 1, EAS = 2, EUR = 3, SAS = 4, ALL = 5, SUPER_POP_COUNT = 6 };
const char* superPopName(int sp);
int superPopIndex(const std::string& name);   // -1 if unknown

// kgl_variant_db.h:25-28
enum class VariantPhase : uint8_t { HAPLOID_PHASED = 0, DIPLOID_PHASE_A = 1, DIPLOID_PHASE_B = 2, UNPHASED = 255 };

// The VCF-record payload shared by every Variant cut from that record (VariantEvidence,
// kgl_evidence/kgl_variant_evidence.h:73-152): PASS flag + INFO allele frequencies.  AF values are
// parsed to float32 and widened to double on read (kgl_parser/kgl_variant_factory_vcf_parse_info.h:27-37);
// NaN marks "field present but undefined for this alt" (infoFloatField -> nullopt, kgl_variant_db_freq.cpp:72-122).
struct RecordEvidence {
  uint64_t record_index = 0;
  bool pass = true;
  uint32_t alt_count = 0;
  std::vector<float> af;   // [SUPER_POP_COUNT][alt_count]
  int info_af_size = -2;   // number of values in the raw "AF" INFO vector; -1 = field absent; -2 = same as alt_count
  // Scalar Float INFO fields that hold a value (a missing value is not stored): what
  // InfoEvidenceAnalysis::getTypedInfoData<double> returns (kgl_variant_factory_vcf_evidence_analysis.h:141-166).
  std::vector<std::pair<std::string, float>> info_scalar;
  // Variant::identifier(): the VCF ID column, verbatim (kgl_variant_db.h:125,164; every parser passes vcf_record_ptr->id).
  std::string identifier;
  // The "vep" INFO string vector (',' separated; empty = key absent) and the sub-field names of the
  // ##INFO=<ID=vep,...Description="... Format: a|b|c"> header line (VEPSubFieldHeader, kgl_variant_factory_vcf_evidence.cpp:24-58).
  std::vector<std::string> vep;
  std::shared_ptr<const std::vector<std::string>> vep_header;
  std::optional<double> infoScalar(const std::string& field) const {
    for (const auto& [name, value] : info_scalar)
      if (name == field) return static_cast<double>(value);
    return std::nullopt;
  }
};

class Variant {
 public:
  Variant(std::string contig, uint64_t offset, VariantPhase phase, std::string ref, std::string alt,
          std::shared_ptr<const RecordEvidence> evidence, uint32_t alt_index)
      : contig_(std::move(contig)), offset_(offset), phase_(phase), ref_(std::move(ref)), alt_(std::move(alt)),
        evidence_(std::move(evidence)), alt_index_(alt_index) {}

  const std::string& contigId() const { return contig_; }
  uint64_t offset() const { return offset_; }
  VariantPhase phaseId() const { return phase_; }
  const std::string& reference() const { return ref_; }
  const std::string& alternate() const { return alt_; }
  const RecordEvidence& evidence() const { return *evidence_; }
  const std::shared_ptr<const RecordEvidence>& evidencePtr() const { return evidence_; }
  uint32_t altVariantIndex() const { return alt_index_; }
  uint32_t altVariantCount() const { return evidence_->alt_count; }
  bool passFilter() const { return evidence_->pass; }
  const std::string& identifier() const { return evidence_->identifier; }

  // kgl_variant_db.cpp:287-298.  std::format("{}", uint8_t) prints the integer.
  std::string HGVS() const;
  std::string HGVS_Phase() const;
  // kgl_variant_db.cpp:121-158
  bool isSNP() const;
  // kgl_variant_db.h:135-143
  bool analogous(const Variant& o) const { return HGVS() == o.HGVS(); }
  bool homozygous(const Variant& o) const { return analogous(o) && phaseId() != o.phaseId(); }
  std::shared_ptr<Variant> clonePhase(VariantPhase phase) const;

  // FrequencyDatabaseRead::superPopFrequency (kgl_variant_db_freq.cpp:13-29) on the float32 store.
  std::optional<double> superPopFrequency(int super_pop) const;

 private:
  std::string contig_;
  uint64_t offset_;
  VariantPhase phase_;
  std::string ref_, alt_;
  std::shared_ptr<const RecordEvidence> evidence_;
  uint32_t alt_index_;
};

using VariantPtr = std::shared_ptr<const Variant>;
using OffsetDBArray = std::vector<VariantPtr>;
using VariantFilter = std::function<bool(const Variant&)>;

class OffsetDB {
 public:
  OffsetDB() { variant_vector_.reserve(2); }
  const OffsetDBArray& getVariantArray() const { return variant_vector_; }
  void addVariant(const VariantPtr& v) { variant_vector_.push_back(v); }
  std::unique_ptr<OffsetDB> viewFilter(const VariantFilter& f) const;   // kgl_variant_db_offset.cpp:13-47
 private:
  OffsetDBArray variant_vector_;
};

using OffsetFilter = std::function<std::unique_ptr<OffsetDB>(const OffsetDB&)>;
// kgl_variant_filter/kgl_variant_filter_db_offset.cpp:17-62, 69-105, 137-157
std::unique_ptr<OffsetDB> homozygousFilter(const OffsetDB& offset);
std::unique_ptr<OffsetDB> heterozygousFilter(const OffsetDB& offset);
std::unique_ptr<OffsetDB> uniqueUnphasedFilter(const OffsetDB& offset);
std::unique_ptr<OffsetDB> uniquePhasedFilter(const OffsetDB& offset);      // kgl_variant_filter_db_offset.cpp:160-181
std::unique_ptr<OffsetDB> diploidFilter(const OffsetDB& offset);          // kgl_variant_filter_db_offset.cpp:110-129

class ContigDB {
 public:
  explicit ContigDB(std::string id) : contig_id_(std::move(id)) {}
  const std::string& contigId() const { return contig_id_; }
  const std::map<uint64_t, std::unique_ptr<OffsetDB>>& getMap() const { return contig_offset_map_; }
  bool addVariant(const VariantPtr& v);                                     // kgl_variant_db_contig.cpp:22-57
  size_t variantCount() const;
  std::optional<OffsetDBArray> findOffsetArray(uint64_t offset) const;      // returns a COPY (:216-232)
  std::unique_ptr<ContigDB> viewFilter(const VariantFilter& f) const;       // :122-150, empty offsets trimmed
 private:
  std::string contig_id_;
  std::map<uint64_t, std::unique_ptr<OffsetDB>> contig_offset_map_;
  mutable std::mutex lock_contig_mutex_;
};

class GenomeDB {
 public:
  explicit GenomeDB(std::string id) : genome_id_(std::move(id)) {}
  const std::string& genomeId() const { return genome_id_; }
  const std::map<std::string, std::shared_ptr<ContigDB>>& getMap() const { return contig_map_; }
  bool addVariant(const VariantPtr& v);                                     // kgl_variant_db_genome.cpp:34-53
  std::shared_ptr<ContigDB> getCreateContig(const std::string& contig_id);
  std::optional<std::shared_ptr<const ContigDB>> getContig(const std::string& contig_id) const;
  size_t variantCount() const;
  std::shared_ptr<GenomeDB> viewFilter(const VariantFilter& f) const;
  bool processAll(const std::function<bool(const VariantPtr&)>& f) const;  // :299-315
 private:
  std::string genome_id_;
  std::map<std::string, std::shared_ptr<ContigDB>> contig_map_;
  mutable std::mutex add_variant_mutex_;
};

// kel_thread/kel_workflow_threads.h:27-149: fixed pool, FIFO queue, futures.
class WorkflowThreads {
 public:
  explicit WorkflowThreads(size_t threads);
  ~WorkflowThreads();
  static size_t defaultThreads() { return std::max<size_t>(std::thread::hardware_concurrency() - 1, 1); }
  static size_t defaultThreads(size_t job_size) { return job_size > 0 ? std::min<size_t>(defaultThreads(), job_size) : 1; }
  template <typename F>
  auto enqueueFuture(F&& f) -> std::future<decltype(f())> {
    using R = decltype(f());
    auto task = std::make_shared<std::packaged_task<R()>>(std::forward<F>(f));
    std::future<R> fut = task->get_future();
    {
      std::lock_guard<std::mutex> lk(mutex_);
      queue_.push([task]() { (*task)(); });
    }
    cv_.notify_one();
    return fut;
  }
 private:
  void worker();
  std::vector<std::thread> threads_;
  std::queue<std::function<void()>> queue_;
  std::mutex mutex_;
  std::condition_variable cv_;
  bool stop_ = false;
};

class PopulationDB {
 public:
  explicit PopulationDB(std::string id) : population_id_(std::move(id)) {}
  const std::string& populationId() const { return population_id_; }
  const std::map<std::string, std::shared_ptr<GenomeDB>>& getMap() const { return genome_map_; }
  std::shared_ptr<GenomeDB> getCreateGenome(const std::string& genome_id);
  bool addGenome(const std::shared_ptr<GenomeDB>& genome);
  // kgl_variant_db_population.cpp:298-325
  bool addVariant(const VariantPtr& v, const std::vector<std::string>& genome_vector);
  size_t variantCount() const;                                              // :98-131
  std::map<std::string, VariantPtr> uniqueVariants() const;                 // :133-161 (serial processAll)
  std::unique_ptr<PopulationDB> viewFilter(const VariantFilter& f) const;   // population_filter.cpp:17-74 (MT over genomes)
  bool processAll(const std::function<bool(const VariantPtr&)>& f) const;  // :368-383
  // :386-433: one pool task per genome, min(G, threads) workers; threads==0 -> defaultThreads().
  bool processAll_MT(const std::function<bool(const std::shared_ptr<const GenomeDB>&, const VariantPtr&)>& f,
                     size_t threads = 0) const;
 private:
  std::string population_id_;
  std::map<std::string, std::shared_ptr<GenomeDB>> genome_map_;
  mutable std::mutex add_variant_mutex_;
};

// Global worker-thread override for every pool the oracle creates (0 = the reference's hw-1 default).
void setThreadOverride(size_t threads);
size_t poolThreads(size_t job_size);

}  // namespace kgo

#endif  // KGO_CORE_H
