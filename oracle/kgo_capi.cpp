// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
// extern "C" surface so tests/ (ctypes) can drive the restatement.  Nothing in the product links this.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <set>
#include <fstream>
#include <thread>

#include "kgo_analysis.h"
#include "kgo_inbreed.h"
#include "kgo_pf7.h"
#include "kgo_sort.h"

using namespace kgo;

struct kgo_pop {
  std::shared_ptr<PopulationDB> pop;
  std::vector<std::string> input_ids;   // genome ids in the caller's order
  uint64_t next_record = 0;
};

struct kgo_vdb {
  std::shared_ptr<const PopulationDB> pop;
  std::unique_ptr<VariantDBVariant> vdb;
};

struct kgo_columns {
  std::vector<std::pair<std::string, ResultsMap>> columns;
  std::vector<std::string> genome_order;
};

static void copyString(const std::string& s, char* buf, size_t len) {
  if (!buf || !len) return;
  const size_t n = std::min(len - 1, s.size());
  std::memcpy(buf, s.data(), n);
  buf[n] = 0;
}

static void fillResults(const LocusResults& r, uint64_t* counts, double* freqs) {
  counts[0] = r.major_hetero_count;
  counts[1] = r.minor_hetero_count;
  counts[2] = r.minor_homo_count;
  counts[3] = r.major_homo_count;
  counts[4] = r.total_allele_count;
  freqs[0] = r.major_hetero_freq;
  freqs[1] = r.minor_hetero_freq;
  freqs[2] = r.minor_homo_freq;
  freqs[3] = r.major_homo_freq;
  freqs[4] = r.inbred_allele_sum;
}

namespace kgo {
std::pair<size_t, size_t> alternateIndex1000(const std::string& contig, const std::string& genotype, size_t n_alt);
long addVcf1000(PopulationDB& population, std::string_view text, std::vector<std::string>* genome_names_out);
long addVcfPf(PopulationDB& population, std::string_view text, std::vector<std::string>* genome_names_out);
long addVcfMonoGenome(PopulationDB& population, std::string_view text, const std::string& source, const std::string& genome_id);
bool p7VariantFilter(const Variant& v);
void canonicalSequences(const std::string& ref, const std::string& alt, uint64_t offset, std::string& c_ref, std::string& c_alt, uint64_t& c_offset);
}

extern "C" {

const char* kgo_banner(void) { return "kgo oracle: CPU restatement of the KGL_Gene hot path; TEST INFRASTRUCTURE; parity unpinned"; }

void kgo_set_threads(int n) { setThreadOverride(n > 0 ? static_cast<size_t>(n) : 0); }
int kgo_default_threads(void) { return static_cast<int>(WorkflowThreads::defaultThreads()); }
int kgo_pool_threads(uint64_t jobs) { return static_cast<int>(poolThreads(jobs)); }

kgo_pop* kgo_population_create(const char* id) {
  auto* p = new kgo_pop();
  p->pop = std::make_shared<PopulationDB>(id ? id : "population");
  return p;
}

void kgo_population_destroy(kgo_pop* p) { delete p; }

// PfVCFImpl::setupPopulationStructure (kgl_variant_factory_pf_impl.cpp:399-425): genomes exist even if
// they carry no variant.  Also fixes the caller's genome order for index-mapped outputs.
int kgo_population_add_genomes(kgo_pop* p, uint64_t n, const char* const* ids, int precreate) {
  if (!p || !ids) return -1;
  for (uint64_t i = 0; i < n; ++i) {
    p->input_ids.emplace_back(ids[i]);
    if (precreate) p->pop->getCreateGenome(ids[i]);
  }
  return 0;
}

// Expand VCF-like records into Variant objects the way the reference's parsers do.
//  mode 0: Genome1000VCFImpl::ParseRecord (kgl_variant_factory_1000_impl.cpp:63-145,274-318): per record one
//          shared Variant per (alt, phase); all phase-A variants are added before phase-B ones.
//  mode 1: PfVCFImpl (kgl_variant_factory_pf_impl.cpp:287-384,427-455): per genome, per allele copy, a fresh
//          UNPHASED Variant; A allele before B allele.
//  mode 2: reference mono-genome (gnomAD-style, no genotypes): every alt added once, UNPHASED, to genome_ids[0].
// gt: [n_records][n_genomes][2] allele indices (0 = reference); ignored for mode 2.
// af_flat: [sum(n_alts)][6] float32 (NaN = missing) or NULL (no AF INFO at all).
int kgo_population_add_records(kgo_pop* p, int mode, const char* contig, uint64_t n_records, const uint64_t* offsets,
                               const char* const* refs, const uint8_t* n_alts, const char* const* alts_flat,
                               const uint8_t* pass, const float* af_flat, uint64_t n_genomes,
                               const char* const* genome_ids, const uint8_t* gt) {
  if (!p || !contig || !offsets || !refs || !n_alts || !alts_flat) return -1;
  uint64_t alt_cursor = 0;
  const std::string contig_id(contig);
  for (uint64_t r = 0; r < n_records; ++r) {
    const uint32_t A = n_alts[r];
    auto ev = std::make_shared<RecordEvidence>();
    ev->record_index = p->next_record++;
    ev->pass = pass ? pass[r] != 0 : true;
    ev->alt_count = A;
    if (af_flat) {
      ev->af.resize(static_cast<size_t>(SUPER_POP_COUNT) * A);
      for (uint32_t a = 0; a < A; ++a)
        for (int sp = 0; sp < SUPER_POP_COUNT; ++sp)
          ev->af[static_cast<size_t>(sp) * A + a] = af_flat[(alt_cursor + a) * SUPER_POP_COUNT + sp];
    }
    const std::string ref(refs[r]);
    if (mode == 2) {
      if (n_genomes < 1) return -1;
      std::vector<std::string> gv{genome_ids[0]};
      for (uint32_t a = 0; a < A; ++a)
        p->pop->addVariant(std::make_shared<const Variant>(contig_id, offsets[r], VariantPhase::UNPHASED, ref,
                                                          alts_flat[alt_cursor + a], ev, a), gv);
    } else if (mode == 0) {
      for (int phase = 0; phase < 2; ++phase) {
        std::map<size_t, std::vector<std::string>> phase_map;
        for (uint64_t g = 0; g < n_genomes; ++g) {
          const uint32_t idx = gt[(r * n_genomes + g) * 2 + phase];
          if (idx != 0 && idx <= A) phase_map[idx - 1].push_back(genome_ids[g]);
        }
        for (const auto& [alt_allele, genome_vector] : phase_map) {
          auto v = std::make_shared<const Variant>(contig_id, offsets[r],
                                                   phase == 0 ? VariantPhase::DIPLOID_PHASE_A : VariantPhase::DIPLOID_PHASE_B,
                                                   ref, alts_flat[alt_cursor + alt_allele], ev, static_cast<uint32_t>(alt_allele));
          p->pop->addVariant(v, genome_vector);
        }
      }
    } else if (mode == 1) {
      for (uint64_t g = 0; g < n_genomes; ++g) {
        std::vector<std::string> gv{genome_ids[g]};
        for (int copy = 0; copy < 2; ++copy) {
          const uint32_t idx = gt[(r * n_genomes + g) * 2 + copy];
          if (idx != 0 && idx <= A)
            p->pop->addVariant(std::make_shared<const Variant>(contig_id, offsets[r], VariantPhase::UNPHASED, ref,
                                                              alts_flat[alt_cursor + idx - 1], ev, idx - 1), gv);
        }
      }
    } else {
      return -1;
    }
    alt_cursor += A;
  }
  return 0;
}

// kgo_population_add_records for large synthetic blocks, with the sequences given as codes instead of C strings
// (ten million Python strings are the slow part of a 5M-locus test, not the oracle): ref_code[r] in 0..3 = "ACGT";
// alt_code (flat, per alt) 0..3 = that base (a SNP), 0x80 | k = an insertion: the reference base followed by k + 1
// bases 'G' (k even) or 'A' (k odd).  Everything else as kgo_population_add_records -- the same Variant objects in the
// same per-genome order -- except that for mode 0 (the phased 1000-Genomes expansion) the genomes are filled by a pool
// of threads, each owning a block of genomes: the store a parser hands over is the INPUT of the path, and filling
// 10^8 map nodes from one thread would dominate every large test and the bench's CPU leg without being timed by either.
int kgo_population_add_records_coded(kgo_pop* p, int mode, const char* contig, uint64_t n_records, const uint64_t* offsets,
                                     const uint8_t* ref_code, const uint8_t* n_alts, const uint8_t* alt_code,
                                     const uint8_t* pass, const float* af_flat, uint64_t n_genomes,
                                     const char* const* genome_ids, const uint8_t* gt) {
  if (!p || !contig || !offsets || !ref_code || !n_alts || !alt_code) return -1;
  static const char bases[] = "ACGT";
  auto sequences = [&](uint64_t r, uint64_t alt_cursor, std::string& ref, std::vector<std::string>& alts) {
    ref.assign(1, bases[ref_code[r] & 3u]);
    alts.clear();
    for (uint32_t a = 0; a < n_alts[r]; ++a) {
      const uint8_t code = alt_code[alt_cursor + a];
      if (code & 0x80u) alts.push_back(ref + std::string((code & 0x7Fu) + 1u, (code & 1u) ? 'A' : 'G'));
      else alts.emplace_back(1, bases[code & 3u]);
    }
  };
  if (mode != 0 || n_genomes < 8 || !gt) {
    constexpr uint64_t kChunk = 1u << 16;
    uint64_t alt_cursor = 0;
    for (uint64_t r0 = 0; r0 < n_records; r0 += kChunk) {
      const uint64_t r1 = std::min(n_records, r0 + kChunk);
      std::vector<std::string> refs, alts, record_alts;
      uint64_t cursor = alt_cursor;
      for (uint64_t r = r0; r < r1; ++r) {
        refs.emplace_back();
        sequences(r, cursor, refs.back(), record_alts);
        for (auto& x : record_alts) alts.push_back(std::move(x));
        cursor += n_alts[r];
      }
      std::vector<const char*> ref_ptrs, alt_ptrs;
      for (const auto& x : refs) ref_ptrs.push_back(x.c_str());
      for (const auto& x : alts) alt_ptrs.push_back(x.c_str());
      const int rc = kgo_population_add_records(p, mode, contig, r1 - r0, offsets + r0, ref_ptrs.data(), n_alts + r0, alt_ptrs.data(),
                                                pass ? pass + r0 : nullptr, af_flat ? af_flat + alt_cursor * SUPER_POP_COUNT : nullptr,
                                                n_genomes, genome_ids, gt ? gt + r0 * n_genomes * 2 : nullptr);
      if (rc != 0) return rc;
      alt_cursor = cursor;
    }
    return 0;
  }
  // mode 0, threaded.  Genome1000VCFImpl::ParseRecord (kgl_variant_factory_1000_impl.cpp:63-145): per record one shared
  // Variant per (alt, phase) that some genome carries; a genome receives its phase-A variant before its phase-B one.
  const std::string contig_id(contig);
  std::vector<uint64_t> first_alt(n_records + 1, 0);
  for (uint64_t r = 0; r < n_records; ++r) first_alt[r + 1] = first_alt[r] + n_alts[r];
  std::vector<VariantPtr> variants(first_alt[n_records] * 2);           // [(first_alt[r] + a) * 2 + phase]
  const uint64_t first_record_index = p->next_record;
  p->next_record += n_records;
  std::vector<std::shared_ptr<GenomeDB>> genomes(n_genomes);
  for (uint64_t g = 0; g < n_genomes; ++g) genomes[g] = p->pop->getCreateGenome(genome_ids[g]);
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>(poolThreads(n_genomes), 64));
  {
    // the Variant objects of every record, records split over the threads
    std::vector<std::thread> pool;
    for (size_t t = 0; t < n_threads; ++t)
      pool.emplace_back([&, t]() {
        std::string ref;
        std::vector<std::string> alts;
        for (uint64_t r = n_records * t / n_threads; r < n_records * (t + 1) / n_threads; ++r) {
          const uint32_t A = n_alts[r];
          auto ev = std::make_shared<RecordEvidence>();
          ev->record_index = first_record_index + r;
          ev->pass = pass ? pass[r] != 0 : true;
          ev->alt_count = A;
          if (af_flat) {
            ev->af.resize(static_cast<size_t>(SUPER_POP_COUNT) * A);
            for (uint32_t a = 0; a < A; ++a)
              for (int sp = 0; sp < SUPER_POP_COUNT; ++sp)
                ev->af[static_cast<size_t>(sp) * A + a] = af_flat[(first_alt[r] + a) * SUPER_POP_COUNT + sp];
          }
          sequences(r, first_alt[r], ref, alts);
          for (uint32_t a = 0; a < A; ++a)
            for (int phase = 0; phase < 2; ++phase)
              variants[(first_alt[r] + a) * 2 + phase] = std::make_shared<const Variant>(
                  contig_id, offsets[r], phase == 0 ? VariantPhase::DIPLOID_PHASE_A : VariantPhase::DIPLOID_PHASE_B, ref, alts[a], ev, a);
        }
      });
    for (auto& th : pool) th.join();
  }
  {
    std::vector<std::thread> pool;
    for (size_t t = 0; t < n_threads; ++t)
      pool.emplace_back([&, t]() {
        const uint64_t g_begin = n_genomes * t / n_threads, g_end = n_genomes * (t + 1) / n_threads;
        for (uint64_t r = 0; r < n_records; ++r)
          for (uint64_t g = g_begin; g < g_end; ++g)
            for (int phase = 0; phase < 2; ++phase) {
              const uint32_t idx = gt[(r * n_genomes + g) * 2 + phase];
              if (idx != 0 && idx <= n_alts[r]) genomes[g]->addVariant(variants[(first_alt[r] + idx - 1) * 2 + phase]);
            }
      });
    for (auto& th : pool) th.join();
  }
  return 0;
}

// VCF text (1000 Genomes flavour) -> Variants in the population; returns the record count (or -1).
long kgo_population_add_vcf_1000(kgo_pop* p, const char* text, uint64_t len) {
  if (!p || !text) return -1;
  std::vector<std::string> names;
  const long n = addVcf1000(*p->pop, std::string_view(text, len), &names);
  if (p->input_ids.empty()) p->input_ids = names;
  return n;
}

// VCF text (unphased P. falciparum flavour) -> Variants in the population; returns the record count (or -1).
long kgo_population_add_vcf_pf(kgo_pop* p, const char* text, uint64_t len) {
  if (!p || !text) return -1;
  std::vector<std::string> names;
  const long n = addVcfPf(*p->pop, std::string_view(text, len), &names);
  if (p->input_ids.empty()) p->input_ids = names;
  return n;
}

// VCF text of a mono-genome frequency source (Gnomad ...) -> UNPHASED Variants in one genome; -1 for an unknown source.
long kgo_population_add_vcf_mono(kgo_pop* p, const char* text, uint64_t len, const char* source, const char* genome_id) {
  if (!p || !text || !source || !genome_id) return -1;
  if (p->input_ids.empty()) p->input_ids = {genome_id};
  return addVcfMonoGenome(*p->pop, std::string_view(text, len), source, genome_id);
}

// PopulationDB::viewFilter(P7VariantFilter()) (kga_analysis_lib_PfFilter.cpp:63-67).
kgo_pop* kgo_population_filter_p7(kgo_pop* p) {
  if (!p) return nullptr;
  auto* out = new kgo_pop();
  out->pop = std::shared_ptr<PopulationDB>(p->pop->viewFilter([](const Variant& v) { return p7VariantFilter(v); }));
  out->input_ids = p->input_ids;
  return out;
}

// Variant::canonicalSequences on text; ref_out / alt_out hold at least len(ref) + len(alt) + 1 bytes.
uint64_t kgo_canonical(const char* ref, const char* alt, uint64_t offset, char* ref_out, char* alt_out) {
  std::string r, a;
  uint64_t o = 0;
  canonicalSequences(ref, alt, offset, r, a, o);
  std::strcpy(ref_out, r.c_str());
  std::strcpy(alt_out, a.c_str());
  return o;
}

int kgo_gt_alternate_index(const char* contig, const char* genotype, uint64_t n_alt, uint64_t out[2]) {
  const auto ab = alternateIndex1000(contig, genotype, n_alt);
  out[0] = ab.first;
  out[1] = ab.second;
  return 0;
}

uint64_t kgo_population_variant_count(kgo_pop* p) { return p ? p->pop->variantCount() : 0; }
uint64_t kgo_population_genome_count(kgo_pop* p) { return p ? p->pop->getMap().size() : 0; }

// Sorted (std::map) genome order -> index into the caller's id list; -1 if the id was never declared.
int kgo_population_genome_order(kgo_pop* p, int64_t* input_index) {
  if (!p || !input_index) return -1;
  std::map<std::string, int64_t> pos;
  for (size_t i = 0; i < p->input_ids.size(); ++i) pos.emplace(p->input_ids[i], static_cast<int64_t>(i));
  size_t k = 0;
  for (const auto& [id, g] : p->pop->getMap()) {
    auto it = pos.find(id);
    input_index[k++] = it == pos.end() ? -1 : it->second;
  }
  return 0;
}

// PopulationDB::viewFilter(AndFilter(SNPFilter(), PassFilter())) (kga_analysis_inbreed.cpp:79).
kgo_pop* kgo_population_filter_snp_pass(kgo_pop* p) {
  if (!p) return nullptr;
  auto* out = new kgo_pop();
  out->pop = std::shared_ptr<PopulationDB>(p->pop->viewFilter([](const Variant& v) { return v.isSNP() && v.passFilter(); }));
  out->input_ids = p->input_ids;
  return out;
}

// ---- VariantDBVariant --------------------------------------------------------------------------

kgo_vdb* kgo_vdb_create(kgo_pop* p, double* seconds) {
  if (!p) return nullptr;
  auto* h = new kgo_vdb();
  h->pop = p->pop;
  const auto t0 = std::chrono::steady_clock::now();
  h->vdb = std::make_unique<VariantDBVariant>(h->pop);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return h;
}

void kgo_vdb_destroy(kgo_vdb* h) { delete h; }
uint64_t kgo_vdb_variants(kgo_vdb* h) { return h ? h->vdb->variantMap().size() : 0; }
uint64_t kgo_vdb_genomes(kgo_vdb* h) { return h ? h->vdb->genomeData().size() : 0; }
uint64_t kgo_vdb_warnings(kgo_vdb* h) { return h ? h->vdb->warnings() : 0; }

// For variant rank i (lexicographic HGVS): the record and alt it was cut from.
int kgo_vdb_variant_keys(kgo_vdb* h, uint64_t* record_index, uint32_t* alt_index) {
  if (!h) return -1;
  for (const auto& [hgvs, rec] : h->vdb->variantMap()) {
    record_index[rec.second] = rec.first->evidence().record_index;
    alt_index[rec.second] = rec.first->altVariantIndex();
  }
  return 0;
}

int kgo_vdb_hgvs(kgo_vdb* h, uint64_t i, char* buf, size_t len) {
  if (!h) return -1;
  for (const auto& [hgvs, rec] : h->vdb->variantMap())
    if (rec.second == i) { copyString(hgvs, buf, len); return 0; }
  return -1;
}

int kgo_vdb_genome_id(kgo_vdb* h, uint64_t i, char* buf, size_t len) {
  if (!h || i >= h->vdb->genomeData().size()) return -1;
  copyString(h->vdb->genomeData()[i].first, buf, len);
  return 0;
}

// summaryByVariant for every variant in index order, the loop of CalcFWS::updateVariantFWSMap
// (kga_analysis_PfEMP_FWS.cpp:41-70).  out[i] = { referenceHomozygous, minorHeterozygous, minorHomozygous }.
int kgo_vdb_summary_by_variant(kgo_vdb* h, uint64_t* out, double* seconds) {
  if (!h || !out) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  for (const auto& [hgvs, rec] : h->vdb->variantMap()) {
    const AlleleSummmary s = h->vdb->summaryByVariant(rec.first);
    out[rec.second * 3 + 0] = s.referenceHomozygous_;
    out[rec.second * 3 + 1] = s.minorHeterozygous_;
    out[rec.second * 3 + 2] = s.minorHomozygous_;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int kgo_vdb_summary_by_genome(kgo_vdb* h, uint64_t* out, double* seconds) {
  if (!h || !out) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  size_t i = 0;
  for (const auto& [genome_id, idx] : h->vdb->genomeMap()) {
    const AlleleSummmary s = h->vdb->summaryByGenome(genome_id);
    out[idx * 3 + 0] = s.referenceHomozygous_;
    out[idx * 3 + 1] = s.minorHeterozygous_;
    out[idx * 3 + 2] = s.minorHomozygous_;
    ++i;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int kgo_vdb_population_summary(kgo_vdb* h, uint64_t out[3]) {
  if (!h || !out) return -1;
  const AlleleSummmary s = h->vdb->populationSummary();
  out[0] = s.referenceHomozygous_;
  out[1] = s.minorHeterozygous_;
  out[2] = s.minorHomozygous_;
  return 0;
}

// The dense matrix itself: out[g][v], genome rank g, variant rank v.
int kgo_vdb_dosage(kgo_vdb* h, uint8_t* out) {
  if (!h || !out) return -1;
  const size_t V = h->vdb->variantMap().size();
  size_t g = 0;
  for (const auto& [id, row] : h->vdb->genomeData()) {
    std::memcpy(out + g * V, row.data(), V);
    ++g;
  }
  return 0;
}

// ---- dense tier: the same summary loops over a caller-supplied dosage matrix ---------------------
// VariantDBGenomeData is G separately allocated uint8 vectors (kgl_variant_db_variant.h:49-51); the
// by-variant loop walks one column across all of them (kgl_variant_db_variant.cpp:143-165).  This tier
// skips the pointer-chasing PopulationDB so that C2-scale inputs (1e9 cells) can be checked and timed.

struct kgo_dense {
  VariantDBGenomeData genome_data;
  size_t n_variants = 0;
};

kgo_dense* kgo_dense_create(const uint8_t* dosage /* [G][V] */, uint64_t G, uint64_t V) {
  auto* d = new kgo_dense();
  d->n_variants = V;
  d->genome_data.reserve(G);
  for (uint64_t g = 0; g < G; ++g)
    d->genome_data.emplace_back("G" + std::to_string(g), std::vector<uint8_t>(dosage + g * V, dosage + (g + 1) * V));
  return d;
}

void kgo_dense_destroy(kgo_dense* d) { delete d; }

int kgo_dense_summary_by_variant(kgo_dense* d, uint64_t v0, uint64_t v1, uint64_t* out /* [v1-v0][3] */, double* seconds) {
  if (!d || !out || v1 > d->n_variants || v0 > v1) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t variant_index = v0; variant_index < v1; ++variant_index) {
    AlleleSummmary s;
    for (const auto& [genome, variant_vector] : d->genome_data) {
      switch (variant_vector[variant_index]) {
        case 0: ++s.referenceHomozygous_; break;
        case 1: ++s.minorHeterozygous_; break;
        case 2: ++s.minorHomozygous_; break;
        default: break;
      }
    }
    uint64_t* o = out + (variant_index - v0) * 3;
    o[0] = s.referenceHomozygous_;
    o[1] = s.minorHeterozygous_;
    o[2] = s.minorHomozygous_;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int kgo_dense_summary_by_genome(kgo_dense* d, const uint8_t* variant_mask, uint64_t* out /* [G][3] */, double* seconds) {
  if (!d || !out) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  size_t g = 0;
  for (const auto& [genome, allele_vector] : d->genome_data) {
    AlleleSummmary s;
    for (size_t v = 0; v < allele_vector.size(); ++v) {
      if (variant_mask && !variant_mask[v]) continue;   // variant absent from the filtered population
      switch (allele_vector[v]) {
        case 0: ++s.referenceHomozygous_; break;
        case 1: ++s.minorHeterozygous_; break;
        case 2: ++s.minorHomozygous_; break;
        default: break;
      }
    }
    out[g * 3 + 0] = s.referenceHomozygous_;
    out[g * 3 + 1] = s.minorHeterozygous_;
    out[g * 3 + 2] = s.minorHomozygous_;
    ++g;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

// ---- CalcFWS / HeteroHomoZygous ----------------------------------------------------------------

// variant_out [V][3] in HGVS order of the UNFILTERED population; genome_out [G][11][3] in genome-id order.
int kgo_fws(kgo_pop* p, uint64_t* variant_out, uint64_t* genome_out) {
  if (!p) return -1;
  CalcFWS fws;
  fws.calcFwsStatistics(p->pop);
  if (variant_out) {
    size_t i = 0;
    for (const auto& [hgvs, s] : fws.getVariantMap()) {
      variant_out[i * 3 + 0] = s.referenceHomozygous_;
      variant_out[i * 3 + 1] = s.minorHeterozygous_;
      variant_out[i * 3 + 2] = s.minorHomozygous_;
      ++i;
    }
  }
  if (genome_out) {
    size_t g = 0;
    for (const auto& [id, arr] : fws.getGenomeMap()) {
      for (size_t b = 0; b < FWS_FREQUENCY_ARRAY_SIZE; ++b) {
        genome_out[(g * FWS_FREQUENCY_ARRAY_SIZE + b) * 3 + 0] = arr[b].referenceHomozygous_;
        genome_out[(g * FWS_FREQUENCY_ARRAY_SIZE + b) * 3 + 1] = arr[b].minorHeterozygous_;
        genome_out[(g * FWS_FREQUENCY_ARRAY_SIZE + b) * 3 + 2] = arr[b].minorHomozygous_;
      }
      ++g;
    }
  }
  return 0;
}

// out[g][7] for one contig, genome-id order: total, snp, indel, hom_minor, het_minor, het_ref_minor, hom_ref.
int kgo_hethom(kgo_pop* p, const char* contig, uint64_t* out) {
  if (!p || !contig || !out) return -1;
  auto result = analyzeVariantPopulation(*p->pop);
  size_t g = 0;
  for (const auto& [genome_id, contig_map] : result) {
    VariantAnalysisType r;
    auto it = contig_map.find(contig);
    if (it != contig_map.end()) r = it->second;
    uint64_t* o = out + g * 7;
    o[0] = r.total_variants_;
    o[1] = r.snp_count_;
    o[2] = r.indel_count_;
    o[3] = r.homozygous_minor_alleles_;
    o[4] = r.heterozygous_minor_alleles_;
    o[5] = r.heterozygous_reference_minor_alleles_;
    o[6] = r.homozygous_reference_alleles_;
    ++g;
  }
  return 0;
}

// 1 where the genome's record map holds the contig at all (a genome holds a contig once a variant was added to it, or
// when the parser created it up front).
int kgo_hethom_present(kgo_pop* p, const char* contig, uint8_t* out) {
  if (!p || !contig || !out) return -1;
  auto result = analyzeVariantPopulation(*p->pop);
  size_t g = 0;
  for (const auto& [genome_id, contig_map] : result) out[g++] = contig_map.count(contig) ? 1 : 0;
  return 0;
}

// The Variant objects each offset filter of kgl_variant_filter_db_offset.cpp leaves of one contig, per genome (genome-id
// order): out[g][4] = { HomozygousFilter, HeterozygousFilter, DiploidFilter, UniqueUnphasedFilter } -- what
// PopulationDB::viewFilter(F) followed by variantCount() gives for that genome and contig.
int kgo_offset_filter_counts(kgo_pop* p, const char* contig, uint64_t* out) {
  if (!p || !contig || !out) return -1;
  size_t g = 0;
  for (const auto& [genome_id, genome_ptr] : p->pop->getMap()) {
    uint64_t* o = out + g * 4;
    o[0] = o[1] = o[2] = o[3] = 0;
    auto found = genome_ptr->getMap().find(contig);
    if (found != genome_ptr->getMap().end())
      for (const auto& [offset, offset_ptr] : found->second->getMap()) {
        o[0] += homozygousFilter(*offset_ptr)->getVariantArray().size();
        o[1] += heterozygousFilter(*offset_ptr)->getVariantArray().size();
        o[2] += diploidFilter(*offset_ptr)->getVariantArray().size();
        o[3] += uniqueUnphasedFilter(*offset_ptr)->getVariantArray().size();
      }
    ++g;
  }
  return 0;
}

// UniquePhasedFilter the same way: out[g] = the Variant objects it leaves of the genome's contig.
int kgo_unique_phased_counts(kgo_pop* p, const char* contig, uint64_t* out) {
  if (!p || !contig || !out) return -1;
  size_t g = 0;
  for (const auto& [genome_id, genome_ptr] : p->pop->getMap()) {
    out[g] = 0;
    auto found = genome_ptr->getMap().find(contig);
    if (found != genome_ptr->getMap().end())
      for (const auto& [offset, offset_ptr] : found->second->getMap()) out[g] += uniquePhasedFilter(*offset_ptr)->getVariantArray().size();
    ++g;
  }
  return 0;
}

double kgo_wrights_fis(const uint64_t location[7], const uint64_t genome[7]) {
  VariantAnalysisType l, g;
  l.total_variants_ = location[0]; l.heterozygous_minor_alleles_ = location[4]; l.heterozygous_reference_minor_alleles_ = location[5];
  g.total_variants_ = genome[0]; g.heterozygous_minor_alleles_ = genome[4]; g.heterozygous_reference_minor_alleles_ = genome[5];
  return wrightsFIS(l, g);
}

// ---- Pf7 sample resources: genome filters, location summary, F_IS ------------------------------------

// FilterPf7::qualityFilter's genome part + squareContigs over the sample / FWS resource files; nullptr when a file does
// not parse.
kgo_pop* kgo_population_filter_pf7_genomes(kgo_pop* p, const char* sample_file, const char* fws_file, int filter_qc, int filter_fws,
                                           double fws_threshold) {
  if (!p || !sample_file || !fws_file) return nullptr;
  Pf7SampleMap samples;
  Pf7FwsMap fws;
  if (!parsePf7SampleFile(sample_file, samples) || !parsePf7FwsFile(fws_file, fws)) return nullptr;
  auto* out = new kgo_pop();
  out->pop = pf7GenomeFilter(*p->pop, samples, fws, filter_qc != 0, filter_fws != 0, fws_threshold);
  out->input_ids = p->input_ids;
  return out;
}

// PfEMPAnalysis::finalizeAnalysis' two HeteroHomoZygous files (kga_analysis_PfEMP.cpp:146-163) for a population.
// radius_km: PfEMPAnalysis::SAMPLE_LOCATION_RADIUS_ is 0 (kga_analysis_PfEMP.h:64).
int kgo_pfemp_location_write(kgo_pop* p, const char* sample_file, const char* fws_file, double radius_km, const char* statistics_csv,
                             const char* location_csv) {
  if (!p || !sample_file || !fws_file || !statistics_csv || !location_csv) return -1;
  Pf7SampleMap samples;
  Pf7FwsMap fws;
  if (!parsePf7SampleFile(sample_file, samples) || !parsePf7FwsFile(fws_file, fws)) return -2;
  try {
    Pf7SampleLocation distance(samples);
    HeteroHomoZygous hethom;
    hethom.analyzeVariantPopulation(*p->pop, fws, samples);
    auto summary = hethom.location_summary(samples, distance, radius_km, fws);
    hethom.UpdateSampleLocation(summary);
    hethom.write_variant_results(statistics_csv, summary);
    hethom.write_location_results(location_csv, summary);
  } catch (std::exception&) {
    return -3;   // a year that is not a number ends the reference's run
  }
  return 0;
}

// ---- inbreeding --------------------------------------------------------------------------------

// AlleleFreqVector::alleleClassFrequencies for raw minor allele frequencies (no Variant objects needed):
// out = { majorHom, majorHet, minorHom, minorHet }.
int kgo_class_frequencies(const double* minor_af, uint32_t n, double inbreeding, int normalize, double out[4]) {
  // The arithmetic of unadjustedAlleleClassFrequencies (_freq.cpp:127-205) on caller-supplied doubles
  // (AlleleFreqVector itself reads float32 INFO values through Variant objects).
  std::vector<double> freqs(minor_af, minor_af + n);
  double sum_minor_freq = 0.0;
  for (double f : freqs) sum_minor_freq += f;
  const double major_frequency = std::max(0.0, (1.0 - sum_minor_freq));
  std::vector<double> m;
  for (double f : freqs) m.push_back(sum_minor_freq > 1.0 ? f / sum_minor_freq : f);
  double minor_homozygous = 0.0;
  for (double f : m) minor_homozygous += (inbreeding * f) + ((1.0 - inbreeding) * f * f);
  double minor_heterozygous = 0.0;
  for (size_t i = 0; i < m.size(); ++i)
    for (size_t j = i + 1; j < m.size(); ++j) minor_heterozygous += (1.0 - inbreeding) * 2.0 * m[i] * m[j];
  const double major_homozygous = (inbreeding * major_frequency) + ((1.0 - inbreeding) * major_frequency * major_frequency);
  double major_heterozygous = 0.0;
  for (double f : m) major_heterozygous += (1.0 - inbreeding) * 2.0 * major_frequency * f;
  AlleleClassFrequencies cf(major_homozygous, major_heterozygous, minor_homozygous, minor_heterozygous, inbreeding);
  if (normalize) cf.normalize();
  out[0] = cf.majorHomozygous();
  out[1] = cf.majorHeterozygous();
  out[2] = cf.minorHomozygous();
  out[3] = cf.minorHeterozygous();
  return 0;
}

// RetrieveLociiVector::getLociiCount / getLociiFromTo on the (single-genome, single-contig) reference population.
int64_t kgo_sample_locii(kgo_pop* reference, int super_pop, int by_count, uint64_t lower, uint64_t upper, uint64_t spacing,
                         uint64_t count, double min_af, double max_af, uint64_t* out, uint64_t out_cap) {
  if (!reference || reference->pop->getMap().size() != 1) return -1;
  const auto& genome = reference->pop->getMap().begin()->second;
  if (genome->getMap().size() != 1) return -1;
  const ContigDB& contig = *genome->getMap().begin()->second;
  LociiVectorArguments a;
  a.lower_offset = lower; a.upper_offset = upper; a.spacing = spacing; a.locii_count = count;
  a.allele_frequency_min = std::clamp(min_af, 0.0, 1.0);
  a.allele_frequency_max = std::clamp(max_af, 0.0, 1.0);
  const auto v = by_count ? getLociiCount(contig, super_pop, a) : getLociiFromTo(contig, super_pop, a);
  for (size_t i = 0; i < v.size() && i < out_cap; ++i) out[i] = v[i];
  return static_cast<int64_t>(v.size());
}

// One window = InbreedingAnalysis::populationInbreedingSample (_diploid.cpp:83-94): locus lists per super
// population from [lower, upper] (getLociiFromTo), then one task per genome.
// super_pop_of_genome: [G] in genome-id (std::map) order; -1 = no PED record (genome skipped).
// counts_out [G][5] = major_het, minor_het, minor_hom, major_hom, total; freqs_out [G][5] = the four class
// frequency sums in the same order + inbred_allele_sum.  Skipped genomes are left untouched.
int kgo_inbreed_window(kgo_pop* reference, kgo_pop* diploid, const int32_t* super_pop_of_genome, const char* algorithm,
                       uint64_t lower, uint64_t upper, uint64_t spacing, uint64_t count, double min_af, double max_af,
                       uint64_t start_seed, uint64_t* counts_out, double* freqs_out, uint8_t* present_out,
                       double* seconds) {
  if (!reference || !diploid || !algorithm) return -1;
  if (reference->pop->getMap().size() != 1) return -1;
  const auto& ref_genome = reference->pop->getMap().begin()->second;
  if (ref_genome->getMap().size() != 1) return -1;
  const auto& [contig_id, contig_ptr] = *ref_genome->getMap().begin();
  InbreedingParameters params;
  params.locii.lower_offset = lower; params.locii.upper_offset = upper; params.locii.spacing = spacing;
  params.locii.locii_count = count;
  params.locii.allele_frequency_min = std::clamp(min_af, 0.0, 1.0);
  params.locii.allele_frequency_max = std::clamp(max_af, 0.0, 1.0);
  params.algorithm = algorithm;
  params.start_seed = start_seed;
  std::map<std::string, int> sp_map;
  size_t g = 0;
  for (const auto& [id, genome] : diploid->pop->getMap()) {
    if (super_pop_of_genome[g] >= 0) sp_map[id] = super_pop_of_genome[g];
    ++g;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::map<int, std::shared_ptr<const ContigDB>> locus_map;
  for (int sp = 0; sp < SUPER_POP_COUNT; ++sp) locus_map[sp] = getLocusList(*contig_ptr, sp, params.locii);
  ResultsMap results = processResults(*diploid->pop, contig_id, locus_map, sp_map, params);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  g = 0;
  for (const auto& [id, genome] : diploid->pop->getMap()) {
    auto it = results.find(id);
    if (present_out) present_out[g] = it != results.end();
    if (it != results.end()) fillResults(it->second, counts_out + g * 5, freqs_out + g * 5);
    ++g;
  }
  return 0;
}

// The 1-D Nelder-Mead the oracle restates nlopt's LN_NELDERMEAD with (neldermead1D), run on a closed-form objective over the
// reference's box [-1, 1] with its stopping rule (xtol_abs 1e-6, 500 evaluations): the points it evaluates, in order, so
// that tests/test_oracle_pins.py can pin every assumption about nlopt's behaviour that DESIGN.md lists.
// objective 0: -(x - a)^2;  1: -|x - a|;  2: a * x (monotone: the optimum on a bound);  3: a two-level step, 1 for x >= a else 0.
int kgo_neldermead_path(int objective, double a, double x0, double* path, int max_path, int* n_path, double* result) {
  if (!path || !n_path || !result || max_path <= 0) return -1;
  int n = 0;
  auto f = [&](double x) {
    if (n < max_path) path[n] = x;
    ++n;
    switch (objective) {
      case 0: return -(x - a) * (x - a);
      case 1: return -std::fabs(x - a);
      case 2: return a * x;
      default: return x >= a ? 1.0 : 0.0;
    }
  };
  int evaluations = 0;
  *result = neldermead1D(f, x0, -1.0, 1.0, 1e-06, 500, &evaluations);
  *n_path = n;
  return evaluations == n ? 0 : -2;
}

// logLikelihood (_calc.cpp:94-129) of every genome at a caller-chosen coefficient: the objective processLogLikelihood
// maximises, over the same window and locus lists as kgo_inbreed_window.  f[G] / out[G] in genome-id order.
int kgo_loglikelihood_at(kgo_pop* reference, kgo_pop* diploid, const int32_t* super_pop_of_genome, uint64_t lower, uint64_t upper,
                         uint64_t spacing, double min_af, double max_af, const double* f, double* out) {
  if (!reference || !diploid || !f || !out) return -1;
  if (reference->pop->getMap().size() != 1) return -1;
  const auto& ref_genome = reference->pop->getMap().begin()->second;
  if (ref_genome->getMap().size() != 1) return -1;
  const auto& [contig_id, contig_ptr] = *ref_genome->getMap().begin();
  LociiVectorArguments args;
  args.lower_offset = lower; args.upper_offset = upper; args.spacing = spacing;
  args.allele_frequency_min = std::clamp(min_af, 0.0, 1.0);
  args.allele_frequency_max = std::clamp(max_af, 0.0, 1.0);
  std::map<int, std::shared_ptr<const ContigDB>> locus_map;
  WorkflowThreads thread_pool(poolThreads(diploid->pop->getMap().size()));
  std::vector<std::future<double>> futures;
  std::vector<size_t> slot;
  size_t g = 0;
  for (const auto& [id, genome] : diploid->pop->getMap()) {
    const int sp = super_pop_of_genome[g];
    auto contig_opt = genome->getContig(contig_id);
    if (sp >= 0 && contig_opt) {
      if (!locus_map.count(sp)) locus_map[sp] = getLocusList(*contig_ptr, sp, args);
      std::shared_ptr<const ContigDB> contig = contig_opt.value();
      std::shared_ptr<const ContigDB> locus_list = locus_map[sp];
      const std::string gid = id;
      const double coefficient = f[g];
      futures.push_back(thread_pool.enqueueFuture([=]() {
        auto [frequency_vector, locus_results] = generateFrequencies(gid, *contig, sp, *locus_list);
        return logLikelihood(coefficient, frequency_vector);
      }));
      slot.push_back(g);
    }
    ++g;
  }
  for (size_t i = 0; i < futures.size(); ++i) out[slot[i]] = futures[i].get();
  return 0;
}

// Dense tier of one window (oracle/kgo_inbreed_dense.cpp): every genome of allele_pairs [n_records][n_genomes][2]
// (raw GT allele indices of the record at record_offsets[r]) against the locus list of `super_pop` from [lower, upper].
// counts_out [G][5] as kgo_inbreed_window; freqs_out [G][6] = the four class-frequency sums, Simple, RitlandLocus.
int kgo_inbreed_dense(kgo_pop* reference_all, kgo_pop* reference_snp_pass, int super_pop, uint64_t lower, uint64_t upper,
                      uint64_t spacing, double min_af, double max_af, const uint64_t* record_offsets, uint64_t n_records,
                      const uint8_t* allele_pairs, uint64_t n_genomes, int phased, uint64_t* counts_out, double* freqs_out,
                      double* seconds) {
  if (!reference_all || !reference_snp_pass || !record_offsets || !allele_pairs || !counts_out || !freqs_out) return -1;
  auto single_contig = [](kgo_pop* p) -> const ContigDB* {
    if (p->pop->getMap().size() != 1) return nullptr;
    const auto& genome = p->pop->getMap().begin()->second;
    return genome->getMap().size() == 1 ? genome->getMap().begin()->second.get() : nullptr;
  };
  const ContigDB* all = single_contig(reference_all);
  const ContigDB* snp_pass = single_contig(reference_snp_pass);
  if (!all || !snp_pass) return -1;
  LociiVectorArguments args;
  args.lower_offset = lower; args.upper_offset = upper; args.spacing = spacing;
  args.allele_frequency_min = std::clamp(min_af, 0.0, 1.0);
  args.allele_frequency_max = std::clamp(max_af, 0.0, 1.0);
  return inbreedDense(*all, *snp_pass, super_pop, args, record_offsets, n_records, allele_pairs, n_genomes, phased != 0, counts_out,
                      freqs_out, seconds);
}

// The restart start points of n per-genome tasks under processResults' seeding (task k: start_seed + k): out[k][restarts].
int kgo_restart_draws(const char* algorithm, uint64_t start_seed, uint64_t n, uint64_t restarts, double* out) {
  auto algo = namedAlgorithm(algorithm ? algorithm : "");
  if (!algo || !out || start_seed == 0) return -1;
  for (uint64_t k = 0; k < n; ++k) {
    const auto draws = restartDraws(algo.value(), start_seed + k, restarts);
    for (uint64_t r = 0; r < restarts; ++r) out[k * restarts + r] = draws[r];
  }
  return 0;
}

// The whole window loop (populationInbreeding, _diploid.cpp:18-79).
kgo_columns* kgo_population_inbreeding(kgo_pop* reference, kgo_pop* diploid, const int32_t* super_pop_of_genome,
                                       const char* algorithm, uint64_t lower, uint64_t upper, uint64_t spacing,
                                       uint64_t count, double min_af, double max_af, uint64_t start_seed) {
  if (!reference || !diploid || !algorithm) return nullptr;
  InbreedingParameters params;
  params.locii.lower_offset = lower; params.locii.upper_offset = upper; params.locii.spacing = spacing;
  params.locii.locii_count = count;
  params.locii.allele_frequency_min = std::clamp(min_af, 0.0, 1.0);
  params.locii.allele_frequency_max = std::clamp(max_af, 0.0, 1.0);
  params.algorithm = algorithm;
  params.start_seed = start_seed;
  std::map<std::string, int> sp_map;
  auto* out = new kgo_columns();
  size_t g = 0;
  for (const auto& [id, genome] : diploid->pop->getMap()) {
    if (super_pop_of_genome[g] >= 0) sp_map[id] = super_pop_of_genome[g];
    out->genome_order.push_back(id);
    ++g;
  }
  out->columns = populationInbreeding(*reference->pop, *diploid->pop, sp_map, params);
  return out;
}

// InbreedingOutput::writePedResults (kga_analysis_inbreed_output.cpp:188-305) on the columns of a window loop, the way
// InbreedAnalysis::finalizeAnalysis calls it (kga_analysis_inbreed.cpp:140-142).  ped: n_ped rows of 9 strings
// { genome, population, population description, super population, super description, relationship, sex, maternal id,
// paternal id } -- the HsGenealogyRecord fields the writer prints.  Same header lines, default stream formatting and the
// trailing delimiter after the last column as the reference.
int kgo_columns_write_ped(kgo_columns* c, const char* path, const char* param_ident, const char* algorithm, double min_af, double max_af,
                          uint64_t spacing, uint64_t count, const char* const* ped, uint64_t n_ped) {
  if (!c || !path || !param_ident || !algorithm || c->columns.empty()) return -1;
  constexpr char DELIMITER_ = ',';
  std::map<std::string, const char* const*> ped_of;
  for (uint64_t i = 0; i < n_ped; ++i) ped_of[ped[i * 9]] = ped + i * 9;
  std::ofstream outfile(path, std::ofstream::out | std::ofstream::trunc);
  if (!outfile.good()) return -1;
  outfile << param_ident << DELIMITER_ << "Algorithm:" << algorithm << DELIMITER_ << "Min_AF:" << std::clamp(min_af, 0.0, 1.0) << DELIMITER_
          << "Max_AF:" << std::clamp(max_af, 0.0, 1.0) << DELIMITER_ << "Spacing:" << spacing << DELIMITER_ << "Count:" << count << '\n';
  outfile << "Sample" << DELIMITER_ << "Population" << DELIMITER_ << "Description" << DELIMITER_ << "SuperPopulation" << DELIMITER_ << "Description"
          << DELIMITER_ << "Relationship" << DELIMITER_ << "Sex" << DELIMITER_ << "Mother" << DELIMITER_ << "Father";
  for (const auto& column : c->columns) outfile << DELIMITER_ << column.first;
  outfile << '\n';
  std::set<std::string> genome_set;
  for (const auto& [genome, data] : c->columns.front().second) genome_set.insert(genome);
  for (const auto& genome_id : genome_set) {
    auto record = ped_of.find(genome_id);
    if (record == ped_of.end()) continue;                 // "does not have a PED record" (:263-268)
    const char* const* f = record->second;
    outfile << genome_id << DELIMITER_;
    for (int k = 1; k <= 4; ++k) outfile << f[k] << DELIMITER_;        // population, description, super population, description
    outfile << f[5] << DELIMITER_ << f[6] << DELIMITER_ << f[7] << DELIMITER_ << f[8] << DELIMITER_;   // relationship, sex, mother, father
    for (const auto& column : c->columns) {
      auto find_result = column.second.find(genome_id);
      if (find_result == column.second.end()) return -2;
      outfile << find_result->second.inbred_allele_sum << DELIMITER_;
    }
    outfile << '\n';
  }
  outfile.flush();
  return 0;
}

void kgo_columns_destroy(kgo_columns* c) { delete c; }
uint64_t kgo_columns_count(kgo_columns* c) { return c ? c->columns.size() : 0; }
int kgo_columns_ident(kgo_columns* c, uint64_t i, char* buf, size_t len) {
  if (!c || i >= c->columns.size()) return -1;
  copyString(c->columns[i].first, buf, len);
  return 0;
}
int kgo_columns_results(kgo_columns* c, uint64_t i, uint64_t* counts_out, double* freqs_out, uint8_t* present_out) {
  if (!c || i >= c->columns.size()) return -1;
  for (size_t g = 0; g < c->genome_order.size(); ++g) {
    auto it = c->columns[i].second.find(c->genome_order[g]);
    if (present_out) present_out[g] = it != c->columns[i].second.end();
    if (it != c->columns[i].second.end()) fillResults(it->second, counts_out + g * 5, freqs_out + g * 5);
  }
  return 0;
}

// Synthetic-inbreeding self-check (SyntheticAnalysis::processSynResults, _synthetic.cpp:74-138) for one super
// population and one window: generate 101 genomes with F = -0.5 .. 0.5, estimate F back.
// syn_out / calc_out: [101]; returns the number of genomes.
int64_t kgo_synthetic_check(kgo_pop* reference, int super_pop, const char* algorithm, uint64_t lower, uint64_t upper,
                            uint64_t spacing, double min_af, double max_af, uint64_t seed, double* syn_out,
                            double* calc_out, uint64_t cap) {
  if (!reference || reference->pop->getMap().size() != 1) return -1;
  const auto& ref_genome = reference->pop->getMap().begin()->second;
  if (ref_genome->getMap().size() != 1) return -1;
  const auto& [contig_id, contig_ptr] = *ref_genome->getMap().begin();
  InbreedingParameters params;
  params.locii.lower_offset = lower; params.locii.upper_offset = upper; params.locii.spacing = spacing;
  params.locii.allele_frequency_min = min_af; params.locii.allele_frequency_max = max_af;
  params.algorithm = algorithm;
  params.start_seed = seed;
  auto locus_list = getLocusList(*contig_ptr, super_pop, params.locii);
  auto population = generateSyntheticPopulation(-0.5, 0.5, 0.01, super_pop, *locus_list, seed);
  std::map<int, std::shared_ptr<const ContigDB>> locus_map{{super_pop, locus_list}};
  std::map<std::string, int> sp_map;
  for (const auto& [id, g] : population->getMap()) sp_map[id] = super_pop;
  ResultsMap results = processResults(*population, contig_id, locus_map, sp_map, params);
  uint64_t i = 0;
  for (const auto& [id, r] : results) {
    if (i >= cap) break;
    syn_out[i] = generateInbreeding(id).second;
    calc_out[i] = r.inbred_allele_sum;
    ++i;
  }
  return static_cast<int64_t>(results.size());
}


// ---- VariantSort / SortedVariantAnalysis (kgo_sort.h) as text -----------------------------------------------------
// what: "ensembl" (EnsemblIndexMap, optional '\n'-separated gene list = ensemblAddIndex), "filter" (filterEnsembl of the
// list), "allele_ensembl" (alleleEnsemblMap), "id" (variantIdIndex), "genome_id" / "genome_id_mt" (variantGenomeIndex[MT]),
// "non_ensembl" (nonEnsemblIdentifiers of the full index).  One entry per line in map order, fields tab separated:
// key, HGVS_Phase of the Variant (or the ','-joined code set); the per-genome maps put the genome id first.  The caller
// frees the text with kgo_free_text.
char* kgo_variant_sort(kgo_pop* p, const char* what, const char* list) {
  if (!p || !what) return nullptr;
  std::vector<std::string> names;
  if (list) {
    std::string item;
    for (const char* c = list; ; ++c) {
      if (*c == '\n' || *c == 0) { if (!item.empty()) names.push_back(item); item.clear(); if (*c == 0) break; }
      else item += *c;
    }
  }
  std::shared_ptr<const PopulationDB> population = p->pop;
  const std::string kind(what);
  std::ostringstream out;
  if (kind == "ensembl") {
    auto index = std::make_shared<EnsemblIndexMap>();
    VariantSort::ensemblAddIndex(population, names, index);
    for (const auto& [gene, variant] : *index) out << gene << '\t' << variant->HGVS_Phase() << '\n';
  } else if (kind == "non_ensembl") {
    out << VariantSort::nonEnsemblIdentifiers(*VariantSort::ensemblIndex(population)) << '\n';
  } else if (kind == "filter") {
    for (const auto& [gene, variant] : SortedVariantAnalysis(population).filterEnsembl(names)) out << gene << '\t' << variant->HGVS_Phase() << '\n';
  } else if (kind == "allele_ensembl") {
    SortedVariantAnalysis sorted(population);
    for (const auto& [id, codes] : *sorted.alleleEnsemblMap()) {
      out << id << '\t';
      bool first = true;
      for (const auto& code : codes) { out << (first ? "" : ",") << code; first = false; }
      out << '\n';
    }
  } else if (kind == "id") {
    const auto index = VariantSort::variantIdIndex(population);
    for (const auto& [id, variant] : *index) out << id << '\t' << variant->HGVS_Phase() << '\n';
  } else if (kind == "genome_id" || kind == "genome_id_mt") {
    const auto index = kind == "genome_id" ? VariantSort::variantGenomeIndex(population) : VariantSort::variantGenomeIndexMT(population);
    for (const auto& [genome, id_map] : *index)
      for (const auto& [id, variant] : *id_map) out << genome << '\t' << id << '\t' << variant->HGVS_Phase() << '\n';
  } else {
    return nullptr;
  }
  const std::string text = out.str();
  char* result = static_cast<char*>(std::malloc(text.size() + 1));
  if (result) std::memcpy(result, text.c_str(), text.size() + 1);
  return result;
}

void kgo_free_text(char* text) { std::free(text); }

}  // extern "C"
