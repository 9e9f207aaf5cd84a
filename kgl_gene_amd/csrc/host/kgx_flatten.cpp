#include "kgx_flatten.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <thread>
#include <unordered_map>

namespace kellerberrin::genome::analysis::gpu {

std::pair<double, double> fwsBinRange(size_t bin) {
  static const double edges[FWS_FREQUENCY_ARRAY_SIZE + 1] = {0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.5, 1.0};
  if (bin >= FWS_FREQUENCY_ARRAY_SIZE) return {0.0, 0.0};
  return {edges[bin], edges[bin + 1]};
}

uint8_t fwsBinOfFrequency(float info_af) {
  if (std::isnan(info_af)) return FWS_NO_BIN;   // missing: passes lower AND upper filter -> NOT(upper) rejects it
  const double af = static_cast<double>(info_af);
  for (size_t b = 0; b < FWS_FREQUENCY_ARRAY_SIZE; ++b) {
    const auto [lo, hi] = fwsBinRange(b);
    if (af >= lo && !(af >= hi)) return static_cast<uint8_t>(b);
  }
  return FWS_NO_BIN;
}

static float infoAF(const Variant& variant) {
  // What P7FrequencyFilter reads (kgl_variant_filter_Pf7.cpp:28-44): the "AF" vector, one value per alt.
  auto info_opt = InfoEvidenceAnalysis::getTypedInfoData<std::vector<double>>(variant, "AF");
  if (!info_opt) return std::numeric_limits<float>::quiet_NaN();
  const std::vector<double>& v = info_opt.value();
  const size_t alt_count = variant.evidence().altVariantCount();
  const size_t alt_index = variant.evidence().altVariantIndex();
  if (v.size() != alt_count || v.size() <= alt_index) return std::numeric_limits<float>::infinity();   // filter errors out: in no bin
  return static_cast<float>(v[alt_index]);
}

FlatPopulation flattenPopulation(const PopulationDB& population, size_t threads) {
  FlatPopulation flat;
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  const bool trace = std::getenv("KGX_FLATTEN_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "kgx flattenPopulation: %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };

  // Row index: distinct HGVS in std::map order, first Variant seen kept (PopulationDB::uniqueVariants,
  // kgl_variant_db_population.cpp:133-161).  Shared Variant objects (one per record/alt/phase in the
  // 1000-Genomes parser) are recognised by address so that HGVS is formatted once per object, not per visit.
  std::map<std::string, std::shared_ptr<const Variant>> unique;
  std::unordered_map<const Variant*, const std::string*> seen;
  std::vector<std::shared_ptr<const GenomeDB>> genome_ptrs;
  for (const auto& [genome_id, genome_ptr] : population.getMap()) {
    flat.genome_ids.push_back(genome_id);
    genome_ptrs.push_back(genome_ptr);
  }
  {
    // The walk over every Variant pointer of every genome (1e7 visits for 1000 genomes x 20,000 variants) in parallel:
    // each worker takes a contiguous run of genomes and notes, per distinct object, where the serial walk would have met it
    // first -- (genome, visit within the genome) -- so that the merge below keeps exactly the object the serial order keeps.
    // (the object's shared_ptr is noted by address: copying it here would bump one reference count per sighting from every
    // thread at once -- 255 threads x 40,000 shared objects on the GPU box)
    struct FirstSeen { uint64_t position; const std::shared_ptr<const Variant>* variant; };
    const size_t n_genomes = genome_ptrs.size();
    // (at most 32 walkers: each of them meets every shared object again and pays its own hash-map insertions for it --
    // measured on the 256-core GPU box at 1000 x 20,000: 0.29 s with 32, 0.46 s with 255)
    const size_t walkers = std::max<size_t>(1, std::min<size_t>(std::min(threads, n_genomes), 32));
    // found[t][p]: what walker t met first, by partition p of the pointer's hash -- every walker meets every shared object
    // again, so the merge is by partition, in parallel as well (walker order within a partition = genome order)
    const size_t partitions = walkers;
    std::vector<std::vector<std::vector<std::pair<const Variant*, FirstSeen>>>> found(walkers);
    std::vector<uint64_t> visits(walkers, 0);
    auto partition_of = [partitions](const Variant* object) { return (((reinterpret_cast<uintptr_t>(object) >> 4) * 0x9E3779B97F4A7C15ull) >> 33) % partitions; };
    auto walk = [&](size_t t) {
      const size_t g_begin = n_genomes * t / walkers, g_end = n_genomes * (t + 1) / walkers;
      found[t].resize(partitions);
      std::unordered_map<const Variant*, char> mine;
      // most visits meet an object met before (one object per record, alt and phase serves every genome): a direct-mapped
      // cache of the last pointers seen answers those without touching the hash map
      constexpr size_t kCache = size_t{1} << 16;
      std::vector<const Variant*> recent(kCache, nullptr);
      for (size_t g = g_begin; g < g_end; ++g) {
        uint64_t visit = 0;
        for (const auto& [contig_id, contig_ptr] : genome_ptrs[g]->getMap())
          for (const auto& [offset, offset_ptr] : contig_ptr->getMap())
            for (const auto& variant_ptr : offset_ptr->getVariantArray()) {
              const Variant* object = variant_ptr.get();
              const Variant*& slot = recent[(reinterpret_cast<uintptr_t>(object) >> 4) & (kCache - 1)];
              if (slot != object) {
                slot = object;
                if (mine.try_emplace(object, 0).second)
                  found[t][partition_of(object)].emplace_back(object, FirstSeen{(static_cast<uint64_t>(g) << 32) | visit, &variant_ptr});
              }
              ++visit;
            }
        visits[t] += visit;
      }
    };
    {
      std::vector<std::thread> pool;
      for (size_t t = 1; t < walkers; ++t) pool.emplace_back(walk, t);
      walk(0);
      for (auto& th : pool) th.join();
    }
    for (size_t t = 0; t < walkers; ++t) flat.variant_objects += visits[t];
    std::vector<std::unordered_map<const Variant*, FirstSeen>> merged(partitions);
    auto merge = [&](size_t p) {
      for (size_t t = 0; t < walkers; ++t)                     // walkers are in genome order: an earlier walker's sighting stands
        for (const auto& [object, first] : found[t][p]) merged[p].try_emplace(object, first);
    };
    {
      std::vector<std::thread> pool;
      for (size_t p = 1; p < partitions; ++p) pool.emplace_back(merge, p);
      merge(0);
      for (auto& th : pool) th.join();
    }
    size_t distinct = 0;
    for (const auto& part : merged) distinct += part.size();
    std::vector<const FirstSeen*> in_order;
    in_order.reserve(distinct);
    for (const auto& part : merged)
      for (const auto& [ptr, first] : part) in_order.push_back(&first);
    std::sort(in_order.begin(), in_order.end(), [](const FirstSeen* a, const FirstSeen* b) { return a->position < b->position; });
    for (const FirstSeen* first : in_order) {
      auto it = unique.try_emplace((*first->variant)->HGVS(), *first->variant).first;
      seen.emplace(first->variant->get(), &it->first);
    }
  }
  lap("distinct variants found");
  std::unordered_map<const Variant*, uint32_t> row_of;
  row_of.reserve(seen.size());
  {
    std::unordered_map<const std::string*, uint32_t> index_of_key;
    index_of_key.reserve(unique.size());
    flat.rows.reserve(unique.size());
    uint32_t index = 0;
    for (const auto& [hgvs, variant_ptr] : unique) {
      VariantRow row;
      row.hgvs = hgvs;
      row.contig = variant_ptr->contigId();
      row.offset = variant_ptr->offset();
      row.is_snp = variant_ptr->isSNP();
      row.info_af = infoAF(*variant_ptr);
      row.variant = variant_ptr;
      flat.rows.push_back(std::move(row));
      index_of_key.emplace(&hgvs, index++);
    }
    for (const auto& [ptr, key] : seen) row_of.emplace(ptr, index_of_key.at(key));
  }

  flat.primary_rows = flat.rows.size();
  // Rows whose Variant objects come from records in different FWS bins get one split row per bin (see VariantRow).
  std::unordered_map<const Variant*, uint32_t> split_row_of;
  {
    std::vector<std::vector<const Variant*>> objects_of_row(flat.primary_rows);
    for (const auto& [ptr, r] : row_of) objects_of_row[r].push_back(ptr);
    for (uint32_t r = 0; r < flat.primary_rows; ++r) {
      if (objects_of_row[r].size() < 2) continue;
      std::map<uint8_t, std::vector<const Variant*>> by_bin;
      for (const Variant* ptr : objects_of_row[r]) by_bin[fwsBinOfFrequency(infoAF(*ptr))].push_back(ptr);
      if (by_bin.size() < 2) continue;
      flat.rows[r].fws_from_splits = true;
      for (const auto& [bin, objects] : by_bin) {
        if (bin == FWS_NO_BIN) continue;                  // copies of a record without a usable AF are in no bin's population
        VariantRow split = flat.rows[r];
        split.fws_from_splits = false;
        split.split_of = r;
        split.info_af = infoAF(*objects.front());
        const uint32_t index = static_cast<uint32_t>(flat.rows.size());
        flat.rows.push_back(std::move(split));
        for (const Variant* ptr : objects) split_row_of.emplace(ptr, index);
      }
    }
  }

  lap("rows indexed");
  const size_t G = flat.genome_ids.size();
  const size_t V = flat.rows.size();
  flat.row_bytes = (G + 3) / 4;
  flat.packed.assign(V * flat.row_bytes, 0);

  // Dosage = number of Variant objects with that HGVS in the genome (kgl_variant_db_variant.cpp:103).
  // Workers own whole bytes (4 consecutive genomes), so no two threads touch the same byte.
  const std::vector<std::shared_ptr<const GenomeDB>>& genomes = genome_ptrs;
  // The phase plane beside the rows (UniquePhasedFilter counts one object per distinct HGVS AND phase,
  // kgl_variant_filter_db_offset.cpp:160-181): one bit per cell, set where the genome's copies of the variant carry more than
  // one distinct phase (a|a in phased data; never in unphased data).  Workers own whole plane bytes = 8 consecutive genomes
  // = two bytes of every packed row, so no two threads touch the same byte of either.
  flat.phase_row_bytes = (G + 7) / 8;
  flat.phase_plane.assign(V * flat.phase_row_bytes, 0);
  const size_t octets = (G + 7) / 8;
  // (every packing worker keeps 3 bytes per row: at most 32 of them, as the discovery walk has)
  threads = std::max<size_t>(1, std::min<size_t>(std::min(threads, octets), 32));
  std::vector<std::vector<NonDiploidCell>> overflow(threads);
  std::vector<uint8_t> any_two_phases(threads, 0);
  std::vector<size_t> three_phases(threads, 0);
  auto phase_bit = [](VariantPhase phase) -> uint8_t {
    switch (phase) {
      case VariantPhase::DIPLOID_PHASE_A: return 1;
      case VariantPhase::DIPLOID_PHASE_B: return 2;
      case VariantPhase::UNPHASED: return 4;
      default: return 8;                                          // HAPLOID_PHASED
    }
  };
  auto worker = [&](size_t t) {
    std::vector<uint32_t> touched;
    std::vector<uint16_t> count(V, 0);
    std::vector<uint8_t> phases(V, 0);
    for (size_t o = t; o < octets; o += threads) {
      for (size_t j = 0; j < 8 && o * 8 + j < G; ++j) {
        const size_t g = o * 8 + j;
        touched.clear();
        for (const auto& [contig_id, contig_ptr] : genomes[g]->getMap())
          for (const auto& [offset, offset_ptr] : contig_ptr->getMap())
            for (const auto& variant_ptr : offset_ptr->getVariantArray()) {
              const uint32_t r = row_of.at(variant_ptr.get());
              if (count[r]++ == 0) touched.push_back(r);
              phases[r] |= phase_bit(variant_ptr->phaseId());
              if (!split_row_of.empty()) {
                auto split = split_row_of.find(variant_ptr.get());
                if (split != split_row_of.end()) {
                  if (count[split->second]++ == 0) touched.push_back(split->second);
                  phases[split->second] |= phase_bit(variant_ptr->phaseId());
                }
              }
            }
        for (uint32_t r : touched) {
          const uint32_t d = count[r];
          const uint8_t seen_phases = phases[r];
          count[r] = 0;
          phases[r] = 0;
          flat.packed[static_cast<size_t>(r) * flat.row_bytes + o * 2 + j / 4] |= static_cast<uint8_t>((d > 2 ? 3u : d) << (2 * (j % 4)));
          if (seen_phases & (seen_phases - 1)) {                  // more than one distinct phase among the copies
            flat.phase_plane[static_cast<size_t>(r) * flat.phase_row_bytes + o] |= static_cast<uint8_t>(1u << j);
            any_two_phases[t] = 1;
            if (__builtin_popcount(seen_phases) > 2) ++three_phases[t];
          }
          if (d > 2 && r < flat.primary_rows) overflow[t].push_back({r, static_cast<uint32_t>(g), d});
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (size_t t = 1; t < threads; ++t) pool.emplace_back(worker, t);
  worker(0);
  for (auto& th : pool) th.join();
  lap("rows packed");
  for (auto& o : overflow) flat.non_diploid.insert(flat.non_diploid.end(), o.begin(), o.end());
  bool two_phases = false;
  for (const uint8_t flag : any_two_phases) two_phases = two_phases || flag;
  for (const size_t n : three_phases) flat.cells_with_three_phases += n;
  if (!two_phases) { flat.phase_plane.clear(); flat.phase_plane.shrink_to_fit(); }    // unphased data: no plane
  std::sort(flat.non_diploid.begin(), flat.non_diploid.end(), [](const NonDiploidCell& a, const NonDiploidCell& b) {
    return a.row != b.row ? a.row < b.row : a.genome < b.genome;
  });
  return flat;
}

}  // namespace kellerberrin::genome::analysis::gpu
