// VCF text -> FlatPopulation without building Variant / PopulationDB objects (SURVEY.md §8f "next" #1).
//
// Behaviour to match: the reference's phased-diploid parser, i.e. what a PopulationDB filled by
//   ParseVCF::moveToVcfRecord            (kgl_genomics/kgl_parser/kgl_variant_vcf_impl.cpp:96-170)
//   Genome1000VCFImpl::ParseRecord       (kgl_parser/kgl_variant_factory_1000_impl.cpp:63-145)
//   Genome1000VCFImpl::alternateIndex    (:148-272)
// would flatten to through flattenPopulation(): genomes exist only if they carry a variant, variants only if some
// genome carries them, rows in lexicographic HGVS order, dosage = copies of the HGVS in the genome.
// Records are independent, so they are parsed by a pool of threads straight into 2-bit rows.
//
// flattenVcfPf is the same for the reference's unphased P. falciparum (Pf7) parser,
//   PfVCFImpl::ParseRecord / setupPopulationStructure   (kgl_parser/kgl_variant_factory_pf_impl.cpp:73-422)
//   Variant::canonicalSequences                         (kgl_variant_db/kgl_variant_db.cpp:165-220)
// optionally followed by the per-record quality filter the PfEMP package applies before it counts,
//   P7VariantFilter::applyFilter                        (kgl_variant_filter/kgl_variant_filter_Pf7.cpp:131-318,
//                                                        called from kga_analysis_lib_PfFilter.cpp:63-67).
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "kgx_flatten.h"
#include "kgx_variant_sort.h"
#include "kgx_vcf_io.h"

namespace kellerberrin::genome::analysis::gpu {

namespace {

std::vector<std::string_view> split(std::string_view s, char delim, size_t reserve = 0) {
  std::vector<std::string_view> out;
  if (reserve) out.reserve(reserve);
  size_t begin = 0;
  for (size_t i = 0; i < s.size(); ++i)
    if (s[i] == delim) { out.push_back(s.substr(begin, i - begin)); begin = i + 1; }
  out.push_back(s.substr(begin));
  return out;
}

std::string_view trim(std::string_view s) {
  while (!s.empty() && std::isspace(static_cast<unsigned char>(s.front()))) s.remove_prefix(1);
  while (!s.empty() && std::isspace(static_cast<unsigned char>(s.back()))) s.remove_suffix(1);
  return s;
}

// std::stoul as the reference uses it: leading white space, optional sign, then at least one digit; trailing text is
// ignored.  ok = false where std::stoul would throw (no digits, or overflow); a leading '-' wraps as strtoul does.
uint64_t parseIndex(std::string_view s, bool& ok) {
  size_t i = 0;
  while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) ++i;
  bool negative = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) { negative = s[i] == '-'; ++i; }
  if (i >= s.size() || !std::isdigit(static_cast<unsigned char>(s[i]))) { ok = false; return 0; }
  uint64_t v = 0;
  for (; i < s.size() && std::isdigit(static_cast<unsigned char>(s[i])); ++i) {
    const uint64_t d = static_cast<uint64_t>(s[i] - '0');
    if (v > (std::numeric_limits<uint64_t>::max() - d) / 10) { ok = false; return 0; }   // out_of_range
    v = v * 10 + d;
  }
  return negative ? (0 - v) : v;
}

enum class Chromosome { Autosome, X, Y };

Chromosome chromosomeOf(std::string_view contig) {
  if (contig == "X" || contig == "chrX") return Chromosome::X;
  if (contig == "Y" || contig == "chrY") return Chromosome::Y;
  return Chromosome::Autosome;
}

// The decision table of Genome1000VCFImpl::alternateIndex for one sample column.
void phasedAlleles(std::string_view genotype, size_t n_alt, Chromosome chrom, uint32_t& a, uint32_t& b) {
  a = b = 0;
  // Fast path for the overwhelmingly common token "d|d" (optionally followed by ":..."): same result as the general
  // decision table below.
  if (genotype.size() >= 3 && genotype[1] == '|' && (genotype.size() == 3 || genotype[3] == ':') &&
      genotype[0] >= '0' && genotype[0] <= '9' && genotype[2] >= '0' && genotype[2] <= '9') {
    const uint32_t pa = static_cast<uint32_t>(genotype[0] - '0'), pb = static_cast<uint32_t>(genotype[2] - '0');
    if (pa <= n_alt && pb <= n_alt) { a = pa; b = pb; }
    return;
  }
  genotype = trim(genotype);
  if (genotype.empty()) return;
  const std::string_view gt = genotype.substr(0, genotype.find(':'));
  const size_t bar = gt.find('|');
  uint64_t pa = 0, pb = 0;
  bool ok = true;
  if (bar == std::string_view::npos) {                 // one phase only: X / Y of a male, anything else stays reference
    if (gt != "." && gt != "-") {
      if (chrom == Chromosome::X) pa = parseIndex(gt, ok);
      else if (chrom == Chromosome::Y) pb = parseIndex(gt, ok);
    }
  } else {
    const std::string_view first = gt.substr(0, bar);
    std::string_view second = gt.substr(bar + 1);
    second = second.substr(0, second.find('|'));       // further phases are ignored
    if (first.find('<') == std::string_view::npos && first != "." && first != "-") pa = parseIndex(first, ok);
    // the reference tests the FIRST phase against "-" before reading the second
    if (ok && second.find('<') == std::string_view::npos && second != "." && first != "-") pb = parseIndex(second, ok);
  }
  if (!ok || pa > n_alt || pb > n_alt) return;          // conversion failure or index past the alt list: all reference
  a = static_cast<uint32_t>(pa);
  b = static_cast<uint32_t>(pb);
}

float infoFloat(std::string_view value) {
  // VCFInfoParser::convertToFloat (kgl_variant_factory_vcf_parse_info.cpp:205-260): "." / nan are missing, std::stof otherwise.
  if (value == ".") return std::numeric_limits<float>::quiet_NaN();
  std::string s(value);
  std::string upper = s;
  for (auto& c : upper) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  if (upper == "NAN") return std::numeric_limits<float>::quiet_NaN();
  try {
    return std::stof(s);
  } catch (const std::out_of_range&) {
    if (upper.find("E-") != std::string::npos) return std::numeric_limits<float>::min();
    if (upper.find('E') != std::string::npos) return std::numeric_limits<float>::max();
    return std::numeric_limits<float>::quiet_NaN();
  } catch (...) {
    return std::numeric_limits<float>::quiet_NaN();
  }
}

bool isSnp(std::string_view ref, std::string_view alt) {   // Variant::isSNP (kgl_variant_db.cpp:121-158)
  if (ref.size() == 1 && alt.size() == 1) return true;
  if (ref.size() != alt.size()) return false;
  bool diff = false;
  for (size_t i = 0; i < ref.size(); ++i)
    if (ref[i] != alt[i]) { if (diff) return false; diff = true; }
  return true;
}

struct RecordRows {            // what one VCF record contributes
  std::vector<VariantRow> rows;          // one per alt
  std::vector<uint8_t> copies;           // [n_alt][ceil(n_samples / 4)] copies of the alt in the sample (0..2), 2 bits each
};
// A sample column calls at most two alleles, so a 2-bit field never carries into its neighbour.
inline size_t copyRowBytes(size_t n_samples) { return (n_samples + 3) / 4; }
inline void addCopy(uint8_t* copies, size_t row_bytes, size_t alt, size_t sample) {
  copies[alt * row_bytes + (sample >> 2)] = static_cast<uint8_t>(copies[alt * row_bytes + (sample >> 2)] + (1u << (2 * (sample & 3))));
}
inline uint32_t copyAt(const uint8_t* row, size_t sample) { return (row[sample >> 2] >> (2 * (sample & 3))) & 3u; }

struct VcfLines {
  std::vector<std::string> samples;            // #CHROM columns 10..
  std::vector<std::string> contigs;            // ##contig=<ID=...> header lines, in file order
  std::vector<std::string_view> records;       // every non-empty line not starting with '#'
};

VcfLines scanLines(std::string_view text) {
  VcfLines out;
  for (size_t begin = 0; begin <= text.size();) {
    size_t end = text.find('\n', begin);
    if (end == std::string_view::npos) end = text.size();
    std::string_view line = text.substr(begin, end - begin);
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (!line.empty()) {
      if (line[0] == '#') {
        if (line.rfind("#CHROM", 0) == 0) {
          const auto f = split(line, '\t');
          for (size_t i = 9; i < f.size(); ++i) out.samples.emplace_back(f[i]);
        } else if (line.rfind("##contig=<", 0) == 0) {
          const size_t id = line.find("ID=");
          if (id != std::string_view::npos) {
            size_t stop = line.find_first_of(",>", id);
            if (stop == std::string_view::npos) stop = line.size();
            out.contigs.emplace_back(line.substr(id + 3, stop - id - 3));
          }
        }
      } else {
        out.records.push_back(line);
      }
    }
    begin = end + 1;
  }
  return out;
}

template <typename ParseRecord>
std::vector<RecordRows> parseRecords(const VcfLines& lines, size_t threads, ParseRecord parse) {
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  std::vector<RecordRows> parsed(lines.records.size());
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (size_t r = next.fetch_add(1); r < lines.records.size(); r = next.fetch_add(1)) parse(lines.records[r], parsed[r]);
  };
  const size_t n = std::max<size_t>(1, std::min(threads, lines.records.size()));
  std::vector<std::thread> pool;
  for (size_t t = 1; t < n; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  return parsed;
}

// Run fn(begin, end) over [0, n) in chunks on `threads` threads.
template <typename Fn>
void parallelChunks(size_t n, size_t chunk, size_t threads, Fn fn) {
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (size_t begin = next.fetch_add(chunk); begin < n; begin = next.fetch_add(chunk)) fn(begin, std::min(n, begin + chunk));
  };
  const size_t workers = std::max<size_t>(1, std::min(threads, (n + chunk - 1) / chunk));
  std::vector<std::thread> pool;
  for (size_t t = 1; t < workers; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
}

FlatPopulation mergeRecords(const std::vector<RecordRows>& parsed, const std::vector<std::string>& samples, bool every_sample, size_t threads = 0,
                            RowSink* sink = nullptr) {
  const size_t S = samples.size();
  if (S == 0) return FlatPopulation{};              // no #CHROM line / no sample columns: no genomes, hence no variants
  struct Key { const std::string* hgvs; uint32_t record, alt; };
  std::vector<Key> keys;
  for (uint32_t r = 0; r < parsed.size(); ++r)
    for (uint32_t a = 0; a < parsed[r].rows.size(); ++a) keys.push_back({&parsed[r].rows[a].hgvs, r, a});
  std::stable_sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return *x.hgvs < *y.hgvs; });

  // Genomes: every sample (the Pf parser creates them up front) or only carriers (the 1000-Genomes parser creates a
  // genome when it first adds a variant to it); std::map order.
  const size_t SB = copyRowBytes(S);
  std::vector<uint8_t> carries(S, every_sample ? 1 : 0);
  if (!every_sample) {
    std::mutex merge_mutex;
    std::vector<uint8_t> any_copy(SB, 0);             // the packed rows OR-ed together: a field is non-zero iff the sample carries something
    parallelChunks(parsed.size(), 64, threads, [&](size_t begin, size_t end) {
      std::vector<uint8_t> local(SB, 0);
      for (size_t r = begin; r < end; ++r) {
        const auto& copies = parsed[r].copies;
        for (size_t i = 0, b = 0; i < copies.size(); ++i, b = (b + 1 == SB ? 0 : b + 1)) local[b] |= copies[i];
      }
      std::lock_guard<std::mutex> lock(merge_mutex);
      for (size_t b = 0; b < SB; ++b) any_copy[b] |= local[b];
    });
    for (size_t smp = 0; smp < S; ++smp) carries[smp] = copyAt(any_copy.data(), smp) ? 1 : 0;
  }
  std::vector<uint32_t> sample_order;
  for (uint32_t s = 0; s < S; ++s) if (carries[s]) sample_order.push_back(s);
  std::sort(sample_order.begin(), sample_order.end(), [&](uint32_t x, uint32_t y) { return samples[x] < samples[y]; });
  // a sample named twice is one genome (std::map): its columns add up
  FlatPopulation flat;
  std::vector<std::vector<uint32_t>> columns_of;      // genome -> sample columns
  for (uint32_t s : sample_order) {
    if (!flat.genome_ids.empty() && flat.genome_ids.back() == samples[s]) { columns_of.back().push_back(s); continue; }
    flat.genome_ids.push_back(samples[s]);
    columns_of.push_back({s});
  }
  const size_t G = flat.genome_ids.size();
  flat.row_bytes = (G + 3) / 4;

  // Variants: one row per distinct HGVS, in lexicographic order; records repeating an HGVS add their copies.
  std::vector<size_t> group_begin;
  for (size_t k = 0; k < keys.size(); ++k)
    if (k == 0 || *keys[k].hgvs != *keys[k - 1].hgvs) group_begin.push_back(k);
  const size_t n_groups = group_begin.size();
  group_begin.push_back(keys.size());
  auto copiesOf = [&](size_t m) { return &parsed[keys[m].record].copies[static_cast<size_t>(keys[m].alt) * SB]; };

  // Phase A, metadata only: which groups anybody carries (a variant nobody carries never reaches the PopulationDB), the
  // record that stands for each (the Variant kept for an HGVS is the first one added, uniqueVariants: the first record
  // with a carrier), and the groups whose records fall in different FWS bins (one split row per bin, holding that bin's
  // copies only).  After it every row has its place, so phase B can hand rows out as they are packed.
  std::vector<uint8_t> has_carrier(keys.size(), 0);
  parallelChunks(keys.size(), 4096, threads, [&](size_t begin, size_t end) {
    for (size_t m = begin; m < end; ++m) {
      const uint8_t* c = copiesOf(m);
      has_carrier[m] = std::any_of(c, c + SB, [](uint8_t x) { return x != 0; }) ? 1 : 0;
    }
  });
  struct SplitPlan { size_t group; uint8_t bin; size_t key; size_t row; };
  std::vector<SplitPlan> split_plan;
  std::vector<int64_t> split_begin(n_groups, -1);     // first entry of the group in split_plan
  std::vector<uint8_t> carried(n_groups, 0);
  std::vector<size_t> first_key(n_groups, 0);
  std::vector<uint32_t> row_of_group(n_groups, 0);
  size_t n_rows = 0;
  auto binOfKey = [&](size_t m) { return fwsBinOfFrequency(parsed[keys[m].record].rows[keys[m].alt].info_af); };
  for (size_t grp = 0; grp < n_groups; ++grp) {
    const size_t k = group_begin[grp], e = group_begin[grp + 1];
    size_t first = e;
    for (size_t m = k; m < e; ++m)
      if (has_carrier[m]) { first = m; break; }
    if (first == e) continue;
    carried[grp] = 1;
    first_key[grp] = first;
    row_of_group[grp] = static_cast<uint32_t>(n_rows++);
    if (e - k > 1) {
      std::map<uint8_t, size_t> first_of_bin;
      for (size_t m = k; m < e; ++m)
        if (has_carrier[m]) first_of_bin.try_emplace(binOfKey(m), m);
      if (first_of_bin.size() > 1) {
        carried[grp] = 2;                               // primary row's bin counts come from its splits
        for (const auto& [bin, m] : first_of_bin) {
          if (bin == FWS_NO_BIN) continue;
          if (split_begin[grp] < 0) split_begin[grp] = static_cast<int64_t>(split_plan.size());
          split_plan.push_back({grp, bin, m, 0});
        }
      }
    }
  }
  for (size_t i = 0; i < split_plan.size(); ++i) split_plan[i].row = n_rows + i;        // split rows follow the primary rows
  flat.rows.reserve(n_rows + split_plan.size());
  for (size_t grp = 0; grp < n_groups; ++grp) {
    if (!carried[grp]) continue;
    flat.rows.push_back(parsed[keys[first_key[grp]].record].rows[keys[first_key[grp]].alt]);
    flat.rows.back().fws_from_splits = carried[grp] == 2;
  }
  flat.primary_rows = flat.rows.size();
  for (const SplitPlan& split : split_plan) {
    VariantRow row = parsed[keys[split.key].record].rows[keys[split.key].alt];
    row.split_of = row_of_group[split.group];
    flat.rows.push_back(std::move(row));
  }

  // Phase B: the rows themselves, a block of groups at a time on every thread; a block's primary rows are consecutive
  // device rows and leave for the sink at once (default: FlatPopulation::packed), so no second copy of the population
  // is ever assembled here.
  struct PackedSink final : RowSink {
    FlatPopulation* flat{nullptr};
    bool begin(const FlatPopulation& meta) override { flat->packed.assign(meta.rows.size() * meta.row_bytes, 0); return true; }
    void rows(uint64_t first_row, uint64_t count, const uint8_t* data) override {
      if (count && flat->row_bytes) std::memcpy(&flat->packed[first_row * flat->row_bytes], data, count * flat->row_bytes);
    }
  } packed_sink;
  packed_sink.flat = &flat;
  RowSink* out = sink ? sink : &packed_sink;
  if (!out->begin(flat)) return flat;
  constexpr size_t kGroupsPerBlock = 1024;
  const size_t n_blocks = (n_groups + kGroupsPerBlock - 1) / kGroupsPerBlock;
  struct Wide { size_t group; uint32_t genome, dosage; };
  std::vector<std::vector<Wide>> wide_of_block(n_blocks);
  std::vector<size_t> objects_of_block(n_blocks, 0);
  parallelChunks(n_groups, kGroupsPerBlock, threads, [&](size_t begin, size_t end) {
    std::vector<uint32_t> total(S);
    std::vector<uint8_t> block_rows, split_row(flat.row_bytes);
    size_t first_row = 0, block_count = 0;
    auto packTotals = [&](uint8_t* row, size_t grp, bool primary) {
      for (size_t g = 0; g < G; ++g) {
        uint32_t d = 0;
        for (uint32_t column : columns_of[g]) d += total[column];
        row[g / 4] |= static_cast<uint8_t>((d > 2 ? 3u : d) << (2 * (g % 4)));
        if (primary) {
          objects_of_block[begin / kGroupsPerBlock] += d;
          if (d > 2) wide_of_block[begin / kGroupsPerBlock].push_back({grp, static_cast<uint32_t>(g), d});
        }
      }
    };
    for (size_t grp = begin; grp < end; ++grp) {
      if (!carried[grp]) continue;
      const size_t k = group_begin[grp], e = group_begin[grp + 1];
      if (block_count == 0) first_row = row_of_group[grp];
      if (carried[grp] == 2) {
        for (int64_t i = split_begin[grp]; i >= 0 && static_cast<size_t>(i) < split_plan.size() && split_plan[i].group == grp; ++i) {
          std::fill(total.begin(), total.end(), 0u);
          for (size_t m = k; m < e; ++m) {
            if (!has_carrier[m] || binOfKey(m) != split_plan[i].bin) continue;
            const uint8_t* c = copiesOf(m);
            for (size_t smp = 0; smp < S; ++smp) total[smp] += copyAt(c, smp);
          }
          std::fill(split_row.begin(), split_row.end(), static_cast<uint8_t>(0));
          packTotals(split_row.data(), grp, false);
          out->rows(split_plan[i].row, 1, split_row.data());
        }
      }
      std::fill(total.begin(), total.end(), 0u);
      for (size_t m = k; m < e; ++m) {
        if (!has_carrier[m]) continue;
        const uint8_t* c = copiesOf(m);
        for (size_t smp = 0; smp < S; ++smp) total[smp] += copyAt(c, smp);
      }
      block_rows.resize((block_count + 1) * flat.row_bytes, 0);
      packTotals(block_rows.data() + block_count * flat.row_bytes, grp, true);
      ++block_count;
    }
    if (block_count) out->rows(first_row, block_count, block_rows.data());
  });
  for (size_t block = 0; block < n_blocks; ++block) {
    flat.variant_objects += objects_of_block[block];
    for (const Wide& w : wide_of_block[block]) flat.non_diploid.push_back({row_of_group[w.group], w.genome, w.dosage});
  }
  return flat;
}

// The streaming counterpart of mergeRecords (kgx_flatten.h: StreamSink): consume() takes one piece's parsed records, gives
// every variant seen for the first time the next row, packs those rows (genome order = sorted sample names) and writes
// them out; a variant seen before is kept aside ("late") and merged into its row by finish(), exactly as mergeRecords adds
// the copies of repeated records -- including the per-bin split rows when the records' AF fall in different FWS bins.
class StreamMerger {
 public:
  StreamMerger(StreamSink& sink, bool every_sample, size_t threads) : sink_(sink), every_sample_(every_sample), threads_(threads) {}
  bool two_phase{false};          // the file cannot be taken this way (why_ says why): not an error
  std::string error;              // sink failure

  bool consume(const std::vector<std::string>& samples, const std::vector<RecordRows>& parsed) {
    if (!error.empty() || two_phase) return false;
    if (!opened_) {
      if (samples.empty()) return true;                   // no #CHROM line yet (or no sample columns at all): nothing to place
      if (!open(samples)) return false;
    }
    const size_t S = column_of_sample_.size(), SB = copyRowBytes(S);
    struct Item { uint32_t record, alt; uint64_t row; bool late; };
    std::vector<Item> items;
    const uint64_t first_new = meta_.size();
    for (uint32_t r = 0; r < parsed.size(); ++r)
      for (uint32_t a = 0; a < parsed[r].rows.size(); ++a) {
        const uint8_t* c = &parsed[r].copies[static_cast<size_t>(a) * SB];
        if (!std::any_of(c, c + SB, [](uint8_t x) { return x != 0; })) continue;        // a variant nobody carries never reaches the PopulationDB
        const VariantRow& row = parsed[r].rows[a];
        auto [found, is_new] = row_of_hgvs_.try_emplace(row.hgvs, meta_.size());
        if (is_new) meta_.push_back(row);
        items.push_back({r, a, found->second, !is_new});
      }
    const uint64_t n_new = meta_.size() - first_new;
    std::vector<uint8_t> block(n_new * row_bytes_, 0);
    std::vector<std::vector<uint8_t>> late_rows(items.size());
    std::vector<size_t> objects(items.size(), 0);
    std::vector<uint8_t> carried_here(S, 0);
    std::mutex carried_mutex;
    parallelChunks(items.size(), 256, threads_, [&](size_t begin, size_t end) {
      std::vector<uint8_t> local(S, 0);
      for (size_t k = begin; k < end; ++k) {
        const Item& item = items[k];
        const uint8_t* c = &parsed[item.record].copies[static_cast<size_t>(item.alt) * SB];
        uint8_t* out;
        if (item.late) { late_rows[k].assign(row_bytes_, 0); out = late_rows[k].data(); }
        else out = &block[(item.row - first_new) * row_bytes_];
        for (size_t smp = 0; smp < S; ++smp) {
          const uint32_t d = copyAt(c, smp);
          if (!d) continue;
          local[smp] = 1;
          objects[k] += d;
          const uint32_t g = column_of_sample_[smp];
          out[g / 4] = static_cast<uint8_t>(out[g / 4] | (d << (2 * (g % 4))));        // one sample per genome here: d <= 2, no carry
        }
      }
      std::lock_guard<std::mutex> lock(carried_mutex);
      for (size_t smp = 0; smp < S; ++smp) carried_here[smp] |= local[smp];
    });
    for (size_t smp = 0; smp < S; ++smp) carries_[smp] |= carried_here[smp];
    for (size_t k = 0; k < items.size(); ++k) {
      variant_objects_ += objects[k];
      if (items[k].late) late_.push_back({items[k].row, parsed[items[k].record].rows[items[k].alt], std::move(late_rows[k])});
    }
    if (n_new && !sink_.write(first_new, n_new, block.data())) { error = "the row sink failed while taking a piece's rows"; return false; }
    return true;
  }

  bool finish(FlatPopulation& flat) {
    flat = FlatPopulation{};
    if (!error.empty() || two_phase) return false;
    if (!opened_) return sink_.open(0, 0) && sink_.close(0);            // no samples: no genomes, hence no variants
    const size_t S = column_of_sample_.size();
    if (!every_sample_)
      for (size_t smp = 0; smp < S; ++smp)
        if (!carries_[smp]) return giveUp("sample " + sample_names_[smp] + " carries no variant: it is no genome, and the rows are one column too wide");
    flat.genome_ids = genome_ids_;
    flat.row_bytes = row_bytes_;
    flat.primary_rows = meta_.size();
    // variants met again: merge into the row of their first appearance, in that row's order
    std::stable_sort(late_.begin(), late_.end(), [](const Late& x, const Late& y) { return x.row < y.row; });
    std::vector<uint8_t> first(row_bytes_), merged(row_bytes_), split_row(row_bytes_);
    std::vector<uint32_t> total(genome_ids_.size());
    std::vector<VariantRow> split_meta;
    std::vector<std::vector<uint8_t>> split_rows;
    auto addCodes = [&](const uint8_t* row) { for (size_t g = 0; g < total.size(); ++g) total[g] += (row[g / 4] >> (2 * (g % 4))) & 3u; };
    auto packTotals = [&](uint8_t* row) {
      std::fill(row, row + row_bytes_, static_cast<uint8_t>(0));
      for (size_t g = 0; g < total.size(); ++g) row[g / 4] = static_cast<uint8_t>(row[g / 4] | ((total[g] > 2 ? 3u : total[g]) << (2 * (g % 4))));
    };
    for (size_t k = 0; k < late_.size();) {
      size_t e = k;
      while (e < late_.size() && late_[e].row == late_[k].row) ++e;
      const uint64_t row = late_[k].row;
      if (!sink_.read(row, first.data())) { error = "the row sink failed to give a row back"; return false; }
      std::fill(total.begin(), total.end(), 0u);
      addCodes(first.data());
      for (size_t m = k; m < e; ++m) addCodes(late_[m].packed.data());
      packTotals(merged.data());
      for (size_t g = 0; g < total.size(); ++g)
        if (total[g] > 2) flat.non_diploid.push_back({static_cast<uint32_t>(row), static_cast<uint32_t>(g), total[g]});
      if (!sink_.write(row, 1, merged.data())) { error = "the row sink failed while taking a merged row"; return false; }
      // records in different FWS bins: one split row per bin, holding that bin's copies only
      std::map<uint8_t, std::vector<const uint8_t*>> by_bin;
      std::map<uint8_t, const VariantRow*> first_of_bin;
      by_bin[fwsBinOfFrequency(meta_[row].info_af)].push_back(first.data());
      first_of_bin.try_emplace(fwsBinOfFrequency(meta_[row].info_af), &meta_[row]);
      for (size_t m = k; m < e; ++m) {
        const uint8_t bin = fwsBinOfFrequency(late_[m].meta.info_af);
        by_bin[bin].push_back(late_[m].packed.data());
        first_of_bin.try_emplace(bin, &late_[m].meta);
      }
      if (by_bin.size() > 1) {
        meta_[row].fws_from_splits = true;                 // primary row's bin counts come from its splits
        for (const auto& [bin, members] : by_bin) {
          if (bin == FWS_NO_BIN) continue;
          std::fill(total.begin(), total.end(), 0u);
          for (const uint8_t* member : members) addCodes(member);
          packTotals(split_row.data());
          VariantRow split = *first_of_bin.at(bin);
          split.fws_from_splits = false;
          split.split_of = static_cast<int64_t>(row);
          split_meta.push_back(std::move(split));
          split_rows.push_back(split_row);
        }
      }
      k = e;
    }
    std::sort(flat.non_diploid.begin(), flat.non_diploid.end(), [](const NonDiploidCell& x, const NonDiploidCell& y) {
      return x.row != y.row ? x.row < y.row : x.genome < y.genome;
    });
    const uint64_t n_primary = meta_.size();
    for (size_t i = 0; i < split_rows.size(); ++i)
      if (!sink_.write(n_primary + i, 1, split_rows[i].data())) { error = "the row sink failed while taking a split row"; return false; }
    flat.rows = std::move(meta_);
    flat.rows.insert(flat.rows.end(), std::make_move_iterator(split_meta.begin()), std::make_move_iterator(split_meta.end()));
    flat.variant_objects = variant_objects_;
    if (!sink_.close(flat.rows.size())) { error = "the row sink failed to close"; return false; }
    return true;
  }

 private:
  struct Late { uint64_t row; VariantRow meta; std::vector<uint8_t> packed; };

  bool giveUp(const std::string& why) { two_phase = true; why_ = why; return false; }
  bool open(const std::vector<std::string>& samples) {
    const size_t S = samples.size();
    sample_names_ = samples;
    std::vector<uint32_t> order(S);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return samples[x] < samples[y]; });
    column_of_sample_.assign(S, 0);
    for (size_t rank = 0; rank < S; ++rank) {
      if (rank && samples[order[rank]] == samples[order[rank - 1]]) return giveUp("sample " + samples[order[rank]] + " is named twice: its columns add up into one genome");
      column_of_sample_[order[rank]] = static_cast<uint32_t>(rank);
      genome_ids_.push_back(samples[order[rank]]);
    }
    carries_.assign(S, every_sample_ ? 1 : 0);
    row_bytes_ = (S + 3) / 4;
    opened_ = true;
    if (!sink_.open(S, row_bytes_)) { error = "the row sink failed to open"; return false; }
    return true;
  }

  StreamSink& sink_;
  bool every_sample_;
  size_t threads_;
  bool opened_{false};
  std::string why_;
  std::vector<std::string> sample_names_;
  std::vector<GenomeId_t> genome_ids_;
  std::vector<uint32_t> column_of_sample_;
  std::vector<uint8_t> carries_;
  uint64_t row_bytes_{0};
  std::unordered_map<std::string, uint64_t> row_of_hgvs_;
  std::vector<VariantRow> meta_;
  std::vector<Late> late_;
  size_t variant_objects_{0};

 public:
  [[nodiscard]] const std::string& why() const { return why_; }
};

// "AF" of the INFO column: one value per alt; a size mismatch is flagged with +inf (P7FrequencyFilter errors out: in no bin).
void readInfoAf(std::string_view info, size_t A, std::vector<float>& af, bool& af_bad_size) {
  af.assign(A, std::numeric_limits<float>::quiet_NaN());
  af_bad_size = false;
  for (const auto item : split(info, ';')) {
    if (item.size() > 3 && item.substr(0, 3) == "AF=") {
      const auto values = split(item.substr(3), ',');
      if (values.size() == A) for (size_t a = 0; a < A; ++a) af[a] = infoFloat(values[a]);
      else af_bad_size = true;
    }
  }
}

}  // namespace

namespace {

// Append b's records to a (a run of whole lines at a time: the text of a piece is gone once its records are parsed).
void appendRecords(std::vector<RecordRows>& a, std::vector<RecordRows>&& b) {
  if (a.empty()) { a = std::move(b); return; }
  a.insert(a.end(), std::make_move_iterator(b.begin()), std::make_move_iterator(b.end()));
}

// next(text): the next run of whole lines of the file, false when there is none.  The sample names are those of the
// first piece that holds a #CHROM line (the header precedes the records).
template <typename NextChunk>
FlatPopulation flattenVcf1000Chunks(NextChunk&& next, size_t threads, RowSink* sink = nullptr, StreamMerger* stream = nullptr) {
  const bool trace = std::getenv("KGX_FLATTEN_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "kgx flattenVcf1000: %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<std::string> samples;
  std::vector<RecordRows> all_parsed;
  std::string_view text;
  while (next(text)) {
  const VcfLines lines = scanLines(text);
  lap("scan lines");
  if (samples.empty()) samples = lines.samples;
  const size_t S = samples.size();
  auto parsed = parseRecords(lines, threads, [&](std::string_view record, RecordRows& out) {
    const auto f = split(record, '\t', S + 10);
    if (f.size() < 8) return;                                      // fewer than the mandatory fields: record dropped
    const std::string_view contig = f[0];
    bool pos_ok = true;
    const uint64_t pos = parseIndex(f[1], pos_ok);
    if (!pos_ok) return;
    const uint64_t offset = pos - 1;                               // VCF positions are 1-based
    const std::string_view ref = f[3];
    const std::string_view alt_field = f[4] == "." ? std::string_view() : f[4];
    const auto alts = split(alt_field, ',');
    const size_t A = alts.size();
    std::vector<float> af;
    bool af_bad_size = false;
    readInfoAf(f[7], A, af, af_bad_size);
    out.rows.resize(A);
    const size_t SB = copyRowBytes(S);
    out.copies.assign(A * SB, 0);
    for (size_t a = 0; a < A; ++a) {
      VariantRow& row = out.rows[a];
      row.contig = std::string(contig);
      row.offset = offset;
      row.hgvs = row.contig + ":g." + std::to_string(offset) + std::string(ref) + ">" + std::string(alts[a]);
      row.is_snp = isSnp(ref, alts[a]);
      row.info_af = af_bad_size ? std::numeric_limits<float>::infinity() : af[a];
    }
    const Chromosome chrom = chromosomeOf(contig);
    uint8_t* copies = out.copies.data();
    for (size_t idx = 9; idx < f.size() && idx - 9 < S; ++idx) {
      uint32_t pa, pb;
      phasedAlleles(f[idx], A, chrom, pa, pb);
      if (pa) addCopy(copies, SB, pa - 1, idx - 9);
      if (pb) addCopy(copies, SB, pb - 1, idx - 9);
    }
  });
  lap("parse records");
  if (stream) {
    if (!stream->consume(samples, parsed)) break;
    lap("stream rows");
    continue;
  }
  appendRecords(all_parsed, std::move(parsed));
  }
  if (stream) {
    FlatPopulation streamed;
    (void)stream->finish(streamed);
    lap("finish stream");
    return streamed;
  }
  FlatPopulation flat = mergeRecords(all_parsed, samples, false, threads, sink);
  lap("merge");
  return flat;
}

// One piece: the whole text.
struct WholeText {
  std::string_view text;
  bool given{false};
  bool operator()(std::string_view& out) { if (given) return false; given = true; out = text; return true; }
};

// Pieces of a file; an I/O or format error ends the run and is kept.
struct FilePieces {
  VcfChunkReader reader;
  std::string piece, error;
  bool operator()(std::string_view& out) {
    if (!error.empty() || !reader.next(piece, error)) return false;
    out = piece;
    return true;
  }
};

}  // namespace

FlatPopulation flattenVcf1000(std::string_view text, size_t threads) { return flattenVcf1000Chunks(WholeText{text}, threads); }

bool flattenVcf1000File(const std::string& file_name, FlatPopulation& flat, std::string& error, size_t threads, size_t chunk_bytes, RowSink* sink) {
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  flat = flattenVcf1000Chunks(pieces, threads, sink);
  error = pieces.error;
  return error.empty();
}

namespace {

// std::stoll as the reference uses it on a GT / AD token: white space, optional sign, digits; anything after is ignored.
// throws = true where std::stoll throws (no digits: invalid_argument; past int64: out_of_range).
int64_t parseLongLong(std::string_view s, bool& throws) {
  size_t i = 0;
  while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) ++i;
  bool negative = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) { negative = s[i] == '-'; ++i; }
  if (i >= s.size() || !std::isdigit(static_cast<unsigned char>(s[i]))) { throws = true; return 0; }
  uint64_t v = 0;
  const uint64_t limit = negative ? (1ull << 63) : (1ull << 63) - 1;
  for (; i < s.size() && std::isdigit(static_cast<unsigned char>(s[i])); ++i) {
    const uint64_t d = static_cast<uint64_t>(s[i] - '0');
    if (v > (limit - d) / 10) { throws = true; return 0; }
    v = v * 10 + d;
  }
  return negative ? -static_cast<int64_t>(v) : static_cast<int64_t>(v);
}

bool allDigits(std::string_view s) { return s.find_first_not_of("0123456789") == std::string_view::npos; }

// Variant::canonicalSequences, on text.
void canonicalSequences(std::string_view ref, std::string_view alt, uint64_t offset, std::string& c_ref, std::string& c_alt, uint64_t& c_offset) {
  const bool canonical = (ref.size() == 1 && alt.size() == 1) || (alt.size() == 1 && ref.size() > 1) || (ref.size() == 1 && alt.size() > 1);
  if (canonical) { c_ref = std::string(ref); c_alt = std::string(alt); c_offset = offset; return; }
  const size_t common = std::min(ref.size(), alt.size());
  size_t prefix = 0;
  while (prefix < common && ref[prefix] == alt[prefix]) ++prefix;
  prefix = prefix > 0 ? prefix - 1 : 0;                              // keeps one base in front: '1MnD' / '1MnI'
  size_t suffix = 0;
  while (suffix < common && ref[ref.size() - 1 - suffix] == alt[alt.size() - 1 - suffix]) ++suffix;
  int64_t adjusted = static_cast<int64_t>(std::min(common - prefix - 1, suffix));      // unsigned, as the reference computes it
  if (adjusted < 0) adjusted = 0;
  auto strip = [&](std::string_view s) {
    const size_t from = std::min(prefix, s.size());
    const size_t to = s.size() - std::min(static_cast<size_t>(adjusted), s.size());
    return to > from ? std::string(s.substr(from, to - from)) : std::string();
  };
  c_ref = strip(ref);
  c_alt = strip(alt);
  c_offset = offset + prefix;
}

// A scalar Float INFO value as getTypedInfoData<double> delivers it: nothing for a missing value.
bool infoScalar(std::string_view info, std::string_view key, double& value) {
  for (const auto item : split(info, ';')) {
    if (item.size() <= key.size() || item[key.size()] != '=' || item.substr(0, key.size()) != key) continue;
    const std::string text(item.substr(key.size() + 1));
    if (text.size() == 3) {
      std::string upper = text;
      for (auto& c : upper) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
      if (upper == "NAN") return false;
    }
    try {
      value = static_cast<double>(std::stof(text));
      return true;
    } catch (const std::out_of_range&) {
      std::string upper = text;
      for (auto& c : upper) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
      if (upper.find("E-") != std::string::npos) { value = std::numeric_limits<float>::min(); return true; }
      if (upper.find('E') != std::string::npos) { value = std::numeric_limits<float>::max(); return true; }
      return false;
    } catch (...) {
      return false;
    }
  }
  return false;
}

bool passesP7VariantFilter(std::string_view info, std::string_view contig) {
  double x = 0.0;
  if (infoScalar(info, "VQSLOD", x)) return x >= 0.0;                 // when present, the only test
  if (infoScalar(info, "QD", x) && !(x >= 2.0)) return false;
  if (infoScalar(info, "MQ", x) && !(x >= (contig == "Pf3D7_MIT_v3" ? 5.0 : 30.0))) return false;
  if (infoScalar(info, "SOR", x) && !(x <= 3.0)) return false;
  if (infoScalar(info, "MQRankSum", x) && !(x >= -12.5)) return false;
  if (infoScalar(info, "ReadPosRankSum", x) && !(x >= -8.0)) return false;
  return true;
}

}  // namespace

namespace {

template <typename NextChunk>
FlatPopulation flattenVcfPfChunks(NextChunk&& next, size_t threads, bool quality_filter, RowSink* sink = nullptr, StreamMerger* stream = nullptr) {
  std::vector<std::string> samples, contigs;
  std::vector<RecordRows> all_parsed;
  std::string_view text;
  while (next(text)) {
  const VcfLines lines = scanLines(text);
  if (samples.empty()) samples = lines.samples;
  contigs.insert(contigs.end(), lines.contigs.begin(), lines.contigs.end());
  const size_t S = samples.size();
  auto parsed = parseRecords(lines, threads, [&](std::string_view record, RecordRows& out) {
    const auto f = split(record, '\t', S + 10);
    if (f.size() < 9) return;
    const std::string_view contig = f[0];
    bool pos_ok = true;
    const uint64_t pos = parseIndex(f[1], pos_ok);
    if (!pos_ok) return;
    const uint64_t offset = pos - 1;
    const std::string_view ref = f[3];
    const auto alts = split(f[4] == "." ? std::string_view() : f[4], ',');   // "." is a missing alt (kgl_variant_vcf_impl.cpp:133-141)
    const size_t A = alts.size();
    const auto format = split(f[8], ':');
    size_t gt_index = format.size(), ad_index = format.size();
    for (size_t i = format.size(); i-- > 0;) {                       // formatIndex: the first match
      if (format[i] == "GT") gt_index = i;
      if (format[i] == "AD") ad_index = i;
    }
    if (gt_index == format.size() || ad_index == format.size()) return;   // both are required
    if (quality_filter && !passesP7VariantFilter(f[7], contig)) return;    // every Variant of the record is filtered out
    std::vector<float> af;
    bool af_bad_size = false;
    readInfoAf(f[7], A, af, af_bad_size);
    out.rows.resize(A);
    const size_t SB = copyRowBytes(S);
    out.copies.assign(A * SB, 0);
    for (size_t a = 0; a < A; ++a) {
      VariantRow& row = out.rows[a];
      std::string c_ref, c_alt;
      uint64_t c_offset = 0;
      canonicalSequences(ref, alts[a], offset, c_ref, c_alt, c_offset);
      row.contig = std::string(contig);
      row.offset = c_offset;
      row.hgvs = row.contig + ":g." + std::to_string(c_offset) + c_ref + ">" + c_alt;
      row.is_snp = isSnp(c_ref, c_alt);
      row.info_af = af_bad_size ? std::numeric_limits<float>::infinity() : af[a];
    }
    uint8_t* copies = out.copies.data();
    std::vector<uint64_t> depth;
    for (size_t idx = 9; idx < f.size() && idx - 9 < S; ++idx) {
      const auto fields = split(f[idx], ':');
      if (fields.size() <= gt_index) continue;
      const std::string_view gt = fields[gt_index];
      auto parts = split(gt, '/');
      if (parts.size() != 2) {
        parts = split(gt, '|');
        if (parts.size() != 2) continue;                              // missing, or not diploid
      }
      // the reference tests parts[0] for digits before converting EITHER part; a conversion that throws ends the record
      bool throws = false;
      int64_t a_allele = 0, b_allele = 0;
      if (allDigits(parts[0])) {
        a_allele = parseLongLong(parts[0], throws);
        if (throws) break;
        b_allele = parseLongLong(parts[1], throws);
        if (throws) break;
      }
      if (a_allele == 0 && b_allele == 0) continue;
      if (fields.size() <= ad_index) continue;
      const auto ad = split(fields[ad_index], ',');
      if (ad.size() != A + 1) continue;
      depth.clear();
      for (const auto token : ad) {
        if (!allDigits(token)) continue;                              // logged, not stored
        depth.push_back(static_cast<uint64_t>(parseLongLong(token, throws)));
        if (throws) break;
      }
      if (throws) break;
      for (const int64_t allele : {a_allele, b_allele}) {
        if (allele == 0) continue;
        // past the alt list or the stored depths the reference reads out of bounds (undefined): such a call is skipped
        if (allele < 0 || static_cast<uint64_t>(allele) > A || static_cast<uint64_t>(allele) >= depth.size()) continue;
        if (alts[allele - 1] == "*") continue;                        // upstream deletion
        if (depth[0] == 0 && depth[allele] == 0) continue;            // the spanning ("downstream") call of one
        addCopy(copies, SB, static_cast<size_t>(allele - 1), idx - 9);
      }
    }
  });
  if (stream) {
    if (!stream->consume(samples, parsed)) break;
    continue;
  }
  appendRecords(all_parsed, std::move(parsed));
  }
  if (stream) {
    FlatPopulation streamed;
    (void)stream->finish(streamed);
    streamed.contig_ids = contigs;
    return streamed;
  }
  FlatPopulation flat = mergeRecords(all_parsed, samples, true, threads, sink);
  flat.contig_ids = contigs;
  return flat;
}

}  // namespace

FlatPopulation flattenVcfPf(std::string_view text, size_t threads, bool quality_filter) { return flattenVcfPfChunks(WholeText{text}, threads, quality_filter); }

bool flattenVcfPfFile(const std::string& file_name, FlatPopulation& flat, std::string& error, size_t threads, bool quality_filter, size_t chunk_bytes,
                      RowSink* sink) {
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  flat = flattenVcfPfChunks(pieces, threads, quality_filter, sink);
  error = pieces.error;
  return error.empty();
}

bool flattenVcf1000FileStreaming(const std::string& file_name, StreamSink& sink, FlatPopulation& flat, std::string& error, bool& two_phase, size_t threads,
                                 size_t chunk_bytes) {
  two_phase = false;
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  StreamMerger stream(sink, false, threads);
  flat = flattenVcf1000Chunks(pieces, threads, nullptr, &stream);
  error = !pieces.error.empty() ? pieces.error : stream.error;
  two_phase = error.empty() && stream.two_phase;
  if (two_phase) error = stream.why();
  return error.empty();
}

bool flattenVcfPfFileStreaming(const std::string& file_name, StreamSink& sink, FlatPopulation& flat, std::string& error, bool& two_phase, size_t threads,
                               bool quality_filter, size_t chunk_bytes) {
  two_phase = false;
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  StreamMerger stream(sink, true, threads);
  flat = flattenVcfPfChunks(pieces, threads, quality_filter, nullptr, &stream);
  error = !pieces.error.empty() ? pieces.error : stream.error;
  two_phase = error.empty() && stream.two_phase;
  if (two_phase) error = stream.why();
  return error.empty();
}

// ---- INBREED inputs ----------------------------------------------------------------------------------------------

namespace {

// FrequencyDatabaseRead::infoFloatField on raw INFO text (kgl_variant_db_freq.cpp:72-122): a vector with one value per
// alt gives this alt's value, a single value serves every alt, any other size gives nothing; missing values are nothing.
double infoFrequency(std::string_view info, std::string_view field, size_t alt_index, size_t alt_count) {
  const double none = std::numeric_limits<double>::quiet_NaN();
  for (const auto item : split(info, ';')) {
    if (item.size() <= field.size() || item[field.size()] != '=' || item.substr(0, field.size()) != field) continue;
    const auto values = split(item.substr(field.size() + 1), ',');
    float f;
    if (values.size() == alt_count) f = infoFloat(values[alt_index]);
    else if (values.size() == 1) f = infoFloat(values[0]);
    else return none;
    return std::isnan(f) ? none : static_cast<double>(f);
  }
  return none;
}

}  // namespace

namespace {

// The INFO field that holds a super population's allele frequency in a data source's VCF: the rows of
// FrequencyDatabaseRead::field_text_map_AF_ (kgl_variant_db/kgl_variant_db_freq.h:84-96; private there, as is its lookup,
// :126-128 -- superPopFrequency reads through them from a Variant, which this flattener never builds).  Empty: no such field.
std::string superPopInfoField(DataSourceEnum data_source, const std::string& super_population) {
  struct FieldText { const char* super_population; const char* gnomad_2_1; const char* gnomad_ex_2_1; const char* gnomad_3_1; const char* gnomadgenome_3_1; const char* genome_1000; };
  static const FieldText table[] = {
      {"AFR", "AF_afr", "AF_afr", "AF_afr", "gnomad_AF_afr", "AFR_AF"}, {"AMR", "AF_amr", "AF_amr", "AF_amr", "gnomad_AF_amr", "AMR_AF"},
      {"EAS", "AF_eas", "AF_eas", "AF_eas", "gnomad_AF_eas", "EAS_AF"}, {"EUR", "AF_nfe", "AF_nfe", "AF_nfe", "gnomad_AF_nfe", "EUR_AF"},
      {"SAS", "AF", "AF_sas", "AF_sas", "gnomad_AF_sas", "SAS_AF"},     {"ALL", "AF", "AF", "AF", "gnomad_AF", "AF"}};
  for (const FieldText& row : table) {
    if (super_population != row.super_population) continue;
    switch (data_source) {                                     // lookupVariantSuperPopField's switch (kgl_variant_db_freq.cpp:33-69)
      case DataSourceEnum::Gnomad2_1: return row.gnomad_2_1;
      case DataSourceEnum::GnomadExomes2_1: return row.gnomad_ex_2_1;
      case DataSourceEnum::Gnomad3_1: case DataSourceEnum::GnomadExomes3_1: case DataSourceEnum::Gnomad3_0: return row.gnomad_3_1;
      case DataSourceEnum::GnomadGenome3_1: return row.gnomadgenome_3_1;
      case DataSourceEnum::Genome1000: return row.genome_1000;
      default: return {};
    }
  }
  return {};
}

template <typename NextChunk>
FlatReference flattenReferenceChunks(NextChunk&& next_piece, DataSourceEnum data_source) {
  const auto& super_pops = FrequencyDatabaseRead::superPopulations();
  std::vector<std::string> fields;
  for (const auto& sp : super_pops) fields.push_back(superPopInfoField(data_source, sp));
  FlatReference out;
  std::map<ContigOffset_t, ReferenceLocusRow> by_offset;
  std::vector<std::string> contigs_seen;
  std::string_view text;
  while (next_piece(text)) {
  const VcfLines lines = scanLines(text);
  for (const auto record : lines.records) {
    const auto f = split(record, '\t', 10);
    if (f.size() < 8) continue;
    bool pos_ok = true;
    const uint64_t pos = parseIndex(f[1], pos_ok);
    if (!pos_ok) continue;
    const std::string contig(f[0]);
    if (std::find(contigs_seen.begin(), contigs_seen.end(), contig) == contigs_seen.end()) contigs_seen.push_back(contig);
    std::string filter(f[6]);
    for (auto& c : filter) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    if (filter != "PASS") continue;                                     // PassFilter
    const std::string_view ref = f[3];
    // "The alt field can be blank": no ',' or an empty field is ONE alt, taken as written (:68-70)
    std::vector<std::string_view> alts;
    const std::string_view alt_field = f[4] == "." ? std::string_view() : f[4];     // "." is a missing alt (kgl_variant_vcf_impl.cpp:133-141)
    if (alt_field.find(',') == std::string_view::npos || alt_field.empty()) alts.push_back(alt_field);
    else alts = split(alt_field, ',');
    for (size_t a = 0; a < alts.size(); ++a) {
      if (!isSnp(ref, alts[a])) continue;                               // SNPFilter
      ReferenceAltRow alt;
      alt.hgvs = contig + ":g." + std::to_string(pos - 1) + std::string(ref) + ">" + std::string(alts[a]);
      for (size_t sp = 0; sp < 6 && sp < fields.size(); ++sp)
        alt.af[sp] = fields[sp].empty() ? std::numeric_limits<double>::quiet_NaN() : infoFrequency(f[7], fields[sp], a, alts.size());
      ReferenceLocusRow& locus = by_offset[pos - 1];
      locus.offset = pos - 1;
      locus.alts.push_back(std::move(alt));
    }
  }
  }
  out.contigs = contigs_seen.size();
  if (!contigs_seen.empty()) out.contig_id = contigs_seen.front();
  for (auto& [offset, locus] : by_offset) {
    out.max_alts = std::max<uint32_t>(out.max_alts, static_cast<uint32_t>(locus.alts.size()));
    out.loci.push_back(std::move(locus));
  }
  return out;
}

}  // namespace

FlatReference flattenReferenceVcf(std::string_view text, DataSourceEnum data_source) { return flattenReferenceChunks(WholeText{text}, data_source); }

bool flattenReferenceVcfFile(const std::string& file_name, DataSourceEnum data_source, FlatReference& reference, std::string& error, size_t threads,
                             size_t chunk_bytes) {
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  reference = flattenReferenceChunks(pieces, data_source);
  error = pieces.error;
  return error.empty();
}

namespace {

// One locus' row from its records (in file order): per genome the count of SNP variants so far, their codes and phases.
struct Gt8RecordCalls {
  int64_t locus{-1};
  std::vector<uint8_t> code;            // [n_alt]
  std::vector<uint8_t> calls;           // [S]: phase A alt | phase B alt << 4  (alt numbers up to 15 fit; larger ones are clipped below)
  std::vector<std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> wide;   // samples whose alt numbers need more than 4 bits
};
// Cell = uint8_t: two 4-bit codes (15 = not in the list); uint16_t: a wide locus, two 8-bit codes (255 = not in the list).
template <typename Cell>
void assembleGt8Locus(const Gt8RecordCalls* const* records, size_t n_records, const std::vector<int64_t>& genome_of_sample, Cell* row,
                      std::vector<uint8_t>& count, std::vector<uint8_t>& first_phase) {
  constexpr unsigned kBits = sizeof(Cell) * 4;
  constexpr unsigned kUnknown = (1u << kBits) - 1u;
  std::fill(count.begin(), count.end(), 0);
  auto add = [&](uint64_t g, uint8_t code, uint8_t phase) {
    const uint8_t n = count[g];
    if (n == 0) { row[g] = code; first_phase[g] = phase; }
    else if (n == 1) {
      const unsigned c0 = row[g] & kUnknown;
      // two copies of one variant on ONE phase (a repeated record): analogous, not homozygous -> the (0, a) cell
      if (c0 == code && code != kUnknown && first_phase[g] == phase) row[g] = static_cast<Cell>(static_cast<unsigned>(code) << kBits);
      else row[g] = static_cast<Cell>(c0 | (static_cast<unsigned>(code) << kBits));
    } else {
      row[g] = static_cast<Cell>(~0u);
    }
    if (n < 3) count[g] = static_cast<uint8_t>(n + 1);
  };
  for (size_t k = 0; k < n_records; ++k) {
    const Gt8RecordCalls& rc = *records[k];
    const size_t A = rc.code.size();
    // Genome1000VCFImpl::addVariants: all phase A variants of the record are added first (by alt), then phase B
    for (uint8_t phase = 0; phase < 2; ++phase) {
      for (size_t s = 0; s < rc.calls.size(); ++s) {
        if (rc.calls[s] == 0 || genome_of_sample[s] < 0) continue;
        const uint32_t alt = phase == 0 ? (rc.calls[s] & 0xFu) : (rc.calls[s] >> 4);
        if (alt == 0 || alt > A) continue;
        const uint8_t code = rc.code[alt - 1];
        if (code == 0) continue;                                        // not a SNP
        add(static_cast<uint64_t>(genome_of_sample[s]), code, phase);
      }
      for (const auto& [s, ab] : rc.wide) {
        if (genome_of_sample[s] < 0) continue;
        const uint32_t alt = phase == 0 ? ab.first : ab.second;
        if (alt == 0 || alt > A) continue;
        const uint8_t code = rc.code[alt - 1];
        if (code == 0) continue;
        add(static_cast<uint64_t>(genome_of_sample[s]), code, phase);
      }
    }
  }
}

// stream (may be null): rows leave for the sink as their loci are complete (Gt8StreamSink); *two_phase is set, with the
// reason in out.error, when the file cannot be taken that way.
template <typename NextChunk>
FlatDiploid flattenVcf1000Gt8Chunks(NextChunk&& next_piece, const FlatReference& reference, size_t threads, Gt8StreamSink* stream = nullptr,
                                    bool* two_phase = nullptr, std::string* stream_error = nullptr) {
  const bool trace = std::getenv("KGX_FLATTEN_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "kgx flattenVcf1000Gt8: %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  FlatDiploid out;
  out.n_loci = reference.loci.size();
  std::unordered_map<ContigOffset_t, uint32_t> locus_of_offset;
  locus_of_offset.reserve(reference.loci.size() * 2);
  for (uint32_t l = 0; l < reference.loci.size(); ++l) locus_of_offset.emplace(reference.loci[l].offset, l);
  std::vector<uint8_t> is_wide(reference.loci.size(), 0);          // more alts than two 4-bit indices address: 16-bit cells (FlatDiploid::wide_*)
  for (uint32_t l = 0; l < reference.loci.size(); ++l) is_wide[l] = reference.loci[l].alts.size() > kGt8NarrowAlts ? 1 : 0;

  // Per record that lands on a reference locus: per alt its code in that locus's list (0 = not a SNP: filtered out
  // before the sweep, _freq.cpp:436), and per sample the two alt numbers.  A record off the loci only says who holds
  // the contig (holds[]) and keeps nothing: what stays in memory between the pieces of a file is a byte per cell of
  // the result.
  using RecordCalls = Gt8RecordCalls;
  std::vector<RecordCalls> parsed;
  // streaming state: the records of the last locus met (it may go on in the next piece), the last locus written
  std::vector<RecordCalls> open_records;
  int64_t last_written = -1;
  bool stream_open = false;
  std::vector<int64_t> stream_genome_of_sample;
  auto giveUp = [&](const std::string& why) {
    if (two_phase) *two_phase = true;
    out.error = why;
  };
  auto streamFailed = [&](const char* what) { if (stream_error) *stream_error = what; };
  // write the rows of the given records' loci (ascending, whole loci) as one dense block: loci without a record in between are zero rows
  auto writeLoci = [&](const std::vector<const RecordCalls*>& records) -> bool {
    if (records.empty()) return true;
    const size_t G = out.genome_ids.size();
    const int64_t first = records.front()->locus, last = records.back()->locus;
    std::vector<uint8_t> block(static_cast<size_t>(last - first + 1) * G, 0);
    std::vector<size_t> group_begin;                    // rows are independent: the loci are assembled on every thread
    for (size_t k = 0; k < records.size(); ++k)
      if (k == 0 || records[k]->locus != records[k - 1]->locus) group_begin.push_back(k);
    group_begin.push_back(records.size());
    // (the wide loci of this block: their 16-bit rows stay in `out`, appended in ascending order before the threads start)
    std::vector<size_t> wide_at(group_begin.size() - 1, ~size_t{0});
    for (size_t grp = 0; grp + 1 < group_begin.size(); ++grp)
      if (is_wide[static_cast<size_t>(records[group_begin[grp]]->locus)]) {
        wide_at[grp] = out.wide_loci.size();
        out.wide_loci.push_back(static_cast<uint32_t>(records[group_begin[grp]]->locus));
      }
    out.wide_cells.resize(out.wide_loci.size() * G, 0);
    parallelChunks(group_begin.size() - 1, 64, threads, [&](size_t begin, size_t end) {
      std::vector<uint8_t> count(G), first_phase(G);
      for (size_t grp = begin; grp < end; ++grp) {
        const size_t k = group_begin[grp];
        uint8_t* row = &block[static_cast<size_t>(records[k]->locus - first) * G];
        if (wide_at[grp] != ~size_t{0}) {
          assembleGt8Locus(&records[k], group_begin[grp + 1] - k, stream_genome_of_sample, &out.wide_cells[wide_at[grp] * G], count, first_phase);
          std::fill(row, row + G, 0xFF);
        } else {
          assembleGt8Locus(&records[k], group_begin[grp + 1] - k, stream_genome_of_sample, row, count, first_phase);
        }
      }
    });
    if (!stream->write(static_cast<uint64_t>(first), static_cast<uint64_t>(last - first + 1), block.data())) { streamFailed("the row sink failed while taking a piece's rows"); return false; }
    last_written = last;
    return true;
  };
  std::vector<std::string> samples;
  std::vector<uint8_t> holds;             // [S] 1 if the sample carries ANY alt (SNP or not) of a record on the contig
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  std::string_view text;
  while (next_piece(text)) {
  const VcfLines lines = scanLines(text);
  lap("scan lines");
  if (samples.empty()) { samples = lines.samples; holds.assign(samples.size(), 0); }
  const size_t S = samples.size();
  std::vector<RecordCalls> piece(lines.records.size());
  {
    std::atomic<size_t> next{0};
    std::mutex holds_mutex;
    auto worker = [&]() {
      std::vector<uint8_t> local_holds(S, 0);
      for (size_t r = next.fetch_add(1); r < lines.records.size(); r = next.fetch_add(1)) {
        RecordCalls& rc = piece[r];
        const auto f = split(lines.records[r], '\t', S + 10);
        if (f.size() < 8) continue;
        if (f[0] != reference.contig_id) continue;
        bool pos_ok = true;
        const uint64_t pos = parseIndex(f[1], pos_ok);
        if (!pos_ok) continue;
        const std::string_view ref = f[3];
        const std::string_view alt_field = f[4] == "." ? std::string_view() : f[4];
        const auto alts = split(alt_field, ',');
        const size_t A = alts.size();
        const auto found = locus_of_offset.find(pos - 1);
        if (found != locus_of_offset.end()) {
          rc.locus = found->second;
          rc.code.assign(A, 0);
          const auto& list = reference.loci[found->second].alts;
          const bool wide_locus = is_wide[found->second];
          for (size_t a = 0; a < A; ++a) {
            if (!isSnp(ref, alts[a])) continue;
            const std::string hgvs = std::string(f[0]) + ":g." + std::to_string(pos - 1) + std::string(ref) + ">" + std::string(alts[a]);
            uint8_t c = wide_locus ? 255 : 15;                                   // not in the list
            for (size_t j = 0; j < list.size() && j < (wide_locus ? kGt8WideAlts : kGt8NarrowAlts); ++j)
              if (list[j].hgvs == hgvs) { c = static_cast<uint8_t>(j + 1); break; }
            rc.code[a] = c;
          }
          rc.calls.assign(S, 0);
        }
        const Chromosome chrom = chromosomeOf(f[0]);
        for (size_t idx = 9; idx < f.size() && idx - 9 < S; ++idx) {
          uint32_t pa, pb;
          phasedAlleles(f[idx], A, chrom, pa, pb);
          if (pa || pb) local_holds[idx - 9] = 1;
          if (rc.locus < 0) continue;
          if (pa > 15 || pb > 15) rc.wide.push_back({static_cast<uint32_t>(idx - 9), {pa, pb}});
          else rc.calls[idx - 9] = static_cast<uint8_t>(pa | (pb << 4));
        }
      }
      std::lock_guard<std::mutex> lock(holds_mutex);
      for (size_t smp = 0; smp < S; ++smp) holds[smp] |= local_holds[smp];
    };
    const size_t n = std::max<size_t>(1, std::min(threads, lines.records.size()));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  if (stream) {
    if (two_phase && *two_phase) break;
    if (!stream_open && S) {
      // genomes: every sample, in id order (whether each holds the contig is known at the end only)
      std::vector<uint32_t> by_name(S);
      std::iota(by_name.begin(), by_name.end(), 0u);
      std::sort(by_name.begin(), by_name.end(), [&](uint32_t x, uint32_t y) { return samples[x] < samples[y]; });
      stream_genome_of_sample.assign(S, -1);
      for (size_t rank = 0; rank < S; ++rank) {
        if (rank && samples[by_name[rank]] == samples[by_name[rank - 1]]) { giveUp("sample " + samples[by_name[rank]] + " is named twice: its columns add up into one genome"); break; }
        stream_genome_of_sample[by_name[rank]] = static_cast<int64_t>(rank);
        out.genome_ids.push_back(samples[by_name[rank]]);
      }
      if (two_phase && *two_phase) break;
      if (!stream->open(out.genome_ids, out.n_loci)) { streamFailed("the row sink failed to open"); break; }
      stream_open = true;
    }
    // the open locus' records, then this piece's, in file order; whole loci are written, the last one stays open
    std::vector<RecordCalls> held = std::move(open_records);
    open_records.clear();
    std::vector<const RecordCalls*> records;
    for (const auto& rc : held) records.push_back(&rc);
    for (const auto& rc : piece) if (rc.locus >= 0) records.push_back(&rc);
    bool ascending = true;
    for (size_t k = 0; k < records.size() && ascending; ++k)
      ascending = records[k]->locus > last_written && (k == 0 || records[k]->locus >= records[k - 1]->locus);
    if (!ascending) { giveUp("the records of the contig are not in ascending position order"); break; }
    size_t keep_from = records.size();
    while (keep_from > 0 && records[keep_from - 1]->locus == records.back()->locus) --keep_from;
    const std::vector<const RecordCalls*> complete(records.begin(), records.begin() + static_cast<std::ptrdiff_t>(keep_from));
    if (stream_open && !writeLoci(complete)) break;
    std::vector<RecordCalls> still_open;
    for (size_t k = keep_from; k < records.size(); ++k) still_open.push_back(*records[k]);
    open_records = std::move(still_open);
    continue;
  }
  for (auto& rc : piece) if (rc.locus >= 0) parsed.push_back(std::move(rc));      // file order kept
  }
  const size_t S = samples.size();
  if (stream) {
    if ((two_phase && *two_phase) || (stream_error && !stream_error->empty())) return out;
    if (!stream_open) {                                   // no sample columns at all: nobody holds the contig
      if (!stream->open(out.genome_ids, out.n_loci) || !stream->close()) streamFailed("the row sink failed");
      return out;
    }
    std::vector<const RecordCalls*> records;
    for (const auto& rc : open_records) records.push_back(&rc);
    if (!writeLoci(records)) return out;
    for (size_t smp = 0; smp < S; ++smp)
      if (!holds[smp]) { giveUp("sample " + samples[smp] + " carries no variant on the contig: it is no genome of it, and the rows are one column too wide"); return out; }
    if (!stream->close()) streamFailed("the row sink failed to close");
    return out;
  }

  lap("parse records");
  // genomes that hold the contig, in id order; a sample named twice is one genome
  std::vector<uint32_t> order;
  for (uint32_t s = 0; s < S; ++s) if (holds[s]) order.push_back(s);
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return samples[x] < samples[y]; });
  std::vector<int64_t> genome_of_sample(S, -1);
  for (uint32_t s : order) {
    if (out.genome_ids.empty() || out.genome_ids.back() != samples[s]) out.genome_ids.push_back(samples[s]);
    genome_of_sample[s] = static_cast<int64_t>(out.genome_ids.size()) - 1;
  }
  // a second column of an already-seen sample name
  for (uint32_t s = 0; s < S; ++s)
    if (genome_of_sample[s] < 0 && holds[s])
      genome_of_sample[s] = std::lower_bound(out.genome_ids.begin(), out.genome_ids.end(), samples[s]) - out.genome_ids.begin();
  const size_t G = out.genome_ids.size();
  out.bytes.assign(out.n_loci * G, 0);
  if (G == 0 || out.n_loci == 0) return out;          // nobody holds the contig, or no reference locus: nothing to assemble
  for (uint32_t l = 0; l < reference.loci.size(); ++l)
    if (is_wide[l]) {                                  // (every wide locus gets its row, carried or not)
      out.wide_loci.push_back(l);
      std::fill(&out.bytes[static_cast<uint64_t>(l) * G], &out.bytes[static_cast<uint64_t>(l) * G] + G, 0xFF);
    }
  out.wide_cells.assign(out.wide_loci.size() * G, 0);

  lap("genome order");
  // Assemble locus by locus (rows are independent; the records of one locus are taken in file order).  Per genome of the
  // row: count of SNP variants so far, their codes and phases.
  std::vector<uint32_t> by_locus;                       // record indices sorted by (locus, record)
  for (uint32_t r = 0; r < parsed.size(); ++r) if (parsed[r].locus >= 0) by_locus.push_back(r);
  std::stable_sort(by_locus.begin(), by_locus.end(), [&](uint32_t a, uint32_t b) { return parsed[a].locus < parsed[b].locus; });
  std::atomic<size_t> next_group{0};
  std::atomic<bool> failed{false};
  std::mutex error_mutex;
  auto assemble = [&]() {
    std::vector<uint8_t> count(G), first_phase(G);
    constexpr size_t kChunk = 256;
    for (size_t begin = next_group.fetch_add(kChunk); begin < by_locus.size() && !failed.load(); begin = next_group.fetch_add(kChunk)) {
      // a chunk boundary must not split a locus: the thread that owns the chunk holding a locus' first record owns the locus
      size_t k = begin;
      if (k > 0 && parsed[by_locus[k]].locus == parsed[by_locus[k - 1]].locus) {
        const int64_t shared = parsed[by_locus[k]].locus;
        while (k < by_locus.size() && parsed[by_locus[k]].locus == shared) ++k;
      }
      const size_t end = std::min(by_locus.size(), begin + kChunk);
      while (k < end) {
        const int64_t locus = parsed[by_locus[k]].locus;
        uint8_t* row = &out.bytes[static_cast<uint64_t>(locus) * G];
        std::vector<const RecordCalls*> records;
        for (; k < by_locus.size() && parsed[by_locus[k]].locus == locus; ++k) records.push_back(&parsed[by_locus[k]]);
        if (is_wide[static_cast<size_t>(locus)]) {
          const size_t at = static_cast<size_t>(std::lower_bound(out.wide_loci.begin(), out.wide_loci.end(), static_cast<uint32_t>(locus)) - out.wide_loci.begin());
          assembleGt8Locus(records.data(), records.size(), genome_of_sample, &out.wide_cells[at * G], count, first_phase);
          std::fill(row, row + G, 0xFF);
        } else {
          assembleGt8Locus(records.data(), records.size(), genome_of_sample, row, count, first_phase);
        }
      }
    }
  };
  {
    const size_t n = std::max<size_t>(1, std::min(threads, (by_locus.size() + 255) / 256));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n; ++t) pool.emplace_back(assemble);
    assemble();
    for (auto& th : pool) th.join();
  }
  lap("assemble");
  return out;
}

}  // namespace

FlatDiploid flattenVcf1000Gt8(std::string_view text, const FlatReference& reference, size_t threads) {
  return flattenVcf1000Gt8Chunks(WholeText{text}, reference, threads);
}

bool flattenVcf1000Gt8File(const std::string& file_name, const FlatReference& reference, FlatDiploid& diploid, std::string& error, size_t threads,
                           size_t chunk_bytes) {
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  diploid = flattenVcf1000Gt8Chunks(pieces, reference, threads);
  error = pieces.error;
  return error.empty();
}

bool flattenVcf1000Gt8FileStreaming(const std::string& file_name, const FlatReference& reference, Gt8StreamSink& sink, FlatDiploid& diploid, std::string& error,
                                    bool& two_phase, size_t threads, size_t chunk_bytes) {
  two_phase = false;
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  std::string sink_error;
  diploid = flattenVcf1000Gt8Chunks(pieces, reference, threads, &sink, &two_phase, &sink_error);
  error = !pieces.error.empty() ? pieces.error : sink_error;
  if (error.empty() && two_phase) { error = diploid.error; diploid.error.clear(); }
  else two_phase = false;
  return error.empty();
}


// ---- record columns for the rsid / Ensembl indexes (kgx_variant_sort.h) --------------------------

namespace {

// The sub-field names of the vep INFO field, from its header line: what VCFParseHeader::parseVcfHeader +
// tokenizeVcfHeaderKeyValues (kgl_parser/kgl_variant_factory_vcf_parse_header.cpp:113-265) and
// VEPSubFieldHeader::parseHeader (kgl_evidence/kgl_variant_factory_vcf_evidence.cpp:24-58) make of
//   ##INFO=<ID=vep,Number=.,Type=String,Description="... Format: Allele|Consequence|...">
// Angle brackets vanish wherever they stand, items separate at commas outside double quotes (backslash escapes the next
// character), an item is cut at '=' signs with empty pieces skipped (so a description stops at its first '='), a line
// without Type or Number is no INFO record, a later line of the same ID replaces an earlier one, and a name listed
// twice voids the header.
std::vector<std::string> vepHeaderNames(std::string_view text) {
  std::vector<std::string> names;
  for (size_t begin = 0; begin < text.size();) {
    size_t end = text.find('\n', begin);
    if (end == std::string_view::npos) end = text.size();
    std::string_view line = text.substr(begin, end - begin);
    begin = end + 1;
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (line.empty() || line[0] != '#') { if (!line.empty()) break; else continue; }
    if (line.rfind("#CHROM", 0) == 0) break;
    const size_t eq = line.find('=');
    if (eq == std::string_view::npos) continue;
    std::string key(line.substr(0, eq));
    if (key.size() < 2 || key[0] != '#') continue;
    key.erase(0, 2);
    for (auto& c : key) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
    if (key != "INFO") continue;
    std::map<std::string, std::string> pairs;
    std::string item;
    bool quoted = false;
    auto closeItem = [&]() {
      std::string name, value;
      int piece = 0;
      for (const auto token : split(item, '=')) {
        if (token.empty()) continue;
        if (piece == 0) name = std::string(token);
        else if (piece == 1) value = std::string(token);
        ++piece;
      }
      item.clear();
      if (piece == 0) return;
      for (auto& c : name) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
      pairs[name] = value;
    };
    const std::string_view body = line.substr(eq + 1);
    for (size_t i = 0; i < body.size(); ++i) {
      const char c = body[i];
      if (c == '<' || c == '>') continue;
      if (c == '\\' && i + 1 < body.size()) { ++i; item += body[i] == 'n' ? '\n' : body[i]; continue; }
      if (c == '"') { quoted = !quoted; continue; }
      if (c == ',' && !quoted) { closeItem(); continue; }
      item += c;
    }
    closeItem();
    const auto id = pairs.find("ID");
    if (id == pairs.end() || id->second != "vep" || !pairs.count("TYPE") || !pairs.count("NUMBER")) continue;
    names.clear();
    const auto description = pairs.find("DESCRIPTION");
    if (description == pairs.end()) continue;
    const size_t format = description->second.find("Format: ");
    if (format == std::string::npos) continue;
    for (const auto name : split(std::string_view(description->second).substr(format + 8), '|')) names.emplace_back(name);
    std::vector<std::string> sorted = names;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) names.clear();
  }
  return names;
}

struct SortParsed {
  SortRecord record;
  bool kept{false};
  struct Call { uint32_t sample; uint16_t a, b; };      // 1-based alts of a sample column that is not all reference
  std::vector<Call> calls;
};

}  // namespace

namespace {

// next_piece(text): the next run of whole lines; the header (the vep line, #CHROM) is in the first piece.
template <typename NextChunk>
SortColumns sortColumnsChunks(NextChunk&& next_piece, SortVcfFlavour flavour, const GenomeId_t& genome_id, size_t threads) {
  SortColumns out;
  const bool phased = flavour == SortVcfFlavour::Phased1000;
  std::vector<SortParsed> parsed;
  std::vector<std::string> samples;
  bool first_piece = true;
  size_t gene_column = 0, vep_columns = 0;
  bool have_gene = false;
  std::string_view text;
  while (next_piece(text)) {
  if (first_piece) {
    out.vep_header = vepHeaderNames(text);
    const auto gene_at = std::find(out.vep_header.begin(), out.vep_header.end(), std::string("Gene"));
    gene_column = static_cast<size_t>(gene_at - out.vep_header.begin());
    have_gene = gene_at != out.vep_header.end();
    vep_columns = out.vep_header.size();
    first_piece = false;
  }
  const VcfLines lines = scanLines(text);
  if (samples.empty()) samples = lines.samples;
  const size_t S = samples.size();
  const size_t parsed_base = parsed.size();
  parsed.resize(parsed_base + lines.records.size());
  parallelChunks(lines.records.size(), 256, threads, [&](size_t begin, size_t end) {
    for (size_t r = begin; r < end; ++r) {
      SortParsed& p = parsed[parsed_base + r];
      const auto f = split(lines.records[r], '\t', phased ? S + 10 : 10);
      if (f.size() < 8) continue;                                        // fewer than the mandatory fields: record dropped
      bool pos_ok = true;
      const uint64_t pos = parseIndex(f[1], pos_ok);
      if (!pos_ok) continue;
      p.kept = true;
      SortRecord& rec = p.record;
      rec.contig = std::string(f[0]);
      rec.offset = pos - 1;
      if (f[2] != ".") {                                                  // "." is no identifier; otherwise trimEndWhiteSpace
        std::string_view id = f[2];
        while (!id.empty() && std::isspace(static_cast<unsigned char>(id.back()))) id.remove_suffix(1);
        rec.identifier = std::string(id);
      }
      rec.ref = std::string(f[3]);
      const std::string_view alt_field = f[4] == "." ? std::string_view() : f[4];
      // GrchVCFImpl takes an ALT without ',' (or empty) as one alt; Genome1000VCFImpl tokenizes: the same list either way
      for (const auto alt : split(alt_field, ',')) rec.alts.emplace_back(alt);
      // the vep INFO vector: fields at ';', key before the FIRST '=', the first of two equal keys counts
      if (have_gene) {
        for (const auto item : split(f[7], ';')) {
          const size_t eq = item.find('=');
          if ((eq == std::string_view::npos ? item : item.substr(0, eq)) != "vep") continue;
          if (eq != std::string_view::npos) {
            for (const auto entry : split(item.substr(eq + 1), ',')) {
              const auto sub_fields = split(entry, '|');
              if (sub_fields.size() != vep_columns) continue;              // the Gnomad 3 work-around: entries of another size are dropped
              rec.vep_usable = true;
              if (!sub_fields[gene_column].empty()) rec.genes.emplace_back(sub_fields[gene_column]);
            }
          }
          break;
        }
        std::sort(rec.genes.begin(), rec.genes.end());
        rec.genes.erase(std::unique(rec.genes.begin(), rec.genes.end()), rec.genes.end());
      }
      if (phased) {
        const Chromosome chrom = chromosomeOf(rec.contig);
        for (size_t idx = 9; idx < f.size() && idx - 9 < S; ++idx) {
          uint32_t pa, pb;
          phasedAlleles(f[idx], rec.alts.size(), chrom, pa, pb);
          if (pa || pb) p.calls.push_back({static_cast<uint32_t>(idx - 9), static_cast<uint16_t>(pa), static_cast<uint16_t>(pb)});
        }
      }
    }
  });
  }
  const size_t S = samples.size();

  // genomes: the one named genome, or the sample names that carry anything (PopulationDB creates a genome on its first variant)
  std::vector<uint32_t> genome_of_sample(S, UINT32_MAX);
  if (!phased) {
    out.genome_ids.push_back(genome_id);
  } else {
    std::vector<uint8_t> carries(S, 0);
    for (const auto& p : parsed) for (const auto& call : p.calls) carries[call.sample] = 1;
    std::map<std::string, uint32_t> by_name;
    for (size_t s = 0; s < S; ++s) if (carries[s]) by_name.emplace(samples[s], 0);
    uint32_t next = 0;
    for (auto& [name, index] : by_name) { index = next++; out.genome_ids.push_back(name); }
    for (size_t s = 0; s < S; ++s) if (carries[s]) genome_of_sample[s] = by_name[samples[s]];
  }

  // records in a genome's visiting order: contig id, offset, then the order the (single-threaded) file gives
  std::vector<uint32_t> order;
  for (size_t r = 0; r < parsed.size(); ++r) if (parsed[r].kept) order.push_back(static_cast<uint32_t>(r));
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    const SortRecord& a = parsed[x].record;
    const SortRecord& b = parsed[y].record;
    if (a.contig != b.contig) return a.contig < b.contig;
    return a.offset < b.offset;
  });
  out.records.reserve(order.size());
  for (const uint32_t r : order) out.records.push_back(std::move(parsed[r].record));

  // Variant objects per genome.  Within one record the parser adds phase A's alts in ascending order, then phase B's
  // (kgl_variant_factory_1000_impl.cpp:118-140), each to its genomes in sample order.
  const size_t G = out.genome_ids.size();
  out.genome_begin.assign(G + 1, 0);
  if (!phased) {
    for (const auto& rec : out.records) out.genome_begin[1] += rec.alts.size();
    out.visits.reserve(out.genome_begin[1]);
    for (size_t r = 0; r < out.records.size(); ++r)
      for (size_t a = 0; a < out.records[r].alts.size(); ++a) out.visits.push_back({static_cast<uint32_t>(r), static_cast<uint16_t>(a), 255});
    return out;
  }
  for (const uint32_t r : order)
    for (const auto& call : parsed[r].calls) out.genome_begin[genome_of_sample[call.sample] + 1] += (call.a ? 1 : 0) + (call.b ? 1 : 0);
  for (size_t g = 0; g < G; ++g) out.genome_begin[g + 1] += out.genome_begin[g];
  out.visits.resize(out.genome_begin[G]);
  std::vector<uint64_t> cursor(out.genome_begin.begin(), out.genome_begin.end() - 1);
  struct Added { uint8_t phase; uint16_t alt; uint32_t sample; };
  std::vector<Added> added;
  for (size_t i = 0; i < order.size(); ++i) {
    added.clear();
    for (const auto& call : parsed[order[i]].calls) {
      if (call.a) added.push_back({1, static_cast<uint16_t>(call.a - 1), call.sample});
      if (call.b) added.push_back({2, static_cast<uint16_t>(call.b - 1), call.sample});
    }
    std::sort(added.begin(), added.end(), [](const Added& x, const Added& y) {
      if (x.phase != y.phase) return x.phase < y.phase;
      if (x.alt != y.alt) return x.alt < y.alt;
      return x.sample < y.sample;
    });
    for (const auto& add : added) out.visits[cursor[genome_of_sample[add.sample]]++] = {static_cast<uint32_t>(i), add.alt, add.phase};
  }
  return out;
}

}  // namespace

SortColumns sortColumnsFromVcf(std::string_view text, SortVcfFlavour flavour, const GenomeId_t& genome_id, size_t threads) {
  return sortColumnsChunks(WholeText{text}, flavour, genome_id, threads);
}

bool sortColumnsFromVcfFile(const std::string& file_name, SortVcfFlavour flavour, SortColumns& columns, std::string& error, const GenomeId_t& genome_id,
                            size_t threads, size_t chunk_bytes) {
  FilePieces pieces;
  if (!pieces.reader.open(file_name, error, threads, chunk_bytes)) return false;
  columns = sortColumnsChunks(pieces, flavour, genome_id, threads);
  error = pieces.error;
  return error.empty();
}

}  // namespace kellerberrin::genome::analysis::gpu
