// VCF text -> FlatPopulation without building Variant / PopulationDB objects (SURVEY.md §8f "next" #1).
//
// Behaviour to match: the reference's phased-diploid parser, i.e. what a PopulationDB filled by
//   ParseVCF::moveToVcfRecord            (kgl_genomics/kgl_parser/kgl_variant_vcf_impl.cpp:96-170)
//   Genome1000VCFImpl::ParseRecord       (kgl_parser/kgl_variant_factory_1000_impl.cpp:63-145)
//   Genome1000VCFImpl::alternateIndex    (:148-272)
// would flatten to through flattenPopulation(): genomes exist only if they carry a variant, variants only if some
// genome carries them, rows in lexicographic HGVS order, dosage = copies of the HGVS in the genome.
// Records are independent, so they are parsed by a pool of threads straight into 2-bit rows.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>

#include "kgx_flatten.h"

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
  std::vector<uint8_t> copies;           // [n_alt][n_samples] copies of the alt in the sample (0..2)
};

}  // namespace

FlatPopulation flattenVcf1000(std::string_view text, size_t threads) {
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;

  // Lines: sample names from the #CHROM header, records = every non-empty line not starting with '#'.
  std::vector<std::string> samples;
  std::vector<std::string_view> records;
  for (size_t begin = 0; begin <= text.size();) {
    size_t end = text.find('\n', begin);
    if (end == std::string_view::npos) end = text.size();
    std::string_view line = text.substr(begin, end - begin);
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    if (!line.empty()) {
      if (line[0] == '#') {
        if (line.rfind("#CHROM", 0) == 0) {
          const auto f = split(line, '\t');
          for (size_t i = 9; i < f.size(); ++i) samples.emplace_back(f[i]);
        }
      } else {
        records.push_back(line);
      }
    }
    begin = end + 1;
  }
  const size_t S = samples.size();

  std::vector<RecordRows> parsed(records.size());
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (size_t r = next.fetch_add(1); r < records.size(); r = next.fetch_add(1)) {
      const auto f = split(records[r], '\t', S + 10);
      if (f.size() < 8) continue;                                    // fewer than the mandatory fields: record dropped
      const std::string_view contig = f[0];
      bool pos_ok = true;
      const uint64_t pos = parseIndex(f[1], pos_ok);
      if (!pos_ok) continue;
      const uint64_t offset = pos - 1;                               // VCF positions are 1-based
      const std::string_view ref = f[3];
      const std::string_view alt_field = f[4] == "." ? std::string_view() : f[4];
      const auto alts = split(alt_field, ',');
      const size_t A = alts.size();
      // INFO "AF": one value per alt, or a single value for all
      std::vector<float> af(A, std::numeric_limits<float>::quiet_NaN());
      bool af_bad_size = false;
      for (const auto item : split(f[7], ';')) {
        if (item.size() > 3 && item.substr(0, 3) == "AF=") {
          const auto values = split(item.substr(3), ',');
          if (values.size() == A) for (size_t a = 0; a < A; ++a) af[a] = infoFloat(values[a]);
          else af_bad_size = true;                                   // P7FrequencyFilter errors out on a size mismatch: in no bin
        }
      }
      RecordRows& out = parsed[r];
      out.rows.resize(A);
      out.copies.assign(A * S, 0);
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
        if (pa) ++copies[(pa - 1) * S + (idx - 9)];
        if (pb) ++copies[(pb - 1) * S + (idx - 9)];
      }
    }
  };
  {
    const size_t n = std::max<size_t>(1, std::min(threads, records.size()));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }

  // Variants: rows in lexicographic HGVS order; records repeating an HGVS add their copies.
  struct Key { const std::string* hgvs; uint32_t record, alt; };
  std::vector<Key> keys;
  for (uint32_t r = 0; r < parsed.size(); ++r)
    for (uint32_t a = 0; a < parsed[r].rows.size(); ++a) keys.push_back({&parsed[r].rows[a].hgvs, r, a});
  std::stable_sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return *x.hgvs < *y.hgvs; });

  // Genomes exist only if they carry something (the parser creates them in addVariant); std::map order.
  std::vector<uint8_t> carries(S, 0);
  for (const auto& rec : parsed)
    for (size_t i = 0; i < rec.copies.size(); ++i)
      if (rec.copies[i]) carries[i % S] = 1;
  std::vector<uint32_t> sample_order;
  for (uint32_t s = 0; s < S; ++s) if (carries[s]) sample_order.push_back(s);
  std::sort(sample_order.begin(), sample_order.end(), [&](uint32_t x, uint32_t y) { return samples[x] < samples[y]; });

  FlatPopulation flat;
  for (uint32_t s : sample_order) flat.genome_ids.push_back(samples[s]);
  const size_t G = flat.genome_ids.size();
  flat.row_bytes = (G + 3) / 4;
  std::vector<uint32_t> total(S);
  for (size_t k = 0; k < keys.size();) {
    size_t e = k + 1;
    while (e < keys.size() && *keys[e].hgvs == *keys[k].hgvs) ++e;
    std::fill(total.begin(), total.end(), 0u);
    bool any = false;
    for (size_t m = k; m < e; ++m) {
      const uint8_t* c = &parsed[keys[m].record].copies[static_cast<size_t>(keys[m].alt) * S];
      for (size_t s = 0; s < S; ++s) { total[s] += c[s]; any = any || c[s]; }
    }
    if (any) {                                      // a variant nobody carries never reaches the PopulationDB
      const uint32_t row_index = static_cast<uint32_t>(flat.rows.size());
      flat.rows.push_back(parsed[keys[k].record].rows[keys[k].alt]);      // the first record's Variant is kept (uniqueVariants)
      const size_t base = flat.packed.size();
      flat.packed.resize(base + flat.row_bytes, 0);
      for (size_t g = 0; g < G; ++g) {
        const uint32_t d = total[sample_order[g]];
        flat.variant_objects += d;
        flat.packed[base + g / 4] |= static_cast<uint8_t>((d > 2 ? 3u : d) << (2 * (g % 4)));
        if (d > 2) flat.non_diploid.push_back({row_index, static_cast<uint32_t>(g), d});
      }
    }
    k = e;
  }
  return flat;
}

}  // namespace kellerberrin::genome::analysis::gpu
