// Small extern "C" window onto the host-side flatteners so that they can be tested without a GPU (ctypes).
// Not part of the device ABI (include/kgx.h): pure host code, no HIP calls.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <chrono>
#include <sstream>
#include <vector>

#include "kga_analysis_gpu_allele.h"
#include "kgx_flatten.h"
#include "kgx_variant_sort.h"
#include "kgx_vcf_io.h"

using kellerberrin::genome::analysis::gpu::FlatPopulation;

extern "C" {

void* kgxh_flatten_vcf1000(const char* text, uint64_t len, int threads) {
  if (!text) return nullptr;
  auto* flat = new FlatPopulation(kellerberrin::genome::analysis::gpu::flattenVcf1000(std::string_view(text, len), threads > 0 ? threads : 0));
  return flat;
}

void* kgxh_flatten_vcf_pf(const char* text, uint64_t len, int threads, int quality_filter) {
  if (!text) return nullptr;
  return new FlatPopulation(kellerberrin::genome::analysis::gpu::flattenVcfPf(std::string_view(text, len), threads > 0 ? threads : 0, quality_filter != 0));
}

// The same from a file, a bounded piece of text at a time (chunk_bytes of text per piece; 0 = the default 64 MiB).
// flavour 0 = 1000 Genomes, 1 = P. falciparum.  Null + message on an I/O or format error.
void* kgxh_flatten_vcf_file(const char* path, int flavour, int threads, int quality_filter, uint64_t chunk_bytes, char* error, size_t error_len) {
  if (!path) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  auto* flat = new FlatPopulation();
  std::string err;
  const size_t piece = chunk_bytes ? static_cast<size_t>(chunk_bytes) : (size_t{64} << 20);
  const bool ok = flavour == 0 ? g::flattenVcf1000File(path, *flat, err, threads > 0 ? threads : 0, piece)
                               : g::flattenVcfPfFile(path, *flat, err, threads > 0 ? threads : 0, quality_filter != 0, piece);
  if (!ok) {
    if (error && error_len) { std::strncpy(error, err.c_str(), error_len - 1); error[error_len - 1] = 0; }
    delete flat;
    return nullptr;
  }
  return flat;
}

// The streaming flatteners (rows leave for a sink piece by piece, in first-appearance order) into memory, then put into the
// two-phase flatteners' order (primary rows by HGVS, split rows by their primary row and bin), so that the result can be
// compared field for field with kgxh_flatten_vcf_file's.  *two_phase = 1 (and null): the file has to take the other path.
void* kgxh_flatten_vcf_file_streaming(const char* path, int flavour, int threads, int quality_filter, uint64_t chunk_bytes, char* error, size_t error_len,
                                      int* two_phase) {
  if (!path) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  struct MemorySink final : g::StreamSink {
    std::vector<uint8_t> packed;
    uint64_t row_bytes{0}, rows{0};
    bool open(uint64_t, uint64_t rb) override { row_bytes = rb; return true; }
    bool write(uint64_t first_row, uint64_t n_rows, const uint8_t* data) override {
      if ((first_row + n_rows) * row_bytes > packed.size()) packed.resize((first_row + n_rows) * row_bytes, 0);
      if (n_rows && row_bytes) std::memcpy(&packed[first_row * row_bytes], data, n_rows * row_bytes);
      return true;
    }
    bool read(uint64_t row, uint8_t* data) override {
      if ((row + 1) * row_bytes > packed.size()) return false;
      std::memcpy(data, &packed[row * row_bytes], row_bytes);
      return true;
    }
    bool close(uint64_t n_rows) override { rows = n_rows; packed.resize(n_rows * row_bytes, 0); return true; }
  } sink;
  auto* flat = new FlatPopulation();
  std::string err;
  bool other_path = false;
  const size_t piece = chunk_bytes ? static_cast<size_t>(chunk_bytes) : (size_t{64} << 20);
  const bool ok = flavour == 0 ? g::flattenVcf1000FileStreaming(path, sink, *flat, err, other_path, threads > 0 ? threads : 0, piece)
                               : g::flattenVcfPfFileStreaming(path, sink, *flat, err, other_path, threads > 0 ? threads : 0, quality_filter != 0, piece);
  if (two_phase) *two_phase = other_path ? 1 : 0;
  if (!ok) {
    if (error && error_len) { std::strncpy(error, err.c_str(), error_len - 1); error[error_len - 1] = 0; }
    delete flat;
    return nullptr;
  }
  // canonical order
  const size_t P = flat->primary_rows, D = flat->rows.size(), RB = flat->row_bytes;
  std::vector<size_t> order(P);
  for (size_t i = 0; i < P; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return flat->rows[x].hgvs < flat->rows[y].hgvs; });
  std::vector<size_t> new_of_old(P);
  for (size_t i = 0; i < P; ++i) new_of_old[order[i]] = i;
  std::vector<size_t> split_order;
  for (size_t i = P; i < D; ++i) split_order.push_back(i);
  std::stable_sort(split_order.begin(), split_order.end(), [&](size_t x, size_t y) {
    const size_t px = new_of_old[static_cast<size_t>(flat->rows[x].split_of)], py = new_of_old[static_cast<size_t>(flat->rows[y].split_of)];
    if (px != py) return px < py;
    return g::fwsBinOfFrequency(flat->rows[x].info_af) < g::fwsBinOfFrequency(flat->rows[y].info_af);
  });
  FlatPopulation sorted;
  sorted.genome_ids = flat->genome_ids;
  sorted.primary_rows = P;
  sorted.row_bytes = RB;
  sorted.variant_objects = flat->variant_objects;
  sorted.contig_ids = flat->contig_ids;
  sorted.packed.resize(D * RB);
  for (size_t i = 0; i < P; ++i) {
    sorted.rows.push_back(flat->rows[order[i]]);
    if (RB) std::memcpy(&sorted.packed[i * RB], &sink.packed[order[i] * RB], RB);
  }
  for (size_t i = 0; i < split_order.size(); ++i) {
    sorted.rows.push_back(flat->rows[split_order[i]]);
    sorted.rows.back().split_of = static_cast<int64_t>(new_of_old[static_cast<size_t>(flat->rows[split_order[i]].split_of)]);
    if (RB) std::memcpy(&sorted.packed[(P + i) * RB], &sink.packed[split_order[i] * RB], RB);
  }
  for (auto cell : flat->non_diploid) {
    cell.row = static_cast<uint32_t>(new_of_old[cell.row]);
    sorted.non_diploid.push_back(cell);
  }
  std::sort(sorted.non_diploid.begin(), sorted.non_diploid.end(), [](const g::NonDiploidCell& x, const g::NonDiploidCell& y) {
    return x.row != y.row ? x.row < y.row : x.genome < y.genome;
  });
  *flat = std::move(sorted);
  return flat;
}

// The streaming flattener into a sink that only counts (scripts/bench_flatten_file.py: the flattener's own footprint and
// time, without a packed copy anywhere on the host).  Rows read back for a merge are all-zero: use on files without
// repeated records.  Returns the number of rows, or -1.
int64_t kgxh_stream_flatten_count(const char* path, int flavour, int threads, uint64_t chunk_bytes, uint64_t* genomes, uint64_t* bytes_written) {
  if (!path) return -1;
  namespace g = kellerberrin::genome::analysis::gpu;
  struct CountingSink final : g::StreamSink {
    uint64_t row_bytes{0}, rows{0}, bytes{0}, genomes{0};
    bool open(uint64_t n, uint64_t rb) override { genomes = n; row_bytes = rb; return true; }
    bool write(uint64_t first_row, uint64_t n_rows, const uint8_t*) override { rows = std::max(rows, first_row + n_rows); bytes += n_rows * row_bytes; return true; }
    bool read(uint64_t, uint8_t* data) override { std::memset(data, 0, row_bytes); return true; }
    bool close(uint64_t n_rows) override { rows = n_rows; return true; }
  } sink;
  FlatPopulation flat;
  std::string err;
  bool other_path = false;
  const size_t piece = chunk_bytes ? static_cast<size_t>(chunk_bytes) : (size_t{64} << 20);
  const bool ok = flavour == 0 ? g::flattenVcf1000FileStreaming(path, sink, flat, err, other_path, threads > 0 ? threads : 0, piece)
                               : g::flattenVcfPfFileStreaming(path, sink, flat, err, other_path, threads > 0 ? threads : 0, false, piece);
  if (!ok) return -1;
  if (genomes) *genomes = sink.genomes;
  if (bytes_written) *bytes_written = sink.bytes;
  return static_cast<int64_t>(sink.rows);
}

void kgxh_flat_destroy(void* h) { delete static_cast<FlatPopulation*>(h); }
uint64_t kgxh_flat_genomes(void* h) { return h ? static_cast<FlatPopulation*>(h)->genomes() : 0; }
uint64_t kgxh_flat_variants(void* h) { return h ? static_cast<FlatPopulation*>(h)->variants() : 0; }
uint64_t kgxh_flat_row_bytes(void* h) { return h ? static_cast<FlatPopulation*>(h)->row_bytes : 0; }
uint64_t kgxh_flat_variant_objects(void* h) { return h ? static_cast<FlatPopulation*>(h)->variant_objects : 0; }
uint64_t kgxh_flat_split_rows(void* h) { return h ? static_cast<FlatPopulation*>(h)->deviceRows() - static_cast<FlatPopulation*>(h)->variants() : 0; }
uint64_t kgxh_flat_non_diploid(void* h) { return h ? static_cast<FlatPopulation*>(h)->non_diploid.size() : 0; }

int kgxh_flat_copy(void* h, uint8_t* packed, float* info_af, uint8_t* is_snp, uint64_t* offsets) {
  if (!h) return -1;
  const FlatPopulation& f = *static_cast<FlatPopulation*>(h);
  if (packed && f.primary_rows) std::memcpy(packed, f.packed.data(), f.primary_rows * f.row_bytes);     // the split rows are not copied out
  for (size_t v = 0; v < f.primary_rows; ++v) {
    if (info_af) info_af[v] = f.rows[v].info_af;
    if (is_snp) is_snp[v] = f.rows[v].is_snp ? 1 : 0;
    if (offsets) offsets[v] = f.rows[v].offset;
  }
  return 0;
}

// The per-bin split rows (see VariantRow): packed[n_split][row_bytes], info_af[n_split], split_of[n_split]; and
// from_splits[n_primary] = 1 where a primary row's FWS bin counts come from its split rows.
int kgxh_flat_copy_splits(void* h, uint8_t* packed, float* info_af, int64_t* split_of, uint8_t* from_splits) {
  if (!h) return -1;
  const FlatPopulation& f = *static_cast<FlatPopulation*>(h);
  const size_t n_split = f.rows.size() - f.primary_rows;
  if (packed && n_split) std::memcpy(packed, f.packed.data() + f.primary_rows * f.row_bytes, n_split * f.row_bytes);
  for (size_t v = 0; v < n_split; ++v) {
    if (info_af) info_af[v] = f.rows[f.primary_rows + v].info_af;
    if (split_of) split_of[v] = f.rows[f.primary_rows + v].split_of;
  }
  if (from_splits)
    for (size_t v = 0; v < f.primary_rows; ++v) from_splits[v] = f.rows[v].fws_from_splits ? 1 : 0;
  return 0;
}

static void copyOut(const std::string& s, char* buf, size_t n) {
  if (!buf || !n) return;
  const size_t k = s.size() < n - 1 ? s.size() : n - 1;
  std::memcpy(buf, s.data(), k);
  buf[k] = 0;
}

int kgxh_flat_hgvs(void* h, uint64_t i, char* buf, size_t n) {
  if (!h || i >= static_cast<FlatPopulation*>(h)->primary_rows) return -1;
  copyOut(static_cast<FlatPopulation*>(h)->rows[i].hgvs, buf, n);
  return 0;
}

int kgxh_flat_genome_id(void* h, uint64_t i, char* buf, size_t n) {
  if (!h || i >= static_cast<FlatPopulation*>(h)->genome_ids.size()) return -1;
  copyOut(static_cast<FlatPopulation*>(h)->genome_ids[i], buf, n);
  return 0;
}

// File -> text (plain / gzip / block gzip).  Returns a malloc'd buffer the caller frees with kgxh_free, or null.
char* kgxh_read_vcf_text(const char* path, uint64_t* len, int threads, char* error, size_t error_len) {
  std::string text, err;
  if (!path || !kellerberrin::genome::analysis::gpu::readVcfText(path, text, err, threads > 0 ? threads : 0)) {
    copyOut(err, error, error_len);
    return nullptr;
  }
  char* out = static_cast<char*>(std::malloc(text.size() + 1));
  if (!out) return nullptr;
  std::memcpy(out, text.data(), text.size());
  out[text.size()] = 0;
  if (len) *len = text.size();
  return out;
}
void kgxh_free(void* p) { std::free(p); }

// GpuHeteroHomoZygous over counters handed in (no device): record i belongs to genomes[i] x contigs[i], counters[i] =
// { total, snp, indel, hom_minor, het_minor, het_ref_minor, hom_ref }.  Writes the two files of the location analysis.
// 0 = written, -1 = a resource file does not parse, -2 = a file could not be written.
int kgxh_pfemp_location_write(const char* sample_file, const char* fws_file, uint64_t n, const char* const* genomes, const char* const* contigs,
                              const uint64_t* counters, double radius_km, const char* statistics_csv, const char* location_csv) {
  namespace kgl = kellerberrin::genome;
  namespace kga = kellerberrin::genome::analysis;
  if (!sample_file || !fws_file || !statistics_csv || !location_csv || (n && (!genomes || !contigs || !counters))) return -1;
  kgl::ParsePf7Sample sample_parser;
  kgl::ParsePf7Fws fws_parser;
  if (!sample_parser.parsePf7SampleFile(sample_file) || !fws_parser.parsePf7FwsFile(fws_file)) return -1;
  const auto samples = std::make_shared<const kgl::Pf7SampleResource>("Pf7Sample", sample_parser.getPf7SampleVector());
  const auto fws = std::make_shared<const kgl::Pf7FwsResource>("Pf7Fws", fws_parser.getPf7FwsVector());
  kga::GpuHeteroHomoZygous hethom;
  hethom.setResources(samples, fws, std::make_shared<const kgl::Pf7SampleLocation>(*samples));
  for (uint64_t i = 0; i < n; ++i) {
    kga::VariantAnalysisType& r = hethom.analysisMap()[genomes[i]][contigs[i]];
    const uint64_t* c = counters + i * 7;
    r.total_variants_ = c[0];
    r.snp_count_ = c[1];
    r.indel_count_ = c[2];
    r.homozygous_minor_alleles_ = c[3];
    r.heterozygous_minor_alleles_ = c[4];
    r.heterozygous_reference_minor_alleles_ = c[5];
    r.homozygous_reference_alleles_ = c[6];
  }
  const auto summary = hethom.locationSummary(radius_km);
  return hethom.writeSampleResults(statistics_csv, summary) && hethom.writeLocationResults(location_csv, summary) ? 0 : -2;
}

// ---- INBREED inputs from VCF text: reference site file + 1000-Genomes population --------------------------------
struct InbreedInputs {
  kellerberrin::genome::analysis::gpu::FlatReference reference;
  kellerberrin::genome::analysis::gpu::FlatDiploid diploid;
};

void* kgxh_inbreed_inputs(const char* reference_text, uint64_t reference_len, int data_source, const char* diploid_text, uint64_t diploid_len,
                          int threads) {
  if (!reference_text || !diploid_text) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  auto* out = new InbreedInputs();
  out->reference = g::flattenReferenceVcf(std::string_view(reference_text, reference_len), static_cast<kellerberrin::genome::DataSourceEnum>(data_source));
  out->diploid = g::flattenVcf1000Gt8(std::string_view(diploid_text, diploid_len), out->reference, threads > 0 ? threads : 0);
  return out;
}
// The population from a FILE, read diploid_chunk_bytes of text at a time (0 = default); null on an I/O error.  With
// reference_len == 0, reference_text names the reference site FILE, read the same way.
void* kgxh_inbreed_inputs_file(const char* reference_text, uint64_t reference_len, int data_source, const char* diploid_path, int threads,
                               uint64_t diploid_chunk_bytes) {
  if (!reference_text || !diploid_path) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  auto* out = new InbreedInputs();
  std::string error;
  const size_t piece = diploid_chunk_bytes ? static_cast<size_t>(diploid_chunk_bytes) : (size_t{64} << 20);
  if (reference_len == 0) {
    if (!g::flattenReferenceVcfFile(reference_text, static_cast<kellerberrin::genome::DataSourceEnum>(data_source), out->reference, error,
                                    threads > 0 ? threads : 0, piece)) {
      delete out;
      return nullptr;
    }
  } else {
    out->reference = g::flattenReferenceVcf(std::string_view(reference_text, reference_len), static_cast<kellerberrin::genome::DataSourceEnum>(data_source));
  }
  if (!g::flattenVcf1000Gt8File(diploid_path, out->reference, out->diploid, error, threads > 0 ? threads : 0,
                                diploid_chunk_bytes ? static_cast<size_t>(diploid_chunk_bytes) : (size_t{64} << 20))) {
    delete out;
    return nullptr;
  }
  return out;
}
// The population through the streaming flattener (rows leave as their loci are complete) into memory: the same InbreedInputs
// as kgxh_inbreed_inputs_file's, for comparison.  *two_phase = 1 (null returned, why filled): the file has to take the other path.
void* kgxh_inbreed_inputs_file_streaming(const char* reference_text, uint64_t reference_len, int data_source, const char* diploid_path, int threads,
                                         uint64_t diploid_chunk_bytes, int* two_phase, char* why, size_t why_len) {
  if (!reference_text || !diploid_path) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  struct MemorySink final : g::Gt8StreamSink {
    std::vector<uint8_t> bytes;
    uint64_t genomes{0};
    bool open(const std::vector<kellerberrin::genome::GenomeId_t>& genome_ids, uint64_t n_loci) override {
      genomes = genome_ids.size();
      bytes.assign(n_loci * genomes, 0);
      return true;
    }
    bool write(uint64_t first_locus, uint64_t n_loci, const uint8_t* rows) override {
      if ((first_locus + n_loci) * genomes > bytes.size()) return false;
      if (n_loci && genomes) std::memcpy(&bytes[first_locus * genomes], rows, n_loci * genomes);
      return true;
    }
    bool close() override { return true; }
  } sink;
  auto* out = new InbreedInputs();
  out->reference = g::flattenReferenceVcf(std::string_view(reference_text, reference_len), static_cast<kellerberrin::genome::DataSourceEnum>(data_source));
  std::string error;
  bool other_path = false;
  const bool ok = g::flattenVcf1000Gt8FileStreaming(diploid_path, out->reference, sink, out->diploid, error, other_path, threads > 0 ? threads : 0,
                                                    diploid_chunk_bytes ? static_cast<size_t>(diploid_chunk_bytes) : (size_t{64} << 20));
  if (two_phase) *two_phase = other_path ? 1 : 0;
  if (!ok) {
    if (why && why_len) { std::strncpy(why, error.c_str(), why_len - 1); why[why_len - 1] = 0; }
    delete out;
    return nullptr;
  }
  out->diploid.bytes = std::move(sink.bytes);
  return out;
}
void kgxh_inbreed_inputs_destroy(void* h) { delete static_cast<InbreedInputs*>(h); }
uint64_t kgxh_inbreed_loci(void* h) { return h ? static_cast<InbreedInputs*>(h)->reference.loci.size() : 0; }
uint64_t kgxh_inbreed_genomes(void* h) { return h ? static_cast<InbreedInputs*>(h)->diploid.genome_ids.size() : 0; }
uint64_t kgxh_inbreed_max_alts(void* h) { return h ? static_cast<InbreedInputs*>(h)->reference.max_alts : 0; }
uint64_t kgxh_inbreed_contigs(void* h) { return h ? static_cast<InbreedInputs*>(h)->reference.contigs : 0; }
// the loci with more than 14 alts and their 16-bit cells (FlatDiploid::wide_*): count; then loci[n_wide], cells[n_wide][genomes]
uint64_t kgxh_inbreed_wide_loci(void* h) { return h ? static_cast<InbreedInputs*>(h)->diploid.wide_loci.size() : 0; }
int kgxh_inbreed_copy_wide(void* h, uint32_t* loci, uint16_t* cells) {
  if (!h) return -1;
  const auto& d = static_cast<InbreedInputs*>(h)->diploid;
  if (loci && !d.wide_loci.empty()) std::memcpy(loci, d.wide_loci.data(), d.wide_loci.size() * sizeof(uint32_t));
  if (cells && !d.wide_cells.empty()) std::memcpy(cells, d.wide_cells.data(), d.wide_cells.size() * sizeof(uint16_t));
  return 0;
}
int kgxh_inbreed_error(void* h, char* buf, size_t n) {
  if (!h) return -1;
  copyOut(static_cast<InbreedInputs*>(h)->diploid.error, buf, n);
  return 0;
}
// offsets[n_loci], n_alts[n_loci], af[n_loci][amax][6] (NaN = no value), bytes[n_loci][genomes]
int kgxh_inbreed_copy(void* h, uint64_t* offsets, uint32_t* n_alts, double* af, uint32_t amax, uint8_t* bytes) {
  if (!h) return -1;
  const InbreedInputs& in = *static_cast<InbreedInputs*>(h);
  for (size_t l = 0; l < in.reference.loci.size(); ++l) {
    const auto& locus = in.reference.loci[l];
    if (offsets) offsets[l] = locus.offset;
    if (n_alts) n_alts[l] = static_cast<uint32_t>(locus.alts.size());
    if (af)
      for (uint32_t a = 0; a < amax; ++a)
        for (int sp = 0; sp < 6; ++sp) af[(l * amax + a) * 6 + sp] = a < locus.alts.size() ? locus.alts[a].af[sp] : std::nan("");
  }
  if (bytes && !in.diploid.bytes.empty()) std::memcpy(bytes, in.diploid.bytes.data(), in.diploid.bytes.size());
  return 0;
}
int kgxh_inbreed_genome_id(void* h, uint64_t i, char* buf, size_t n) {
  if (!h || i >= static_cast<InbreedInputs*>(h)->diploid.genome_ids.size()) return -1;
  copyOut(static_cast<InbreedInputs*>(h)->diploid.genome_ids[i], buf, n);
  return 0;
}


// ---- the rsid / Ensembl indexes (kgx_variant_sort.h) as text, one entry per line in index order --------------------
// flavour: 0 = mono-genome site file (genome_id names the genome), 1 = phased 1000-Genomes population.
// what: "ensembl" (list = optional '\n'-separated gene codes to keep), "filter" (filterEnsembl of the list),
// "allele_ensembl", "non_ensembl", "id", "genome_id".  Fields are tab separated: key, Variant HGVS_Phase (or the
// ','-joined codes); "genome_id" lines start with the genome.  Returns a malloc'd text the caller frees with kgxh_free.
static std::vector<std::string> splitNames(const char* list) {
  std::vector<std::string> names;
  if (list) {
    std::string item;
    for (const char* c = list;; ++c) {
      if (*c == '\n' || *c == 0) { if (!item.empty()) names.push_back(item); item.clear(); if (*c == 0) break; }
      else item += *c;
    }
  }
  return names;
}

static char* dumpVariantSort(const kellerberrin::genome::analysis::gpu::SortColumns& columns, const char* what, const std::vector<std::string>& names,
                             int threads, std::chrono::steady_clock::time_point t_begin) {
  namespace g = kellerberrin::genome::analysis::gpu;
  const std::string kind(what);
  std::ostringstream out;
  auto dumpEnsembl = [&](const g::EnsemblIndex& index) {
    for (size_t k = 0; k < index.genes().size(); ++k) {
      const auto [first, last] = index.equalRange(index.genes()[k]);
      for (size_t e = first; e < last; ++e) out << index.genes()[k] << '\t' << columns.hgvsPhase(index.variants()[e]) << '\n';
    }
  };
  if (kind == "ensembl") {
    dumpEnsembl(g::VariantSortIndex::ensemblIndex(columns, names));
  } else if (kind == "filter") {
    dumpEnsembl(g::VariantSortIndex::ensemblIndex(columns).filterEnsembl(names));
  } else if (kind == "non_ensembl") {
    out << g::VariantSortIndex::ensemblIndex(columns).nonEnsemblIdentifiers() << '\n';
  } else if (kind == "allele_ensembl") {
    for (const auto& [id, codes] : g::VariantSortIndex::ensemblIndex(columns).alleleEnsemblMap(columns)) {
      out << id << '\t';
      bool first = true;
      for (const auto& code : codes) { out << (first ? "" : ",") << code; first = false; }
      out << '\n';
    }
  } else if (kind == "id") {
    const auto index = g::VariantSortIndex::variantIdIndex(columns);
    for (size_t k = 0; k < index.size(); ++k) out << index.ids()[k] << '\t' << columns.hgvsPhase(index.variants()[k]) << '\n';
  } else if (kind == "timing") {      // build times in ms, no dump: columns, then each index
    auto ms = [](std::chrono::steady_clock::time_point from) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - from).count(); };
    out << "columns\t" << ms(t_begin) << '\n';
    auto t = std::chrono::steady_clock::now();
    const auto ensembl = g::VariantSortIndex::ensemblIndex(columns, names);
    out << "ensembl\t" << ms(t) << '\t' << ensembl.size() << '\n';
    t = std::chrono::steady_clock::now();
    const auto ids = g::VariantSortIndex::variantIdIndex(columns);
    out << "id\t" << ms(t) << '\t' << ids.size() << '\n';
    t = std::chrono::steady_clock::now();
    const auto by_genome = g::VariantSortIndex::variantGenomeIndex(columns, threads > 0 ? threads : 0);
    size_t entries = 0;
    for (size_t gi = 0; gi < by_genome.genomes(); ++gi) entries += by_genome.size(gi);
    out << "genome_id\t" << ms(t) << '\t' << entries << '\n';
  } else if (kind == "genome_id") {
    const auto index = g::VariantSortIndex::variantGenomeIndex(columns, threads > 0 ? threads : 0);
    for (size_t gi = 0; gi < index.genomes(); ++gi)
      for (size_t e = 0; e < index.size(gi); ++e) {
        // every entry must also be found by bisection
        const g::SortVariant* found = index.find(gi, index.id(gi, e));
        if (!found || !(*found == index.variant(gi, e))) return nullptr;
        out << columns.genome_ids[gi] << '\t' << index.id(gi, e) << '\t' << columns.hgvsPhase(index.variant(gi, e)) << '\n';
      }
  } else {
    return nullptr;
  }
  const std::string dump = out.str();
  char* result = static_cast<char*>(std::malloc(dump.size() + 1));
  if (result) std::memcpy(result, dump.c_str(), dump.size() + 1);
  return result;
}

char* kgxh_variant_sort(const char* text, uint64_t len, int flavour, const char* genome_id, const char* what, const char* list, int threads) {
  if (!text || !what) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  const auto t_begin = std::chrono::steady_clock::now();
  const g::SortColumns columns = g::sortColumnsFromVcf(std::string_view(text, len), flavour == 0 ? g::SortVcfFlavour::MonoGenome : g::SortVcfFlavour::Phased1000,
                                                       genome_id ? genome_id : "Reference", threads > 0 ? threads : 0);
  return dumpVariantSort(columns, what, splitNames(list), threads, t_begin);
}

// The same with the VCF read from a file chunk_bytes of text at a time (0 = default); null on an I/O error.
char* kgxh_variant_sort_file(const char* path, int flavour, const char* genome_id, const char* what, const char* list, int threads, uint64_t chunk_bytes) {
  if (!path || !what) return nullptr;
  namespace g = kellerberrin::genome::analysis::gpu;
  const auto t_begin = std::chrono::steady_clock::now();
  g::SortColumns columns;
  std::string error;
  if (!g::sortColumnsFromVcfFile(path, flavour == 0 ? g::SortVcfFlavour::MonoGenome : g::SortVcfFlavour::Phased1000, columns, error,
                                 genome_id ? genome_id : "Reference", threads > 0 ? threads : 0, chunk_bytes ? static_cast<size_t>(chunk_bytes) : (size_t{64} << 20)))
    return nullptr;
  return dumpVariantSort(columns, what, splitNames(list), threads, t_begin);
}

}  // extern "C"
