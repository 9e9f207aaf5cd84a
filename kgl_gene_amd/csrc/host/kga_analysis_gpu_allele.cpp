#include "kga_analysis_gpu_allele.h"

#include <chrono>
#include <algorithm>
#include <fstream>
#include <limits>
#include <mutex>
#include <iterator>

#include "../../../include/kgx.h"
#include "kgx_device_binding.h"
#include "kgx_vcf_io.h"

namespace kga = kellerberrin::genome::analysis;
namespace kgl = kellerberrin::genome;
using kellerberrin::ExecEnv;

namespace {

// Owns a kgx_pop for the duration of one file's sweeps.
struct DevicePopulation {
  kgx_pop* handle{nullptr};
  ~DevicePopulation() { if (handle) kgx_population_destroy(handle); }
};

// The flatteners' rows straight into HBM as they are packed (kgx_flatten.h: RowSink): the population is created when its
// shape is known and every block of rows is uploaded by the thread that packed it, one upload at a time.
struct DeviceRowSink final : kgl::analysis::gpu::RowSink {
  DevicePopulation& dev;
  uint64_t row_bytes{0};
  std::mutex upload;
  std::string error;
  explicit DeviceRowSink(DevicePopulation& d) : dev(d) {}
  bool begin(const kgl::analysis::gpu::FlatPopulation& meta) override {
    row_bytes = meta.row_bytes;
    if (meta.genomes() == 0 || meta.deviceRows() == 0) return true;
    dev.handle = kgx_population_create(meta.genomes(), meta.deviceRows());
    if (!dev.handle) error = std::string("kgx_population_create failed: ") + kgx_last_error();
    return dev.handle != nullptr;
  }
  void rows(uint64_t first_row, uint64_t n_rows, const uint8_t* data) override {
    std::lock_guard<std::mutex> lock(upload);
    if (!dev.handle || !error.empty()) return;
    if (kgx_population_load_dosage2(dev.handle, data, row_bytes, first_row, first_row + n_rows) != KGX_OK)
      error = std::string("upload failed: ") + kgx_last_error();
  }
};

std::string joinPath(const std::string& dir, const std::string& stem) {
  if (dir.empty()) return stem + ".csv";
  return dir + (dir.back() == '/' ? "" : "/") + stem + ".csv";
}

}  // namespace

// ... and the streaming flatteners' rows while the file is still being read (kgx_flatten.h: StreamSink): the population
// grows on the device as pieces arrive (kgx_population_resize re-allocates by half as much again when it must), a row
// that a later record adds copies to is read back, merged on the host and written again.
struct DeviceStreamSink final : kgl::analysis::gpu::StreamSink {
  DevicePopulation& dev;
  uint64_t row_bytes{0}, rows{0};
  std::string error;
  explicit DeviceStreamSink(DevicePopulation& d) : dev(d) {}
  bool fail(const char* what) { error = std::string(what) + ": " + kgx_last_error(); return false; }
  bool open(uint64_t n_genomes, uint64_t bytes_per_row) override {
    row_bytes = bytes_per_row;
    if (n_genomes == 0) return true;
    dev.handle = kgx_population_create(n_genomes, 0);
    return dev.handle ? true : fail("kgx_population_create failed");
  }
  bool write(uint64_t first_row, uint64_t n_rows, const uint8_t* data) override {
    if (!dev.handle) return n_rows == 0;
    if (first_row + n_rows > rows) {
      rows = first_row + n_rows;
      if (kgx_population_resize(dev.handle, rows) != KGX_OK) return fail("growing the device population failed");
    }
    return kgx_population_load_dosage2(dev.handle, data, row_bytes, first_row, first_row + n_rows) == KGX_OK ? true : fail("upload failed");
  }
  bool read(uint64_t row, uint8_t* data) override {
    if (!dev.handle || row >= rows) return false;
    return kgx_population_read_dosage2(dev.handle, data, row_bytes, row, row + 1) == KGX_OK ? true : fail("reading a row back failed");
  }
  bool close(uint64_t n_rows) override {
    if (!dev.handle) return n_rows == 0;
    rows = n_rows;
    return kgx_population_resize(dev.handle, n_rows) == KGX_OK ? true : fail("setting the final row count failed");
  }
};

bool kga::GpuAlleleAnalysis::initializeAnalysis(const std::string& work_directory, const ActiveParameterList& named_parameters,
                                                const std::shared_ptr<const AnalysisResources>& resource_ptr) {
  ExecEnv::log().info("Analysis Id: {} initialized with work directory: {}", ident(), work_directory);
  work_directory_ = work_directory;
  bool quality_filter_given = false;
  for (const auto& [block_name, named_vector] : named_parameters.getMap()) {
    for (const auto& parameter_map : named_vector.second) {
      if (auto v = parameter_map.getString("VariantFile")) variant_file_ = v.value().front();
      if (auto v = parameter_map.getString("GenomeFile")) genome_file_ = v.value().front();
      if (auto v = parameter_map.getString("HetHomFile")) hethom_file_ = v.value().front();
      // "FileNameOnly" VCF data files: which of the reference's parsers the text is for, and whether the PfEMP package's
      // per-record quality filter (P7VariantFilter) runs before the counting
      if (auto v = parameter_map.getString("VcfFlavour")) vcf_flavour_ = v.value().front();
      if (auto v = parameter_map.getBool("Pf7QualityFilter")) { pf7_quality_filter_ = v.value(); quality_filter_given = true; }
      // the genome-level filters of FilterPf7 (kga_analysis_lib_PfFilter.h:58-66) and the location summary's radius
      if (auto v = parameter_map.getString("LocationFile")) location_file_ = v.value().front();
      if (auto v = parameter_map.getBool("Pf7FilterQC")) filter_qc_ = v.value();
      if (auto v = parameter_map.getBool("Pf7FilterFWS")) filter_fws_ = v.value();
      if (auto v = parameter_map.getFloat("Pf7FwsThreshold")) fws_monoclonal_threshold_ = v.value().front();
      if (auto v = parameter_map.getFloat("LocationRadiusKm")) location_radius_km_ = v.value().front();
    }
  }
  // The Pf7 sample resources are optional here (PfEMPAnalysis requires them, kga_analysis_PfEMP.cpp:24-28): both or neither.
  if (resource_ptr) {
    const auto samples = resource_ptr->getResources(ResourceProperties::PF7SAMPLE_RESOURCE_ID_);
    const auto fws = resource_ptr->getResources(ResourceProperties::PF7FWS_RESOURCE_ID_);
    if (samples.size() > 1 || fws.size() > 1 || samples.size() != fws.size()) {
      ExecEnv::log().error("GpuAlleleAnalysis::initializeAnalysis; expected one Pf7Sample and one Pf7Fws resource (or neither), found: {} and {}",
                           samples.size(), fws.size());
      return false;
    }
    if (samples.size() == 1) {
      pf7_sample_ptr_ = std::dynamic_pointer_cast<const Pf7SampleResource>(samples.front());
      pf7_fws_ptr_ = std::dynamic_pointer_cast<const Pf7FwsResource>(fws.front());
      if (!pf7_sample_ptr_ || !pf7_fws_ptr_) {
        ExecEnv::log().error("GpuAlleleAnalysis::initializeAnalysis; invalid Pf7Sample / Pf7Fws resource type");
        return false;
      }
      // as PfEMPAnalysis: FilterPf7::qualityFilter always applies P7VariantFilter (kga_analysis_lib_PfFilter.cpp:63-67)
      if (!quality_filter_given) pf7_quality_filter_ = true;
      const auto physical_distance_ptr = std::make_shared<const Pf7SampleLocation>(*pf7_sample_ptr_);
      hetero_homo_zygous_.setResources(pf7_sample_ptr_, pf7_fws_ptr_, physical_distance_ptr);
      ExecEnv::log().info("GpuAlleleAnalysis; Pf7 sample resources: {} samples, {} FWS values, {} locations; QC filter: {}, monoclonal FWS filter: {} (>= {})",
                          pf7_sample_ptr_->getMap().size(), pf7_fws_ptr_->getMap().size(), physical_distance_ptr->locationMap().size(),
                          filter_qc_ ? "on" : "off", filter_fws_ ? "on" : "off", fws_monoclonal_threshold_);
    }
  }
  std::string binding, binding_error;
  if (!gpu::bindDevices(named_parameters, binding, binding_error)) {
    // No CPU fallback: the analysis is disabled, other packages continue (kgl_package_analysis.cpp:41-42).
    ExecEnv::log().error("GpuAlleleAnalysis::initializeAnalysis; cannot bind the MI355X devices: {}", binding_error);
    return false;
  }
  ExecEnv::log().info("GpuAlleleAnalysis; genomes sharded over {}", binding);
  device_ready_ = true;
  return true;
}

bool kga::GpuAlleleAnalysis::fileReadAnalysis(std::shared_ptr<const DataDB> data_object_ptr) {
  ExecEnv::log().info("Analysis: {}, begin processing data file: {}", ident(), data_object_ptr->fileId());
  if (!device_ready_) {
    ExecEnv::log().error("GpuAlleleAnalysis::fileReadAnalysis; no device bound");
    return false;
  }
  const auto file_characteristic = data_object_ptr->dataCharacteristic();
  if (std::dynamic_pointer_cast<const FilenameDataDB>(data_object_ptr) || file_characteristic.data_structure == DataStructureEnum::NoStructure)
    return sweepVcfFile(data_object_ptr->fileId());                   // a "FileNameOnly" data file: the package reads the VCF itself
  if (file_characteristic.data_structure != DataStructureEnum::DiploidPhased &&
      file_characteristic.data_structure != DataStructureEnum::DiploidUnphased) {
    ExecEnv::log().info("Analysis: {}, file: {} is not a diploid population; ignored", ident(), data_object_ptr->fileId());
    return true;
  }
  auto population = std::dynamic_pointer_cast<const PopulationDB>(data_object_ptr);
  if (!population) {
    ExecEnv::log().error("GpuAlleleAnalysis::fileReadAnalysis; Analysis: {}, file: {} is not a PopulationDB", ident(), data_object_ptr->fileId());
    return false;
  }
  return sweepPopulation(*population);
}

bool kga::GpuAlleleAnalysis::sweepPopulation(const PopulationDB& population) {
  // K1: what VariantDBVariant::createVariantDB does for the CPU path (kgl_variant_db_variant.cpp:11-123: variant index,
  // genome index, one dosage row per genome) -- here the variant-major 2-bit rows, then their upload (sweepFlat)
  const auto flatten_begin = std::chrono::steady_clock::now();
  const gpu::FlatPopulation flat = gpu::flattenPopulation(population);
  if (flat.cells_with_three_phases)
    ExecEnv::log().warn("GpuAlleleAnalysis; {} (genome, variant) cells hold copies on three or more distinct phases: the phase plane says \"more than one\" "
                        "there, UniquePhasedFilter counts would be one short per extra phase", flat.cells_with_three_phases);
  k1_flatten_seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - flatten_begin).count();
  // contigs a genome holds without any variant still get a (zero) record (heterozygous.cpp:38-41)
  for (const auto& [genome_id, genome_ptr] : population.getMap()) {
    if (!keepGenome(genome_id)) continue;                           // the genome-level Pf7 filters: absent from every result
    auto& contig_map = hetero_homo_zygous_.analysisMap()[genome_id];
    for (const auto& [contig_id, contig_ptr] : genome_ptr->getMap()) contig_map.try_emplace(contig_id);
  }
  // A PopulationDB is counted as delivered: the per-record P7VariantFilter is the caller's viewFilter (the "FileNameOnly"
  // entry applies it itself, Pf7QualityFilter).
  return sweepFlat(flat, population.populationId());
}

bool kga::GpuAlleleAnalysis::sweepVcfFile(const std::string& file_name) {
  // plain text, .gz or .bgz, read and flattened a bounded piece at a time: the text never has to fit in memory
  std::string io_error;
  gpu::FlatPopulation flat;
  if (vcf_flavour_ != "Genome1000" && vcf_flavour_ != "Falciparum") {
    ExecEnv::log().error("GpuAlleleAnalysis; unknown VcfFlavour: {} (Genome1000 or Falciparum)", vcf_flavour_);
    return false;
  }
  // First the streaming flattener: every piece's rows go to the device while the next piece is read, the host never
  // holds more than a piece.  A file it cannot take (a sample named twice, a sample without any variant) goes through the
  // two-phase flattener instead, whose packed rows leave block by block once every row's place is known.
  DevicePopulation dev;
  bool two_phase = false;
  {
    DeviceStreamSink stream(dev);
    const bool streamed = vcf_flavour_ == "Genome1000"
                              ? gpu::flattenVcf1000FileStreaming(file_name, stream, flat, io_error, two_phase)
                              : gpu::flattenVcfPfFileStreaming(file_name, stream, flat, io_error, two_phase, 0, pf7_quality_filter_);
    if (!streamed && !two_phase) {
      ExecEnv::log().error("GpuAlleleAnalysis; {}{}", io_error, stream.error.empty() ? std::string() : " (" + stream.error + ")");
      return false;
    }
  }
  if (two_phase) {
    ExecEnv::log().warn("GpuAlleleAnalysis; {}: flattened in two phases ({})", file_name, io_error);
    if (dev.handle) { kgx_population_destroy(dev.handle); dev.handle = nullptr; }
    io_error.clear();
    flat = gpu::FlatPopulation{};
    DeviceRowSink sink(dev);
    const bool read_ok = vcf_flavour_ == "Genome1000" ? gpu::flattenVcf1000File(file_name, flat, io_error, 0, size_t{64} << 20, &sink)
                                                      : gpu::flattenVcfPfFile(file_name, flat, io_error, 0, pf7_quality_filter_, size_t{64} << 20, &sink);
    if (!read_ok) {
      ExecEnv::log().error("GpuAlleleAnalysis; {}", io_error);
      return false;
    }
    if (!sink.error.empty()) {
      ExecEnv::log().error("GpuAlleleAnalysis; {}", sink.error);
      return false;
    }
  }
  if (vcf_flavour_ == "Genome1000") return sweepFlat(flat, file_name, dev.handle);
  // every genome holds every contig of the header, carrier or not (PfVCFImpl::setupPopulationStructure): zero records
  for (const auto& genome_id : flat.genome_ids) {
    if (!keepGenome(genome_id)) continue;
    auto& contig_map = hetero_homo_zygous_.analysisMap()[genome_id];
    for (const auto& contig_id : flat.contig_ids) contig_map.try_emplace(contig_id);
  }
  return sweepFlat(flat, file_name, dev.handle);
}

// uploaded: the population already on the device (its rows arrived through a DeviceRowSink), or null: create it here and
// upload flat.packed, a bounded block of rows at a time.
bool kga::GpuAlleleAnalysis::sweepFlat(const gpu::FlatPopulation& flat, const std::string& label, kgx_pop* uploaded) {
  // V: the population's distinct variants; D >= V: rows on the device (the extra ones are per-bin splits, see VariantRow)
  const uint64_t G = flat.genomes(), V = flat.variants(), D = flat.deviceRows();
  ExecEnv::log().info("GpuAlleleAnalysis; population: {}, genomes: {}, distinct variants: {}, Variant objects: {}",
                      label, G, V, flat.variant_objects);
  if (G == 0) return true;
  // FilterPf7::qualityFilter's genome part (kga_analysis_lib_PfFilter.cpp:26-58): the genomes that pass QC and are
  // monoclonal take part, the others are absent from everything below -- a genome mask on the device population.
  const bool masked = genomeFilterActive();
  std::vector<uint8_t> keep(G, 1);
  if (masked) {
    uint64_t kept = 0;
    for (uint64_t g = 0; g < G; ++g) kept += keep[g] = keepGenome(flat.genome_ids[g]) ? 1 : 0;
    ExecEnv::log().info("GpuAlleleAnalysis; Pf7 genome filters (QC pass: {}, FWS >= {}: {}) keep {} of {} genomes", filter_qc_ ? "on" : "off",
                        fws_monoclonal_threshold_, filter_fws_ ? "on" : "off", kept, G);
  }
  for (uint64_t g = 0; g < G; ++g) {
    if (!keep[g]) continue;
    genome_fws_map_.try_emplace(flat.genome_ids[g], GpuFwsFrequencyArray());
    hetero_homo_zygous_.analysisMap().try_emplace(flat.genome_ids[g]);
  }
  if (V == 0) return true;

  DevicePopulation owned;
  struct { kgx_pop* handle; } dev{uploaded};
  if (!dev.handle) {
    const auto upload_begin = std::chrono::steady_clock::now();
    owned.handle = kgx_population_create(G, D);
    dev.handle = owned.handle;
    if (!dev.handle) {
      ExecEnv::log().error("GpuAlleleAnalysis; kgx_population_create failed: {}", kgx_last_error());
      return false;
    }
    const uint64_t block = std::max<uint64_t>(1, (uint64_t{256} << 20) / std::max<uint64_t>(1, flat.row_bytes));   // 256 MiB of rows per copy
    for (uint64_t r0 = 0; r0 < D; r0 += block) {
      const uint64_t r1 = std::min(D, r0 + block);
      if (kgx_population_load_dosage2(dev.handle, flat.packed.data() + r0 * flat.row_bytes, flat.row_bytes, r0, r1) != KGX_OK) {
        ExecEnv::log().error("GpuAlleleAnalysis; upload failed: {}", kgx_last_error());
        return false;
      }
    }
    (void)kgx_synchronize();
    const double upload_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - upload_begin).count();
    ExecEnv::log().info("GpuAlleleAnalysis; K1 (createVariantDB's work): flatten {} s, device rows created and uploaded {} s, {} genomes x {} rows",
                        k1_flatten_seconds_, upload_seconds, G, D);
  }

  if (masked && kgx_population_set_genome_mask(dev.handle, keep.data()) != KGX_OK) {
    ExecEnv::log().error("GpuAlleleAnalysis; setting the genome mask failed: {}", kgx_last_error());
    return false;
  }

  // ---- K2: CalcFWS::updateVariantFWSMap -- summaryByVariant for every variant ------------------
  std::vector<uint32_t> by_variant(D * 4);
  if (kgx_allele_count_by_locus(dev.handle, by_variant.data()) != KGX_OK) {
    ExecEnv::log().error("GpuAlleleAnalysis; allele count sweep failed: {}", kgx_last_error());
    return false;
  }
  for (uint64_t v = 0; v < V; ++v) {
    // no carrier among the genomes taking part: the filtered population does not hold the variant
    if (masked && (by_variant[v * 4 + 1] | by_variant[v * 4 + 2] | by_variant[v * 4 + 3]) == 0) continue;
    AlleleSummmary summary;
    summary.referenceHomozygous_ = by_variant[v * 4 + 0];
    summary.minorHeterozygous_ = by_variant[v * 4 + 1];
    summary.minorHomozygous_ = by_variant[v * 4 + 2];
    if (by_variant[v * 4 + 3] != 0)   // the reference's warning (kgl_variant_db_variant.cpp:158-161,168-174)
      ExecEnv::log().warn("GpuAlleleAnalysis; Variant: {} has {} genomes with a non diploid allele count", flat.rows[v].hgvs, by_variant[v * 4 + 3]);
    variant_fws_map_[flat.rows[v].hgvs] += summary;
  }

  // ---- K3: CalcFWS::updateGenomeFWSMap over the 11 allele-frequency bins -----------------------
  {
    // The P7FrequencyFilter pair of every bin is evaluated on the device, from the AF of each row's own record; a row whose
    // copies sit in several bins takes no part itself (no value), its split rows do.
    std::vector<float> fws_af(D);
    for (uint64_t v = 0; v < D; ++v) fws_af[v] = flat.rows[v].fws_from_splits ? std::numeric_limits<float>::quiet_NaN() : flat.rows[v].info_af;
    std::vector<double> bin_edges(gpu::FWS_FREQUENCY_ARRAY_SIZE + 1);
    for (size_t b = 0; b < gpu::FWS_FREQUENCY_ARRAY_SIZE; ++b) {
      bin_edges[b] = gpu::fwsBinRange(b).first;
      bin_edges[b + 1] = gpu::fwsBinRange(b).second;
    }
    std::vector<uint64_t> by_genome(G * gpu::FWS_FREQUENCY_ARRAY_SIZE * 4);
    if (kgx_population_set_af(dev.handle, fws_af.data()) != KGX_OK ||
        kgx_count_by_genome_af_bins(dev.handle, bin_edges.data(), gpu::FWS_FREQUENCY_ARRAY_SIZE, by_genome.data()) != KGX_OK) {
      ExecEnv::log().error("GpuAlleleAnalysis; by-genome sweep failed: {}", kgx_last_error());
      return false;
    }
    for (uint64_t g = 0; g < G; ++g) {
      if (!keep[g]) continue;
      auto& freq_array = genome_fws_map_[flat.genome_ids[g]];
      for (size_t b = 0; b < gpu::FWS_FREQUENCY_ARRAY_SIZE; ++b) {
        const uint64_t* c = &by_genome[(g * gpu::FWS_FREQUENCY_ARRAY_SIZE + b) * 4];
        AlleleSummmary summary;
        summary.referenceHomozygous_ = c[0];
        summary.minorHeterozygous_ = c[1];
        summary.minorHomozygous_ = c[2];
        freq_array[b] += summary;
      }
    }
  }

  // ---- K3 + K8: HeteroHomoZygous::updateVariantAnalysisType per genome x contig ---------------
  {
    std::map<ContigId_t, uint32_t> contig_index;
    for (const auto& row : flat.rows) contig_index.try_emplace(row.contig, 0);
    uint32_t n_contigs = 0;
    std::vector<ContigId_t> contig_ids;
    for (auto& [contig_id, index] : contig_index) { index = n_contigs++; contig_ids.push_back(contig_id); }
    // Offsets holding >= 2 distinct variants, as lists of their rows: adjacent in a population flattened in HGVS order,
    // anywhere in one streamed in file order (the Pf flavour's canonical variants need not start where their record does).
    std::vector<uint32_t> member_rows, first_member, n_rows, group_bin;
    std::vector<uint8_t> compound(V, 0);
    {
      std::map<std::pair<uint32_t, uint64_t>, std::vector<uint32_t>> rows_of_offset;
      for (uint64_t v = 0; v < V; ++v) rows_of_offset[{contig_index.at(flat.rows[v].contig), flat.rows[v].offset}].push_back(static_cast<uint32_t>(v));
      for (const auto& [key, rows] : rows_of_offset) {
        if (rows.size() < 2) continue;
        first_member.push_back(static_cast<uint32_t>(member_rows.size()));
        n_rows.push_back(static_cast<uint32_t>(rows.size()));
        group_bin.push_back(key.first);
        for (uint32_t r : rows) { member_rows.push_back(r); compound[r] = 1; }
      }
    }
    // bin = contig*4 + is_snp*2 + compound: dosage-weighted totals per class of row.  One sweep holds 254 bins, 63 contigs;
    // a population with more (an assembly with its unplaced scaffolds) is swept once per 63 of them.
    constexpr uint32_t kContigsPerSweep = 63;
    const uint32_t n_bins = n_contigs * 4;
    std::vector<uint64_t> by_genome(G * n_bins * 4);
    std::vector<uint8_t> bin_of_variant(D);
    std::vector<uint64_t> sweep_counts;
    for (uint32_t c0 = 0; c0 < n_contigs; c0 += kContigsPerSweep) {
      const uint32_t c1 = std::min(c0 + kContigsPerSweep, n_contigs), sweep_bins = (c1 - c0) * 4;
      std::fill(bin_of_variant.begin(), bin_of_variant.end(), static_cast<uint8_t>(0xFF));    // split rows, and the other sweeps' contigs, take no part
      for (uint64_t v = 0; v < V; ++v) {
        const uint32_t c = contig_index.at(flat.rows[v].contig);
        if (c >= c0 && c < c1) bin_of_variant[v] = static_cast<uint8_t>((c - c0) * 4 + (flat.rows[v].is_snp ? 2 : 0) + compound[v]);
      }
      uint64_t* counts = by_genome.data();
      if (sweep_bins != n_bins) {
        sweep_counts.assign(G * sweep_bins * 4, 0);
        counts = sweep_counts.data();
      }
      if (kgx_count_by_genome_binned(dev.handle, bin_of_variant.data(), sweep_bins, counts) != KGX_OK) {
        ExecEnv::log().error("GpuAlleleAnalysis; by-genome (contig) sweep failed: {}", kgx_last_error());
        return false;
      }
      if (sweep_bins != n_bins)
        for (uint64_t g = 0; g < G; ++g)
          std::copy_n(&sweep_counts[g * sweep_bins * 4], static_cast<size_t>(sweep_bins) * 4, &by_genome[(g * n_bins + c0 * 4) * 4]);
    }
    std::vector<uint64_t> compound_counts(G * n_contigs * 3);
    if (kgx_compound_offsets_listed(dev.handle, member_rows.data(), member_rows.size(), first_member.data(), n_rows.data(), group_bin.data(),
                                    first_member.size(), n_contigs, compound_counts.data()) != KGX_OK) {
      ExecEnv::log().error("GpuAlleleAnalysis; compound offset sweep failed: {}", kgx_last_error());
      return false;
    }
    // exact copy numbers of the (rare) non-diploid cells: code 3 counted them as nothing in het/hom
    std::vector<uint64_t> extra_total(G * n_contigs, 0), extra_snp(G * n_contigs, 0);
    for (const auto& cell : flat.non_diploid) {
      const uint32_t c = contig_index.at(flat.rows[cell.row].contig);
      extra_total[cell.genome * n_contigs + c] += cell.dosage;
      if (flat.rows[cell.row].is_snp) extra_snp[cell.genome * n_contigs + c] += cell.dosage;
    }
    for (uint64_t g = 0; g < G; ++g) {
      if (!keep[g]) continue;
      auto& contig_map = hetero_homo_zygous_.analysisMap()[flat.genome_ids[g]];
      for (uint32_t c = 0; c < n_contigs; ++c) {
        VariantAnalysisType record;
        uint64_t copies_snp = 0, copies_indel = 0;
        for (uint32_t cls = 0; cls < 4; ++cls) {
          const uint64_t* k = &by_genome[(g * n_bins + c * 4 + cls) * 4];   // refHom, het, hom, nonDiploid
          const uint64_t copies = k[1] + 2 * k[2];
          if (cls & 2) copies_snp += copies; else copies_indel += copies;
          if ((cls & 1) == 0) {   // offsets with a single distinct variant: 1 copy -> (A;a), >= 2 copies -> one unique "homozygous" alt
            record.heterozygous_reference_minor_alleles_ += k[1];
            record.homozygous_minor_alleles_ += k[2] + k[3];
          }
        }
        const uint64_t x_total = extra_total[g * n_contigs + c], x_snp = extra_snp[g * n_contigs + c];
        record.snp_count_ += copies_snp + x_snp;
        record.indel_count_ += copies_indel + (x_total - x_snp);
        record.total_variants_ += copies_snp + copies_indel + x_total;
        const uint64_t* k8 = &compound_counts[(g * n_contigs + c) * 3];
        record.heterozygous_reference_minor_alleles_ += k8[0];
        record.homozygous_minor_alleles_ += k8[1];
        record.heterozygous_minor_alleles_ += k8[2];
        // homozygous_reference_alleles_ is never incremented by the reference.
        // A genome holds a contig only if it carries a variant there (or the parser created it up front): no carrier,
        // no pre-existing record -> no record, as when the reference walks the genome's own contig map.
        auto found = contig_map.find(contig_ids[c]);
        if (found == contig_map.end()) {
          if (record.total_variants_ == 0) continue;
          found = contig_map.try_emplace(contig_ids[c]).first;
        }
        VariantAnalysisType& sum = found->second;
        sum.total_variants_ += record.total_variants_;
        sum.snp_count_ += record.snp_count_;
        sum.indel_count_ += record.indel_count_;
        sum.heterozygous_reference_minor_alleles_ += record.heterozygous_reference_minor_alleles_;
        sum.homozygous_minor_alleles_ += record.homozygous_minor_alleles_;
        sum.heterozygous_minor_alleles_ += record.heterozygous_minor_alleles_;
      }
    }
  }
  return true;
}

bool kga::GpuAlleleAnalysis::iterationAnalysis() {
  ExecEnv::log().info("Iteration Analysis called for Analysis Id: {}", ident());
  return true;
}

bool kga::GpuAlleleAnalysis::finalizeAnalysis() {
  ExecEnv::log().info("Finalize Analysis called for Analysis Id: {}", ident());
  bool ok = writeVariantResults(joinPath(work_directory_, variant_file_));
  ok = writeGenomeResults(joinPath(work_directory_, genome_file_)) && ok;
  if (pf7_sample_ptr_) {
    // PfEMPAnalysis::finalizeAnalysis (kga_analysis_PfEMP.cpp:146-163)
    const GpuLocationSummaryMap location_summary = hetero_homo_zygous_.locationSummary(location_radius_km_);
    ok = hetero_homo_zygous_.writeSampleResults(joinPath(work_directory_, hethom_file_), location_summary) && ok;
    ok = hetero_homo_zygous_.writeLocationResults(joinPath(work_directory_, location_file_), location_summary) && ok;
  } else {
    ok = hetero_homo_zygous_.writeContigResults(joinPath(work_directory_, hethom_file_)) && ok;
  }
  return ok;
}

// Column layout of CalcFWS::writeVariantResults (kga_analysis_PfEMP_FWS.cpp:235-309).
bool kga::GpuAlleleAnalysis::writeVariantResults(const std::string& file_name) const {
  std::ofstream out(file_name);
  if (!out.good()) {
    ExecEnv::log().error("GpuAlleleAnalysis::writeVariantResults; Unable to open results file: {}", file_name);
    return false;
  }
  out << "Variant" << CSV_DELIMITER_ << "Hom/Het" << CSV_DELIMITER_ << "Minor Hom/Het" << CSV_DELIMITER_ << "Genome Count"
      << CSV_DELIMITER_ << "Hom Ref (A;A)" << CSV_DELIMITER_ << "Het Ref Minor (A;a)" << CSV_DELIMITER_ << "Hom Minor (a;a)" << '\n';
  for (const auto& [hgvs, s] : variant_fws_map_) {
    const size_t het = s.minorHeterozygous_;
    const double hom_het = het > 0 ? static_cast<double>(s.minorHomozygous_ + s.referenceHomozygous_) / static_cast<double>(het) : 0.0;
    const double minor_hom_het = het > 0 ? static_cast<double>(s.minorHomozygous_) / static_cast<double>(het) : 0.0;
    out << hgvs << CSV_DELIMITER_ << hom_het << CSV_DELIMITER_ << minor_hom_het << CSV_DELIMITER_ << (s.minorHeterozygous_ + s.minorHomozygous_)
        << CSV_DELIMITER_ << s.referenceHomozygous_ << CSV_DELIMITER_ << s.minorHeterozygous_ << CSV_DELIMITER_ << s.minorHomozygous_ << '\n';
  }
  return out.good();
}

// Column layout of CalcFWS::writeGenomeResults (kga_analysis_PfEMP_FWS.cpp:147-232) without the Pf7 FWS join.
bool kga::GpuAlleleAnalysis::writeGenomeResults(const std::string& file_name) const {
  std::ofstream out(file_name);
  if (!out.good()) {
    ExecEnv::log().error("GpuAlleleAnalysis::writeGenomeResults; Unable to open results file: {}", file_name);
    return false;
  }
  out << "Genome";
  for (size_t i = 0; i < gpu::FWS_FREQUENCY_ARRAY_SIZE; ++i)
    out << CSV_DELIMITER_ << "LowerFreq" << CSV_DELIMITER_ << "UpperFreq" << CSV_DELIMITER_ << "Hom/Het" << CSV_DELIMITER_ << "Minor Hom/Het"
        << CSV_DELIMITER_ << "Variant Count" << CSV_DELIMITER_ << "Hom Ref (A;A)" << CSV_DELIMITER_ << "Het Ref Minor (A;a)"
        << CSV_DELIMITER_ << "Hom Minor (a;a)";
  out << '\n';
  for (const auto& [genome_id, freq_array] : genome_fws_map_) {
    out << genome_id;
    for (size_t i = 0; i < gpu::FWS_FREQUENCY_ARRAY_SIZE; ++i) {
      const AlleleSummmary& s = freq_array[i];
      const size_t het = s.minorHeterozygous_;
      const double hom_het = het > 0 ? static_cast<double>(s.minorHomozygous_ + s.referenceHomozygous_) / static_cast<double>(het) : 0.0;
      const double minor_hom_het = het > 0 ? static_cast<double>(s.minorHomozygous_) / static_cast<double>(het) : 0.0;
      const auto [lower_range, upper_range] = gpu::fwsBinRange(i);
      out << CSV_DELIMITER_ << lower_range << CSV_DELIMITER_ << upper_range << CSV_DELIMITER_ << hom_het << CSV_DELIMITER_ << minor_hom_het
          << CSV_DELIMITER_ << (s.minorHeterozygous_ + s.minorHomozygous_) << CSV_DELIMITER_ << s.referenceHomozygous_
          << CSV_DELIMITER_ << s.minorHeterozygous_ << CSV_DELIMITER_ << s.minorHomozygous_;
    }
    out << '\n';
  }
  return out.good();
}
