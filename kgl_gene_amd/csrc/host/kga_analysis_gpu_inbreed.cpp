#include "kga_analysis_gpu_inbreed.h"

#include <algorithm>
#include <future>
#include <atomic>
#include <thread>
#include <cmath>
#include <fstream>
#include <functional>
#include <iomanip>
#include <limits>
#include <sstream>
#include <unordered_map>

#include "../../../include/kgx.h"
#include "kgx_device_binding.h"
#include "kgx_vcf_io.h"

namespace kga = kellerberrin::genome::analysis;
namespace kgl = kellerberrin::genome;
using kellerberrin::ExecEnv;

namespace {

constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();

struct DeviceMatrix {
  kgx_gt8* handle{nullptr};
  ~DeviceMatrix() { if (handle) kgx_gt8_destroy(handle); }
};

int algorithmCode(const std::string& name) {   // InbreedingCalculation::algoMap keys (kga_analysis_inbreed_calc.h:93-101)
  if (name == "RitlandLocus") return KGX_ALGO_RITLAND_LOCUS;
  if (name == "Simple") return KGX_ALGO_SIMPLE;
  if (name == "HallME") return KGX_ALGO_HALL_ME;
  if (name == "Loglikelihood") return KGX_ALGO_LOGLIKELIHOOD;
  return -1;
}

}  // namespace

// ---- parameters ----------------------------------------------------------------------------------------

std::vector<kga::GpuInbreedingParameters> kga::GpuInbreedAnalysis::extractParameters(const ActiveParameterList& named_parameters) {
  std::vector<GpuInbreedingParameters> param_vector;
  for (const auto& [block_name, named_vector] : named_parameters.getMap()) {
    for (const auto& xml_vector : named_vector.second) {
      GpuInbreedingParameters p;
      p.parameter_ident = block_name;
      auto analysis_opt = xml_vector.getBool("AnalysisType");
      auto output_opt = xml_vector.getString("OutputFile");
      auto algo_opt = xml_vector.getString("Algorithm");
      auto min_opt = xml_vector.getFloat("MinAlleleFreq");
      auto max_opt = xml_vector.getFloat("MaxAlleleFreq");
      auto low_opt = xml_vector.getSize("LowerWindow");
      auto high_opt = xml_vector.getSize("UpperWindow");
      auto count_opt = xml_vector.getSize("LociiCount");
      auto spacing_opt = xml_vector.getSize("SamplingDistance");
      if (!analysis_opt || !output_opt || !algo_opt || !min_opt || !max_opt || !low_opt || !high_opt || !count_opt || !spacing_opt) {
        ExecEnv::log().error("GpuInbreedAnalysis::extractParameters; bad or missing value in parameter block: {}", block_name);
        continue;
      }
      p.analyze_synthetic = analysis_opt.value();
      p.output_file = output_opt.value().front();
      p.inbreeding_algorithm = algo_opt.value().front();
      p.locii.allele_frequency_min = std::clamp(min_opt.value().front(), 0.0, 1.0);
      p.locii.allele_frequency_max = std::clamp(max_opt.value().front(), 0.0, 1.0);
      p.locii.lower_offset = low_opt.value().front();
      p.locii.upper_offset = high_opt.value().front();
      p.locii.locii_count = count_opt.value().front();
      p.locii.spacing = spacing_opt.value().front();
      param_vector.push_back(p);
    }
    ExecEnv::log().info("Inbreeding Analysis, parsed named argument vector: {}, size: {}", block_name, param_vector.size());
  }
  return param_vector;
}

// ---- reference contig ----------------------------------------------------------------------------------

kga::GpuReferenceContig kga::GpuInbreedAnalysis::buildReference(const PopulationDB& unphased_population, bool& ok) {
  GpuReferenceContig ref;
  ok = false;
  if (unphased_population.getMap().size() != 1) {
    ExecEnv::log().error("GpuInbreedAnalysis; Unphased Population: {} has unexpected Genome count: {}", unphased_population.populationId(),
                         unphased_population.getMap().size());
    return ref;
  }
  const auto& genome_ptr = unphased_population.getMap().begin()->second;
  if (genome_ptr->getMap().size() != 1) {
    ExecEnv::log().error("GpuInbreedAnalysis; Unphased Population: {} has more than 1 contig: {}", unphased_population.populationId(),
                         genome_ptr->getMap().size());
    return ref;
  }
  const auto& [contig_id, contig_ptr] = *genome_ptr->getMap().begin();
  ref.contig_id = contig_id;
  const auto& super_pops = FrequencyDatabaseRead::superPopulations();
  for (const auto& [offset, offset_ptr] : contig_ptr->getMap()) {
    GpuReferenceLocus locus;
    locus.offset = offset;
    for (const auto& variant_ptr : offset_ptr->getVariantArray()) {
      if (!variant_ptr->isSNP() || !variant_ptr->evidence().passFilter()) continue;   // AndFilter(SNPFilter(), PassFilter())
      GpuReferenceAlt alt;
      alt.hgvs = variant_ptr->HGVS();
      for (size_t sp = 0; sp < 6 && sp < super_pops.size(); ++sp) {
        auto f = FrequencyDatabaseRead::superPopFrequency(*variant_ptr, super_pops[sp]);
        alt.af[sp] = f ? f.value() : kNaN;
      }
      locus.alts.push_back(std::move(alt));
    }
    if (locus.alts.empty()) continue;    // trimEmpty (kgl_variant_db_contig.cpp:135-148)
    ref.max_alts = std::max<uint32_t>(ref.max_alts, static_cast<uint32_t>(locus.alts.size()));
    ref.loci.push_back(std::move(locus));
  }
  ok = true;
  return ref;
}

uint32_t kga::GpuReferenceContig::narrowAlts() const {
  uint32_t most = 1;
  for (const auto& locus : loci)
    if (locus.alts.size() <= kMaxAlts) most = std::max<uint32_t>(most, static_cast<uint32_t>(locus.alts.size()));
  return most;
}

size_t kga::GpuReferenceContig::limitAlts(uint32_t kMost) {
  size_t lossy = 0;
  if (max_alts <= kMost) return lossy;
  max_alts = 0;
  for (auto& locus : loci) {
    if (locus.alts.size() > kMost) {
      auto bears_frequency = [](const GpuReferenceAlt& alt) {
        for (const double f : alt.af)
          if (!std::isnan(f)) return true;
        return false;
      };
      const auto first_without = std::stable_partition(locus.alts.begin(), locus.alts.end(), bears_frequency);
      if (static_cast<size_t>(first_without - locus.alts.begin()) > kMost) ++lossy;
      locus.alts.resize(kMost);
    }
    max_alts = std::max<uint32_t>(max_alts, static_cast<uint32_t>(locus.alts.size()));
  }
  return lossy;
}

void kga::GpuReferenceContig::alleleFreqRow(size_t l, int sp, double* row, uint32_t amax) const {
  const auto& alts = loci[l].alts;
  for (uint32_t j = 0; j < amax; ++j) row[j] = kNaN;
  for (size_t j = 0; j < alts.size() && j < amax; ++j) {
    const double f = alts[j].af[sp];
    if (std::isnan(f)) continue;
    bool duplicate = false;
    for (size_t k = 0; k < j; ++k)
      if (!std::isnan(row[k]) && alts[k].hgvs == alts[j].hgvs) { duplicate = true; break; }
    if (!duplicate) row[j] = std::clamp(f, 0.0, 1.0);
  }
}

bool kga::GpuReferenceContig::validForSampling(size_t l, int sp, double& minor_sum) const {
  double row[16];
  const uint32_t amax = std::min<uint32_t>(16, std::max<uint32_t>(1, static_cast<uint32_t>(loci[l].alts.size())));
  alleleFreqRow(l, sp, row, amax);
  double sum = 0.0;
  size_t n = 0;
  for (uint32_t j = 0; j < amax; ++j)
    if (!std::isnan(row[j])) { sum += row[j]; ++n; }
  if ((sum - 1.0) > 1.0e-5 || n == 0) return false;     // checkValidAlleleVector
  minor_sum = std::clamp(sum, 0.0, 1.0);                // minorAlleleFrequencies
  return true;
}

std::vector<uint32_t> kga::GpuReferenceContig::sampleLocii(int sp, const GpuLociiArguments& args, bool by_count) const {
  std::vector<uint32_t> out;
  auto it = std::lower_bound(loci.begin(), loci.end(), args.lower_offset,
                             [](const GpuReferenceLocus& locus, ContigOffset_t v) { return locus.offset < v; });
  ContigOffset_t previous_offset{0};
  for (; it != loci.end(); ++it) {
    const ContigOffset_t offset = it->offset;
    if (by_count ? out.size() >= args.locii_count : offset > args.upper_offset) break;
    if (offset >= previous_offset + args.spacing || previous_offset == 0) {
      double sum_frequencies = 0.0;
      const size_t l = static_cast<size_t>(it - loci.begin());
      if (!validForSampling(l, sp, sum_frequencies)) continue;
      if (sum_frequencies == 0.0 || sum_frequencies < args.allele_frequency_min || sum_frequencies > args.allele_frequency_max) continue;
      previous_offset = offset;
      out.push_back(static_cast<uint32_t>(l));
    }
  }
  return out;
}

// ---- VirtualAnalysis -----------------------------------------------------------------------------------

bool kga::GpuInbreedAnalysis::initializeAnalysis(const std::string& work_directory, const ActiveParameterList& named_parameters,
                                                 const std::shared_ptr<const AnalysisResources>& resource_ptr) {
  ExecEnv::log().info("Analysis Id: {} initialized with work directory: {}", ident(), work_directory);
  genealogy_data_ = resource_ptr->getSingleResource<HsGenomeGenealogyData>(ResourceProperties::GENEALOGY_RESOURCE_ID_);
  work_directory_ = work_directory;
  for (const auto& [block_name, named_vector] : named_parameters.getMap())
    for (const auto& parameter_map : named_vector.second)
    {
      if (auto v = parameter_map.getSize("SyntheticSeed")) synthetic_seed_ = v.value().front();
      if (auto v = parameter_map.getSize("StartSeed")) start_seed_ = v.value().front();
      if (auto v = parameter_map.getSize("WindowBatch")) window_batch_ = std::max<size_t>(1, v.value().front());
      if (auto v = parameter_map.getString("StartPoints")) start_midpoints_ = v.value().front() == "Midpoint";
    }
  for (const auto& parameter : extractParameters(named_parameters)) {
    GpuParamOutput out;
    out.parameters = parameter;
    parameter_output_vector_.push_back(std::move(out));
  }
  std::string binding, binding_error;
  if (!gpu::bindDevices(named_parameters, binding, binding_error)) {
    ExecEnv::log().error("GpuInbreedAnalysis::initializeAnalysis; cannot bind the MI355X devices: {}", binding_error);
    return false;
  }
  ExecEnv::log().info("GpuInbreedAnalysis; genomes sharded over {}", binding);
  device_ready_ = true;
  return true;
}

bool kga::GpuInbreedAnalysis::fileReadAnalysis(std::shared_ptr<const DataDB> data_object_ptr) {
  ExecEnv::log().info("Analysis: {}, begin processing data file: {}", ident(), data_object_ptr->fileId());
  const auto file_characteristic = data_object_ptr->dataCharacteristic();
  if (std::dynamic_pointer_cast<const FilenameDataDB>(data_object_ptr)) {
    // a VCF the package reads itself: the 1000-Genomes population, or a mono-genome frequency source
    if (data_object_ptr->dataSource() == DataSourceEnum::Genome1000) diploid_vcf_ = data_object_ptr->fileId();
    else if (file_characteristic.data_structure == DataStructureEnum::UnphasedMonoGenome) {
      reference_vcf_ = data_object_ptr->fileId();
      reference_vcf_source_ = data_object_ptr->dataSource();
    } else {
      ExecEnv::log().error("GpuInbreedAnalysis::fileReadAnalysis, Analysis: {}, VCF file: {} is neither a Genome1000 population nor a frequency source", ident(), data_object_ptr->fileId());
      return false;
    }
    return true;
  }
  if (file_characteristic.data_structure == DataStructureEnum::DiploidPhased ||
      file_characteristic.data_structure == DataStructureEnum::DiploidUnphased) {
    diploid_population_ = std::dynamic_pointer_cast<const PopulationDB>(data_object_ptr);
    if (!diploid_population_) {
      ExecEnv::log().error("GpuInbreedAnalysis::fileReadAnalysis, Analysis: {}, file: {} is not a Diploid Population", ident(), data_object_ptr->fileId());
      return false;
    }
  }
  if (file_characteristic.data_structure == DataStructureEnum::UnphasedMonoGenome) {
    unphased_population_ = std::dynamic_pointer_cast<const PopulationDB>(data_object_ptr);
    if (!unphased_population_) {
      ExecEnv::log().error("GpuInbreedAnalysis::fileReadAnalysis, Analysis: {}, file: {} is not an Unphased Population", ident(), data_object_ptr->fileId());
      return false;
    }
  }
  return true;
}

bool kga::GpuInbreedAnalysis::iterationAnalysis() {
  ExecEnv::log().info("Iteration Analysis called for Analysis Id: {}", ident());
  if (!device_ready_) return false;
  bool ok = true;
  for (auto& param_output : parameter_output_vector_) {
    // ExecuteInbreedingAnalysis::executeAnalysis (kga_analysis_inbreed_execute.cpp:16-45)
    if (param_output.parameters.analyze_synthetic) {
      if (!haveReference()) {
        ExecEnv::log().error("InbreedAnalysis::iterationAnalysis; Insufficient data, cannot process synthetic diploid inbreeding");
        ok = false;
        continue;
      }
      ok = syntheticInbreeding(param_output) && ok;
    } else {
      if (!haveDiploid() || !haveReference() || !genealogy_data_) {
        ExecEnv::log().error("ExecuteInbreedingAnalysis::processDiploid; Insufficient data, cannot process diploid inbreeding");
        ok = false;
        continue;
      }
      ok = populationInbreeding(param_output) && ok;
    }
  }
  diploid_population_ = nullptr;
  unphased_population_ = nullptr;
  reference_vcf_.clear();
  diploid_vcf_.clear();
  return ok;
}

bool kga::GpuInbreedAnalysis::referenceInput(GpuReferenceContig& reference, uint32_t most) const {
  const bool ok = referenceSource(reference);
  // An offset may hold any number of SNP alts (AlleleFreqVector has no cap): up to 14 fit the matrix bytes, up to 254 a wide row
  // (`most`: 254 for the population sweep; the synthetic self-check draws its genomes on the device from 4-bit indices: 14).
  if (ok && reference.max_alts > most) {
    const uint32_t widest = reference.max_alts;
    const size_t lossy = reference.limitAlts(most);
    ExecEnv::log().warn("GpuInbreedAnalysis; a reference offset holds {} SNP alts, {} fit the allele index: offsets cut to their first {} "
                        "frequency-bearing alts, {} of them lost a frequency-bearing alt (its carriers count as carriers of an unknown alt)",
                        widest, most, most, lossy);
  }
  return ok;
}

bool kga::GpuInbreedAnalysis::referenceSource(GpuReferenceContig& reference) const {
  if (unphased_population_) {
    bool ok = false;
    reference = buildReference(*unphased_population_, ok);
    return ok;
  }
  std::string io_error;
  gpu::FlatReference flat;                                           // plain text, .gz or .bgz, a bounded piece at a time
  if (!gpu::flattenReferenceVcfFile(reference_vcf_, reference_vcf_source_, flat, io_error)) {
    ExecEnv::log().error("GpuInbreedAnalysis; reference VCF: {}", io_error);
    return false;
  }
  if (flat.contigs != 1) {
    ExecEnv::log().error("InbreedingAnalysis::populationInbreeding; Unphased Population: {} has unexpected contig count: {}", reference_vcf_, flat.contigs);
    return false;
  }
  reference.contig_id = std::move(flat.contig_id);
  reference.max_alts = flat.max_alts;
  reference.loci = std::move(flat.loci);
  return true;
}

// Allele-index bytes [locus][genome]: each genome's SNP variants at each reference offset, in OffsetDB order.
kgl::analysis::gpu::FlatDiploid kga::GpuInbreedAnalysis::diploidBytes(const PopulationDB& diploid_population, const GpuReferenceContig& reference,
                                                                        bool phased) {
  gpu::FlatDiploid out;
  const uint64_t n_loci = reference.loci.size();
  out.n_loci = n_loci;
  std::vector<std::shared_ptr<const ContigDB>> contigs;
  for (const auto& [genome_id, genome_ptr] : diploid_population.getMap()) {
    auto contig_opt = genome_ptr->getContig(reference.contig_id);
    if (!contig_opt) continue;
    out.genome_ids.push_back(genome_id);
    contigs.push_back(contig_opt.value());
  }
  const uint64_t G = out.genome_ids.size();
  std::unordered_map<ContigOffset_t, uint32_t> locus_of_offset;
  locus_of_offset.reserve(n_loci * 2);
  for (uint32_t l = 0; l < n_loci; ++l) locus_of_offset.emplace(reference.loci[l].offset, l);
  out.bytes.assign(n_loci * G, 0);
  // offsets with more than 14 alts: 16-bit cells beside the bytes (gpu::FlatDiploid::wide_*), their byte rows 0xFF throughout
  std::vector<int64_t> wide_row_of(n_loci, -1);
  for (uint32_t l = 0; l < n_loci; ++l)
    if (reference.isWide(l)) {
      wide_row_of[l] = static_cast<int64_t>(out.wide_loci.size());
      out.wide_loci.push_back(l);
      std::fill(&out.bytes[static_cast<uint64_t>(l) * G], &out.bytes[static_cast<uint64_t>(l) * G] + G, 0xFF);
    }
  out.wide_cells.assign(out.wide_loci.size() * G, 0);
  for (uint64_t g = 0; g < G; ++g) {
    for (const auto& [offset, offset_ptr] : contigs[g]->getMap()) {
      auto lit = locus_of_offset.find(offset);
      if (lit == locus_of_offset.end()) continue;
      const auto& alts = reference.loci[lit->second].alts;
      const bool wide = wide_row_of[lit->second] >= 0;
      const uint32_t unknown = wide ? 255u : 15u;                     // an alt the reference list does not hold
      uint32_t n = 0, code[2] = {0, 0};
      VariantPhase phase[2] = {VariantPhase::UNPHASED, VariantPhase::UNPHASED};
      for (const auto& variant_ptr : offset_ptr->getVariantArray()) {
        if (!variant_ptr->isSNP()) continue;                         // contig_ptr->viewFilter(SNPFilter()) (_freq.cpp:436)
        if (n < 2) {
          const std::string hgvs = variant_ptr->HGVS();
          uint32_t c = unknown;
          for (size_t j = 0; j < alts.size(); ++j)
            if (alts[j].hgvs == hgvs) { c = static_cast<uint32_t>(j + 1); break; }
          code[n] = c;
          phase[n] = variant_ptr->phaseId();
        }
        ++n;
      }
      if (n == 0) continue;
      const uint32_t bits = wide ? 8u : 4u;
      uint32_t cell;
      if (n >= 3) cell = wide ? 0xFFFFu : 0xFFu;
      else {
        // two copies of one variant on ONE phase (a repeated record): analogous, not homozygous -> the (0, a) cell
        if (n == 2 && code[0] == code[1] && code[0] != unknown && phased && phase[0] == phase[1]) cell = code[0] << bits;
        else cell = code[0] | (code[1] << bits);
      }
      if (wide) out.wide_cells[static_cast<uint64_t>(wide_row_of[lit->second]) * G + g] = static_cast<uint16_t>(cell);
      else out.bytes[static_cast<uint64_t>(lit->second) * G + g] = static_cast<uint8_t>(cell);
    }
  }
  return out;
}

bool kga::GpuInbreedAnalysis::diploidInput(const GpuReferenceContig& reference, gpu::FlatDiploid& diploid, bool& phased) const {
  if (diploid_population_) {
    phased = diploid_population_->dataCharacteristic().data_structure == DataStructureEnum::DiploidPhased;
    diploid = diploidBytes(*diploid_population_, reference, phased);
  } else {
    phased = true;                                                   // Genome1000: DiploidPhased (kgl_data_file_type.h:118-134)
    gpu::FlatReference flat;
    flat.contig_id = reference.contig_id;
    flat.loci = reference.loci;
    std::string io_error;                                            // the file is read a bounded piece at a time: its text need not fit in memory
    if (!gpu::flattenVcf1000Gt8File(diploid_vcf_, flat, diploid, io_error)) {
      ExecEnv::log().error("GpuInbreedAnalysis; population VCF: {}", io_error);
      return false;
    }
  }
  if (!diploid.error.empty()) {
    ExecEnv::log().error("GpuInbreedAnalysis; {}", diploid.error);
    return false;
  }
  return true;
}

bool kga::GpuInbreedAnalysis::populationInbreeding(GpuParamOutput& param_output) {
  const GpuInbreedingParameters& params = param_output.parameters;
  const int algorithm = algorithmCode(params.inbreeding_algorithm);
  if (algorithm < 0) {
    ExecEnv::log().error("InbreedingAnalysis::populationInbreeding, Inbreeding algorithm not found: {}", params.inbreeding_algorithm);
    return true;   // the reference logs and returns an empty results map (_diploid.cpp:107-112)
  }
  GpuReferenceContig reference;
  if (!referenceInput(reference, GpuReferenceContig::kMaxWideAlts)) return false;
  // the frequency table's columns: the most alts of an offset that fits a matrix byte -- a window that samples a WIDE offset (more
  // than 14 alts, rare) gets a table as wide as that offset and is swept on its own, by the generic kernels
  const uint32_t amax = reference.narrowAlts();
  const uint64_t n_loci = reference.loci.size();
  const auto& super_pops = FrequencyDatabaseRead::superPopulations();
  gpu::FlatDiploid diploid;
  bool phased = true;

  // Genomes with the contig and a PED record, grouped by super population; each group starts on a multiple of 16.
  struct DeviceGenome { uint64_t column; GenomeId_t id; uint64_t stream; };   // stream: its place in processResults' fan-out
  std::vector<std::vector<DeviceGenome>> by_super_pop(super_pops.size());
  std::vector<uint64_t> range_begin(super_pops.size(), 0), range_end(super_pops.size(), 0);
  uint64_t device_genomes = 0;
  std::vector<int64_t> column_of;                               // device column -> column of the flattened input (genome-id order)
  bool planned_before = false;                                  // a second plan (two-phase after a streaming attempt) repeats no message
  auto planColumns = [&](const std::vector<GenomeId_t>& genome_ids) {
    const bool log_missing = !planned_before;
    planned_before = true;
    for (auto& group : by_super_pop) group.clear();
    uint64_t enqueued = 0;                                       // future_vector.size() at the genome's turn (_diploid.cpp:117-148)
    for (uint64_t column = 0; column < genome_ids.size(); ++column) {
      const GenomeId_t& genome_id = genome_ids[column];
      auto record_opt = genealogy_data_->getGenomeGenealogyRecord(genome_id);
      if (!record_opt) {
        if (log_missing) ExecEnv::log().error("InbreedingAnalysis::populationInbreeding, Genome sample: {} does not have a PED record", genome_id);
        continue;
      }
      const auto sp_it = std::find(super_pops.begin(), super_pops.end(), record_opt.value().superPopulation());
      if (sp_it == super_pops.end()) {
        if (log_missing) ExecEnv::log().error("InbreedingAnalysis::populationInbreeding, Locus set not found for super population: {}", record_opt.value().superPopulation());
        continue;
      }
      by_super_pop[static_cast<size_t>(sp_it - super_pops.begin())].push_back({column, genome_id, enqueued++});
    }
    device_genomes = 0;
    for (size_t sp = 0; sp < super_pops.size(); ++sp) {
      device_genomes = (device_genomes + 15) / 16 * 16;   // 16-genome boundary: the widest sweep kernel applies
      range_begin[sp] = device_genomes;
      device_genomes += by_super_pop[sp].size();
      range_end[sp] = device_genomes;
    }
    // Device column order = super-population groups; the flattened input is in genome-id order.
    column_of.assign(device_genomes, -1);                        // -1: padding up to the next group's 16-genome boundary
    for (size_t sp = 0; sp < super_pops.size(); ++sp)
      for (size_t k = 0; k < by_super_pop[sp].size(); ++k) column_of[range_begin[sp] + k] = static_cast<int64_t>(by_super_pop[sp][k].column);
  };
  // rows of the flattened input (genome-id order, input_genomes wide) into device column order
  auto permuteRows = [&](const uint8_t* in, uint64_t input_genomes, uint64_t n_rows, uint8_t* out) {
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
      for (uint64_t begin = next.fetch_add(1024); begin < n_rows; begin = next.fetch_add(1024))
        for (uint64_t l = begin; l < std::min<uint64_t>(n_rows, begin + 1024); ++l) {
          const uint8_t* from = in + l * input_genomes;
          uint8_t* row = out + l * device_genomes;
          for (uint64_t g = 0; g < device_genomes; ++g) row[g] = column_of[g] >= 0 ? from[column_of[g]] : 0;
        }
    };
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::max(2u, std::thread::hardware_concurrency()) - 1, (n_rows + 1023) / 1024));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  };

  DeviceMatrix dev;
  bool on_device = false;
  if (!diploid_population_) {
    // A population VCF file: its rows go to the device as their loci are complete (flattenVcf1000Gt8FileStreaming), the
    // host holds a piece of the file at a time.  A file that cannot be streamed takes the two-phase flattener below.
    struct DeviceSink final : gpu::Gt8StreamSink {
      std::function<void(const std::vector<GenomeId_t>&)> plan;
      std::function<void(const uint8_t*, uint64_t, uint64_t, uint8_t*)> permute;
      DeviceMatrix* dev{nullptr};
      const uint64_t* device_genomes{nullptr};
      uint64_t input_genomes{0};
      std::vector<uint8_t> block;
      std::string error;
      bool open(const std::vector<GenomeId_t>& genome_ids, uint64_t loci) override {
        plan(genome_ids);
        input_genomes = genome_ids.size();
        if (*device_genomes == 0 || loci == 0) return true;
        dev->handle = kgx_gt8_create(*device_genomes, loci);
        if (!dev->handle) error = std::string("kgx_gt8_create failed: ") + kgx_last_error();
        return dev->handle != nullptr;
      }
      bool write(uint64_t first_locus, uint64_t n, const uint8_t* rows) override {
        if (!dev->handle) return true;
        block.resize(n * *device_genomes);
        permute(rows, input_genomes, n, block.data());
        if (kgx_gt8_load_rows(dev->handle, block.data(), *device_genomes, first_locus, first_locus + n) == KGX_OK) return true;
        error = std::string("genotype upload failed: ") + kgx_last_error();
        return false;
      }
      bool close() override { return true; }
    } sink;
    sink.plan = planColumns;
    sink.permute = permuteRows;
    sink.dev = &dev;
    sink.device_genomes = &device_genomes;
    gpu::FlatReference flat;
    flat.contig_id = reference.contig_id;
    flat.loci = reference.loci;
    std::string io_error;
    bool two_phase = false;
    if (gpu::flattenVcf1000Gt8FileStreaming(diploid_vcf_, flat, sink, diploid, io_error, two_phase)) {
      on_device = true;
    } else if (two_phase) {
      ExecEnv::log().warn("GpuInbreedAnalysis; {}: flattened in two phases ({})", diploid_vcf_, io_error);
      if (dev.handle) { kgx_gt8_destroy(dev.handle); dev.handle = nullptr; }
      diploid = gpu::FlatDiploid{};
    } else {
      ExecEnv::log().error("GpuInbreedAnalysis; population VCF: {}{}", io_error, sink.error.empty() ? std::string() : " (" + sink.error + ")");
      return false;
    }
  }
  if (!on_device) {
    if (!diploidInput(reference, diploid, phased)) return false;
    planColumns(diploid.genome_ids);
  }
  if (device_genomes == 0 || n_loci == 0) return true;
  if (!on_device) {
    std::vector<uint8_t> bytes(n_loci * device_genomes, 0);
    permuteRows(diploid.bytes.data(), diploid.genome_ids.size(), n_loci, bytes.data());
    diploid.bytes.clear();
    diploid.bytes.shrink_to_fit();
    dev.handle = kgx_gt8_create(device_genomes, n_loci);
    if (!dev.handle || kgx_gt8_load_rows(dev.handle, bytes.data(), device_genomes, 0, n_loci) != KGX_OK) {
      ExecEnv::log().error("GpuInbreedAnalysis; genotype upload failed: {}", kgx_last_error());
      return false;
    }
  }
  // the offsets with more than 14 alts: their 16-bit rows, in the device's column order (every flattener leaves them in `diploid`)
  if (!diploid.wide_loci.empty()) {
    const uint64_t input_genomes = diploid.genome_ids.size(), n_wide = diploid.wide_loci.size();
    std::vector<uint16_t> cells(n_wide * device_genomes, 0);
    for (uint64_t w = 0; w < n_wide; ++w)
      for (uint64_t g = 0; g < device_genomes; ++g)
        if (column_of[g] >= 0) cells[w * device_genomes + g] = diploid.wide_cells[w * input_genomes + static_cast<uint64_t>(column_of[g])];
    if (kgx_gt8_set_wide_rows(dev.handle, n_wide, diploid.wide_loci.data(), cells.data(), device_genomes) != KGX_OK) {
      ExecEnv::log().error("GpuInbreedAnalysis; upload of the wide rows failed: {}", kgx_last_error());
      return false;
    }
    ExecEnv::log().warn("GpuInbreedAnalysis; {} reference offsets hold more than 14 SNP alts (the widest {}): their cells are kept as 16-bit rows and the "
                        "windows that sample them are swept on their own", n_wide, reference.max_alts);
  }

  // The window loop of InbreedingAnalysis::populationInbreeding (_diploid.cpp:43-75).  Windows are independent once sampled --
  // a window's bounds follow from the reference contig alone, its super populations differ in their frequency rows and
  // genome range -- so WindowBatch windows (default 16) are sampled ahead on the host and go to the device together
  // (kgx_inbreed_batch: one copy in, two launches, one copy out for all their (window, super population) tasks), and the
  // next batch is sampled while the device works on this one.
  const int all_slot = static_cast<int>(std::find(super_pops.begin(), super_pops.end(), std::string(FrequencyDatabaseRead::SUPER_POP_ALL_)) - super_pops.begin());
  GpuLociiArguments local = params.locii;
  std::vector<uint32_t> locii_vector = reference.sampleLocii(all_slot, local, true);
  if (locii_vector.empty()) return true;
  local.upper_offset = reference.loci[locii_vector.back()].offset;
  struct PendingTask { size_t sp; uint32_t amax; std::vector<uint32_t> selected; std::vector<double> af, start; std::vector<kgx_locus_results> results; };
  struct PendingWindow { std::string column_ident; std::vector<PendingTask> tasks; };
  std::vector<uint64_t> streams;
  bool start_failed = false;
  // the window at `local`, then `local` moved on to the next; false once the reference's loop would have ended
  auto nextWindow = [&](PendingWindow& window) -> bool {
    if (!(local.upper_offset < params.locii.upper_offset && locii_vector.size() >= 100)) return false;
    {
      std::stringstream ss;
      ss << reference.contig_id << "_" << local.lower_offset << "_" << local.upper_offset;
      window.column_ident = ss.str();
    }
    for (size_t sp = 0; sp < super_pops.size(); ++sp) {
      const uint64_t n = range_end[sp] - range_begin[sp];
      if (n == 0) continue;
      PendingTask task;
      task.sp = sp;
      task.selected = reference.sampleLocii(static_cast<int>(sp), local, false);   // getLocusList (_locus.cpp:263-326)
      task.amax = amax;
      for (const uint32_t l : task.selected)
        if (reference.isWide(l)) task.amax = std::max<uint32_t>(task.amax, static_cast<uint32_t>(reference.loci[l].alts.size()));
      task.af.assign(task.selected.size() * task.amax, kNaN);
      for (size_t s = 0; s < task.selected.size(); ++s) reference.alleleFreqRow(task.selected[s], static_cast<int>(sp), &task.af[s * task.amax], task.amax);
      task.results.assign(n, kgx_locus_results{});
      streams.resize(n);
      for (uint64_t k = 0; k < n; ++k) streams[k] = by_super_pop[sp][k].stream;
      task.start = startPoints(algorithm, streams);
      const bool wants_start = !start_midpoints_ && (algorithm == KGX_ALGO_HALL_ME || algorithm == KGX_ALGO_LOGLIKELIHOOD);
      if (wants_start && task.start.empty()) start_failed = true;
      window.tasks.push_back(std::move(task));
    }
    local.lower_offset = local.upper_offset;
    locii_vector = reference.sampleLocii(all_slot, local, true);
    if (locii_vector.empty()) local.upper_offset = params.locii.upper_offset;      // (the loop's `break`)
    else local.upper_offset = reference.loci[locii_vector.back()].offset;
    return true;
  };
  auto sampleBatch = [&](std::vector<PendingWindow>& batch) {
    batch.clear();
    while (batch.size() < window_batch_) {
      PendingWindow window;
      if (!nextWindow(window)) break;
      batch.push_back(std::move(window));
    }
  };
  auto runBatch = [&](std::vector<PendingWindow>& batch) -> std::string {      // on the device; "" = fine
    std::vector<kgx_inbreed_task> tasks;
    for (PendingWindow& window : batch)
      for (PendingTask& task : window.tasks) {
        if (task.amax != amax) {                                    // a window over a wide offset: on its own
          if (kgx_inbreed(dev.handle, range_begin[task.sp], range_end[task.sp], task.selected.data(), task.selected.size(), task.af.data(), task.amax,
                          phased ? 1 : 0, algorithm, task.start.empty() ? nullptr : task.start.data(), task.results.data()) != KGX_OK)
            return kgx_last_error();
          continue;
        }
        kgx_inbreed_task t;
        t.g0 = range_begin[task.sp];
        t.g1 = range_end[task.sp];
        t.locus_index = task.selected.data();
        t.n_selected = task.selected.size();
        t.minor_af = task.af.data();
        t.start = task.start.empty() ? nullptr : task.start.data();
        t.out = task.results.data();
        tasks.push_back(t);
      }
    if (tasks.empty()) return std::string();
    if (kgx_inbreed_batch(dev.handle, tasks.data(), static_cast<uint32_t>(tasks.size()), amax, phased ? 1 : 0, algorithm) != KGX_OK) return kgx_last_error();
    return std::string();
  };
  std::vector<PendingWindow> current, ahead;
  sampleBatch(current);
  while (!current.empty()) {
    if (start_failed) return false;
    // (kgx_last_error is the calling thread's: the worker hands its message back)
    std::future<std::string> on_device = std::async(std::launch::async, [&]() { return runBatch(current); });
    sampleBatch(ahead);
    const std::string failure = on_device.get();
    if (!failure.empty()) {
      ExecEnv::log().error("GpuInbreedAnalysis; inbreeding sweep failed: {}", failure);
      return false;
    }
    for (PendingWindow& window : current) {
      GpuResultColumn column;
      column.column_ident = window.column_ident;
      for (const PendingTask& task : window.tasks)
        for (uint64_t k = 0; k < task.results.size(); ++k) {
          const kgx_locus_results& d = task.results[k];
          GpuLocusResults r;
          r.genome = by_super_pop[task.sp][k].id;
          r.major_hetero_count = d.major_hetero_count;  r.major_hetero_freq = d.major_hetero_freq;
          r.minor_hetero_count = d.minor_hetero_count;  r.minor_hetero_freq = d.minor_hetero_freq;
          r.minor_homo_count = d.minor_homo_count;      r.minor_homo_freq = d.minor_homo_freq;
          r.major_homo_count = d.major_homo_count;      r.major_homo_freq = d.major_homo_freq;
          r.total_allele_count = d.total_allele_count;  r.inbred_allele_sum = d.inbred_allele_sum;
          column.results[r.genome] = r;
        }
      param_output.columns.push_back(std::move(column));
    }
    current.swap(ahead);
  }
  return !start_failed;
}

std::vector<double> kga::GpuInbreedAnalysis::startPoints(int algorithm, const std::vector<uint64_t>& streams) const {
  std::vector<double> start;
  if (start_midpoints_ || (algorithm != KGX_ALGO_HALL_ME && algorithm != KGX_ALGO_LOGLIKELIHOOD)) return start;
  start.resize(streams.size());
  if (start_seed_ == 0) {                                  // fresh entropy for every task of every window, as in the reference
    if (!streams.empty() && kgx_inbreed_reference_starts(algorithm, 0, 0, streams.size(), start.data()) != KGX_OK) {
      ExecEnv::log().error("GpuInbreedAnalysis; start points: {}", kgx_last_error());
      return {};
    }
    return start;
  }
  // Seeded: the k-th task of every window owns the same stream, so its start is drawn once (seeding a twister per genome
  // per window would cost more than the window's sweep).
  std::vector<double>& known = seeded_starts_[algorithm == KGX_ALGO_HALL_ME ? 0 : 1];
  uint64_t needed = 0;
  for (const uint64_t stream : streams) needed = std::max(needed, stream + 1);
  if (needed > known.size()) {
    const uint64_t have = known.size();
    known.resize(needed);
    if (kgx_inbreed_reference_starts(algorithm, start_seed_, have, needed - have, known.data() + have) != KGX_OK) {
      ExecEnv::log().error("GpuInbreedAnalysis; start points: {}", kgx_last_error());
      known.resize(have);
      return {};
    }
  }
  for (size_t k = 0; k < streams.size(); ++k) start[k] = known[streams[k]];
  return start;
}

// ---- synthetic self-check ------------------------------------------------------------------------------

kgl::GenomeId_t kga::GpuInbreedAnalysis::generateSyntheticGenomeId(double inbreeding, const std::string& super_population, size_t counter) {
  std::stringstream genome_id_stream;
  genome_id_stream << std::fixed << std::setprecision(0);
  if (inbreeding >= 0) genome_id_stream << super_population << "_" << (inbreeding * 1000000.0) << "_" << counter;
  else genome_id_stream << super_population << "_N" << (-inbreeding * 1000000.0) << "_" << counter;
  return genome_id_stream.str();
}

std::pair<bool, double> kga::GpuInbreedAnalysis::generateInbreeding(const GenomeId_t& genome_id) {
  bool valid_value = false, negative = false;
  double inbreed_coefficient = -1000.0;
  auto first_pos = genome_id.find_first_of("N");
  if (first_pos == std::string::npos) first_pos = genome_id.find_first_of("_");
  else negative = true;
  if (first_pos != std::string::npos) {
    ++first_pos;
    const auto second_pos = genome_id.find_first_of("_", first_pos);
    if (second_pos != std::string::npos) {
      try {
        const size_t coefficient = static_cast<size_t>(std::stod(genome_id.substr(first_pos, second_pos - first_pos)));
        inbreed_coefficient = static_cast<double>(coefficient) / 1000000.0;
        valid_value = true;
        if (negative) inbreed_coefficient = -1.0 * inbreed_coefficient;
      } catch (std::exception&) {
      }
    }
  }
  return {valid_value, inbreed_coefficient};
}

bool kga::GpuInbreedAnalysis::syntheticInbreeding(GpuParamOutput& param_output) {
  const GpuInbreedingParameters& params = param_output.parameters;
  const int algorithm = algorithmCode(params.inbreeding_algorithm);
  if (algorithm < 0) {
    ExecEnv::log().error("InbreedingAnalysis::populationInbreeding, Inbreeding algorithm not found: {}", params.inbreeding_algorithm);
    return true;
  }
  GpuReferenceContig reference;
  if (!referenceInput(reference, GpuReferenceContig::kMaxAlts)) return false;
  const uint32_t amax = std::max<uint32_t>(1, reference.max_alts);
  const auto& super_pops = FrequencyDatabaseRead::superPopulations();
  const int all_slot = static_cast<int>(std::find(super_pops.begin(), super_pops.end(), std::string(FrequencyDatabaseRead::SUPER_POP_ALL_)) - super_pops.begin());

  // The inbreeding grid of generateSyntheticPopulation (_syngen.cpp:44-56): repeated addition, as the reference does.
  std::vector<double> grid;
  for (double inbreeding = -0.5; inbreeding <= (0.5 + 0.000001); inbreeding += 0.01) grid.push_back(inbreeding);

  // One column's worth of work: SyntheticAnalysis::processSynResults uses the ORIGINAL window of the parameter block
  // (_synthetic.cpp:53), so every column of a block analyses the same locus lists; only the draws differ in the
  // reference.  Here the draws are keyed by (seed, column).
  auto process = [&](uint64_t column, GpuResultsMap& results_map) -> bool {
    for (size_t sp = 0; sp < super_pops.size(); ++sp) {
      // processResults walks the synthetic population in genome-id (std::map) order: the grid genome's place in it
      std::vector<GenomeId_t> ids(grid.size());
      for (size_t g = 0; g < grid.size(); ++g) ids[g] = generateSyntheticGenomeId(grid[g], super_pops[sp], g);
      std::vector<size_t> by_id(grid.size());
      for (size_t g = 0; g < grid.size(); ++g) by_id[g] = g;
      std::sort(by_id.begin(), by_id.end(), [&](size_t a, size_t b) { return ids[a] < ids[b]; });
      std::vector<uint64_t> streams(grid.size());
      for (size_t rank = 0; rank < by_id.size(); ++rank) streams[by_id[rank]] = rank;
      const std::vector<double> start = startPoints(algorithm, streams);
      const std::vector<uint32_t> selected = reference.sampleLocii(static_cast<int>(sp), params.locii, false);
      std::vector<double> af_table(selected.size() * amax, kNaN);
      for (size_t s = 0; s < selected.size(); ++s) reference.alleleFreqRow(selected[s], static_cast<int>(sp), &af_table[s * amax], amax);
      DeviceMatrix dev;
      dev.handle = kgx_gt8_create(grid.size(), selected.size());
      std::vector<kgx_locus_results> device_results(grid.size());
      if (!dev.handle ||
          kgx_gt8_synth_inbred(dev.handle, af_table.data(), amax, grid.data(), synthetic_seed_ + 1000003ull * column + sp) != KGX_OK ||
          kgx_inbreed(dev.handle, 0, grid.size(), nullptr, selected.size(), af_table.data(), amax, 1, algorithm,
                      start.empty() ? nullptr : start.data(), device_results.data()) != KGX_OK) {
        ExecEnv::log().error("GpuInbreedAnalysis; synthetic sweep failed: {}", kgx_last_error());
        return false;
      }
      for (size_t g = 0; g < grid.size(); ++g) {
        const kgx_locus_results& d = device_results[g];
        GpuLocusResults r;
        r.genome = ids[g];
        r.major_hetero_count = d.major_hetero_count;  r.major_hetero_freq = d.major_hetero_freq;
        r.minor_hetero_count = d.minor_hetero_count;  r.minor_hetero_freq = d.minor_hetero_freq;
        r.minor_homo_count = d.minor_homo_count;      r.minor_homo_freq = d.minor_homo_freq;
        r.major_homo_count = d.major_homo_count;      r.major_homo_freq = d.major_homo_freq;
        r.total_allele_count = d.total_allele_count;  r.inbred_allele_sum = d.inbred_allele_sum;
        results_map[r.genome] = r;
      }
    }
    return true;
  };

  GpuLociiArguments local = params.locii;
  std::vector<uint32_t> locii_vector = reference.sampleLocii(all_slot, local, true);
  if (locii_vector.empty()) return true;
  local.upper_offset = reference.loci[locii_vector.back()].offset;
  uint64_t column_index = 0;
  while (local.upper_offset < params.locii.upper_offset && locii_vector.size() >= 100) {
    GpuResultColumn column;
    std::stringstream ss;
    ss << reference.contig_id << "_" << local.lower_offset << "_" << local.upper_offset;
    column.column_ident = ss.str();
    if (!process(column_index++, column.results)) return false;
    param_output.columns.push_back(std::move(column));
    local.lower_offset = local.upper_offset;
    locii_vector = reference.sampleLocii(all_slot, local, true);
    if (locii_vector.empty()) break;
    local.upper_offset = reference.loci[locii_vector.back()].offset;
  }
  return true;
}

bool kga::GpuInbreedAnalysis::finalizeAnalysis() {
  ExecEnv::log().info("Finalize called for Analysis Id: {}", ident());
  (void)kgx_release_scratch();            // the sweep's arena is not needed past the last iteration
  return writeResults();
}

// InbreedingOutput::writeSynthetic / writePedResults / writeNoPedResults (kga_analysis_inbreed_output.cpp:85-395), chosen as
// InbreedAnalysis::finalizeAnalysis chooses (kga_analysis_inbreed.cpp:134-150): the same header lines, the same columns
// (Sample, Population, Description, SuperPopulation, Description, Relationship, Sex, Mother, Father, then one per window),
// the same default stream formatting and the trailing delimiter of every data row -- the files diff clean against the
// reference's.  Beside it "<OutputFile>_detail.csv" (not a reference file) carries every LocusResults field at full precision.
bool kga::GpuInbreedAnalysis::writeResults() const {
  for (const auto& param_output : parameter_output_vector_) {
    if (param_output.columns.empty()) {
      ExecEnv::log().error("InbreedingOutput::writePedResults; No results to output for parameter ident: {}", param_output.parameters.parameter_ident);
      continue;
    }
    const auto& p = param_output.parameters;
    const std::string stem = work_directory_ + (work_directory_.empty() || work_directory_.back() == '/' ? "" : "/") + p.output_file;
    std::ofstream outfile(stem + ".csv", std::ofstream::out | std::ofstream::trunc);
    std::ofstream detail(stem + "_detail.csv", std::ofstream::out | std::ofstream::trunc);
    if (!outfile.good() || !detail.good()) {
      ExecEnv::log().error("InbreedingAnalysis::writeColumnResults; could not open output file: {}", stem);
      return false;
    }
    outfile << p.parameter_ident << DELIMITER_ << "Algorithm:" << p.inbreeding_algorithm << DELIMITER_ << "Min_AF:" << p.locii.allele_frequency_min
            << DELIMITER_ << "Max_AF:" << p.locii.allele_frequency_max << DELIMITER_ << "Spacing:" << p.locii.spacing << DELIMITER_
            << "Count:" << p.locii.locii_count << '\n';
    const bool with_ped = !p.analyze_synthetic && genealogy_data_;
    if (p.analyze_synthetic) {
      outfile << "Sample" << DELIMITER_ << "SynInbreed" << DELIMITER_ << "CalcInbreed" << '\n';                      // writeSynthetic (:353)
    } else {
      outfile << "Sample";
      if (with_ped)                                                                                                    // writePedResults (:232-240)
        outfile << DELIMITER_ << "Population" << DELIMITER_ << "Description" << DELIMITER_ << "SuperPopulation" << DELIMITER_ << "Description"
                << DELIMITER_ << "Relationship" << DELIMITER_ << "Sex" << DELIMITER_ << "Mother" << DELIMITER_ << "Father";
      for (const auto& column : param_output.columns) outfile << DELIMITER_ << column.column_ident;
      outfile << '\n';
    }
    detail.precision(17);
    detail << "Column,Sample,MajorHetCount,MajorHetFreq,MinorHetCount,MinorHetFreq,MinorHomCount,MinorHomFreq,MajorHomCount,MajorHomFreq,Total,Inbreeding\n";
    for (const auto& [genome_id, first] : param_output.columns.front().results) {
      if (p.analyze_synthetic) {
        outfile << genome_id << DELIMITER_ << generateInbreeding(genome_id).second << DELIMITER_;
      } else if (with_ped) {
        auto record_opt = genealogy_data_->getGenomeGenealogyRecord(genome_id);
        if (!record_opt) {
          ExecEnv::log().error("InbreedingAnalysis::writeColumnResults, Genome sample: {} does not have a PED record", genome_id);
          continue;
        }
        const auto& ped_record = record_opt.value();
        outfile << genome_id << DELIMITER_ << ped_record.population() << DELIMITER_ << ped_record.populationDescription() << DELIMITER_
                << ped_record.superPopulation() << DELIMITER_ << ped_record.superDescription() << DELIMITER_ << ped_record.relationship() << DELIMITER_
                << ped_record.sex() << DELIMITER_ << ped_record.maternalId() << DELIMITER_ << ped_record.paternalId() << DELIMITER_;
      } else {
        outfile << genome_id << DELIMITER_;
      }
      for (const auto& column : param_output.columns) {
        auto found = column.results.find(genome_id);
        if (found == column.results.end()) {
          ExecEnv::log().error("InbreedingAnalysis::writeColumnResults, Column: {}, Genome sample: {} not found", column.column_ident, genome_id);
          return false;
        }
        outfile << found->second.inbred_allele_sum << DELIMITER_;
      }
      outfile << '\n';
    }
    for (const auto& column : param_output.columns)
      for (const auto& [genome_id, r] : column.results)
        detail << column.column_ident << DELIMITER_ << genome_id << DELIMITER_ << r.major_hetero_count << DELIMITER_ << r.major_hetero_freq << DELIMITER_
               << r.minor_hetero_count << DELIMITER_ << r.minor_hetero_freq << DELIMITER_ << r.minor_homo_count << DELIMITER_ << r.minor_homo_freq
               << DELIMITER_ << r.major_homo_count << DELIMITER_ << r.major_homo_freq << DELIMITER_ << r.total_allele_count << DELIMITER_
               << r.inbred_allele_sum << '\n';
  }
  return true;
}
