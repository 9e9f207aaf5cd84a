// GpuAlleleAnalysis — the allele-count sweeps of the PfEMP package on the MI355X, as a drop-in
// VirtualAnalysis (kgl_app/kgl_package_analysis_virtual.h:20-55).  It produces what
//   CalcFWS::calcFwsStatistics            (kga_analytic/kga_PfEMP/kga_analysis_PfEMP_FWS.cpp:15-101)
//   HeteroHomoZygous::analyzeVariantPopulation (kga_analysis_PfEMP_heterozygous.cpp:16-105)
// produce — per-variant AlleleSummmary, per-genome x 11 AF-bin AlleleSummmary, per-genome x contig
// VariantAnalysisType — by flattening the PopulationDB once and running K2/K3/K8 through include/kgx.h.
// Register it next to the reference's packages (kga_analytic/kga_analysis_factory.cpp:31-43):
//   { kga::GpuAlleleAnalysis::IDENT, kga::GpuAlleleAnalysis::factory }
#ifndef KGA_ANALYSIS_GPU_ALLELE_H
#define KGA_ANALYSIS_GPU_ALLELE_H

#include <array>

#ifdef KGX_WITH_REFERENCE_HEADERS
#include "kgl_package_analysis_virtual.h"
#include "kgl_variant_db_variant.h"               // AlleleSummmary
#include "kga_analysis_PfEMP_heterozygous.h"      // VariantAnalysisType
#else
#include "kgx_refshim.h"
#endif
#include "kgx_flatten.h"

struct kgx_pop;                                    // include/kgx.h

namespace kellerberrin::genome::analysis {

#ifndef KGX_WITH_REFERENCE_HEADERS
// kgl_variant_db_variant.h:26-41
struct AlleleSummmary {
  size_t referenceHomozygous_{0};    // (A,A)
  size_t minorHeterozygous_{0};      // (a,A)
  size_t minorHomozygous_{0};        // (a,a)
  void operator+=(const AlleleSummmary& rhs) {
    minorHomozygous_ += rhs.minorHomozygous_;
    referenceHomozygous_ += rhs.referenceHomozygous_;
    minorHeterozygous_ += rhs.minorHeterozygous_;
  }
};
// kga_analysis_PfEMP_heterozygous.h:22-32
struct VariantAnalysisType {
  size_t total_variants_{0};
  size_t snp_count_{0};
  size_t indel_count_{0};
  size_t homozygous_minor_alleles_{0};
  size_t heterozygous_minor_alleles_{0};
  size_t heterozygous_reference_minor_alleles_{0};
  size_t homozygous_reference_alleles_{0};
};
#endif

using GpuFwsFrequencyArray = std::array<AlleleSummmary, gpu::FWS_FREQUENCY_ARRAY_SIZE>;
using GpuGenomeFWSMap = std::map<GenomeId_t, GpuFwsFrequencyArray>;
using GpuVariantFWSMap = std::map<std::string, AlleleSummmary>;
using GpuVariantAnalysisContigMap = std::map<std::string, VariantAnalysisType>;
using GpuVariantAnalysisMap = std::map<GenomeId_t, GpuVariantAnalysisContigMap>;

class GpuAlleleAnalysis : public VirtualAnalysis {
 public:
  GpuAlleleAnalysis() = default;
  ~GpuAlleleAnalysis() override = default;

  inline static const std::string IDENT{"GPU_ALLELE"};
  [[nodiscard]] std::string ident() const override { return IDENT; }
  [[nodiscard]] static std::unique_ptr<VirtualAnalysis> factory() { return std::make_unique<GpuAlleleAnalysis>(); }

  // Parameters (all optional, first parameter block wins): "DeviceList" / "Devices" / "Device" (which MI355X devices the
  // genomes are sharded over: kgx_device_binding.h; default device 0),
  // "VariantFile" / "GenomeFile" / "HetHomFile" (output file stems, default VariantFWS / GenomeFWS / VariantStatistics).
  [[nodiscard]] bool initializeAnalysis(const std::string& work_directory, const ActiveParameterList& named_parameters,
                                        const std::shared_ptr<const AnalysisResources>& resource_ptr) override;
  // A diploid PopulationDB (DiploidPhased / DiploidUnphased): flattened, swept on the GPU, accumulated.
  // A "FileNameOnly" data file (NoStructure, kgl_parser/kgl_data_file_type.h:133) is read as a phased-diploid VCF
  // and flattened directly, bypassing Variant/PopulationDB construction.
  [[nodiscard]] bool fileReadAnalysis(std::shared_ptr<const DataDB> data_object_ptr) override;
  [[nodiscard]] bool iterationAnalysis() override;
  // Writes the three CSV files into the work directory.
  [[nodiscard]] bool finalizeAnalysis() override;

  [[nodiscard]] const GpuGenomeFWSMap& getGenomeMap() const { return genome_fws_map_; }
  [[nodiscard]] const GpuVariantFWSMap& getVariantMap() const { return variant_fws_map_; }
  [[nodiscard]] const GpuVariantAnalysisMap& getVariantAnalysisMap() const { return variant_analysis_map_; }
  // Wright's F_IS of one genome against an aggregate (UpdateSampleLocation, heterozygous.cpp:400-406).
  [[nodiscard]] static double wrightsInbreeding(const VariantAnalysisType& location, const VariantAnalysisType& genome);

 private:
  bool sweepPopulation(const PopulationDB& population);
  bool sweepVcfFile(const std::string& file_name);
  bool sweepFlat(const gpu::FlatPopulation& flat, const std::string& label, kgx_pop* uploaded = nullptr);
  bool writeVariantResults(const std::string& file_name) const;
  bool writeGenomeResults(const std::string& file_name) const;
  bool writeHetHomResults(const std::string& file_name) const;

  std::string work_directory_;
  std::string vcf_flavour_{"Genome1000"};
  bool pf7_quality_filter_{false};
  std::string variant_file_{"VariantFWS"}, genome_file_{"GenomeFWS"}, hethom_file_{"VariantStatistics"};
  bool device_ready_{false};
  GpuGenomeFWSMap genome_fws_map_;
  GpuVariantFWSMap variant_fws_map_;
  GpuVariantAnalysisMap variant_analysis_map_;
  constexpr static const char CSV_DELIMITER_ = ',';
};

}  // namespace kellerberrin::genome::analysis

#endif  // KGA_ANALYSIS_GPU_ALLELE_H
