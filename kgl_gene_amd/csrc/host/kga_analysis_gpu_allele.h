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
#include "kgx_pf7_resources.h"

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

// One sampling site or country: kga_analysis_PfEMP_heterozygous.h:85-106 (LocationSummary), same members.
struct GpuLocationSummary {
  std::string location_;
  LocationType location_type_{LocationType::City};
  std::string city_;
  std::string country_;
  std::string region_;
  double radius_km_{0.0};
  size_t radii_samples_{0};
  size_t radii_samples_OK_{0};
  std::map<std::string, size_t> studies_;
  double monoclonal_Fst_{0.0};
  double hom_het_ratio_{0.0};
  size_t total_variants_{0};
  double variant_rate_{0.0};
  size_t homozygous_reference_alleles_{0};
  size_t heterozygous_reference_minor_alleles_{0};
  size_t homozygous_minor_alleles_{0};
  size_t heterozygous_minor_alleles_{0};
  size_t snp_count_{0};
  size_t indel_count_{0};
};
using GpuLocationSummaryMap = std::map<std::string, GpuLocationSummary>;

// The per-genome x contig counters of HeteroHomoZygous (kga_analysis_PfEMP_heterozygous.h:108-135) once the device has
// produced them, joined with the Pf7 sample resources: per-site summaries, location F_IS, the result files.  Host only.
class GpuHeteroHomoZygous {
 public:
  // Both or neither; with them the location analysis and the reference's file layouts are available.
  void setResources(std::shared_ptr<const Pf7SampleResource> sample_ptr, std::shared_ptr<const Pf7FwsResource> fws_ptr,
                    std::shared_ptr<const Pf7SampleLocation> physical_distance_ptr);
  [[nodiscard]] bool hasResources() const { return static_cast<bool>(pf7_sample_ptr_); }
  [[nodiscard]] GpuVariantAnalysisMap& analysisMap() { return variant_analysis_map_; }
  [[nodiscard]] const GpuVariantAnalysisMap& analysisMap() const { return variant_analysis_map_; }

  // HeteroHomoZygous::aggregateResults (kga_analysis_PfEMP_heterozygous.cpp:229-263): the counters of the listed genomes,
  // every genome once, over all its contigs.
  [[nodiscard]] VariantAnalysisType aggregateResults(const std::vector<GenomeId_t>& sample_vector) const;
  // HeteroHomoZygous::location_summary (:266-360); needs the Pf7 resources.
  [[nodiscard]] GpuLocationSummaryMap locationSummary(double radius_km) const;
  // HeteroHomoZygous::UpdateSampleLocation (:363-414): genome -> Wright's F_IS against its site (its country when fewer
  // than 20 of the site's samples passed QC); genomes whose site / country is unknown are left out (F_IS 0 in the file).
  [[nodiscard]] std::map<GenomeId_t, double> locationInbreeding(const GpuLocationSummaryMap& location_summary) const;
  // Wright's F_IS of one genome against an aggregate (:400-406).
  [[nodiscard]] static double wrightsInbreeding(const VariantAnalysisType& location, const VariantAnalysisType& genome);

  // HeteroHomoZygous::write_variant_results (:108-226) / write_location_results (:418-510), byte for byte.
  bool writeSampleResults(const std::string& file_name, const GpuLocationSummaryMap& location_summary) const;
  bool writeLocationResults(const std::string& file_name, const GpuLocationSummaryMap& location_summary) const;
  // Without the resources: one line per genome x contig, F_IS against the whole population's aggregate of the contig.
  bool writeContigResults(const std::string& file_name) const;

  constexpr static const size_t MINIMUM_LOCATION_SAMPLES_ = 20;   // kga_analysis_PfEMP_heterozygous.h:127
 private:
  GpuVariantAnalysisMap variant_analysis_map_;
  std::shared_ptr<const Pf7SampleResource> pf7_sample_ptr_;
  std::shared_ptr<const Pf7FwsResource> pf7_fws_ptr_;
  std::shared_ptr<const Pf7SampleLocation> pf7_physical_distance_ptr_;
  constexpr static const char CSV_DELIMITER_ = ',';
};

class GpuAlleleAnalysis : public VirtualAnalysis {
 public:
  GpuAlleleAnalysis() = default;
  ~GpuAlleleAnalysis() override = default;

  inline static const std::string IDENT{"GPU_ALLELE"};
  [[nodiscard]] std::string ident() const override { return IDENT; }
  [[nodiscard]] static std::unique_ptr<VirtualAnalysis> factory() { return std::make_unique<GpuAlleleAnalysis>(); }

  // Parameters (all optional, first parameter block wins): "DeviceList" / "Devices" / "Device" (which MI355X devices the
  // genomes are sharded over: kgx_device_binding.h; default device 0),
  // "VariantFile" / "GenomeFile" / "HetHomFile" / "LocationFile" (output file stems, default VariantFWS / GenomeFWS /
  // VariantStatistics / VariantLocation), "Pf7FilterQC" / "Pf7FilterFWS" / "Pf7FwsThreshold" (the genome-level filters of
  // FilterPf7, default on / on / 0.95 once the Pf7 resources are there), "LocationRadiusKm" (default 0); "Pf7QualityFilter"
  // (the per-record P7VariantFilter of "FileNameOnly" Pf7 VCF files) defaults to on with the resources, off without.
  // Resources (optional, both or neither): one Pf7SampleResource and one Pf7FwsResource.  With them the package works as
  // PfEMPAnalysis does (kga_analysis_PfEMP.cpp:24-26,90,105,146-163): only genomes that pass QC and are monoclonal take part
  // (a genome mask on the device population), VariantStatistics.csv takes the reference's one-line-per-genome layout with
  // the sample's site, study, published FWS and location F_IS, and VariantLocation.csv is written.  Without them every
  // genome takes part and VariantStatistics.csv is one line per genome x contig with F_IS against the whole population.
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
  [[nodiscard]] const GpuVariantAnalysisMap& getVariantAnalysisMap() const { return hetero_homo_zygous_.analysisMap(); }
  [[nodiscard]] const GpuHeteroHomoZygous& heteroHomoZygous() const { return hetero_homo_zygous_; }
  [[nodiscard]] static double wrightsInbreeding(const VariantAnalysisType& location, const VariantAnalysisType& genome) {
    return GpuHeteroHomoZygous::wrightsInbreeding(location, genome);
  }

 private:
  bool sweepPopulation(const PopulationDB& population);
  bool sweepVcfFile(const std::string& file_name);
  bool sweepFlat(const gpu::FlatPopulation& flat, const std::string& label, kgx_pop* uploaded = nullptr);
  bool writeVariantResults(const std::string& file_name) const;
  bool writeGenomeResults(const std::string& file_name) const;
  // FilterPf7::qualityFilter's genome part (kga_analysis_lib_PfFilter.cpp:26-58): does the genome take part?
  [[nodiscard]] bool keepGenome(const GenomeId_t& genome_id) const;
  [[nodiscard]] bool genomeFilterActive() const { return pf7_sample_ptr_ && (filter_qc_ || filter_fws_); }

  std::string work_directory_;
  std::string vcf_flavour_{"Genome1000"};
  bool pf7_quality_filter_{false};
  std::string variant_file_{"VariantFWS"}, genome_file_{"GenomeFWS"}, hethom_file_{"VariantStatistics"}, location_file_{"VariantLocation"};
  std::shared_ptr<const Pf7SampleResource> pf7_sample_ptr_;
  std::shared_ptr<const Pf7FwsResource> pf7_fws_ptr_;
  bool filter_qc_{true}, filter_fws_{true};
  double fws_monoclonal_threshold_{Pf7FwsResource::MONOCLONAL_FWS_THRESHOLD};
  double location_radius_km_{0.0};                             // PfEMPAnalysis::SAMPLE_LOCATION_RADIUS_ (kga_analysis_PfEMP.h:64)
  bool device_ready_{false};
  double k1_flatten_seconds_{0.0};            // the last PopulationDB -> 2-bit rows flattening (logged with the upload time)
  GpuGenomeFWSMap genome_fws_map_;
  GpuVariantFWSMap variant_fws_map_;
  GpuHeteroHomoZygous hetero_homo_zygous_;
  constexpr static const char CSV_DELIMITER_ = ',';
};

}  // namespace kellerberrin::genome::analysis

#endif  // KGA_ANALYSIS_GPU_ALLELE_H
