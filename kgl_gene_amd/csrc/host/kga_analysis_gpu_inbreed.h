// GpuInbreedAnalysis — the INBREED package (kga_analytic/kga_inbreed/kga_analysis_inbreed.{h,cpp}) with
// the per-genome x per-locus sweep on the MI355X.  Same sequencing, same XML parameters
// (kga_analysis_inbreed_args.h:24-55,164-172), same window loop (kga_analysis_inbreed_diploid.cpp:18-79),
// same locus sampling (kga_analysis_inbreed_locus.cpp), same estimators (kga_analysis_inbreed_calc.cpp);
// what the reference fans out as one thread-pool task per genome (_diploid.cpp:98-166) is one kgx_inbreed()
// call per super population.  Register next to InbreedAnalysis in kga_analytic/kga_analysis_factory.cpp:31-43.
#ifndef KGA_ANALYSIS_GPU_INBREED_H
#define KGA_ANALYSIS_GPU_INBREED_H

#include <array>

#ifdef KGX_WITH_REFERENCE_HEADERS
#include "kgl_package_analysis_virtual.h"
#include "kgl_hsgenealogy_parser.h"
#include "kgl_variant_db_freq.h"
#else
#include "kgx_refshim.h"
#endif
#include "kgx_flatten.h"

namespace kellerberrin::genome::analysis {

// LociiVectorArguments + InbreedingParameters (kga_analysis_inbreed_args.h:69-160), defaults included.
struct GpuLociiArguments {
  ContigOffset_t lower_offset{0};
  ContigOffset_t upper_offset{1000000000};
  size_t spacing{1000};
  size_t locii_count{1000};
  double allele_frequency_min{0.0};
  double allele_frequency_max{1.0};
};

struct GpuInbreedingParameters {
  std::string parameter_ident{"ParamIdent"};
  GpuLociiArguments locii;
  std::string inbreeding_algorithm{"Loglikelihood"};
  std::string output_file{"output"};
  bool analyze_synthetic{true};
};

// LocusResults (kga_analysis_inbreed_output.h:21-35)
struct GpuLocusResults {
  GenomeId_t genome;
  size_t major_hetero_count{0};
  double major_hetero_freq{0.0};
  size_t minor_hetero_count{0};
  double minor_hetero_freq{0.0};
  size_t minor_homo_count{0};
  double minor_homo_freq{0.0};
  size_t major_homo_count{0};
  double major_homo_freq{0.0};
  size_t total_allele_count{0};
  double inbred_allele_sum{0.0};
};
using GpuResultsMap = std::map<GenomeId_t, GpuLocusResults>;

struct GpuResultColumn {
  std::string column_ident;          // InbreedingResultColumn::generateIdent: contig_lower_upper
  GpuResultsMap results;
};

struct GpuParamOutput {
  GpuInbreedingParameters parameters;
  std::vector<GpuResultColumn> columns;
};

// The SNP & PASS view of the reference (unphased, mono-genome) contig that locus sampling and the AF tables
// are cut from (InbreedAnalysis::fileReadAnalysis, kga_analysis_inbreed.cpp:79).
using GpuReferenceAlt = gpu::ReferenceAltRow;       // hgvs + af[6] (NaN = no value)
using GpuReferenceLocus = gpu::ReferenceLocusRow;   // offset + alts in OffsetDB array order
class GpuReferenceContig {
 public:
  ContigId_t contig_id;
  std::vector<GpuReferenceLocus> loci;     // ascending offset
  uint32_t max_alts{0};
  // AlleleFreqVector for locus l and super population slot sp as a fixed-width row: af of alt j, NaN when the
  // alt is not in the vector (no value, or analogous to an earlier alt) (kga_analysis_inbreed_freq.cpp:18-57).
  void alleleFreqRow(size_t l, int sp, double* row, uint32_t amax) const;
  // checkValidAlleleVector + the sampling predicate (kga_analysis_inbreed_locus.cpp:46-62): returns false when the
  // locus cannot be sampled; minor_sum = minorAlleleFrequencies().
  [[nodiscard]] bool validForSampling(size_t l, int sp, double& minor_sum) const;
  // RetrieveLociiVector::getLociiCount / getLociiFromTo: indices into loci.
  [[nodiscard]] std::vector<uint32_t> sampleLocii(int sp, const GpuLociiArguments& args, bool by_count) const;
  // The device's allele index is 4 bits: 14 alts per offset.  An offset with more keeps the alts that carry a frequency
  // for some super population first (the others can be in no AlleleFreqVector and index as "unknown alt", the same to
  // every estimator) and is cut to 14; returns the number of offsets that lost a frequency-bearing alt that way (their
  // carriers of such an alt then count as carriers of an unknown one).
  static constexpr uint32_t kMaxAlts = 14;          // two 4-bit indices of a matrix byte; offsets with more take a WIDE ROW (kgx_gt8_set_wide_rows) ...
  static constexpr uint32_t kMaxWideAlts = 254;     // ... of two 8-bit indices
  [[nodiscard]] bool isWide(size_t l) const { return loci[l].alts.size() > kMaxAlts; }
  [[nodiscard]] uint32_t narrowAlts() const;        // the most alts of any offset that fits a byte (the frequency table's columns where no wide offset is selected)
  size_t limitAlts(uint32_t most = kMaxWideAlts);
};

class GpuInbreedAnalysis : public VirtualAnalysis {
 public:
  GpuInbreedAnalysis() = default;
  ~GpuInbreedAnalysis() override = default;

  inline static const std::string IDENT{"GPU_INBREED"};
  [[nodiscard]] std::string ident() const override { return IDENT; }
  [[nodiscard]] static std::unique_ptr<VirtualAnalysis> factory() { return std::make_unique<GpuInbreedAnalysis>(); }

  [[nodiscard]] bool initializeAnalysis(const std::string& work_directory, const ActiveParameterList& named_parameters,
                                        const std::shared_ptr<const AnalysisResources>& resource_ptr) override;
  [[nodiscard]] bool fileReadAnalysis(std::shared_ptr<const DataDB> data_object_ptr) override;
  [[nodiscard]] bool iterationAnalysis() override;
  [[nodiscard]] bool finalizeAnalysis() override;

  [[nodiscard]] const std::vector<GpuParamOutput>& parameterOutput() const { return parameter_output_vector_; }

  // InbreedArguments::extractParameters (kga_analysis_inbreed_args.cpp:8-152)
  [[nodiscard]] static std::vector<GpuInbreedingParameters> extractParameters(const ActiveParameterList& named_parameters);
  [[nodiscard]] static GpuReferenceContig buildReference(const PopulationDB& unphased_population, bool& ok);
  // The diploid population as allele-index bytes against the reference: [n_loci][genomes holding the contig, id order].
  [[nodiscard]] static gpu::FlatDiploid diploidBytes(const PopulationDB& diploid_population, const GpuReferenceContig& reference, bool phased);

 private:
  bool populationInbreeding(GpuParamOutput& param_output);
  // Either source of the two inputs: the PopulationDB objects a parser delivered, or "FileNameOnly" VCF files the
  // package flattens itself (no Variant objects).
  bool referenceInput(GpuReferenceContig& reference, uint32_t most_alts) const;       // referenceSource + GpuReferenceContig::limitAlts(most_alts)
  bool referenceSource(GpuReferenceContig& reference) const;
  bool diploidInput(const GpuReferenceContig& reference, gpu::FlatDiploid& diploid, bool& phased) const;
  [[nodiscard]] bool haveReference() const { return unphased_population_ != nullptr || !reference_vcf_.empty(); }
  [[nodiscard]] bool haveDiploid() const { return diploid_population_ != nullptr || !diploid_vcf_.empty(); }
  // SyntheticAnalysis::syntheticInbreeding (kga_analysis_inbreed_synthetic.cpp:17-138)
  bool syntheticInbreeding(GpuParamOutput& param_output);

 public:
  // InbreedSynthetic::generateSyntheticGenomeId / generateInbreeding (kga_analysis_inbreed_syngen.cpp:202-275)
  [[nodiscard]] static GenomeId_t generateSyntheticGenomeId(double inbreeding, const std::string& super_population, size_t counter);
  [[nodiscard]] static std::pair<bool, double> generateInbreeding(const GenomeId_t& genome_id);
  // Seed of the synthetic draws (parameter "SyntheticSeed", default 1111 = DeterministicEntropySource, kel_distribution.h:60;
  // the reference itself seeds from std::random_device).
  uint64_t synthetic_seed_{1111};
  // Entropy of the HallME / Loglikelihood start points.  The reference gives every per-genome task its own
  // RandomEntropySource (a std::mt19937_64 seeded from std::random_device, kel_math/kel_distribution.h:25-43;
  // _calc.cpp:163,235) -- that is the default here too (parameter "StartSeed" absent or 0).  "StartSeed" = s > 0 makes the
  // runs repeatable: the k-th task processResults enqueues (genome-id order over the genomes with the contig, a PED record
  // and a locus list, _diploid.cpp:117-148) owns the stream std::mt19937_64(s + k), the DeterministicEntropySource idea
  // (kel_distribution.h:53-70) with one stream per task.  "StartPoints" = "Midpoint": no draws, the midpoints of the
  // reference's start intervals (0.25 / 0.0) -- a deterministic mode the reference does not have.
  uint64_t start_seed_{0};
  bool start_midpoints_{false};
  size_t window_batch_{16};                       // parameter WindowBatch: windows sampled ahead and handed to the device together (kgx_inbreed_batch)
  mutable std::array<std::vector<double>, 2> seeded_starts_;    // [HallME, Loglikelihood][stream]: drawn once under a StartSeed
  // The start points of n tasks, the first of which is the first_stream-th enqueued; empty = midpoints (or not iterative).
  [[nodiscard]] std::vector<double> startPoints(int algorithm, const std::vector<uint64_t>& streams) const;

 private:
  bool writeResults() const;

  std::vector<GpuParamOutput> parameter_output_vector_;
  std::string work_directory_;
  bool device_ready_{false};
  std::shared_ptr<const PopulationDB> diploid_population_;
  std::shared_ptr<const PopulationDB> unphased_population_;
  std::shared_ptr<const HsGenomeGenealogyData> genealogy_data_;
  std::string reference_vcf_, diploid_vcf_;            // FileNameOnly data files (kgl_variant_factory_parsers.cpp:65-66)
  DataSourceEnum reference_vcf_source_{DataSourceEnum::NotImplemented};
  constexpr static const char DELIMITER_ = ',';
};

}  // namespace kellerberrin::genome::analysis

#endif  // KGA_ANALYSIS_GPU_INBREED_H
