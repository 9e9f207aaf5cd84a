// PopulationDB -> columnar SoA: the step that replaces VariantDBVariant::createVariantDB
// (kgl_genomics/kgl_variant_db/kgl_variant_db_variant.cpp:11-123).  Host C++; no counting of alleles
// across genomes happens here — that is the GPU's job through include/kgx.h.
#ifndef KGX_FLATTEN_H
#define KGX_FLATTEN_H

#ifdef KGX_WITH_REFERENCE_HEADERS
#include "kgl_variant_db_population.h"
#include "kgl_variant_db_freq.h"
#include "kgl_variant_factory_vcf_evidence_analysis.h"
#else
#include "kgx_refshim.h"
#endif

#include <string>
#include <string_view>
#include <vector>

namespace kellerberrin::genome::analysis::gpu {

// One distinct variant (HGVS) of the population = one dosage row on the device.
struct VariantRow {
  std::string hgvs;                        // Variant::HGVS(), the reference's variant identity
  ContigId_t contig;
  ContigOffset_t offset{0};
  bool is_snp{false};                      // Variant::isSNP()
  float info_af{0.0f};                     // "AF" INFO value for this alt (float32 as stored); NaN = missing
  std::shared_ptr<const Variant> variant;  // first Variant seen with this HGVS (uniqueVariants semantics)
};

// A (genome, row) cell whose real dosage exceeds 2: stored as code 3 on the device; kept here so that
// totals that need the exact copy number (HeteroHomoZygous total/SNP/indel counts) stay exact.
struct NonDiploidCell {
  uint32_t row;
  uint32_t genome;
  uint32_t dosage;
};

struct FlatPopulation {
  std::vector<GenomeId_t> genome_ids;      // std::map order = VariantDBGenomeIndex order (kgl_variant_db_variant.cpp:36-51)
  std::vector<VariantRow> rows;            // lexicographic HGVS order = VariantDBVariantIndex order (:17-30)
  uint64_t row_bytes{0};                   // ceil(G/4)
  std::vector<uint8_t> packed;             // [rows][row_bytes] 2-bit dosage codes, genome g in bits 2*(g%4) of byte g/4
  std::vector<NonDiploidCell> non_diploid;
  size_t variant_objects{0};               // Variant visits (= PopulationDB::variantCount())
  std::vector<ContigId_t> contig_ids;      // contigs EVERY genome holds, carrier or not (Pf flavour: the ##contig header lines)

  [[nodiscard]] size_t genomes() const { return genome_ids.size(); }
  [[nodiscard]] size_t variants() const { return rows.size(); }
};

// threads == 0: the reference's default, hardware_concurrency() - 1 (kel_thread/kel_workflow_threads.h:40).
[[nodiscard]] FlatPopulation flattenPopulation(const PopulationDB& population, size_t threads = 0);

// VCF text (phased diploid, 1000-Genomes flavour) straight to the same FlatPopulation a PopulationDB filled by the
// reference's Genome1000VCFImpl parser would flatten to — without creating Variant objects.
// VariantRow::variant is left null.  Implemented in kgx_vcf_flatten.cpp.
[[nodiscard]] FlatPopulation flattenVcf1000(std::string_view text, size_t threads = 0);

// The same for the unphased P. falciparum (Pf7) flavour, PfVCFImpl (kgl_parser/kgl_variant_factory_pf_impl.cpp): every
// sample is a genome, every called alt copy a canonical UNPHASED Variant unless it is the '*' allele or a spanning call
// (AD 0,0).  quality_filter: records failing P7VariantFilter (kgl_variant_filter_Pf7.cpp:131-318) contribute nothing, as
// after FilterPf7::qualityFilter's viewFilter (kga_analysis_lib_PfFilter.cpp:63-67).  Not reproduced: the check of REF
// against the reference genome's sequence (ParseVCFRecord) -- the VCF is taken at its word.
[[nodiscard]] FlatPopulation flattenVcfPf(std::string_view text, size_t threads = 0, bool quality_filter = false);

// P7FrequencyFilter / CalcFWS bins on the "AF" INFO value of a row (kgl_variant_filter_Pf7.cpp:20-66,
// kga_analysis_PfEMP_FWS.cpp:15-38,104-145): bin index 0..10, or 0xFF when the row is in no bin
// (missing AF passes both filters and is excluded by the NOT).
constexpr size_t FWS_FREQUENCY_ARRAY_SIZE = 11;
constexpr uint8_t FWS_NO_BIN = 0xFF;
[[nodiscard]] std::pair<double, double> fwsBinRange(size_t bin);
[[nodiscard]] uint8_t fwsBinOfFrequency(float info_af);

}  // namespace kellerberrin::genome::analysis::gpu

#endif  // KGX_FLATTEN_H
