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

#include <array>
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
  // CalcFWS filters Variant OBJECTS by the AF of their own VCF record (kga_analysis_PfEMP_FWS.cpp:27-29), so when the
  // copies of one HGVS come from records whose AF fall in different bins, each bin's population holds only its own
  // copies.  Such a row is flagged (its by-genome bin counts come from its split rows) and one extra "split" row per bin
  // -- the copies of that bin only -- follows the primary rows.
  bool fws_from_splits{false};
  int64_t split_of{-1};                    // >= 0: this is a split row of primary row split_of; info_af places it in its bin
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
  std::vector<VariantRow> rows;            // primary rows: lexicographic HGVS order = VariantDBVariantIndex order (:17-30); then split rows
  size_t primary_rows{0};                  // rows[0 .. primary_rows) are the population's distinct variants
  uint64_t row_bytes{0};                   // ceil(G/4)
  std::vector<uint8_t> packed;             // [rows][row_bytes] 2-bit dosage codes, genome g in bits 2*(g%4) of byte g/4
  std::vector<NonDiploidCell> non_diploid;
  // [rows][phase_row_bytes] one bit per cell, genome g in bit g%8 of byte g/8: the genome's copies of the variant carry more
  // than one distinct phase (what UniquePhasedFilter counts beyond UniqueUnphasedFilter; kgx_population_load_phase_plane).
  // Empty where no cell does (unphased data).  flattenPopulation fills it; the VCF flatteners do not.
  std::vector<uint8_t> phase_plane;
  uint64_t phase_row_bytes{0};
  // Cells whose copies carry THREE or more distinct phases (A, B and UNPHASED or HAPLOID: no parser of the reference makes
  // such a genome): UniquePhasedFilter would keep three Variant objects there, the one bit of the plane says "two".
  size_t cells_with_three_phases{0};
  size_t variant_objects{0};               // Variant visits (= PopulationDB::variantCount())
  std::vector<ContigId_t> contig_ids;      // contigs EVERY genome holds, carrier or not (Pf flavour: the ##contig header lines)

  [[nodiscard]] size_t genomes() const { return genome_ids.size(); }
  [[nodiscard]] size_t variants() const { return primary_rows; }
  [[nodiscard]] size_t deviceRows() const { return rows.size(); }
};

// Where a VCF flattener's finished device rows go.  Without one they are assembled in FlatPopulation::packed.  With one,
// `packed` stays empty: the sink is first shown the population's metadata (genome ids, row metadata, row_bytes -- the
// place of every row is settled before the first one is packed), then receives blocks of consecutive rows as they are
// packed, possibly from several threads at once, every row exactly once.  GpuAlleleAnalysis uploads them straight to the
// device, so the host never holds the packed population next to the parsed records it is made from.
class RowSink {
 public:
  virtual ~RowSink() = default;
  virtual bool begin(const FlatPopulation& meta) = 0;          // false: stop here (the flattener returns the metadata alone)
  virtual void rows(uint64_t first_row, uint64_t n_rows, const uint8_t* data) = 0;      // n_rows * meta.row_bytes bytes
};

// Streaming: rows leave for the sink piece by piece WHILE the file is read, so the host holds one piece of parsed records
// at a time instead of all of them.  The rows are then in the order the variants first appear in the file (not HGVS
// order: nothing on the path reads it -- the CSV writers key by HGVS, the compound-offset sweep takes its groups as row
// lists), their number is known at the end only (the sink grows), and a variant met again in a later
// record is merged into its row at the end (the row is read back: exact, a single record puts at most two copies into a
// cell).  Not every file can be taken this way -- see flattenVcf1000FileStreaming.
class StreamSink {
 public:
  virtual ~StreamSink() = default;
  virtual bool open(uint64_t n_genomes, uint64_t row_bytes) = 0;
  virtual bool write(uint64_t first_row, uint64_t n_rows, const uint8_t* data) = 0;      // may extend the population
  virtual bool read(uint64_t row, uint8_t* data) = 0;                                     // one row written earlier
  virtual bool close(uint64_t n_rows) = 0;                                                // the final row count
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

// The same two straight from a file (plain / gzip / block gzip), read a bounded piece of whole lines at a time
// (VcfChunkReader, kgx_vcf_io.h): a piece's text is dropped once its records are parsed into 2-bit rows, so the working
// set is the genotypes, not the text.  chunk_bytes = text per piece.  false + error on an I/O or format error.
[[nodiscard]] bool flattenVcf1000File(const std::string& file_name, FlatPopulation& flat, std::string& error, size_t threads = 0,
                                      size_t chunk_bytes = size_t{64} << 20, RowSink* sink = nullptr);
[[nodiscard]] bool flattenVcfPfFile(const std::string& file_name, FlatPopulation& flat, std::string& error, size_t threads = 0,
                                    bool quality_filter = false, size_t chunk_bytes = size_t{64} << 20, RowSink* sink = nullptr);

// The streaming forms (StreamSink above).  flat comes back without `packed`, rows in first-appearance order.  Returns false
// with `error` on an I/O, format or sink error -- and false with `two_phase` set (no error) for a file that has to go through
// the two-phase flatteners above instead: a sample named twice, or a sample that carries no variant at all (Genome1000
// flavour: it is then no genome, which changes every row's width).
[[nodiscard]] bool flattenVcf1000FileStreaming(const std::string& file_name, StreamSink& sink, FlatPopulation& flat, std::string& error, bool& two_phase,
                                               size_t threads = 0, size_t chunk_bytes = size_t{64} << 20);
[[nodiscard]] bool flattenVcfPfFileStreaming(const std::string& file_name, StreamSink& sink, FlatPopulation& flat, std::string& error, bool& two_phase,
                                             size_t threads = 0, bool quality_filter = false, size_t chunk_bytes = size_t{64} << 20);

// ---- the INBREED package's two inputs straight from VCF text (SURVEY.md §8f #1 for the K5 path) ------------------
//
// The unphased mono-genome reference (Gnomad / 1000-Genomes site files; GrchVCFImpl::ProcessVCFRecord,
// kgl_parser/kgl_variant_factory_grch_impl.cpp:53-156: one UNPHASED Variant per alt, as written, in one genome), cut
// down to what InbreedAnalysis keeps of it -- AndFilter(SNPFilter(), PassFilter()) (kga_analysis_inbreed.cpp:79) -- with
// the six super-population frequencies FrequencyDatabaseRead::superPopFrequency reads for the data source
// (kgl_variant_db_freq.cpp:13-122; NaN = no value).  Alts of one offset are in file order (the reference's order is
// the order its parser threads happen to add them in).
struct ReferenceAltRow {
  std::string hgvs;
  std::array<double, 6> af{};              // per FrequencyDatabaseRead::superPopulations() slot, NaN = no value
};
struct ReferenceLocusRow {
  ContigOffset_t offset{0};
  std::vector<ReferenceAltRow> alts;       // OffsetDB array order
};
struct FlatReference {
  ContigId_t contig_id;                    // of the first record
  size_t contigs{0};                       // distinct contigs seen (the package insists on exactly 1)
  uint32_t max_alts{0};
  std::vector<ReferenceLocusRow> loci;     // ascending offset
};
[[nodiscard]] FlatReference flattenReferenceVcf(std::string_view text, DataSourceEnum data_source);
// The same from a file read a bounded piece at a time (threads: the block-gzip inflate; the records are walked in order).
[[nodiscard]] bool flattenReferenceVcfFile(const std::string& file_name, DataSourceEnum data_source, FlatReference& reference, std::string& error,
                                           size_t threads = 0, size_t chunk_bytes = size_t{64} << 20);

// The phased diploid population (1000-Genomes flavour, as flattenVcf1000) as the allele-index bytes of the inbreeding
// sweep: for every reference locus and genome, the genome's SNP variants at that offset in the order the parser adds
// them (phase A alts, then phase B, record by record), each as 1 + its index in the locus's reference alt list (15 =
// not in the list), two per byte, 0xFF for three or more; two copies of one variant on the SAME phase (a repeated record)
// are the byte (0, code): the reference treats them as two analogous, not homozygous, variants.  Genomes = samples carrying any variant on the reference's
// contig (the parser creates a genome's contig when it first adds a variant to it), in id order.
struct FlatDiploid {
  std::vector<GenomeId_t> genome_ids;
  uint64_t n_loci{0};
  std::vector<uint8_t> bytes;              // [n_loci][genome_ids.size()]
  // Reference loci with MORE than 14 alts do not fit two 4-bit indices: their cells are 16-bit here -- a1 | a2 << 8, indices up
  // to 254, 255 = not in the list, 0xFFFF = three or more (kgx_gt8_set_wide_rows) -- and their byte rows are 0xFF throughout.
  std::vector<uint32_t> wide_loci;         // ascending
  std::vector<uint16_t> wide_cells;        // [wide_loci.size()][genome_ids.size()]
  std::string error;                       // reserved: every population the parser accepts is representable
};
constexpr size_t kGt8NarrowAlts = 14, kGt8WideAlts = 254;
[[nodiscard]] FlatDiploid flattenVcf1000Gt8(std::string_view text, const FlatReference& reference, size_t threads = 0);
// The same from a file read a bounded piece at a time (see flattenVcf1000File): between pieces only the calls of the
// records that land on a reference locus are kept, a byte per cell of the result.
[[nodiscard]] bool flattenVcf1000Gt8File(const std::string& file_name, const FlatReference& reference, FlatDiploid& diploid, std::string& error,
                                         size_t threads = 0, size_t chunk_bytes = size_t{64} << 20);

// Streaming: the rows of the loci a piece completes leave for the sink while the next piece is read; what stays on the
// host between pieces is the records of the one locus that may go on.  Genomes are every sample, in id order, as soon as
// the header is read (open); rows are [genome_ids.size()] bytes each and arrive as dense runs of consecutive loci in
// ascending order (a locus no record lands on is never written: zero); the wide rows (loci of more than 14 alts) stay in
// `diploid` and are complete at close.  Returns false with two_phase set (error = why) for
// a file that has to take flattenVcf1000Gt8File instead: a sample named twice, a sample that carries nothing on the contig
// (it is then no genome of it), records not in ascending position order.
class Gt8StreamSink {
 public:
  virtual ~Gt8StreamSink() = default;
  virtual bool open(const std::vector<GenomeId_t>& genome_ids, uint64_t n_loci) = 0;
  virtual bool write(uint64_t first_locus, uint64_t n_loci, const uint8_t* rows) = 0;
  virtual bool close() = 0;
};
[[nodiscard]] bool flattenVcf1000Gt8FileStreaming(const std::string& file_name, const FlatReference& reference, Gt8StreamSink& sink, FlatDiploid& diploid,
                                                  std::string& error, bool& two_phase, size_t threads = 0, size_t chunk_bytes = size_t{64} << 20);

// P7FrequencyFilter / CalcFWS bins on the "AF" INFO value of a row (kgl_variant_filter_Pf7.cpp:20-66,
// kga_analysis_PfEMP_FWS.cpp:15-38,104-145): bin index 0..10, or 0xFF when the row is in no bin
// (missing AF passes both filters and is excluded by the NOT).
constexpr size_t FWS_FREQUENCY_ARRAY_SIZE = 11;
constexpr uint8_t FWS_NO_BIN = 0xFF;
[[nodiscard]] std::pair<double, double> fwsBinRange(size_t bin);
[[nodiscard]] uint8_t fwsBinOfFrequency(float info_af);

}  // namespace kellerberrin::genome::analysis::gpu

#endif  // KGX_FLATTEN_H
