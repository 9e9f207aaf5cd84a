/*
 * kgx.h — C ABI of the MI355X-native population allele-count / inbreeding sweep.
 *
 * This is the drop-in boundary for the one hot path of kellerberrin/KGL_Gene that this
 * repository accelerates.  The reference has no FFI of its own: analyses are
 * C++ classes compiled into the executable behind `VirtualAnalysis`
 * (kgl_app/kgl_package_analysis_virtual.h:20-55).  The C++ analysis classes shipped in
 * kgl_gene_amd/csrc/host/ (GpuAlleleAnalysis, GpuInbreedAnalysis) implement that interface and
 * call ONLY the functions declared here; tests and bench.py bind the same symbols via ctypes.
 *
 * Conventions
 *   - plain pointers and sizes; no C++/torch types; caller owns every host buffer
 *   - every function returning int returns 0 on success, a negative KGX_E* code on failure;
 *     kgx_last_error() then holds a message (thread-local).  The C++ layer maps failure to
 *     `return false` + log().error, the reference's error convention
 *     (kgl_app/kgl_package_analysis.cpp:72-76).
 *   - the library is bound to ONE OR MORE devices by kgx_init().  An opaque handle (kgx_pop, kgx_gt8) owns device
 *     memory on every bound device: its genomes are split into contiguous shards, one per device, and every entry
 *     point works on the whole population -- per-variant counts are summed over the shards by ONE exchange step
 *     (a direct RCCL ncclAllReduce(sum, uint32) over xGMI when the shards sit on different devices), per-genome
 *     results are concatenated in genome order.  This is the shape of the reference, which is ONE process fanning
 *     out one task per genome (kgl_variant_db_population.cpp:386-433).  A process that owns a single GPU (one rank
 *     of a torch.distributed job) binds that one device and exchanges counts itself.
 *   - functions are not re-entrant per handle (the reference calls its analysis virtuals from a single thread,
 *     kgl_app/kgl_package.cpp:41-75); different handles may be used from different threads.
 *   - "variant row" v: one distinct HGVS variant (locus × alt); biallelic ⇒ one row per locus.
 *     Row order is the caller's; the reference's lexicographic-HGVS order
 *     (kgl_variant_db_variant.cpp:17-30) is applied host-side by the C++ layer.
 *   - dosage code per (genome, variant), 2 bits: 0 = (A,A) reference homozygous, 1 = (a,A)
 *     heterozygous, 2 = (a,a) minor homozygous, 3 = non-diploid (>2 copies; the reference warns
 *     and does not count it, kgl_variant_db_variant.cpp:158-161).
 *   - there is NO CPU fallback: every compute entry point fails with KGX_ENODEVICE when no
 *     gfx950 device is usable.
 */
#ifndef KGX_H
#define KGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGX_OK          0
#define KGX_EINVAL     -1   /* bad argument (null pointer, range, shape)            */
#define KGX_ENODEVICE  -2   /* no usable HIP device / wrong architecture            */
#define KGX_EHIP       -3   /* a HIP runtime call failed; see kgx_last_error()      */
#define KGX_ENOMEM     -4   /* device or host allocation failed                     */
#define KGX_ESTATE     -5   /* call order violated (e.g. genotypes not loaded)      */

typedef struct kgx_pop kgx_pop;           /* 2-bit dosage population shard (K2/K3/K4/K8)      */

/* ---- library / device ------------------------------------------------------------------- */

const char* kgx_version(void);
const char* kgx_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). Never fails. */
int kgx_device_count(void);
/* Bind the library to device_count devices: device_ids[i] = HIP ordinal of slot i (NULL = ordinals 0..device_count-1;
 * device_count 0 = every visible device).  Must precede all other calls; every handle created afterwards is sharded
 * over these devices.  With more than one slot a RCCL communicator over the slots is created
 * (ncclCommInitAll) for the count exchange.  Calling it again rebinds the library; handles created before keep their
 * devices until they are destroyed.  An ordinal may be listed more than once (two shards on one device: a way to
 * exercise the sharded paths on a one-GPU box); RCCL cannot span such a binding, so the counts are then summed by
 * device-to-device copies + an add kernel ("peer" exchange). */
int kgx_init(int device_count, const int* device_ids);
/* The library's KGX_* environment switches (launch shapes, the path selectors the tests and comparisons use: DESIGN.md) are read
 * at kgx_init, once; a process that changes them afterwards calls this to have them read again.  Never fails. */
int kgx_reload_options(void);
/* Slots bound by the last successful kgx_init (0 before). */
int kgx_bound_devices(void);
/* How per-variant counts of different shards are summed: "none" (one slot), "rccl" or "peer". */
const char* kgx_exchange_kind(void);
/* Device properties of one slot a caller needs for roofline reporting. */
int kgx_device_info(int slot, char* name, size_t name_len, char* arch, size_t arch_len,
                    int* compute_units, uint64_t* hbm_bytes);
/* The library's own (non-blocking) hipStream_t of a slot, used by every host-returning entry point. */
void* kgx_stream(int slot);
/* Wait for all work of the library on every bound device. */
int kgx_synchronize(void);

/* ---- population shard: replaces PopulationDB→VariantDBVariant (kgl_variant_db_variant.h:53-76)
 *      D[g][v] uint8 rows become variant-major 2-bit rows in HBM.                                */

/* n_genomes genomes (split over the bound devices in contiguous shards of whole 64-genome chunks), n_variants rows.
 * Rows are zero (all reference-homozygous) after create. */
kgx_pop* kgx_population_create(uint64_t n_genomes, uint64_t n_variants);
/* Shard geometry: how many shards, and for shard i its device slot, first genome and genome count. */
uint32_t kgx_population_shards(const kgx_pop* pop);
int kgx_population_shard_info(const kgx_pop* pop, uint32_t shard, int* slot, uint64_t* genome_base, uint64_t* n_genomes);
/* Change the number of variant rows: the rows held so far keep their content, new rows are zero (nobody carries them),
 * the AF column is dropped.  Growing past what is allocated re-allocates with at least half as much again, so a
 * population filled piece by piece (rows uploaded as a VCF is read, the final count unknown until its end) is copied a
 * bounded number of times; shrinking only lowers the count. */
int      kgx_population_resize(kgx_pop* pop, uint64_t n_variants);
void     kgx_population_destroy(kgx_pop* pop);
uint64_t kgx_population_genomes(const kgx_pop* pop);
uint64_t kgx_population_variants(const kgx_pop* pop);
/* Device row pitch in bytes of the first shard: ceil(shard genomes/4) rounded up to 16, or to 128 when a row exceeds
 * 512 B (line-aligned rows: every 1 KiB wave load then covers whole 128-B lines). */
uint64_t kgx_population_row_pitch(const kgx_pop* pop);
/* Algorithmic HBM bytes of one allele-count sweep, summed over the shards: n_variants*ceil(shard genomes/4) +
 * 16*n_variants each (one shard: n_variants*ceil(n_genomes/4) + 16*n_variants). */
uint64_t kgx_population_sweep_bytes(const kgx_pop* pop);

/* Host → device, packed 2-bit rows [v0,v1): src row r at src + r*src_pitch, ceil(G/4) bytes used,
 * genome g of the shard in bits (2*(g%4)) of byte g/4. */
int kgx_population_load_dosage2(kgx_pop* pop, const uint8_t* src, uint64_t src_pitch,
                                uint64_t v0, uint64_t v1);
/* Host → device from the reference's own layout: VariantDBGenomeData rows, one uint8 dosage vector
 * of n_variants per genome (kgl_variant_db_variant.h:49-51).  Genomes [g0,g1) of this shard;
 * src row (g-g0) at src + (g-g0)*n_variants.  Dosage >2 is stored as code 3. Packed on device. */
int kgx_population_load_dosage_u8(kgx_pop* pop, const uint8_t* src, uint64_t g0, uint64_t g1);
/* Device → host copy of packed rows [v0,v1) with dst_pitch >= ceil(G/4) (tests). */
int kgx_population_read_dosage2(const kgx_pop* pop, uint8_t* dst, uint64_t dst_pitch,
                                uint64_t v0, uint64_t v1);
/* Genome-level filter, evaluated on the device by every sweep: keep[g] != 0 = genome g takes part, NULL = all do again.
 * Replaces the genome filters the PfEMP package applies before counting -- Pf7SampleResource::filterPassQCGenomes
 * (kga_analytic/kga_analysis_library/kga_analysis_lib_PfFilter.cpp:124-158) and Pf7FwsResource::viewFilterFWS
 * (kgl_genomics/kgl_parser/kgl_pf7_fws_parser.cpp:55-70), both PopulationDB::viewFilter(GenomeListFilter) -- without
 * re-flattening or re-uploading the population.  Under a mask the results are those of the filtered PopulationDB:
 * K2 counts over the kept genomes (referenceHomozygous = kept - carriers; a row left with no carrier -- kept,0,0,0 -- is a
 * variant the filtered population does not hold: the reference has no entry for it, the caller skips it), K3 leaves such
 * rows out of every kept genome's referenceHomozygous, and the genomes left out read as zero in K3 / K8 results. */
int kgx_population_set_genome_mask(kgx_pop* pop, const uint8_t* keep /* host [n_genomes] or NULL */);
/* Per-variant allele frequency as parsed from VCF INFO (float32, the reference's storage type,
 * kgl_parser/kgl_variant_factory_vcf_parse_info.h:27-37); NaN = missing. */
int kgx_population_set_af(kgx_pop* pop, const float* af /* [n_variants] */);
int kgx_population_get_af(const kgx_pop* pop, float* af /* [n_variants] */);

/* Fill the shard with the synthetic biallelic population of SURVEY.md §8(d) directly in HBM:
 * Philox4x32-10 keyed by `seed`, AF[v] = float32(U[0.01,0.5]), genome (genome_base+g) has
 * inbreeding F = -0.5 + 0.01*((genome_base+g) % 101), genotype drawn from the reference's
 * Hardy-Weinberg-with-inbreeding class probabilities (kga_analysis_inbreed_freq.cpp:127-205,221-261).
 * The host twin kgx_synth_biallelic_host() produces identical bits. */
int kgx_population_synth_biallelic(kgx_pop* pop, uint64_t seed, uint64_t genome_base,
                                   uint64_t variant_base);
/* Host twin of the device generator (no device needed): packed rows [v0,v1) for genomes
 * [genome_base, genome_base+n_genomes), dst_pitch bytes per row; af_out may be NULL. */
int kgx_synth_biallelic_host(uint64_t seed, uint64_t genome_base, uint64_t n_genomes,
                             uint64_t v0, uint64_t v1, uint8_t* dst, uint64_t dst_pitch,
                             float* af_out);

/* ---- K2: per-variant allele summary = VariantDBVariant::summaryByVariant for every variant
 *      (kgl_variant_db_variant.cpp:126-178, caller kga_analysis_PfEMP_FWS.cpp:41-70).
 *      out[v] = { referenceHomozygous, minorHeterozygous, minorHomozygous, nonDiploid } (uint32 each). */
int kgx_allele_count_by_locus(kgx_pop* pop, uint32_t* out /* host [n_variants][4] */);
/* Same, result left in device memory on the FIRST slot's device (caller's buffer, e.g. a torch tensor the caller
 * all-reduces across processes); asynchronous on `stream`, taken literally as a hipStream_t of that device (NULL = the
 * legacy default stream, which is what torch.cuda.current_stream().cuda_stream is unless a side stream is current).
 * With several shards the other devices sweep on the library's streams, the exchange follows, and `stream` is made to
 * wait for the summed counts. */
int kgx_allele_count_by_locus_dev(kgx_pop* pop, void* d_out /* device [n_variants][4] u32 */,
                                  void* stream);
/* Epilogue on (all-reduced) device counts: af[v] = (het + 2*hom) / (2*total_genomes) in fp64. */
int kgx_allele_frequency_dev(const void* d_counts /* device [n][4] u32 */, uint64_t n_variants,
                             uint64_t total_genomes, void* d_af /* device [n] f64 */, void* stream);
/* Timed repetition of the K2 launch with HIP events on the launch stream: per-iteration kernel
 * durations (ms) into ms_each[iters] (several shards: the slowest shard's kernel of each iteration; no exchange).
 * d_out as above. */
int kgx_allele_count_timed(kgx_pop* pop, void* d_out, void* stream, int warmup, int iters,
                           float* ms_each);

/* ---- K3: per-genome allele summary = VariantDBVariant::summaryByGenome for every genome
 *      (kgl_variant_db_variant.cpp:180-231), restricted to variants with mask[v] != 0
 *      (mask NULL = all): the FWS allele-frequency bins of kga_analysis_PfEMP_FWS.cpp:15-38,72-101.
 *      out[g] = { referenceHomozygous, minorHeterozygous, minorHomozygous, nonDiploid } (uint64 each). */
int kgx_count_by_genome(kgx_pop* pop, const uint8_t* variant_mask /* host [n_variants] or NULL */,
                        uint64_t* out /* host [n_genomes][4] */);
/* All bins in one call: bin_of_variant[v] in [0,n_bins) or 0xFF = in no bin.
 * out[g][b] = 4 x uint64 as above. */
int kgx_count_by_genome_binned(kgx_pop* pop, const uint8_t* bin_of_variant /* host [n_variants] */,
                               uint32_t n_bins, uint64_t* out /* host [n_genomes][n_bins][4] */);

/* The same with the bin of every row decided ON THE DEVICE from the population's AF column (kgx_population_set_af), the
 * P7FrequencyFilter pair of kga_analysis_PfEMP_FWS.cpp:15-38 (kgl_variant_filter_Pf7.cpp:20-66) as a per-launch
 * predicate: row v is in bin b iff  af[v] >= bin_edges[b]  and not  af[v] >= bin_edges[b + 1]  (compared as doubles: a
 * row whose AF passes the lower filter and fails the upper one); NaN = no value = in no bin.  Nothing but the n_bins + 1
 * edges travels to the device per call. */
int kgx_count_by_genome_af_bins(kgx_pop* pop, const double* bin_edges /* host [n_bins + 1] */, uint32_t n_bins,
                                uint64_t* out /* host [n_genomes][n_bins][4] */);

/* Device time (HIP events on the library streams) of the k_count_by_genome kernel of the most recent by-genome sweep
 * (the slowest shard's); 0 before the first or when no row was selected.  Algorithmic bytes of that launch:
 * selected rows x ceil(G/4) + 32 x G x n_bins. */
double kgx_count_by_genome_last_ms(void);

/* ---- K8: compound offsets of HeteroHomoZygous::updateVariantAnalysisType
 *      (kga_analytic/kga_PfEMP/kga_analysis_PfEMP_heterozygous.cpp:61-105).  A group is a contig offset at
 *      which the population holds n_rows[i] >= 2 distinct variants, stored as adjacent rows starting at
 *      first_row[i]; bin[i] in [0,n_bins) selects the output slot (the contig).  Per genome and bin:
 *      out[g][bin] = { heterozygous_reference_minor (exactly one copy there),
 *                      homozygous_minor (distinct variants carried when >= 2 copies; the reference's quirk),
 *                      heterozygous_minor (variants carried exactly once when >= 2 copies) }.
 *      Offsets with one row follow from kgx_count_by_genome_binned and are not passed here. */
int kgx_compound_offsets(kgx_pop* pop, const uint32_t* first_row, const uint32_t* n_rows, const uint32_t* bin,
                         uint64_t n_groups, uint32_t n_bins, uint64_t* out /* host [n_genomes][n_bins][3] */);
/* The same for groups whose rows are NOT adjacent (a population whose rows are in file order rather than HGVS order):
 * group i is the rows member_rows[first_member[i] .. first_member[i] + n_rows[i]). */
int kgx_compound_offsets_listed(kgx_pop* pop, const uint32_t* member_rows, uint64_t n_members, const uint32_t* first_member,
                                const uint32_t* n_rows, const uint32_t* bin, uint64_t n_groups, uint32_t n_bins,
                                uint64_t* out /* host [n_genomes][n_bins][3] */);

/* The offset filters of kgl_genomics/kgl_variant_filter/kgl_variant_filter_db_offset.cpp as device-side counting predicates:
 * out[g][b] = { HomozygousFilter (:17-58), HeterozygousFilter (:66-101), DiploidFilter (:110-129), UniqueUnphasedFilter
 * (:137-156) } = the Variant objects PopulationDB::viewFilter(F) leaves genome g in bin b (its variantCount() there).
 * Offsets holding >= 2 distinct variants come as row lists, as for kgx_compound_offsets_listed; every other row is an
 * offset of its own and single_bin[row] names its bin (0xFF: not counted; required for the rows of the lists).  A cell
 * with more than two copies of one variant (code 3) counts as "more than two" wherever the exact number does not matter
 * -- it never does for these four filters.  Honours the genome mask. */
int kgx_offset_filter_counts(kgx_pop* pop, const uint8_t* single_bin /* host [n_variants] */, const uint32_t* member_rows, uint64_t n_members,
                             const uint32_t* first_member, const uint32_t* n_rows, const uint32_t* bin, uint64_t n_groups, uint32_t n_bins,
                             uint64_t* out /* host [n_genomes][n_bins][4] */);

/* UniquePhasedFilter (kgl_variant_filter_db_offset.cpp:160-...): one Variant object per distinct (HGVS, phase) of an offset.
 * Phase is not in the 2-bit rows; it comes as a PHASE PLANE beside them -- one bit per (variant row, genome), variant-major,
 * genome g in bit g % 8 of byte g / 8 of its row: set where the genome's copies of the variant sit on BOTH phases (a|a
 * homozygote; never for unphased data, where every copy carries VariantPhase::UNPHASED).  The filter then leaves, per
 * genome and bin, (variants carried) + (plane bits set): out[g][b] -- exact for copies on one or two distinct phases, which is
 * all the reference's parsers produce (A and B of a phased diploid; UNPHASED alone); a genome whose copies of ONE variant sat on
 * three or more distinct phases would count one short per extra phase (the host flattener counts such cells and the package
 * warns: FlatPopulation::cells_with_three_phases).  bin_of_variant: host [n_variants], 0xFF = row not
 * counted; NULL with n_bins == 1: every row.  Honours the genome mask.  The plane is optional (half the size of the rows);
 * kgx_population_resize carries it along. */
int kgx_population_load_phase_plane(kgx_pop* pop, const uint8_t* src, uint64_t src_pitch /* >= ceil(n_genomes / 8) */, uint64_t v0, uint64_t v1);
int kgx_unique_phased_counts(kgx_pop* pop, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out /* host [n_genomes][n_bins] */);

/* ---- Genome-major row lists: for every genome of [g0, g1) the rows it carries (dosage > 0), ascending -- the visit
 *      GenomeDB::processAll makes of one genome, on which VariantSort::variantGenomeIndexMT builds its per-genome
 *      identifier maps (kgl_genomics/kgl_variant_analysis/kgl_variant_sort.cpp:234-306, one pool task per genome; the
 *      IndexMap keeps Variants with a non-empty identifier, :245-255).  A sparse transpose of the bit matrix on the device.
 *      row_selected: host [n_variants] or NULL (every row) -- e.g. the rows whose variant bears an identifier.
 *      begin: host [g1 - g0 + 1], begin[i] .. begin[i+1] = genome g0 + i's entries in rows; rows: host [capacity], or NULL
 *      to size the call (begin[g1 - g0] = entries needed).  Genomes a genome mask leaves out have no entries. */
int kgx_genome_row_lists(kgx_pop* pop, uint64_t g0, uint64_t g1, const uint8_t* row_selected /* or NULL */, uint64_t* begin,
                         uint32_t* rows /* or NULL */, uint64_t capacity);

/* ---- K4: VariantDBVariant::populationSummary (kgl_variant_db_variant.cpp:234-279). */
int kgx_population_summary(kgx_pop* pop, uint64_t out[4]);

/* ---- inbreeding: kga_analytic/kga_inbreed on a locus-major allele-index matrix -------------------------
 *
 * kgx_gt8: one byte per (locus, genome).  Low nibble = the first SNP variant the genome carries at the
 * locus, as 1 + its index in the locus's reference alt list (0 = none, 15 = a SNP alt the reference list does
 * not hold); high nibble = the second SNP variant (0 = none); 0xFF = three or more.  Indels never enter:
 * INBREED filters the reference population to SNP & PASS (kga_analysis_inbreed.cpp:79) and each genome's
 * contig to SNPs (kga_analysis_inbreed_freq.cpp:436).  The order of the two nibbles is the order of the
 * genome's OffsetDB array (front()/back() at _freq.cpp:462,476,493).  With phased input a byte (a, a) is one variant
 * on both phases (homozygous); two copies of one variant on the SAME phase -- a repeated VCF record -- are written
 * (0, a): analogous but not homozygous() (kgl_variant_db.h:135-143), which the reference classifies as a minor
 * heterozygote with that allele twice.  Bytes outside this description are skipped, never read through. */
typedef struct kgx_gt8 kgx_gt8;
/* Genomes are split over the bound devices in contiguous shards of whole 128-genome units (kgx_init). */
kgx_gt8* kgx_gt8_create(uint64_t n_genomes, uint64_t n_loci);
uint32_t kgx_gt8_shards(const kgx_gt8* gt);
int kgx_gt8_shard_info(const kgx_gt8* gt, uint32_t shard, int* slot, uint64_t* genome_base, uint64_t* n_genomes);
void     kgx_gt8_destroy(kgx_gt8* gt);
uint64_t kgx_gt8_genomes(const kgx_gt8* gt);
uint64_t kgx_gt8_loci(const kgx_gt8* gt);
/* Algorithmic bytes of one frequency sweep: G*n_selected + 8*amax*n_selected + 80*G. */
uint64_t kgx_gt8_sweep_bytes(uint64_t n_genomes, uint64_t n_selected, uint32_t amax);
/* Host -> device.  kgx_gt8_load: genome-major source, genomes [g0,g1), row (g-g0) = n_loci bytes (transposed
 * on the device).  kgx_gt8_load_rows: locus-major source rows [l0,l1) of n_genomes bytes at src_pitch. */
int kgx_gt8_load(kgx_gt8* gt, const uint8_t* src, uint64_t g0, uint64_t g1);
int kgx_gt8_load_rows(kgx_gt8* gt, const uint8_t* src, uint64_t src_pitch, uint64_t l0, uint64_t l1);
int kgx_gt8_read_rows(const kgx_gt8* gt, uint8_t* dst, uint64_t dst_pitch, uint64_t l0, uint64_t l1);
/* Offsets with MORE than 14 reference alts (AlleleFreqVector has no cap, kga_analysis_inbreed_freq.cpp:18-57; a multi-base SNP
 * record can spell out dozens): their cells do not fit two 4-bit indices.  Such a row of the matrix gets a WIDE ROW beside it:
 * cells[i][g] = a1 | a2 << 8 for row locus[i] (ascending) -- two 8-bit indices into the locus's reference alt list, 0 = none,
 * 255 = an alt the list does not hold, 0xFFFF = three or more variants, the pairs as the bytes' (kgx_gt8_load_rows); what the
 * byte row holds there is ignored by a call whose amax exceeds 14.  kgx_inbreed then takes minor_af with amax up to 254 columns
 * and runs its generic per-cell kernels (no table passes, no one-launch iteration, no moments: such offsets are rare).
 * n_wide = 0 drops the wide rows.  Replaces any set before. */
int kgx_gt8_set_wide_rows(kgx_gt8* gt, uint64_t n_wide, const uint32_t* locus /* [n_wide] */, const uint16_t* cells /* [n_wide][cells_pitch] */,
                          uint64_t cells_pitch /* >= n_genomes */);

/* LocusResults (kga_analysis_inbreed_output.h:21-35) without the genome id, same field order. */
typedef struct kgx_locus_results {
  uint64_t major_hetero_count;  double major_hetero_freq;
  uint64_t minor_hetero_count;  double minor_hetero_freq;
  uint64_t minor_homo_count;    double minor_homo_freq;
  uint64_t major_homo_count;    double major_homo_freq;
  uint64_t total_allele_count;  double inbred_allele_sum;
} kgx_locus_results;

/* InbreedingCalculation::algoMap() (kga_analysis_inbreed_calc.h:93-117). */
#define KGX_ALGO_RITLAND_LOCUS  0
#define KGX_ALGO_SIMPLE         1
#define KGX_ALGO_HALL_ME        2
#define KGX_ALGO_LOGLIKELIHOOD  3

/* K6 alone: per locus { majorAlleleFrequency, majorHom, majorHet, minorHom, minorHet } =
 * AlleleFreqVector::alleleClassFrequencies(inbreeding) (kga_analysis_inbreed_freq.cpp:119-217) from the locus's
 * minor allele frequencies minor_af[l][amax] (amax <= 254) (NaN = alt absent from the AlleleFreqVector); valid[l] =
 * checkValidAlleleVector() (:61-75).  valid may be NULL. */
int kgx_locus_class_frequencies(const double* minor_af, uint64_t n_loci, uint32_t amax, double inbreeding,
                                double* out /* [n_loci][5] */, uint8_t* valid /* [n_loci] */);

/* K5+K7: InbreedingAnalysis::processResults for genomes [g0,g1) (g0 a multiple of 4) over one locus list
 * (kga_analysis_inbreed_diploid.cpp:98-166 -> InbreedingCalculation::process*, _calc.cpp).
 * locus_index[n_selected]: rows of the matrix, ascending (NULL = rows 0..n_selected-1);
 * minor_af[n_selected][amax]: super-population allele frequency of each reference alt (double(float32)), NaN = the
 * alt is not in the locus's AlleleFreqVector; phased != 0 when the two copies of a homozygous alt carry different
 * phases (1000-Genomes style); algorithm = KGX_ALGO_*.  out[g1-g0] (host).  locus_index and minor_af may be host
 * pointers or pointers to memory of a bound device (a table kept resident between calls is then copied device to
 * device).  Genomes are independent: every shard the range touches is swept on its own device at the same time, no exchange.
 * start[g1-g0] (host) or NULL: where HallME / Loglikelihood start for each genome.  The reference draws its starts from
 * std::random_device-seeded Mersenne twisters (kel_math/kel_distribution.h:25-43; _calc.cpp:163-166,180,235-237) and
 * restarts five times, of which -- RetryCalcResult::checkTolerance compares every entry with itself (_calc.cpp:45-68) --
 * the FIFTH alone decides the result: kgx_inbreed_reference_starts() makes exactly those draws, so passing its output is
 * the reference's algorithm at no extra pass.  NULL: the midpoints of its start intervals (0.25 / 0.0), a deterministic
 * mode the reference does not have.  HallME runs the reference's 50 expectation steps from the start -- over a selection
 * of more than 8192 loci (up to there the whole iteration is one kernel launch) on per-genome moments of the homozygous
 * cells' allele frequencies (ONE more pass over the genotype bytes, which leaves the hits of every class of homozygous cell
 * as rows of bits; the moments from those rows as an exact integer product on the matrix cores; then the 50 steps on those
 * numbers: the step's sum is expanded about the centre of each frequency bin and cut below 1e-12 of a term, measured 6e-16
 * of F from the 50 passes; KGX_K7_HALL_PASSES=1 or a frequency outside [2^-20, 1] u {0} makes the 50 passes over the bytes
 * instead; KGX_K7_CLASS_BYTES=1 / KGX_K7_CLASS_SWEEPS=1: the moments by a pass over the bytes per class, on the matrix cores /
 * by vector adds -- the checkers of the bit rows); Loglikelihood
 * walks nlopt's 1-D Nelder-Mead from it with the reference's stopping rule (absolute simplex width 1e-6, at most 500
 * evaluations; _calc.cpp:131-144); a pass over the genotype bytes serves two evaluations (a simplex' reflection and
 * inside contraction) of every genome still searching.  Ignored by Simple and RitlandLocus. */
int kgx_inbreed(kgx_gt8* gt, uint64_t g0, uint64_t g1, const uint32_t* locus_index, uint64_t n_selected,
                const double* minor_af, uint32_t amax, int phased, int algorithm, const double* start, kgx_locus_results* out);
/* MANY such calls in one: the INBREED package makes one per (window of ~1000 sampled loci, super population), thousands per
 * contig, and windows are independent once sampled (kga_analysis_inbreed_diploid.cpp:53-75; a window's super populations
 * differ only in their frequency rows and genome range) -- so the package samples several windows ahead and hands them over
 * together.  task i is what kgx_inbreed would be given (host pointers only); amax, phased and algorithm are the batch's.
 * Where every task selects 1..8192 loci the batch is ONE copy in, TWO launches (the tasks' class-frequency tables, then one
 * kernel in which a wave or a workgroup per (task, genome) classifies the genome's cells, sums its counts and class
 * frequencies and runs the estimator's whole iteration) and ONE copy out per device; otherwise it is made of kgx_inbreed
 * calls.  Results: those of the n_tasks kgx_inbreed calls -- the integers bit for bit, the fp64 sums up to the order of
 * their additions (<= 1e-12 relative), the iterative estimators as between any two of the library's paths. */
typedef struct {
  uint64_t g0, g1;                 /* genomes [g0, g1), g0 a multiple of 4                       */
  const uint32_t* locus_index;     /* [n_selected] rows of the matrix, ascending; NULL = 0..     */
  uint64_t n_selected;
  const double* minor_af;          /* [n_selected][amax]                                         */
  const double* start;             /* [g1 - g0] or NULL (kgx_inbreed)                            */
  kgx_locus_results* out;          /* [g1 - g0]                                                  */
} kgx_inbreed_task;
int kgx_inbreed_batch(kgx_gt8* gt, const kgx_inbreed_task* tasks, uint32_t n_tasks, uint32_t amax, int phased, int algorithm);
/* The start points the reference's processHallME / processLogLikelihood end up using, for n genomes.  seed > 0: genome i
 * owns the stream std::mt19937_64(seed + first_stream + i) and draws std::uniform_real_distribution<>(0.5, 0) (HallME,
 * _calc.cpp:237) or (0.5, -0.5) (Loglikelihood, :166) once per restart; out[i] = the fifth draw.  seed 0: fresh entropy
 * from std::random_device, the reference's RandomEntropySource, i.e. production behaviour -- the same five draws per
 * genome, all genomes of the call from one twister seeded from the random device (the draws are independent uniforms
 * either way; seeding a twister per genome would cost more than a window-sized sweep).  Host only, no device. */
int kgx_inbreed_reference_starts(int algorithm, uint64_t seed, uint64_t first_stream, uint64_t n, double* out /* [n] */);
/* kgx_inbreed and the by-genome sweeps keep their per-call device buffers in one grow-only arena per device between calls
 * (a window loop calls kgx_inbreed thousands of times), and a large Loglikelihood call two more buffers holding the
 * genotype columns of the genomes still searching (at most ~3/4 of the swept bytes together); this frees them (they are
 * re-created when needed). */
int kgx_release_scratch(void);
/* (A HallME / Loglikelihood call over a large selection adds its moments -- 40 B per 1024 selected loci and genome, 100 KB of
 * frequency bins per genome -- and, Loglikelihood, a bit per cell of the bins its floor can reach to those buffers: ~6 + 15 GB
 * at 10,000 genomes x 5 M loci.  They are kept for the next such call while together they stay under a quarter of the device's
 * memory (KGX_KEEP_SCRATCH_GB=n: under n GiB) and freed at the end of the call otherwise; a call whose moments would not
 * fit into half of what is free makes the passes over the bytes instead, which need none of it.) */
/* Device time (HIP events on the library streams) of the frequency sweep -- locus helpers + the K5 kernel, i.e. the one
 * pass over the genotype bytes that every estimator makes -- of the most recent successful kgx_inbreed call (the
 * slowest shard's); 0 before the first, and 0 after a call over fewer than 2^24 cells (selected loci x genomes of a
 * shard): those are not timed, four event records cost such a call more than its sweep takes (KGX_TIME_SMALL_CALLS=1
 * times them all the same).  For bench.py / profiles: algorithmic bytes = kgx_gt8_sweep_bytes(). */
double kgx_inbreed_last_sweep_ms(void);
/* ... and of the one kernel inside it that reads the genotype bytes (k_inbreed_eval_lut<3|4>, or the SWAR / generic
 * sweep): the sweep without the per-locus helper kernels (tables, entries, segment defaults). */
double kgx_inbreed_last_kernel_ms(void);
/* Where the most recent successful kgx_inbreed call ran on per-genome moments (HallME, Loglikelihood over more than 8192
 * loci): device time of its class passes over the genotype bytes (one per class of homozygous cell, with the merges of
 * their items) and of the kernel that then iterates / searches on the moments; 0 otherwise. */
double kgx_inbreed_last_moments_ms(void);
double kgx_inbreed_last_search_ms(void);
/* Objective evaluations the most recent KGX_ALGO_LOGLIKELIHOOD call needed (the most any genome made; where the call
 * made passes over the genotype bytes, the passes). */
int kgx_inbreed_last_evaluations(void);
/* What the most recent successful kgx_inbreed call ran on -- so that a test of one path cannot quietly exercise another. */
#define KGX_PATH_NONE                       0
#define KGX_PATH_FREQUENCY_SWEEP            1   /* Simple, RitlandLocus: the one sweep every estimator starts with            */
#define KGX_PATH_ONE_LAUNCH                 2   /* HallME / Loglikelihood, <= 8192 loci: the whole iteration in one launch     */
#define KGX_PATH_HALL_MOMENTS               3   /* HallME on per-genome moments (one pass per class of homozygous cell)        */
#define KGX_PATH_HALL_PASSES                4   /* HallME by 50 passes over the bytes                                          */
#define KGX_PATH_LOGLIK_MOMENTS             5   /* Loglikelihood on the same moments + the exact walk next to the floor        */
#define KGX_PATH_LOGLIK_MOMENTS_AND_PASSES  6   /* ... and passes for the genomes the statistics could not serve               */
#define KGX_PATH_LOGLIK_PASSES              7   /* Loglikelihood by passes over the bytes (two evaluations per pass)           */
int kgx_inbreed_last_path(void);
/* A diagnostic (tests, scripts): the log-likelihood logLikelihood(F) (kga_analysis_inbreed_calc.cpp:94-129) of genomes
 * [g0,g1) at the points at[g1-g0] (each in [-1, 1]) over the selection kgx_inbreed would be given -- by_passes == 0: from the
 * per-genome moments and the exact walk next to the floor, as a large Loglikelihood call evaluates it (KGX_ESTATE when
 * such a call would not run on them; NaN for a genome it would hand to the passes); by_passes != 0: by one table pass over
 * the genotype bytes.  value[g1-g0] (host). */
int kgx_inbreed_objective(kgx_gt8* gt, uint64_t g0, uint64_t g1, const uint32_t* locus_index, uint64_t n_selected,
                          const double* minor_af, uint32_t amax, int phased, const double* at, int by_passes, double* value);

/* Synthetic multi-allelic SNP+indel population (BASELINE.json configs[4]) written straight into
 * the matrix: 1/2/3 alts (70/20/10 %), 15 % of alts are indels, AFs rescaled to sum <= 0.6, genotypes drawn from
 * the reference's class probabilities at F = -0.5 + 0.01*((genome_base+g) % 101).  af_table (host, may be NULL)
 * receives the SNP alt frequencies [n_loci][3], NaN padded.  kgx_synth_multiallelic_host is the bit-identical
 * host twin for loci [l0,l1): gt8 bytes [l1-l0][pitch] and/or the raw allele pairs [l1-l0][n_genomes][2]
 * (1-based over ALL alts, 0 = reference); kgx_synth_locus_host describes one locus. */
int kgx_gt8_synth_multiallelic(kgx_gt8* gt, uint64_t seed, uint64_t genome_base, uint64_t locus_base, double* af_table);
int kgx_synth_multiallelic_host(uint64_t seed, uint64_t genome_base, uint64_t n_genomes, uint64_t l0, uint64_t l1,
                                uint8_t* gt8, uint64_t pitch, double* af_table, uint8_t* alleles);
int kgx_synth_locus_host(uint64_t seed, uint64_t l, int* n_alt, float af[3], int is_indel[3]);
/* The same for loci [l0,l1) at once: n_alt[l1-l0], af[l1-l0][3], is_indel[l1-l0][3] (uint8). */
int kgx_synth_loci_host(uint64_t seed, uint64_t l0, uint64_t l1, uint8_t* n_alt, float* af, uint8_t* is_indel);

/* The reference's synthetic-inbreeding self-check population (InbreedSynthetic::generateSyntheticPopulation,
 * kga_analysis_inbreed_syngen.cpp:20-196) written into the matrix: locus l has the minor allele frequencies
 * minor_af[l][amax] (NaN = alt not in the list), genome g the inbreeding coefficient inbreeding[g]; allele classes
 * and alleles are drawn as the reference does, from Philox4x32-10 keyed by `seed` (the reference uses std::random_device).
 * Homozygous pairs carry two phases: analyse with phased = 1. */
int kgx_gt8_synth_inbred(kgx_gt8* gt, const double* minor_af, uint32_t amax, const double* inbreeding /* [n_genomes] */,
                         uint64_t seed);

#ifdef __cplusplus
}
#endif

#endif /* KGX_H */
