// HIP kernels for gfx950 (MI355X, CDNA4): the inbreeding sweeps (K5/K6/K7) over allele-index bytes.  Byte / fp64 work
// per cell: HBM- or VALU-bound, no MFMA.  Included only by kgx_inbreed.hip.
#ifndef KGX_KERNELS_INBREED_H
#define KGX_KERNELS_INBREED_H
// (Two translation units include this header -- kgx_inbreed.hip, kgx_window.hip: the kernels that are not templates are
// `static`, each unit keeping its own copy of those it launches.)

#include "kgx_kernels_common.h"
#include "kgx_synth.h"
#include "kgx_synth_multi.h"

namespace kgx {

// ---------------------------------------------------------------------------------------------
// K5/K6/K7  inbreeding sweep (kga_analytic/kga_inbreed).
//
// HBM layout ("gt8"): locus-major byte rows, row l at gt + l*pitch, one byte per genome:
//   low nibble  = first SNP variant the genome carries at the locus, as 1 + index into the locus's
//                 reference alt list (0 = none, 15 = a SNP alt the reference list does not hold)
//   high nibble = second SNP variant (0 = none);  0xFF = three or more SNP variants.
// Indel alleles never appear: INBREED filters both sides to SNPs (kga_analysis_inbreed.cpp:79,
// kga_analysis_inbreed_freq.cpp:436).  A lane owns 4 consecutive genomes (one dword per locus row), a wave
// 256 genomes; everything indexed by locus (allele frequencies, class frequencies) is wave-uniform.
//
// Per-locus table row (K6, computed once per locus instead of once per genome per locus as the
// reference does at _freq.cpp:444,552):  af[amax] (NaN = alt not in the AlleleFreqVector), p_major =
// majorAlleleFrequency(), cf[4] = alleleClassFrequencies(0.0) {majorHom, majorHet, minorHom, minorHet}.
// ---------------------------------------------------------------------------------------------
constexpr int kTableExtra = 5;   // p_major + 4 class frequencies after the af[amax] slots
// The sweep's table row is wider: [af[amax] | p_major | cf[4] | 1/af[amax] | 1/p_major]  (reciprocals once per locus
// instead of one fp64 divide per genome per locus in processRitlandLocus, _calc.cpp:398).
__host__ __device__ constexpr uint32_t sweep_stride(uint32_t amax) { return 2 * amax + kTableExtra + 1; }
enum : uint8_t { kLocusValid = 1, kLocusDefault = 2, kLocusRitlandDefault = 4 };   // valid[] flag bits

__device__ __forceinline__ double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// K6: AlleleFreqVector (ctor clamp, _freq.cpp:47), checkValidAlleleVector (:61-75), majorAlleleFrequency
// (:119-123), unadjustedAlleleClassFrequencies(0.0) + normalize (:127-217), in the reference's operation order.
// valid[l] = 0 marks a locus generateFrequencies skips (:445-449).
template <bool WIDE>
__global__ void __launch_bounds__(kBlock)
k_locus_tables(const double* __restrict__ af_in, uint64_t n_loci, uint32_t amax, double inbreeding,
               double* __restrict__ table, uint8_t* __restrict__ valid) {
  const uint32_t stride = WIDE ? sweep_stride(amax) : amax + kTableExtra;
  // A thread's row is `stride` doubles: written straight to memory, neighbouring threads' stores are `stride` apart
  // (0.84 ms for 5 M loci at amax = 3: 480 MB at a tenth of the bandwidth).  Rows of up to kTileStride doubles go through
  // LDS instead and leave as whole lines.
  constexpr uint32_t kTileStride = 16;
  __shared__ double tile[kBlock * kTileStride];
  const bool tiled = stride <= kTileStride;
  for (uint64_t l0 = static_cast<uint64_t>(blockIdx.x) * blockDim.x; l0 < n_loci; l0 += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
  const uint64_t l = l0 + threadIdx.x;
  if (l < n_loci) {
    const double* in = af_in + l * amax;
    double* out = tiled ? tile + threadIdx.x * stride : table + l * stride;
    double sum_minor = 0.0;
    uint32_t n = 0;
    for (uint32_t a = 0; a < amax; ++a) {
      double f = in[a];
      if (f == f) { f = clamp01(f); sum_minor += f; ++n; }
      out[a] = f;
    }
    const bool is_valid = n > 0 && !((sum_minor - 1.0) > 1.0e-5);
    const double p_major = clamp01(1.0 - clamp01(sum_minor));             // majorAlleleFrequency()
    out[amax] = p_major;
    // flag bits: a reference-homozygous genome is MAJOR_HOMOZYGOUS here iff p_major > 0.01 (_freq.cpp:531-539) and
    // enters the Ritland sum iff p_major > 0.001 (_calc.cpp:380,396)
    valid[l] = !is_valid ? 0 : static_cast<uint8_t>(kLocusValid | (p_major > 0.01 ? kLocusDefault : 0) |
                                                    ((p_major > 0.01 && p_major > 0.001) ? kLocusRitlandDefault : 0));
    if constexpr (WIDE) {
      for (uint32_t a = 0; a < amax; ++a) out[amax + kTableExtra + a] = 1.0 / out[a];
      out[2 * amax + kTableExtra] = 1.0 / p_major;
    }
    const double major_frequency = (1.0 - sum_minor) > 0.0 ? (1.0 - sum_minor) : 0.0;
    const bool rescale = sum_minor > 1.0;
    double minor_hom = 0.0, major_het = 0.0, minor_het = 0.0;
    for (uint32_t a = 0; a < amax; ++a) {
      const double f0 = out[a];
      if (!(f0 == f0)) continue;
      const double f = rescale ? f0 / sum_minor : f0;
      minor_hom += (inbreeding * f) + ((1.0 - inbreeding) * f * f);
    }
    for (uint32_t a = 0; a < amax; ++a) {
      const double fa0 = out[a];
      if (!(fa0 == fa0)) continue;
      const double fa = rescale ? fa0 / sum_minor : fa0;
      for (uint32_t b = a + 1; b < amax; ++b) {
        const double fb0 = out[b];
        if (!(fb0 == fb0)) continue;
        const double fb = rescale ? fb0 / sum_minor : fb0;
        minor_het += (1.0 - inbreeding) * 2.0 * fa * fb;
      }
    }
    double major_hom = (inbreeding * major_frequency) + ((1.0 - inbreeding) * major_frequency * major_frequency);
    for (uint32_t a = 0; a < amax; ++a) {
      const double f0 = out[a];
      if (!(f0 == f0)) continue;
      const double f = rescale ? f0 / sum_minor : f0;
      major_het += (1.0 - inbreeding) * 2.0 * major_frequency * f;
    }
    major_hom = major_hom > 0.0 ? major_hom : 0.0;
    major_het = major_het > 0.0 ? major_het : 0.0;
    minor_hom = minor_hom > 0.0 ? minor_hom : 0.0;
    minor_het = minor_het > 0.0 ? minor_het : 0.0;
    const double sum_freqs = major_hom + major_het + minor_hom + minor_het;
    out[amax + 1] = major_hom / sum_freqs;
    out[amax + 2] = major_het / sum_freqs;
    out[amax + 3] = minor_hom / sum_freqs;
    out[amax + 4] = minor_het / sum_freqs;
  }
  if (tiled) {                                              // block-uniform
    __syncthreads();
    const uint64_t rows = n_loci - l0 < blockDim.x ? n_loci - l0 : blockDim.x;
    for (uint64_t i = threadIdx.x; i < rows * stride; i += blockDim.x) table[l0 * stride + i] = tile[i];
    __syncthreads();
  }
  }
}

enum : int { kClassNone = 0, kMajorHom = 1, kMajorHet = 2, kMinorHom = 3, kMinorHet = 4 };

// generateFrequencies' per-locus decision (_freq.cpp:452-543) for one cell: the genome's (up to) two SNP variants at the offset
// as 1 + their place in the locus's reference alt list (0 none; an index past amax: an alt the list does not hold -- the
// matrix bytes' 15, a wide row's 255 -- or the "three or more variants" marker).
__device__ __forceinline__ int classify_pair(uint32_t a1, uint32_t a2, const double* __restrict__ row, uint32_t amax, bool phased,
                                             double& f1, double& f2) {
  const double p_major = row[amax];
  if ((a1 | a2) == 0u) {
    if (p_major > 0.01) { f1 = p_major; f2 = p_major; return kMajorHom; }   // minimum_major_frequency (:531-539)
    return kClassNone;
  }
  if (a1 > amax || a2 > amax) return kClassNone;                              // unknown alt, >= 3 variants, past the table
  if (a1 == 0u) {
    // (0, a): two copies of alt a on ONE phase (a repeated VCF record).  They are analogous but not homozygous()
    // (kgl_variant_db.h:135-143), so the offset takes the two-variant branch with the same allele found twice
    // (_freq.cpp:488-506): MINOR_HETEROZYGOUS with both frequencies that allele's.
    f1 = row[a2 - 1];
    if (!(f1 == f1)) return kClassNone;
    f2 = f1;
    return kMinorHet;
  }
  f1 = row[a1 - 1];
  if (!(f1 == f1)) return kClassNone;                                         // front() not in the AF list
  if (a2 == 0) { f2 = p_major; return kMajorHet; }
  if (a1 == a2 && phased) { f2 = f1; return kMinorHom; }                      // homozygous(): same HGVS, different phase
  f2 = row[a2 - 1];
  if (!(f2 == f2)) return kClassNone;                                         // second minor allele not found (:503-506)
  return kMinorHet;
}

// ... for one genotype byte of the matrix: two 4-bit indices (15: an alt the list does not hold; 0xFF: three or more variants).
// (amax <= 14 for every call that reads bytes alone, so 15 is past the table; a call with wider loci -- amax up to 254 --
// reads those loci's cells from the matrix's WIDE ROWS, 8-bit indices, and a byte's 15 must not index its table.)
__device__ __forceinline__ int classify_cell(uint32_t b, const double* __restrict__ row, uint32_t amax, bool phased,
                                             double& f1, double& f2) {
  const uint32_t a1 = b & 15u, a2 = b >> 4;
  if (a1 == 15u || a2 == 15u) return kClassNone;
  return classify_pair(a1, a2, row, amax, phased, f1, f2);
}

// MODE 0: class counts + class-frequency sums at F=0 + Ritland terms   (generateFrequencies, processRitlandLocus)
// MODE 1: one Hall expectation step:  sum over hom loci of F/(F+(1-F)p)  (processHallME :255-285)
// MODE 2: log-likelihood at F                                            (logLikelihood :94-129)
// Grid: x = genome quads of 256 threads (1024 genomes), y = locus segment.  Per (segment, genome) partials
// go to part[(seg*n_genomes + g)*kParts + k]; integer class counts are added atomically to counts[g][6].
constexpr int kParts0 = 5;   // majorHom, majorHet, minorHom, minorHet frequency sums, Ritland sum
// wide_of_row (may be null): for every row of the matrix the index of its WIDE ROW, 0xFFFFFFFF for none -- an offset with more than
// 14 reference alts: its cells are 16-bit, two 8-bit indices, in wide_cells[wide row][wide_pitch] (kgx_gt8_set_wide_rows).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_inbreed_sweep(const uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t g0, uint64_t n_genomes,
                const uint32_t* __restrict__ locus_index, uint64_t n_sel, uint64_t loci_per_seg,
                const double* __restrict__ table, const uint8_t* __restrict__ valid, uint32_t amax, int phased,
                const double* __restrict__ f_in, unsigned long long* __restrict__ counts, double* __restrict__ part,
                const uint32_t* __restrict__ wide_of_row = nullptr, const uint16_t* __restrict__ wide_cells = nullptr, uint64_t wide_pitch = 0) {
  const uint64_t quad = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // 4 genomes g0 + 4*quad ..
  if (quad * 4 >= n_genomes) return;
  const uint64_t seg = blockIdx.y;
  const uint64_t s_begin = seg * loci_per_seg;
  const uint64_t s_end = s_begin + loci_per_seg < n_sel ? s_begin + loci_per_seg : n_sel;
  const uint32_t stride = sweep_stride(amax);
  const uint64_t col = (g0 >> 2) + quad;                  // g0 is a multiple of 4

  uint32_t cnt[4][6];        // majorHom, majorHet, minorHom, minorHet, total, ritland count
  double acc[4][MODE == 0 ? kParts0 : 1];
  double F[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int k = 0; k < 6; ++k) cnt[j][k] = 0;
#pragma unroll
    for (int k = 0; k < (MODE == 0 ? kParts0 : 1); ++k) acc[j][k] = 0.0;
    const uint64_t g = quad * 4 + j;
    F[j] = (MODE != 0 && g < n_genomes) ? f_in[g] : 0.0;
  }

  for (uint64_t s = s_begin; s < s_end; ++s) {
    if (!(valid[s] & kLocusValid)) continue;
    const uint64_t l = locus_index ? static_cast<uint64_t>(locus_index[s]) : s;
    const uint32_t w = gt[l * dwords_per_row + col];
    const double* row = table + s * stride;
    const uint32_t wide = wide_of_row ? wide_of_row[l] : 0xFFFFFFFFu;              // (the same for the whole workgroup)
    const uint16_t* wide_row = wide != 0xFFFFFFFFu ? wide_cells + static_cast<uint64_t>(wide) * wide_pitch + g0 + quad * 4 : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double f1 = 0.0, f2 = 0.0;
      int cls;
      if (wide_row) {
        const uint32_t pair = quad * 4 + j < n_genomes ? wide_row[j] : 0u;
        cls = classify_pair(pair & 0xFFu, pair >> 8, row, amax, phased != 0, f1, f2);
      } else {
        cls = classify_cell((w >> (8 * j)) & 0xFFu, row, amax, phased != 0, f1, f2);
      }
      if (cls == kClassNone) continue;
      if constexpr (MODE == 0) {
        ++cnt[j][cls - 1];
        ++cnt[j][4];
        acc[j][0] += row[amax + 1];
        acc[j][1] += row[amax + 2];
        acc[j][2] += row[amax + 3];
        acc[j][3] += row[amax + 4];
        if (cls == kMajorHom || cls == kMinorHom) {
          if (f1 > 0.001) {                     // minimum_frequency (_calc.cpp:380,396)
            acc[j][4] += 1.0 / f1;
            acc[j][4] -= 1.0;
            ++cnt[j][5];
          }
        } else {
          acc[j][4] -= 1.0;
          ++cnt[j][5];
        }
      } else if constexpr (MODE == 1) {
        if (cls == kMajorHom || cls == kMinorHom) {
          const double denominator = F[j] + ((1.0 - F[j]) * f1);
          if (denominator != 0) acc[j][0] += F[j] / denominator;
        }
      } else {
        double prob;
        if (cls == kMajorHom || cls == kMinorHom) prob = (F[j] * f1) + ((1.0 - F[j]) * (f1 * f1));
        else prob = 2 * (1.0 - F[j]) * f1 * f2;
        prob = prob < 1e-10 ? 1e-10 : (prob > 1.0 ? 1.0 : prob);
        acc[j][0] += log(prob);
      }
    }
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t g = quad * 4 + j;
    if (g >= n_genomes) continue;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (cnt[j][k]) atomicAdd(&counts[g * 6 + k], static_cast<unsigned long long>(cnt[j][k]));
#pragma unroll
      for (int k = 0; k < kParts0; ++k) part[(seg * n_genomes + g) * kParts0 + k] = acc[j][k];
    } else {
      part[seg * n_genomes + g] = acc[j][0];
    }
  }
}

// The table passes, amax <= 7: the frequency sweep every estimator starts with (MODE 4; MODE 3 for RitlandLocus, with its
// terms from the same read) and the evaluation passes (MODE 1: one Hall expectation step, run 50 times per call; MODE 2:
// the log-likelihood at F, ~56 times).  What a cell contributes depends only on (locus, byte value, F of the genome): an
// entry per (locus, allele index pair) is tabulated once per call (k_eval_entries -- classify_cell decides every entry,
// so the class logic is the generic kernel's own), a pass holds the entries of the batch of 8 loci it is walking in LDS,
// and each cell is one LDS read and two or three more operations:
//   MODE 2  v = the cell's probability.  hom: y = f1*f1, d = f1 - f1*f1   (F*f + (1-F)*f*f,  _calc.cpp:94-129)
//                                        het: y = 2*f1*f2, d = -y         (2*(1-F)*f1*f2)
//                                        unclassified: (1, 0) -> probability 1, log 0.
//           The probabilities, clamped to [1e-10, 1] as the reference clamps them, are multiplied across the batch, the
//           exponent of the running product is peeled off into an integer after every batch, and ONE log per
//           (segment, genome) is taken at the end: sum(log p) = log(prod p).
//   MODE 1  v = the denominator F + (1-F)*f1 of a homozygous cell: y = f1 (8-byte entries: d = 1 - y for every entry),
//           v = fma(1-F, y, F); every other cell y = 1, i.e. v = 1 exactly.  The terms 1/v of ALL cells of a segment are
//           summed as one fraction N/D (N <- N*v + D,
//           D <- D*v, both rescaled by a power of two every other batch; one division per segment), so part[] holds
//           sum_hom 1/v + #(other cells the lane walked);
//           k_hall_update subtracts that count (it is known: loci walked - homozygous cells counted by the frequency
//           sweep) and multiplies by F.  The reference's zero-denominator guard (_calc.cpp:272) can only fire at
//           F = 0, where every term F/v is 0: such a genome is walked with F = 1 (v = 1 everywhere) and the
//           multiplication by its F = 0 happens in k_hall_update.
//   MODE 3  RitlandLocus in ONE pass over the bytes: generateFrequencies (_freq.cpp:425-583) and processRitlandLocus
//           (_calc.cpp:367-431) from the same table read.  y = the cell's Ritland term -- homozygous with f1 > 0.001:
//           1/f1 - 1, heterozygous: -1, anything else 0 -- summed per genome in fp64; the other 8 bytes of the entry are
//           two packed integer words added to two packed counters per genome: lo = majorHom | majorHet << 12,
//           hi = minorHom | minorHet << 12 | odd << 28 (12-bit fields: a segment holds at most kRitlandSegment loci).
//           The four class-frequency sums are class independent, so they are the segment's defaults (pre-filled into
//           part[], k_fill_defaults) corrected where a cell is "odd": unclassified at a locus whose reference-
//           homozygote is classified, classified at a locus whose reference-homozygote is not, a homozygote of an
//           allele with f <= 0.001 (no Ritland term), or a byte past the table.  An odd entry adds 1 to the top four
//           bits of hi and says in the top bits of lo what is odd; once per batch the wave looks at the top of hi and,
//           if any lane saw one, walks the batch's entries again and lets the odd ones adjust the lane's own
//           (segment, genome) partial slot in memory -- a second look at 32 entries, only where such a cell is.
//   MODE 4  MODE 3 without the Ritland term (Simple, and the first pass of HallME / Loglikelihood): 8-byte entries, the
//           two packed words alone.  In both, a segment's class counters leave as one packed word per (segment, genome)
//           (k_reduce_class_counts).
// Bit 7 of the byte (second allele index >= 8) is folded onto bit 3, and every entry with bit 3 set is "unclassified".
// FOLD = false leaves the fold's three operations per dword out: the matrix is known to hold no allele index 8..14
// (kgx_gt8::wide_nibbles == 1) and amax <= 6, so a second index with bit 3 set is 15 and, read without bit 7, 7 > amax --
// an entry nobody tabulates, "unclassified" from the prefill.
// Entry (a1, a2) sits at slot a1 + 20*a2 of the locus's 160-slot table (a1 < 16, a2 < 8: injective), so the 16-byte
// slots of the cells a 16-lane ds_read_b128 group meets together -- a1, a2 in 0..3 -- fall on 16 different bank quads
// ((a1 + 4*a2) mod 16); at slot a1 + 16*a2 every a2 shared a1's banks (SQ_LDS_BANK_CONFLICT: 2.7 extra cycles a read).
// GPL genomes per lane (4 or 8: one dword / dwordx2 load per locus).
#ifndef KGX_EVAL_DEPTH
#define KGX_EVAL_DEPTH 1            // table reads in flight ahead of the arithmetic, in groups of four (mode 2: 16-byte entries, 40 registers of state)
#endif
#ifndef KGX_EVAL_DEPTH_PAIR
#define KGX_EVAL_DEPTH_PAIR 0       // ... mode 2 with two values of F: 80 registers of state, and 24 fp64 operations per group to wait under
#endif
#ifndef KGX_EVAL_DEPTH3
#define KGX_EVAL_DEPTH3 2           // ... modes 1, 3, 4, which have the registers for it
#endif
struct alignas(16) EvalEntry { double y, d; };
constexpr int kEvalBatch = 8;
constexpr uint32_t kEvalSlots = 160;
constexpr uint64_t kRitlandSegment = 4088;        // MODE 3: loci per segment, below the 12-bit class counters' range
constexpr uint32_t kOddCell = 1u << 28;           // MODE 3: in the entry's hi word: one more odd cell (a 4-bit count per batch)
// ... and what is odd about it, in the entry's lo word (bits the 12-bit class counters never reach):
constexpr uint32_t kOddMinus = 1u << 24;          // unclassified carrier at a locus with defaults: its class frequencies come off
constexpr uint32_t kOddPlus = 1u << 25;           // classified at a locus without defaults: they are added
constexpr uint32_t kOddNoRitland = 1u << 26;      // a homozygote of an allele with f <= 0.001: classified, no Ritland term
constexpr uint32_t kOddOutside = 1u << 27;        // a byte past the table: off wherever the locus has defaults
constexpr uint32_t kOddBigHet = 1u << 28;         // entries of mode 5 only: a heterozygous cell with 2*f1*f2 > 1/2 (see k_eval_entries)
// Entries of mode 5 (walked by k_inbreed_eval_lut<3>: its fp64 term is whatever the entry says): what the log-likelihood by
// moments (kgx_kernels_loglik.h) needs of the HETEROZYGOUS cells.  Their probability is u*w, u = 1 - F, w = 2*f1*f2, clamped
// to [1e-10, 1]: with kLoglikTinyHet <= w <= 1/2 the clamp can only bind from below and only for u < 1e-10 / w, so the sum of
// their logs is  sum(log w) + H * log u  and the term is log w.  A cell with w < kLoglikTinyHet is under the floor for every
// u <= 2 (term 0, marked like a cell without Ritland term: it comes off the genome's sixth counter); one with w > 1/2 (two
// copies of an allele more frequent than 1/2 counted as heterozygous: a repeated record, an unphased homozygote) can meet the
// UPPER bound: term 0, and 2^32 onto the sixth counter -- such a genome takes the passes.
// Mode 6: the same marks without the terms, as 8-byte entries for the cheaper <4> pass: sum(log w) does not depend on F, so a SEARCH
// for the maximum does without it (the objective it climbs is the reference's minus that constant); kgx_inbreed_objective, which
// reports the value, takes mode 5.
constexpr double kLoglikTinyHet = 5e-11;
// Slot of one byte value (the walk computes four at once: slots_of in the kernel).
template <bool FOLD>
__host__ __device__ constexpr uint32_t eval_slot(uint32_t byte) {
  const uint32_t folded = FOLD ? ((byte & 0x7Fu) | ((byte >> 4) & 0x08u)) : (byte & 0x7Fu);
  return folded + ((folded >> 2) & 0x1Cu);
}
__host__ __device__ constexpr uint32_t eval_bits(uint32_t amax) { return amax <= 1 ? 1u : amax <= 3 ? 2u : 3u; }

// The entries of the selected loci, tabulated ONCE per call: modes 1 and 2 do not depend on F, so the 50 / ~35 passes of
// a call read the same ones, and the pass itself no longer classifies anything (building them per batch inside the pass
// was a sixth of its vector instructions, repeated by every workgroup of a segment).  (1 << 2*bits) entries per locus,
// entry a1 | a2 << bits with bits = eval_bits(amax): 64 / 256 / 1024 bytes per locus.  classify_cell decides every entry, so the class logic is the
// generic kernel's own.  An index pair past amax is "unclassified", like the slots the pass never refills.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_eval_entries(const double* __restrict__ table, const uint8_t* __restrict__ valid, uint64_t n_sel, uint32_t amax, int phased,
               void* __restrict__ entries_out, unsigned long long* __restrict__ smallest_het = nullptr) {
  const uint32_t stride = sweep_stride(amax);
  const uint32_t bits = eval_bits(amax), mask = (1u << bits) - 1u;
  const uint64_t total = n_sel << (2u * bits);
  double smallest = 1.0;                                       // MODE 5: this thread's smallest 2*f1*f2 with a term
  for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t s = idx >> (2u * bits);
    const uint32_t a1 = static_cast<uint32_t>(idx) & mask, a2 = (static_cast<uint32_t>(idx) >> bits) & mask;
    constexpr bool kPacked = MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6;      // the frequency sweeps' entries: packed class counters
    constexpr bool kHetMarks = MODE == 5 || MODE == 6;         // MODE 6: mode 5's packed words alone (8-byte entries for the <4> pass), no term
    double y = kPacked ? 0.0 : 1.0, d = 0.0;
    const uint8_t flag = valid[s];
    if (a1 > amax || a2 > amax) {
      // MODE 3, 4, 5: a byte past the table is counted as nothing but is odd wherever the locus has defaults
      if constexpr (kPacked) d = __builtin_bit_cast(double, (static_cast<uint64_t>(kOddCell) << 32) | kOddOutside);
    } else if (flag & kLocusValid) {
      double f1 = 0.0, f2 = 0.0;
      const int cls = classify_cell(a1 | (a2 << 4), table + s * stride, amax, phased != 0, f1, f2);
      if constexpr (kPacked) {
        uint32_t lo = 0, hi = 0;
        if (cls == kMajorHom) lo = 1u; else if (cls == kMajorHet) lo = 1u << 12;
        else if (cls == kMinorHom) hi = 1u; else if (cls == kMinorHet) hi = 1u << 12;
        // odd: the cell's share of the class-frequency sums is not the segment default's
        if (flag & kLocusDefault) { if (cls == kClassNone && (a1 | a2) != 0u) lo |= kOddMinus; }
        else if (cls != kClassNone) lo |= kOddPlus;
        if constexpr (kHetMarks) {
          if (cls == kMajorHet || cls == kMinorHet) {                  // (see kLoglikTinyHet)
            const double w = 2.0 * f1 * f2;
            if (w < kLoglikTinyHet) lo |= kOddNoRitland;
            else if (w > 0.5) lo |= kOddBigHet;
            else {
              if constexpr (MODE == 5) y = log(w);
              smallest = w < smallest ? w : smallest;
            }
          }
        } else if (cls == kMajorHom || cls == kMinorHom) {
          if (f1 > 0.001) { y = 1.0 / f1; y -= 1.0; }                  // minimum_frequency (_calc.cpp:380,396)
          else lo |= kOddNoRitland;
        } else if (cls != kClassNone) {
          y = -1.0;
        }
        if (lo >> 24) hi |= kOddCell;
        d = __builtin_bit_cast(double, (static_cast<uint64_t>(hi) << 32) | lo);
      } else if (cls == kMajorHom || cls == kMinorHom) {
        if constexpr (MODE == 2) { y = f1 * f1; d = f1 - y; }
        else { y = f1; d = 1.0 - f1; }
      } else if (cls != kClassNone) {
        if constexpr (MODE == 2) { y = 2.0 * f1 * f2; d = -y; }
      }
    }
    if constexpr (MODE == 4 || MODE == 6) {
      static_cast<uint64_t*>(entries_out)[idx] = __builtin_bit_cast(uint64_t, d);      // the packed words alone
    } else if constexpr (MODE == 1) {
      static_cast<double*>(entries_out)[idx] = y;                                     // d = 1 - y for every entry: y alone
    } else {
      EvalEntry* entries = static_cast<EvalEntry*>(entries_out);
      entries[idx].y = y;
      entries[idx].d = d;
    }
  }
  if constexpr (MODE == 5 || MODE == 6) {
    // one atomic per wave (all threads of the launch on one word took 14 ms at 5 M loci); positive doubles order as their bits
    for (int off = 32; off > 0; off >>= 1) { const double other = __shfl_xor(smallest, off); smallest = other < smallest ? other : smallest; }
    if (smallest_het && (threadIdx.x & (kWave - 1)) == 0 && smallest < 1.0)
      atomicMin(smallest_het, static_cast<unsigned long long>(__double_as_longlong(smallest)));
  }
}

// (byte B of w) << shift, shift = log2 of the entry size: the LDS byte offset of a cell's entry within its locus's
// table, one SDWA shift (which takes its count from a register).
template <int B>
__device__ __forceinline__ uint32_t byte_shifted(uint32_t w, uint32_t four) {
  uint32_t r;
  if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(four), "v"(w));
  else if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(four), "v"(w));
  else if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(four), "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(four), "v"(w));
  return r;
}

// The pass.  Per batch of 8 loci the block copies the batch's entries from `entries` into one of two LDS tables (through
// registers, fetched two batches ahead; the genotype dwords one batch ahead), and each cell is then one LDS read and a
// few fp64 / integer operations.  The loop is unrolled over the two tables, so a cell's LDS address is one SDWA shift
// of its slot byte plus an immediate offset.
// PAIR (MODE 2 only): TWO values of F per genome in one pass -- f_in[g] and f_in[n_genomes + g], results in
// part[(2 * seg + k) * n_genomes + g] -- from the same table read: the Nelder-Mead search almost always (97 % of its steps)
// needs the reflected point and the inside-contraction point of the same simplex, both known before either is evaluated.
template <int MODE, int GPL, bool FOLD, bool PAIR = false>
__global__ void __launch_bounds__(kBlock)
k_inbreed_eval_lut(const uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t g0, uint64_t n_genomes,
                   const uint32_t* __restrict__ locus_index, uint64_t n_sel, uint64_t loci_per_seg,
                   const void* __restrict__ entries_in, const double* __restrict__ table, const uint8_t* __restrict__ valid,
                   uint32_t amax, const double* __restrict__ f_in, double* __restrict__ part,
                   unsigned long long* __restrict__ counts, unsigned long long* __restrict__ seg_counts, uint32_t xcds) {
  constexpr int DW = GPL / 4;
  constexpr bool kCounts = MODE == 3 || MODE == 4;           // the one-pass frequency sweeps: packed class counters, odd cells
  // MODE 4 is MODE 3 without the Ritland term: 8-byte entries (the packed words), half the LDS traffic.  MODE 1 as well:
  // its d is 1 - y for every entry (homozygous: y = f1; anything else: y = 1), so the entry is y and the cell's
  // denominator F + (1-F)*y is ONE fma of the genome's F and 1-F -- one rounding closer to the reference's
  // F + ((1.0 - F) * f) (_calc.cpp:270) than y + F*(1-y) was.
  using Entry = std::conditional_t<MODE == 4, uint64_t, std::conditional_t<MODE == 1, double, EvalEntry>>;
  const Entry* __restrict__ entries = static_cast<const Entry*>(entries_in);
  // Where the entries are 8 bytes (modes 1, 4) a locus's table has a slot for EVERY byte value, entry (a1, a2) at slot
  // a1 | a2 << 4 = the cell's byte itself: a cell's LDS address is its byte times 8, one SDWA shift and nothing else (the
  // four slot-number operations per dword are gone: 5.6 -> 4.6 vector instructions per cell in the HallME step, which
  // they bound), and a byte no entry was tabulated for -- bit 3 or bit 7 set: an allele index past the table -- meets the
  // prefill, so there is nothing to fold either.  A ds_read_b64's bank is (2 * slot) mod 64: bytes 32 apart share one
  // (the same first allele under second alleles two apart -- a rare pair of cells in one 32-lane group).  With 16-byte
  // entries a 256-slot table would be 4 KB a locus (two workgroups a CU) and neighbouring second alleles would share
  // banks: those modes keep the 160-slot table at a1 + 20 * a2.
  constexpr bool kDirect = sizeof(Entry) == 8;
  constexpr uint32_t kSlots = kDirect ? 256u : kEvalSlots;
  __shared__ Entry lut[2][kEvalBatch * kSlots];
  constexpr uint64_t kOutside = (static_cast<uint64_t>(kOddCell) << 32) | kOddOutside;
  auto nothing = []() {                                      // the entry of a locus past the segment, whatever the byte
    if constexpr (MODE == 4) return static_cast<uint64_t>(0);
    else if constexpr (MODE == 1) return 1.0;
    else return EvalEntry{MODE == 3 ? 0.0 : 1.0, 0.0};
  };
  auto packed_of = [](const Entry& e) {
    if constexpr (MODE == 4) return e;
    else if constexpr (MODE == 1) return static_cast<uint64_t>(0);
    else return __builtin_bit_cast(uint64_t, e.d);
  };
  // Workgroup -> (genome chunk, segment); with xcds > 1, XCD-aware.  The genome chunks of one segment read the same entries (and, in
  // MODE 3, the same per-locus rows on odd cells); the hardware deals consecutive workgroup ids round-robin over the
  // XCDs, each with its own L2, so with the plain id every chunk of a segment would fetch them again from memory.
  // Workgroup b = (round * n_chunks + chunk) * xcds + x works on segment round * xcds + x: XCD x takes every xcds-th
  // segment with all its chunks (consecutive in its own queue, so they run together and share its L2), and the XCDs
  // together still sweep neighbouring segments of the matrix at any time.  (1-D grid of xcds * n_chunks * ceil(n_seg /
  // xcds) workgroups; in the last round some have no segment.)
  const uint64_t n_chunks = ((n_genomes + GPL - 1) / GPL + kBlock - 1) / kBlock;
  const uint64_t n_seg = (n_sel + loci_per_seg - 1) / loci_per_seg;
  const uint64_t turn = blockIdx.x / xcds;
  const uint64_t seg = (turn / n_chunks) * xcds + blockIdx.x % xcds;
  if (seg >= n_seg) return;
  const uint64_t lane = (turn % n_chunks) * blockDim.x + threadIdx.x;                    // genomes g0 + GPL*lane ..
  const bool active = lane * GPL < n_genomes;
  const uint64_t s_begin = seg * loci_per_seg;
  const uint64_t s_end = s_begin + loci_per_seg < n_sel ? s_begin + loci_per_seg : n_sel;
  const uint32_t stride = sweep_stride(amax);
  const uint64_t col = (g0 >> 2) + lane * DW;              // g0 is a multiple of GPL
  const uint32_t bits = eval_bits(amax), mask = (1u << bits) - 1u;
  const uint32_t in_batch = static_cast<uint32_t>(kEvalBatch) << (2u * bits);              // 32, 128 or 512 entries

  double F[kCounts ? 1 : GPL], one_minus_F[MODE == 1 ? GPL : 1], run_a[MODE == 4 ? 1 : GPL], run_b[MODE == 1 ? GPL : 1];
  int expo[MODE == 2 ? GPL : 1];   // MODE 2: run_a = product;  MODE 1: run_a / run_b = N / D;  MODE 3: run_a = Ritland sum
  static_assert(!PAIR || MODE == 2, "two values of F per pass: the log-likelihood only");
  double F_pair[PAIR ? GPL : 1], run_pair[PAIR ? GPL : 1];         // PAIR: the second value of F and its running product
  int expo_pair[PAIR ? GPL : 1];
  uint32_t cnt_lo[kCounts ? GPL : 1], cnt_hi[kCounts ? GPL : 1];   // MODE 3, 4: packed class counters (see above)
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const uint64_t g = lane * GPL + j;
    if constexpr (!kCounts) F[j] = g < n_genomes ? f_in[g] : 0.0;
    if constexpr (MODE != 4) run_a[j] = MODE == 2 ? 1.0 : 0.0;
    if constexpr (MODE == 1) {
      run_b[j] = 1.0;
      if (!(F[j] > 0.0)) F[j] = 1.0;                        // see above: v = 1 everywhere, k_hall_update multiplies by the real F
      one_minus_F[j] = 1.0 - F[j];
    } else if constexpr (MODE == 2) {
      expo[j] = 0;
      if constexpr (PAIR) {
        F_pair[j] = g < n_genomes ? f_in[n_genomes + g] : 0.0;
        run_pair[j] = 1.0;
        expo_pair[j] = 0;
      }
    } else {
      cnt_lo[j] = cnt_hi[j] = 0u;
    }
  }

  // Every slot starts "unclassified"; the batches refill the (1 << bits)^2 slots their entries have.
  for (uint32_t e = threadIdx.x; e < 2u * kEvalBatch * kSlots; e += kBlock) {
    if constexpr (MODE == 4) lut[0][e] = kOutside;
    else if constexpr (MODE == 1) lut[0][e] = 1.0;
    else lut[0][e] = EvalEntry{MODE == 3 ? 0.0 : 1.0, MODE == 3 ? __builtin_bit_cast(double, kOutside) : 0.0};
  }
  // Positions within the segment are 32-bit (the scalar unit compares those; 64-bit ones go through the vector unit).
  const uint32_t seg_len = static_cast<uint32_t>(s_end - s_begin);
  const Entry* __restrict__ seg_entries = entries + (s_begin << (2u * bits));
  const uint32_t* __restrict__ seg_index = locus_index ? locus_index + s_begin : nullptr;
  // Entry threadIdx.x of a batch travels through a register pair: fetched at the start of the batch before its own, put
  // into the other table at that batch's end.  At bits == 3 a batch has 512 entries and the second of the thread goes
  // straight from memory to the table (four-to-seven-allele loci only).  The load is unconditional (a thread without an
  // entry reads the segment's first and drops it): no branch between the fetch and the put, see `batch`.
  Entry staged;                                               // as loaded: has_entry() says whether it is the thread's own
  auto has_entry = [&](uint32_t r0, uint32_t idx) { return idx < in_batch && r0 + (idx >> (2u * bits)) < seg_len; };
  auto load_entry = [&](uint32_t r0, uint32_t idx) { return seg_entries[has_entry(r0, idx) ? (r0 << (2u * bits)) + idx : 0u]; };
  auto entry_of = [&](uint32_t r0, uint32_t idx) {
    const Entry e = load_entry(r0, idx);
    return has_entry(r0, idx) ? e : nothing();
  };
  auto fetch = [&](uint32_t r0) { staged = load_entry(r0, threadIdx.x); };
  auto put = [&](int rb, uint32_t idx, Entry e) {
    if (idx < in_batch) {
      const uint32_t a1 = idx & mask, a2 = (idx >> bits) & mask, i = idx >> (2u * bits);
      lut[rb][i * kSlots + (kDirect ? (a1 | (a2 << 4)) : a1 + 20u * a2)] = e;
    }
  };
  auto stash = [&](int rb, uint32_t r0) {                     // r0: the batch the registers hold
    put(rb, threadIdx.x, has_entry(r0, threadIdx.x) ? staged : nothing());
    if (bits == 3u) put(rb, threadIdx.x + kBlock, entry_of(r0, threadIdx.x + kBlock));
  };
  // The cells of the batch at r0.  A locus past the segment reads the segment's last row (or, indexed, one of its first
  // eight) against a table of nothing; a lane past the genomes reads the first lane's columns and stores nothing.
  const uint64_t row_base = seg_index ? 0ull : s_begin;
  const uint64_t col_read = active ? col : (g0 >> 2);
  // The rows of the batch at r0 as scalars, relative to row_base.  With a locus index: its eight entries by ONE scalar
  // load, waited for on the spot -- on the scalar memory counter, so that no wait on the vector memory counter stands
  // between the row loads in flight and their use (the index is padded by 8 entries past its end; a batch past the
  // segment takes the segment's first eight).
  typedef uint32_t v8u __attribute__((ext_vector_type(8)));
  auto take_rows = [&](uint32_t (&rows)[kEvalBatch], uint32_t r0) {
    if (seg_index) {
      const uint32_t* at = seg_index + (r0 < seg_len ? r0 : 0u);
      v8u loaded;
      asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(loaded) : "s"(at) : "memory");
#pragma unroll
      for (int i = 0; i < kEvalBatch; ++i) rows[i] = loaded[i];
    } else {
#pragma unroll
      for (int i = 0; i < kEvalBatch; ++i) rows[i] = r0 + i < seg_len ? r0 + i : seg_len - 1u;
    }
  };
  auto load_row = [&](uint32_t (&wi)[DW], uint32_t row) {
    const uint32_t* src = gt + (row_base + row) * dwords_per_row + col_read;
    if constexpr (DW == 1) {
      wi[0] = __builtin_nontemporal_load(src);
    } else {
      typedef uint32_t v2u __attribute__((ext_vector_type(2)));
      const v2u v = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(src));
      wi[0] = v.x; wi[1] = v.y;
    }
  };
  // A cell's table entry: slot a1 + 20*a2 of its locus (bit 7 of the byte folded onto bit 3: "unclassified").
  auto slots_of = [](uint32_t x) {
    const uint32_t xf = FOLD ? ((x & 0x7F7F7F7Fu) | ((x >> 4) & 0x08080808u)) : (x & 0x7F7F7F7Fu);
    return xf + ((FOLD ? xf >> 2 : x >> 2) & 0x1C1C1C1Cu);               // a1 + 16*a2 + 4*a2 per byte, < 156: no carry
  };
  uint32_t four = sizeof(Entry) == 8 ? 3u : 4u;               // log2 of the entry size
  asm volatile("" : "+v"(four));                              // the SDWA shift takes its count from a register

  // One batch: the table lut[BUF] holds its entries, w its cells.  The next batch's entries are fetched at its start and
  // put into the other table at its end.  The cells are loaded TWO batches ahead and in place: as soon as the last table
  // read of a locus' cells has its addresses, the same registers take the load of that locus' row two batches on (the other
  // register set holds the batch in between), so a wave keeps up to sixteen row loads in flight, spread over the batch.
  // Nothing in the batch waits for ALL vector loads: between the entry's fetch (the oldest load in flight when it is
  // needed) and its put lies straight-line code -- no branch, every lane loads (see col_read) -- so the compiler's counter
  // bookkeeping holds (s_waitcnt vmcnt(8) where it used to drain the queue with vmcnt(0) once per batch).  Measured at C5:
  // no change in the passes' time -- they run against the socket's 1400 W cap at ~1.7 GHz (profiles/r03_power_cap.md), so
  // neither load latency nor issue slots are what is short -- but the structure is what a sweep that is NOT power-bound
  // needs, and it costs nothing.  (A row past the segment is read for nothing: 16 rows per segment of thousands.)
  auto batch = [&](auto buf_c, uint32_t r0, uint32_t (&w)[kEvalBatch][DW]) {
    constexpr int BUF = decltype(buf_c)::value;
    fetch(r0 + kEvalBatch);
    uint32_t rows[kEvalBatch];
    take_rows(rows, r0 + 2 * kEvalBatch);
    {
      auto reload = [&](int i) { load_row(w[i], rows[i]); };   // locus i of the batch two on, into the registers locus i just left
      const char* cur = reinterpret_cast<const char*>(&lut[BUF][0]);
      auto entry_at = [&](int i, uint32_t offset16) {
        return *reinterpret_cast<const Entry*>(cur + i * static_cast<int>(kSlots * sizeof(Entry)) + offset16);
      };
      // The batch's cells in groups of four (one dword), a group's four table reads issued kEvalDepth groups before its
      // arithmetic: the reads' latency passes under that of the groups before it, and the registers stay those of
      // kEvalDepth + 1 groups -- left alone the compiler either waits for every read where it issues it or (machine
      // sinking: nothing in this block reads the sums) carries all 64 reads of the batch past the batch.
      constexpr int kGroups = kEvalBatch * DW, kDepth = PAIR ? KGX_EVAL_DEPTH_PAIR : (kCounts || sizeof(Entry) == 8) ? KGX_EVAL_DEPTH3 : KGX_EVAL_DEPTH;
      auto read_group = [&](int q, Entry (&e)[4]) {
        const uint32_t slots = kDirect ? w[q / DW][q % DW] : slots_of(w[q / DW][q % DW]);
        e[0] = entry_at(q / DW, byte_shifted<0>(slots, four));
        e[1] = entry_at(q / DW, byte_shifted<1>(slots, four));
        e[2] = entry_at(q / DW, byte_shifted<2>(slots, four));
        e[3] = entry_at(q / DW, byte_shifted<3>(slots, four));
      };
      auto walk = [&](auto&& cell) {
        Entry e[kDepth + 1][4];
#pragma unroll
        for (int q = 0; q < kDepth; ++q) {
          read_group(q, e[q]);
          if (q % DW == DW - 1) reload(q / DW);
        }
#pragma unroll
        for (int q = 0; q < kGroups; ++q) {
          if (q + kDepth < kGroups) {
            read_group(q + kDepth, e[(q + kDepth) % (kDepth + 1)]);
            if ((q + kDepth) % DW == DW - 1) reload((q + kDepth) / DW);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int j = 4 * (q % DW) + b;
            cell(j, e[q % (kDepth + 1)][b]);
            // the empty asm reads the sums here, so the group's arithmetic stays here
            if constexpr (MODE == 4) asm volatile("" : "+v"(cnt_lo[j]), "+v"(cnt_hi[j]));
            else if constexpr (MODE == 3) asm volatile("" : "+v"(run_a[j]), "+v"(cnt_lo[j]), "+v"(cnt_hi[j]));
            else if constexpr (MODE == 1) asm volatile("" : "+v"(run_a[j]), "+v"(run_b[j]));
            else if constexpr (PAIR) asm volatile("" : "+v"(run_a[j]), "+v"(run_pair[j]));
            else asm volatile("" : "+v"(run_a[j]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if constexpr (MODE == 2) {
        // The clamp of logLikelihood (:117-121) to [1e-10, 1]: the fma's clamp bit takes the value into [0, 1] for
        // nothing (min(max(x, 0), 1) is folded into it), the fp64 max lifts it to 1e-10.
        walk([&](int j, const EvalEntry& e) {
          const double p = __builtin_fmin(__builtin_fmax(__builtin_fma(F[kCounts ? 0 : j], e.d, e.y), 0.0), 1.0);
          run_a[j] *= __builtin_fmax(p, 1e-10);
          if constexpr (PAIR) {
            const double q = __builtin_fmin(__builtin_fmax(__builtin_fma(F_pair[j], e.d, e.y), 0.0), 1.0);
            run_pair[j] *= __builtin_fmax(q, 1e-10);
          }
        });
      } else if constexpr (kCounts) {
        walk([&](int j, const Entry& e) {
          if constexpr (MODE == 3) run_a[j] += e.y;
          const uint64_t packed = packed_of(e);
          cnt_lo[j] += static_cast<uint32_t>(packed);
          cnt_hi[j] += static_cast<uint32_t>(packed >> 32);
        });
      } else {
        walk([&](int j, const Entry& y) {
          const double v = __builtin_fma(one_minus_F[MODE == 1 ? j : 0], y, F[kCounts ? 0 : j]);
          run_a[j] = __builtin_fma(run_a[j], v, run_b[j]);
          run_b[j] *= v;
        });
      }
      if constexpr (kCounts) {
        uint32_t seen = 0;
#pragma unroll
        for (int j = 0; j < GPL; ++j) seen |= cnt_hi[j];
        if (!active) seen = 0u;                               // a lane past the genomes walks the first lane's cells for nothing
        if (__any((seen >> 28) != 0u)) {                    // some lane of the wave met an odd cell in this batch
#pragma unroll
          for (int j = 0; j < GPL; ++j) cnt_hi[j] &= kOddCell - 1u;
          // The lanes that met one look at their cells of the batch once more, this time for what is odd about them: an odd
          // cell adjusts the lane's own (segment, genome) partial slot in memory (single writer: the order of its adds is
          // the program's) or the genome's Ritland count.  Rolled loops, bytes re-read from the matrix (L2 hits), the
          // locus's class frequencies from the per-locus table in memory.
          if ((seen >> 28) != 0u) {
            const Entry* cur_entries = &lut[BUF][0];
#pragma nounroll
            for (int i = 0; i < kEvalBatch; ++i) {
              if (r0 + i >= seg_len) break;
              const uint64_t s = s_begin + r0 + i;
              const uint8_t flag = valid[s];
              const double* row = table + s * stride;
              const uint64_t l = locus_index ? static_cast<uint64_t>(locus_index[s]) : s;
              const uint8_t* bytes = reinterpret_cast<const uint8_t*>(gt + l * dwords_per_row + col);
#pragma nounroll
              for (int j = 0; j < GPL; ++j) {
                const uint64_t g = lane * GPL + j;
                if (g >= n_genomes) break;
                const uint32_t odd = static_cast<uint32_t>(packed_of(cur_entries[i * kSlots + (kDirect ? static_cast<uint32_t>(bytes[j]) : eval_slot<FOLD>(bytes[j]))])) >> 24;
                if (odd == 0u) continue;
                const double sign = (odd & (kOddMinus >> 24)) ? -1.0 : (odd & (kOddPlus >> 24)) ? 1.0
                                    : ((odd & (kOddOutside >> 24)) && (flag & kLocusDefault)) ? -1.0 : 0.0;
                if (sign != 0.0) {
                  double* p = part + (seg * n_genomes + g) * kParts0;
#pragma nounroll
                  for (uint32_t k = 0; k < 4u; ++k) unsafeAtomicAdd(p + k, sign * row[amax + 1u + k]);
                }
                if (odd & (kOddNoRitland >> 24)) atomicAdd(&counts[g * 6 + 5], ~0ull);       // counted with the classes, not by Ritland
                if (odd & (kOddBigHet >> 24)) atomicAdd(&counts[g * 6 + 5], 1ull << 32);      // (mode 5 entries)
              }
            }
          }
        }
      }
      // MODE 2: 8 factors >= 1e-10 a batch: no underflow before the exponent is peeled.
      // MODE 1: D is a product of denominators F + (1-F)*f1 <= 1 and shrinks, so every other batch N and D are scaled by
      // the power of two that takes D back into [1/2, 1) -- exact, and the 16 factors in between would have to average
      // 1e-19 to take D under.  One division per (segment, genome), at the end.
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        if constexpr (MODE == 2) {
          expo[j] += __builtin_amdgcn_frexp_exp(run_a[j]);
          run_a[j] = __builtin_amdgcn_frexp_mant(run_a[j]);
          if constexpr (PAIR) {
            expo_pair[j] += __builtin_amdgcn_frexp_exp(run_pair[j]);
            run_pair[j] = __builtin_amdgcn_frexp_mant(run_pair[j]);
          }
        } else if constexpr (MODE == 1 && BUF == 1) {
          run_a[j] = __builtin_ldexp(run_a[j], -__builtin_amdgcn_frexp_exp(run_b[j]));
          run_b[j] = __builtin_amdgcn_frexp_mant(run_b[j]);
        }
      }
    }
    stash(BUF ^ 1, r0 + kEvalBatch);                         // the other table was last read in the batch before this one
    __syncthreads();     // the other table is complete, and this one is free for the batch after next
  };

  __syncthreads();                                           // the prefill, before the first entries land on it
  uint32_t w_a[kEvalBatch][DW], w_b[kEvalBatch][DW];
  fetch(0);
  stash(0, 0);
  {
    uint32_t rows[kEvalBatch];
    take_rows(rows, 0);
#pragma unroll
    for (int i = 0; i < kEvalBatch; ++i) load_row(w_a[i], rows[i]);
    take_rows(rows, kEvalBatch);
#pragma unroll
    for (int i = 0; i < kEvalBatch; ++i) load_row(w_b[i], rows[i]);
  }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < seg_len; r0 += 2 * kEvalBatch) {
    batch(std::integral_constant<int, 0>{}, r0, w_a);
    if (r0 + kEvalBatch >= seg_len) break;
    batch(std::integral_constant<int, 1>{}, r0 + kEvalBatch, w_b);
  }

  if (!active) return;
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const uint64_t g = lane * GPL + j;
    if (g >= n_genomes) continue;
    if constexpr (MODE == 2) {
      part[(PAIR ? 2 * seg : seg) * n_genomes + g] = log(run_a[j]) + static_cast<double>(expo[j]) * 0.6931471805599453;
      if constexpr (PAIR) part[(2 * seg + 1) * n_genomes + g] = log(run_pair[j]) + static_cast<double>(expo_pair[j]) * 0.6931471805599453;
    } else if constexpr (MODE == 1) {
      part[seg * n_genomes + g] = run_a[j] / run_b[j];
    } else {
      // slot 4: the Ritland sum (MODE 4: none -- the slot held the segment's default, which is RitlandLocus' alone)
      part[(seg * n_genomes + g) * kParts0 + 4] = MODE == 3 ? run_a[MODE == 4 ? 0 : j] : 0.0;
      // The segment's class counters leave as ONE packed word per (segment, genome) -- a lane's eight are 64 contiguous
      // bytes -- and k_reduce_class_counts adds them up (six 8-byte atomics per (segment, genome) were 5 % of the sweep's
      // HBM traffic, each a line of its own).  Ritland counts every classified cell but the few taken off above.
      seg_counts[seg * n_genomes + g] = (static_cast<unsigned long long>(cnt_hi[j] & 0xFFFFFFu) << 32) | (cnt_lo[j] & 0xFFFFFFu);
    }
  }
}

// counts[g][0..5] += the class counters the frequency table pass (k_inbreed_eval_lut<3|4>) left per (segment, genome):
// majorHom | majorHet << 12 in the low word, minorHom | minorHet << 12 in the high word; total and Ritland count = their sum.
// blockIdx.y strides the segments.
static __global__ void __launch_bounds__(kBlock)
k_reduce_class_counts(const unsigned long long* __restrict__ seg_counts, uint64_t n_seg, uint64_t n_genomes,
                      unsigned long long* __restrict__ counts) {
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_genomes) return;
  unsigned long long major_hom = 0, major_het = 0, minor_hom = 0, minor_het = 0;
  for (uint64_t seg = blockIdx.y; seg < n_seg; seg += gridDim.y) {
    const unsigned long long packed = seg_counts[seg * n_genomes + g];
    const uint32_t lo = static_cast<uint32_t>(packed), hi = static_cast<uint32_t>(packed >> 32);
    major_hom += lo & 0xFFFu; major_het += (lo >> 12) & 0xFFFu;
    minor_hom += hi & 0xFFFu; minor_het += (hi >> 12) & 0xFFFu;
  }
  const unsigned long long total = major_hom + major_het + minor_hom + minor_het;
  unsigned long long* c = counts + g * 6;
  if (major_hom) atomicAdd(c + 0, major_hom);
  if (major_het) atomicAdd(c + 1, major_het);
  if (minor_hom) atomicAdd(c + 2, minor_hom);
  if (minor_het) atomicAdd(c + 3, minor_het);
  if (total) { atomicAdd(c + 4, total); atomicAdd(c + 5, total); }
}

// Per-segment totals of what a genome that is reference-homozygous at EVERY locus of the segment would collect:
// def[seg] = { sum majorHom cf, sum majorHet cf, sum minorHom cf, sum minorHet cf, Ritland sum, #default loci,
// #Ritland-default loci, 0 }.  One wave per segment; lanes stride the loci, then a wave reduction.
constexpr int kSegDefaults = 8;
static __global__ void __launch_bounds__(kWave)
k_segment_defaults(const double* __restrict__ table, const uint8_t* __restrict__ flags, uint64_t n_sel, uint64_t loci_per_seg,
                   uint32_t amax, int class_sums_elsewhere, double* __restrict__ seg_def) {
  const uint64_t seg = blockIdx.x;
  const uint64_t s_begin = seg * loci_per_seg;
  const uint64_t s_end = s_begin + loci_per_seg < n_sel ? s_begin + loci_per_seg : n_sel;
  const uint32_t stride = sweep_stride(amax);
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (uint64_t s = s_begin + threadIdx.x; s < s_end; s += kWave) {
    const uint8_t f = flags[s];
    if (!(f & kLocusDefault)) continue;
    const double* row = table + s * stride;
    acc[0] += row[amax + 1]; acc[1] += row[amax + 2]; acc[2] += row[amax + 3]; acc[3] += row[amax + 4];
    acc[5] += 1.0;
    if (f & kLocusRitlandDefault) { acc[4] += row[2 * amax + kTableExtra]; acc[4] -= 1.0; acc[6] += 1.0; }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    // class_sums_elsewhere: the four class-frequency sums of the defaults come from the sequential-sum emulation
    // (k_seq_* below), once per call; the per-segment partials then hold the genomes' corrections only
    if (threadIdx.x == 0) seg_def[seg * kSegDefaults + k] = (class_sums_elsewhere && k < 4) ? 0.0 : v;
  }
  if (threadIdx.x == 0) seg_def[seg * kSegDefaults + 7] = 0.0;
}

// K5 frequency pass, the idea shared by the SWAR kernels below (same results as MODE 0 of k_inbreed_sweep up to fp64
// summation order).  Every class-frequency sum of generateFrequencies is class independent (_freq.cpp:549-556) and a
// reference-homozygous genome behaves the same at a locus for every genome, so a genome's results are the segment
// defaults (k_segment_defaults) corrected at the loci where it carries a variant.  The per-cell decision
// (_freq.cpp:452-543) is branch-free integer logic on per-locus bit masks held in scalar registers (which alts are in
// the AlleleFreqVector, which have AF > 0.001), so a wave never diverges on genotype; only the rare cells whose
// classification disagrees with the locus default touch fp64 class sums.  amax <= 4 (wider loci take the generic kernel).

// Sets *found when some byte of the matrix holds an allele index 8..14 (bit 3 of a nibble set, the nibble not 15).
static __global__ void __launch_bounds__(kBlock)
k_scan_wide_nibbles(const kgx_v4u* __restrict__ gt, uint64_t n_chunks, unsigned int* __restrict__ found) {
  uint32_t any = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_chunks; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const kgx_v4u v = __builtin_nontemporal_load(gt + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t x = v[k];
      const uint32_t fifteen = x & (x >> 1) & (x >> 2) & (x >> 3) & 0x11111111u;      // 1 at the nibbles that are 15
      any |= ((x >> 3) & 0x11111111u) & ~fifteen;
    }
  }
  if (__any(any != 0) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(found, 1u);
}

// meta[s] (for amax <= 7): flag bits (kLocus*) | in_list bits 0..7 << 8 | rit_ok bits << 16 — one scalar dword per locus.
static __global__ void __launch_bounds__(kBlock)
k_locus_bits(const double* __restrict__ table, const uint8_t* __restrict__ flags, uint64_t n_sel, uint32_t amax,
             uint32_t* __restrict__ meta) {
  const uint32_t stride = sweep_stride(amax);
  for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < n_sel;
       s += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const double* row = table + s * stride;
    uint32_t in_list = 0, rit_ok = 0;
    for (uint32_t a = 0; a < amax; ++a) {
      const double f = row[a];
      if (f == f) { in_list |= 1u << (a + 1); if (f > 0.001) rit_ok |= 1u << (a + 1); }
    }
    meta[s] = static_cast<uint32_t>(flags[s]) | ((in_list & 0xFFu) << 8) | ((rit_ok & 0xFFFFu) << 16);
  }
}

// K5 SWAR path (Simple / HallME / Loglikelihood frequency pass; amax <= 6): the four genotype bytes of a dword
// are classified together.  v_perm_b32 is a 4-lane byte LUT: the per-locus "alt is in the AlleleFreqVector" table
// (8 bytes, wave-uniform, in a scalar register pair) is indexed by the low / high nibbles of all four bytes in one
// instruction; the rest is byte-parallel boolean algebra, and the class counters are byte lanes of a dword flushed
// to wide counters every 255 loci.  ~40 VALU operations per dword instead of ~25 per byte.
__device__ __forceinline__ uint32_t bytes_nonzero(uint32_t x) {   // x: bytes <= 0x0F -> 0x01 where byte != 0
  return ((x + 0x0F0F0F0Fu) >> 4) & 0x01010101u;
}
// The class decision of generateFrequencies (_freq.cpp:452-543) for the four genotype bytes of a dword at one locus.
// All outputs are 0x01-per-byte masks.  LUT lookups are v_perm_b32 with the 3-bit nibbles as selectors:
//   in-list LUT (lut_hi:lut_lo)  entry a = 1 if alt a is in the AlleleFreqVector (entries 0 and 7 are 0);
//   "is zero" LUT {entry 0 = 1}  applied to the high nibble       -> no second variant;
//   "same"    LUT {entry 0 = ph} applied to lo ^ hi               -> same variant on both phases (phased input only).
// A nibble >= 8 is past this path's 8-entry table (amax <= 6), "unknown alt" (15) or ">= 3 variants" (0xFF): the
// cell is skipped whatever the other nibble holds.  Bit 3 of either nibble clears ok1, and every class needs ok1, so
// the aliasing of 8..15 onto 0..7 in the selectors never shows.
struct SwarClasses {
  uint32_t major_het, minor_hom, minor_het, nonzero;
};
// GUARD = false: the matrix is known to hold no allele index 8..14 (kgx_gt8::wide_nibbles == 1).  15 aliases onto
// selector 7, whose LUT entry is 0, so "unknown alt" and 0xFF need no guard; only 8..14 would alias onto real entries.
template <bool GUARD>
__device__ __forceinline__ SwarClasses swar_classify(uint32_t x, uint32_t lut_lo, uint32_t lut_hi, uint32_t same_lut) {
  const uint32_t xs = x >> 4;
  const uint32_t lo = x & 0x07070707u, hi = xs & 0x07070707u;
  uint32_t ok1, nonzero;
  if constexpr (GUARD) {
    const uint32_t t = x | xs;                                                   // low nibble of each byte: lo | hi
    ok1 = __builtin_amdgcn_perm(lut_hi, lut_lo, lo) & ~(t >> 3);
    nonzero = bytes_nonzero(t & 0x0F0F0F0Fu);
  } else {
    // nibbles are 0..7 or 15 here, so lo | hi (3-bit) is 0 exactly where both nibbles are 0: one more byte LUT
    ok1 = __builtin_amdgcn_perm(lut_hi, lut_lo, lo);
    nonzero = __builtin_amdgcn_perm(0x01010101u, 0x01010100u, lo | hi);
  }
  const uint32_t ok2 = __builtin_amdgcn_perm(lut_hi, lut_lo, hi);
  const uint32_t hom = __builtin_amdgcn_perm(0u, same_lut, lo ^ hi);             // homozygous(): same HGVS, different phase
  const uint32_t single = __builtin_amdgcn_perm(0u, 1u, hi);
  SwarClasses c;
  c.major_het = ok1 & single;
  c.minor_hom = ok1 & hom;
  c.minor_het = (ok1 & ok2) ^ c.minor_hom;       // hom => ok2 == ok1; ok2 => a second variant (LUT entry 0 is 0)
  c.nonzero = nonzero;
  return c;
}

// The cells the byte algebra cannot settle, decided one by one by classify_cell: the (0, a) pair (two copies of one
// variant on one phase) at a default locus, and every carrier of a non-default locus (no defaults to correct there).
// Kept out of line: it runs for repeated records and near-monomorphic loci only, and inlining it sixteen times cost
// the sweep a third of its registers.  Returns 0x01-per-byte masks: count one more minor heterozygote / take the
// locus's class frequencies off / add them.
struct SwarSpecial { uint32_t minor_het, minus, plus; };
__device__ __noinline__ SwarSpecial swar_special_cells(uint32_t x, uint32_t special, uint32_t classed, int is_default,
                                                       const double* __restrict__ row, uint32_t amax, int phased) {
  SwarSpecial r{0u, 0u, 0u};
  for (int j = 0; j < 4; ++j) {
    if (!((special >> (8 * j)) & 1u)) continue;
    double f1 = 0.0, f2 = 0.0;
    const int cls = classify_cell((x >> (8 * j)) & 0xFFu, row, amax, phased != 0, f1, f2);
    const uint32_t bit = 1u << (8 * j);
    if (cls != kClassNone && !(classed & bit)) r.minor_het |= bit;
    if (is_default) { if (cls == kClassNone) r.minus |= bit; }
    else if (cls != kClassNone) r.plus |= bit;
  }
  return r;
}

// locus_index and meta are padded by 8 entries past n_sel (whole batches are fetched with one scalar load each);
// loci_per_seg is a multiple of 8.
template <bool INDEXED, bool GUARD>
__global__ void __launch_bounds__(kBlock)
k_inbreed_sweep_swar(const uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t g0, uint64_t n_genomes,
                     const uint32_t* __restrict__ locus_index, uint64_t n_sel, uint64_t loci_per_seg,
                     const double* __restrict__ table, const uint32_t* __restrict__ meta,
                     uint32_t amax, int phased, const double* __restrict__ seg_def, unsigned long long* __restrict__ counts,
                     double* __restrict__ part) {
  const uint64_t quad = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (quad * 4 >= n_genomes) return;
  const uint64_t seg = blockIdx.y;
  const uint64_t s_begin = seg * loci_per_seg;
  const uint64_t s_end = s_begin + loci_per_seg < n_sel ? s_begin + loci_per_seg : n_sel;
  const uint32_t stride = sweep_stride(amax);
  const uint64_t col = (g0 >> 2) + quad;
  const uint32_t same_lut = phased ? 1u : 0u;

  uint32_t b_major_het = 0, b_minor_hom = 0, b_minor_het = 0, b_miss = 0;     // 4 x 8-bit lanes
  uint32_t n_major_het[4] = {0, 0, 0, 0}, n_minor_hom[4] = {0, 0, 0, 0}, n_minor_het[4] = {0, 0, 0, 0}, n_miss[4] = {0, 0, 0, 0};
  double cf_corr[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cf_corr[j][0] = cf_corr[j][1] = cf_corr[j][2] = cf_corr[j][3] = 0.0;
  uint32_t since_flush = 0;

  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      n_major_het[j] += (b_major_het >> (8 * j)) & 0xFFu;
      n_minor_hom[j] += (b_minor_hom >> (8 * j)) & 0xFFu;
      n_minor_het[j] += (b_minor_het >> (8 * j)) & 0xFFu;
      n_miss[j] += (b_miss >> (8 * j)) & 0xFFu;
    }
    b_major_het = b_minor_hom = b_minor_het = b_miss = 0;
    since_flush = 0;
  };

  constexpr int kBatch = 8;
  for (uint64_t s0 = s_begin; s0 < s_end; s0 += kBatch) {
    uint32_t w[kBatch], m[kBatch], idx[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {          // wave-uniform: two scalar loads of 8 dwords
      m[i] = meta[s0 + i];
      idx[i] = INDEXED ? locus_index[s0 + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const uint64_t s = s0 + i;
      const uint64_t l = INDEXED ? static_cast<uint64_t>(idx[i]) : s;
      w[i] = (s < s_end) ? __builtin_nontemporal_load(gt + l * dwords_per_row + col) : 0u;
    }
    if (since_flush + kBatch > 255) flush();            // a locus adds at most 1 to a byte lane (the rare path included)
    since_flush += kBatch;
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const uint64_t s = s0 + i;
      if (s >= s_end) break;
      const uint32_t f = m[i] & 0xFFu;                   // wave-uniform
      if (!(f & kLocusValid)) continue;
      // byte LUT: entry a = 1 if alt a is in the list (entries 0 and 7 are 0: "no allele" and "unknown alt")
      const uint32_t in_list = (m[i] >> 8) & 0xFFu;
      const uint32_t lut_lo = ((in_list >> 0) & 1u) | (((in_list >> 1) & 1u) << 8) | (((in_list >> 2) & 1u) << 16) | (((in_list >> 3) & 1u) << 24);
      const uint32_t lut_hi = ((in_list >> 4) & 1u) | (((in_list >> 5) & 1u) << 8) | (((in_list >> 6) & 1u) << 16);
      const bool is_default = (f & kLocusDefault) != 0;       // wave-uniform
      const SwarClasses c = swar_classify<GUARD>(w[i], lut_lo, lut_hi, same_lut);
      b_major_het += c.major_het;
      b_minor_hom += c.minor_hom;
      b_minor_het += c.minor_het;
      if (is_default) b_miss += c.nonzero;
      const uint32_t classed = c.major_het | c.minor_hom | c.minor_het;    // a subset of nonzero
      // Cells the defaults do not cover: at a default locus the carriers the byte algebra left unclassified (their class
      // frequencies come off), at any other locus every carrier.
      const uint32_t rare = is_default ? (c.nonzero ^ classed) : c.nonzero;
      if (rare) {
        const double* row = table + s * stride;
        // the (0, a) pairs among them, and all of a non-default locus, are decided by classify_cell
        const uint32_t special = is_default ? (rare & (bytes_nonzero(w[i] & 0x0F0F0F0Fu) ^ 0x01010101u)) : rare;
        uint32_t minus = is_default ? (rare ^ special) : 0u, plus = 0u;
        if (special) {
          const SwarSpecial sp = swar_special_cells(w[i], special, classed, is_default ? 1 : 0, row, amax, phased);
          b_minor_het += sp.minor_het;
          minus |= sp.minus;
          plus |= sp.plus;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double sign = ((plus >> (8 * j)) & 1u) ? 1.0 : (((minus >> (8 * j)) & 1u) ? -1.0 : 0.0);
          if (sign == 0.0) continue;
          cf_corr[j][0] += sign * row[amax + 1]; cf_corr[j][1] += sign * row[amax + 2];
          cf_corr[j][2] += sign * row[amax + 3]; cf_corr[j][3] += sign * row[amax + 4];
        }
      }
    }
  }
  flush();

  const double* def_row = seg_def + seg * kSegDefaults;
  const unsigned long long n_def = static_cast<unsigned long long>(def_row[5]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t g = quad * 4 + j;
    if (g >= n_genomes) continue;
    const unsigned long long major_hom = n_def - n_miss[j];
    const unsigned long long total = major_hom + n_major_het[j] + n_minor_hom[j] + n_minor_het[j];
    unsigned long long* c = counts + g * 6;
    if (major_hom) atomicAdd(c + 0, major_hom);
    if (n_major_het[j]) atomicAdd(c + 1, static_cast<unsigned long long>(n_major_het[j]));
    if (n_minor_hom[j]) atomicAdd(c + 2, static_cast<unsigned long long>(n_minor_hom[j]));
    if (n_minor_het[j]) atomicAdd(c + 3, static_cast<unsigned long long>(n_minor_het[j]));
    if (total) atomicAdd(c + 4, total);
    double* p = part + (seg * n_genomes + g) * kParts0;
    p[0] = def_row[0] + cf_corr[j][0];
    p[1] = def_row[1] + cf_corr[j][1];
    p[2] = def_row[2] + cf_corr[j][2];
    p[3] = def_row[3] + cf_corr[j][3];
    p[4] = 0.0;                       // RitlandLocus' terms come from k_inbreed_eval_lut<3>
  }
}

// K5 SWAR path, 16 genomes per lane: one 16-byte load per locus row (a wave moves a full 1 KiB line-aligned
// segment), four dwords classified as in k_inbreed_sweep_swar.  Class counters: byte lanes flushed into 16-bit
// halves of registers (a segment holds < 65536 loci).  The rare cells whose classification disagrees with the locus
// default adjust the lane's own (segment, genome) partial slot in memory directly — the slot has exactly one writer,
// so the result is deterministic — which keeps fp64 out of the register file.  part[] must be pre-filled with the
// segment defaults (k_fill_defaults).
template <bool INDEXED, bool GUARD>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3)))      // three waves per SIMD: <= 170 VGPRs
k_inbreed_sweep_swar16(const kgx_v4u* __restrict__ gt, uint64_t chunks_per_row, uint64_t g0, uint64_t n_genomes,
                       const uint32_t* __restrict__ locus_index, uint64_t n_sel, uint64_t loci_per_seg,
                       const double* __restrict__ table, const uint32_t* __restrict__ meta, uint32_t amax, int phased,
                       const double* __restrict__ seg_def, unsigned long long* __restrict__ counts, double* __restrict__ part) {
  const uint64_t lane16 = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;     // 16 genomes g0 + 16*lane16 ..
  if (lane16 * 16 >= n_genomes) return;
  const uint64_t seg = blockIdx.y;
  const uint64_t s_begin = seg * loci_per_seg;
  const uint64_t s_end = s_begin + loci_per_seg < n_sel ? s_begin + loci_per_seg : n_sel;
  const uint32_t stride = sweep_stride(amax);
  const uint64_t col = (g0 >> 4) + lane16;                 // g0 is a multiple of 16
  const uint32_t same_lut = phased ? 1u : 0u;

  uint32_t b_mhet[4] = {0, 0, 0, 0}, b_mhom[4] = {0, 0, 0, 0}, b_nhet[4] = {0, 0, 0, 0};   // byte lanes
  // wide counters: [dword d][pair p] holds genomes 4d+p (low 16 bits) and 4d+p+2 (high 16 bits)
  uint32_t n_mhet[4][2], n_mhom[4][2], n_nhet[4][2];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int p = 0; p < 2; ++p) n_mhet[d][p] = n_mhom[d][p] = n_nhet[d][p] = 0;
  // Carriers at default loci (what the major-homozygote count is the complement of) are not counted cell by cell: they
  // are the classified cells of ALL loci, plus the unclassified carriers of default loci, minus the classified cells
  // of the other loci -- and those two corrections are exactly the cells the rare path below visits with sign -1 / +1.
  // It keeps their balance in slot 4 of the lane's own partial (zero on entry, read back and zeroed at the end).
  uint32_t since_flush = 0;

  auto flush = [&]() {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      // bytes 0,2 -> pair 0 (low/high halves), bytes 1,3 -> pair 1
      n_mhet[d][0] += b_mhet[d] & 0x00FF00FFu;  n_mhet[d][1] += (b_mhet[d] >> 8) & 0x00FF00FFu;
      n_mhom[d][0] += b_mhom[d] & 0x00FF00FFu;  n_mhom[d][1] += (b_mhom[d] >> 8) & 0x00FF00FFu;
      n_nhet[d][0] += b_nhet[d] & 0x00FF00FFu;  n_nhet[d][1] += (b_nhet[d] >> 8) & 0x00FF00FFu;
      b_mhet[d] = b_mhom[d] = b_nhet[d] = 0;
    }
    since_flush = 0;
  };

  constexpr int kBatch = 4;
  for (uint64_t s0 = s_begin; s0 < s_end; s0 += kBatch) {
    kgx_v4u w[kBatch];
    uint32_t m[kBatch], idx[kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      m[i] = meta[s0 + i];
      idx[i] = INDEXED ? locus_index[s0 + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const uint64_t s = s0 + i;
      const uint64_t l = INDEXED ? static_cast<uint64_t>(idx[i]) : s;
      kgx_v4u v = {0u, 0u, 0u, 0u};
      if (s < s_end) v = __builtin_nontemporal_load(gt + l * chunks_per_row + col);
      w[i] = v;
    }
    if (since_flush + kBatch > 255) flush();
    since_flush += kBatch;
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const uint64_t s = s0 + i;
      if (s >= s_end) break;
      const uint32_t f = m[i] & 0xFFu;
      if (!(f & kLocusValid)) continue;
      const uint32_t in_list = (m[i] >> 8) & 0xFFu;
      const uint32_t lut_lo = ((in_list >> 0) & 1u) | (((in_list >> 1) & 1u) << 8) | (((in_list >> 2) & 1u) << 16) | (((in_list >> 3) & 1u) << 24);
      const uint32_t lut_hi = ((in_list >> 4) & 1u) | (((in_list >> 5) & 1u) << 8) | (((in_list >> 6) & 1u) << 16);
      const bool is_default = (f & kLocusDefault) != 0;       // wave-uniform
      uint32_t rare[4], classed[4];
      uint32_t rare_any = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const SwarClasses c = swar_classify<GUARD>(w[i][d], lut_lo, lut_hi, same_lut);
        b_mhet[d] += c.major_het;
        b_mhom[d] += c.minor_hom;
        b_nhet[d] += c.minor_het;
        classed[d] = c.major_het | c.minor_hom | c.minor_het;                     // a subset of nonzero
        // cells the defaults do not cover: at a default locus the carriers the byte algebra left unclassified, at any
        // other locus every carrier
        rare[d] = is_default ? (c.nonzero ^ classed[d]) : c.nonzero;
        rare_any |= rare[d];
      }
      if (rare_any) {
        const double* row = table + s * stride;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          if (!rare[d]) continue;
          // the (0, a) pairs among them, and all of a non-default locus, are decided by classify_cell
          const uint32_t special = is_default ? (rare[d] & (bytes_nonzero(w[i][d] & 0x0F0F0F0Fu) ^ 0x01010101u)) : rare[d];
          uint32_t minus = is_default ? (rare[d] ^ special) : 0u, plus = 0u;
          if (special) {
            const SwarSpecial sp = swar_special_cells(w[i][d], special, classed[d], is_default ? 1 : 0, row, amax, phased);
            b_nhet[d] += sp.minor_het;
            minus |= sp.minus;
            plus |= sp.plus;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double sign = ((plus >> (8 * j)) & 1u) ? 1.0 : (((minus >> (8 * j)) & 1u) ? -1.0 : 0.0);
            if (sign == 0.0) continue;
            const uint64_t g = lane16 * 16 + d * 4 + j;
            if (g >= n_genomes) continue;
            double* p = part + (seg * n_genomes + g) * kParts0;     // single writer: this lane
            p[0] += sign * row[amax + 1]; p[1] += sign * row[amax + 2]; p[2] += sign * row[amax + 3]; p[3] += sign * row[amax + 4];
            p[4] -= sign;                                           // carriers at default loci beyond / short of the classified cells
          }
        }
      }
    }
  }
  flush();

  const unsigned long long n_def = static_cast<unsigned long long>(seg_def[seg * kSegDefaults + 5]);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t g = lane16 * 16 + d * 4 + j;
      if (g >= n_genomes) continue;
      const int pair = j & 1, shift = (j & 2) ? 16 : 0;
      const unsigned long long mhet = (n_mhet[d][pair] >> shift) & 0xFFFFu, mhom = (n_mhom[d][pair] >> shift) & 0xFFFFu;
      const unsigned long long nhet = (n_nhet[d][pair] >> shift) & 0xFFFFu;
      double* balance = part + (seg * n_genomes + g) * kParts0 + 4;
      const long long miss = static_cast<long long>(mhet + mhom + nhet) + static_cast<long long>(*balance);   // small integers: exact
      *balance = 0.0;                                              // the slot is RitlandLocus' afterwards (k_inbreed_eval_lut<3>)
      const unsigned long long major_hom = n_def - static_cast<unsigned long long>(miss);
      const unsigned long long total = major_hom + mhet + mhom + nhet;
      unsigned long long* c = counts + g * 6;
      if (major_hom) atomicAdd(c + 0, major_hom);
      if (mhet) atomicAdd(c + 1, mhet);
      if (mhom) atomicAdd(c + 2, mhom);
      if (nhet) atomicAdd(c + 3, nhet);
      if (total) atomicAdd(c + 4, total);
    }
  }
}

// part[(seg, g)][0..4] = segment defaults (class-frequency sums of a genome that is reference-homozygous throughout).
static __global__ void __launch_bounds__(kBlock)
k_fill_defaults(const double* __restrict__ seg_def, uint64_t n_seg, uint64_t n_genomes, double* __restrict__ part) {
  const uint64_t total = n_seg * n_genomes;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const double* d = seg_def + (i / n_genomes) * kSegDefaults;
    double* p = part + i * kParts0;
    p[0] = d[0]; p[1] = d[1]; p[2] = d[2]; p[3] = d[3]; p[4] = 0.0;
  }
}

// Sum the per-segment partials (deterministic: a workgroup takes 16 neighbouring items, its 16 slices of 16 lanes each add
// a contiguous sixteenth of the segments in ascending order, and the sixteen slice sums are added in order -- the same
// additions whatever the launch; one thread walking all ~1000 segments of an item was 0.3 ms of latency per pass).
// base (may be null): kParts0 values added to every genome's kParts0 sums -- the class-frequency sums of the defaults
// when they are kept apart from the genomes' corrections (k_seq_chain).
constexpr uint32_t kReduceItems = 16, kReduceSlices = kBlock / kReduceItems;
static __global__ void __launch_bounds__(kBlock)
k_reduce_parts(const double* __restrict__ part, uint64_t n_seg, uint64_t n_items, const double* __restrict__ base, double* __restrict__ out) {
  __shared__ double slice_sum[kReduceSlices][kReduceItems + 1];
  const uint32_t item = threadIdx.x % kReduceItems, slice = threadIdx.x / kReduceItems;
  const uint64_t per_slice = (n_seg + kReduceSlices - 1) / kReduceSlices;
  const uint64_t k_begin = slice * per_slice < n_seg ? slice * per_slice : n_seg;
  const uint64_t k_end = k_begin + per_slice < n_seg ? k_begin + per_slice : n_seg;
  for (uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * kReduceItems; i0 < n_items; i0 += static_cast<uint64_t>(gridDim.x) * kReduceItems) {
    const uint64_t i = i0 + item;
    double s = 0.0;
    if (i < n_items)
      for (uint64_t k = k_begin; k < k_end; ++k) s += part[k * n_items + i];
    slice_sum[slice][item] = s;
    __syncthreads();
    if (slice == 0 && i < n_items) {
      double total = 0.0;
#pragma unroll
      for (uint32_t k = 0; k < kReduceSlices; ++k) total += slice_sum[k][item];
      out[i] = base ? base[i % kParts0] + total : total;
    }
    __syncthreads();
  }
}

// ---- the reference's summation order for the class-frequency sums, at any size ---------------------------------------
// generateFrequencies adds the four class frequencies of every classified locus to four running doubles in ascending
// locus order (_freq.cpp:548-556): at 5M loci that sequential sum sits ~1.5e-12 (relative) off the exactly rounded sum
// -- ten times the tolerance of SURVEY.md 8a -- while a tree reduction lands within 1e-15 of the exact one.  So the tree
// is the wrong target; the sequential result is reproduced instead.  Within one binade of the running sum S (ulp u),
// fl(S + x) = S + (x rounded to a multiple of u): the sequential sum over a run of loci that stays inside a binade is
// u * sum(rint(x / u)), an exact integer sum that any order of additions gives (a tie, x an odd multiple of u/2, rounds
// to the even neighbour of x/u here and of (S + x)/u there: one ulp, a few dozen times in 5M loci).  Loci go in blocks
// of kSeqBlock: k_seq_block_sums adds each block plainly, k_seq_block_predict turns the blocks' prefix into the binade
// every block will be summed in (or "walk it": the running sum is near a power of two there, or the block is among the
// first), k_seq_block_quantize takes the integer sums, and k_seq_chain walks the blocks in order -- one wave per class
// sum -- adding u * N where the running sum is in the predicted binade before and after, and the block's loci one by one,
// as the reference does, everywhere else.  The sums are those of a genome that is reference-homozygous at every default
// locus; what a genome's own cells change (rare: _freq.cpp:452-543) stays in the per-segment partials, added at the end.
constexpr int kSeqBlock = 1024;
constexpr int kSeqWalk = -(1 << 30);          // e_pred: sum this block locus by locus

static __global__ void __launch_bounds__(kBlock)
k_seq_block_sums(const double* __restrict__ table, const uint8_t* __restrict__ flags, uint64_t n_sel, uint32_t amax,
                 double* __restrict__ block_sum /* [n_blocks][4] */) {
  __shared__ double sh[4][kBlock / kWave];
  const uint32_t stride = sweep_stride(amax);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < kSeqBlock / kBlock; ++j) {
    const uint64_t s = static_cast<uint64_t>(blockIdx.x) * kSeqBlock + j * kBlock + threadIdx.x;
    if (s < n_sel && (flags[s] & kLocusDefault)) {
      const double* row = table + s * stride + amax + 1;
      acc[0] += row[0]; acc[1] += row[1]; acc[2] += row[2]; acc[3] += row[3];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) sh[k][threadIdx.x / kWave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int w = 0; w < kBlock / kWave; ++w) v += sh[threadIdx.x][w];
    block_sum[static_cast<uint64_t>(blockIdx.x) * 4 + threadIdx.x] = v;
  }
}

// One workgroup: the exclusive prefix of the block sums, and from it the binade (exponent of the running sum) block b
// of class sum k is added in -- if the prefix and the prefix plus the block lie in one binade with a margin (1e-9
// relative: the sequential sum is within ~1e-11 of these plainly added prefixes), else kSeqWalk.
static __global__ void __launch_bounds__(kBlock)
k_seq_block_predict(const double* __restrict__ block_sum, uint64_t n_blocks, int* __restrict__ e_pred /* [n_blocks][4] */) {
  __shared__ double scan[4][kBlock];
  double carry[4] = {0.0, 0.0, 0.0, 0.0};
  for (uint64_t b0 = 0; b0 < n_blocks; b0 += kBlock) {
    const uint64_t b = b0 + threadIdx.x;
    double mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mine[k] = b < n_blocks ? block_sum[b * 4 + k] : 0.0;
      scan[k][threadIdx.x] = mine[k];
    }
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {            // inclusive Hillis-Steele scan of the chunk
      double add[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) add[k] = threadIdx.x >= static_cast<unsigned>(off) ? scan[k][threadIdx.x - off] : 0.0;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) scan[k][threadIdx.x] += add[k];
      __syncthreads();
    }
    if (b < n_blocks) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double after = carry[k] + scan[k][threadIdx.x], before = after - mine[k];
        int e = kSeqWalk;
        if (before > 0.0 && mine[k] >= 0.0) {
          const int lo = ilogb(before * (1.0 - 1.0e-9)), hi = ilogb(after * (1.0 + 1.0e-9));
          if (lo == hi && lo > -900) e = lo;
        }
        e_pred[b * 4 + k] = e;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) carry[k] += scan[k][kBlock - 1];
    __syncthreads();
  }
}

// N[b][k] = sum over the block's default loci of rint(x * 2^(52 - e)): the block's sum in units of the binade's ulp.
static __global__ void __launch_bounds__(kBlock)
k_seq_block_quantize(const double* __restrict__ table, const uint8_t* __restrict__ flags, uint64_t n_sel, uint32_t amax,
                     const int* __restrict__ e_pred, long long* __restrict__ block_n /* [n_blocks][4] */) {
  __shared__ long long sh[4][kBlock / kWave];
  const uint32_t stride = sweep_stride(amax);
  int e[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = e_pred[static_cast<uint64_t>(blockIdx.x) * 4 + k];
  long long acc[4] = {0, 0, 0, 0};
  for (int j = 0; j < kSeqBlock / kBlock; ++j) {
    const uint64_t s = static_cast<uint64_t>(blockIdx.x) * kSeqBlock + j * kBlock + threadIdx.x;
    if (s < n_sel && (flags[s] & kLocusDefault)) {
      const double* row = table + s * stride + amax + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (e[k] != kSeqWalk) acc[k] += static_cast<long long>(__builtin_rint(ldexp(row[k], 52 - e[k])));   // x <= 1 < the prefix: below 2^53
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    long long v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) sh[k][threadIdx.x / kWave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    long long v = 0;
    for (int w = 0; w < kBlock / kWave; ++w) v += sh[threadIdx.x][w];
    block_n[static_cast<uint64_t>(blockIdx.x) * 4 + threadIdx.x] = v;
  }
}

// One workgroup of four waves, wave k the running sum of class frequency k over all blocks in order (see above).
// out[0..3] = the four sums, out[4] = 0 (the Ritland slot of a kParts0 row).  Everything a wave decides on is
// wave-uniform: every lane carries the same running sum.  Blocks go 64 at a time: where all 64 share one predicted
// binade their integer sums are added up across the lanes and taken in one step (every partial sum lies between the two
// ends, so it stays in the binade as well); otherwise block by block; a block that has to be walked has its 1024 values
// fetched first (16 loads in flight) and then added one by one, as the reference adds them.
static __global__ void __launch_bounds__(kBlock)
k_seq_chain(const double* __restrict__ table, const uint8_t* __restrict__ flags, uint64_t n_sel, uint32_t amax,
            const int* __restrict__ e_pred, const long long* __restrict__ block_n, uint64_t n_blocks, double* __restrict__ out /* [kParts0] */) {
  const int k = threadIdx.x / kWave;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t stride = sweep_stride(amax);
  double running = 0.0;
  auto take = [&](int e, long long n) {                    // add n ulps of binade e if the sum is in that binade before and after
    if (e == kSeqWalk || !(running > 0.0) || ilogb(running) != e) return false;
    const double next = running + ldexp(static_cast<double>(n), e - 52);        // exact while it stays in the binade
    if (ilogb(next) != e) return false;
    running = next;
    return true;
  };
  for (uint64_t b0 = 0; b0 < n_blocks; b0 += kWave) {
    const uint64_t mine = b0 + lane;
    const int e_lane = mine < n_blocks ? e_pred[mine * 4 + k] : kSeqWalk;
    const long long n_lane = mine < n_blocks ? block_n[mine * 4 + k] : 0;
    const int in_batch = static_cast<int>(n_blocks - b0 < static_cast<uint64_t>(kWave) ? n_blocks - b0 : kWave);
    const int e_first = __shfl(e_lane, 0, kWave);
    if (in_batch == kWave && e_first != kSeqWalk && __all(e_lane == e_first)) {
      long long total = n_lane;
      for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, kWave);
      if (total < (1ll << 52) && take(e_first, total)) continue;
    }
    for (int i = 0; i < in_batch; ++i) {
      if (take(__shfl(e_lane, i, kWave), __shfl(n_lane, i, kWave))) continue;
      // the reference's own loop over this block's loci
      const uint64_t s0 = (b0 + i) * static_cast<uint64_t>(kSeqBlock);
      double x[kSeqBlock / kWave];
#pragma unroll
      for (int c = 0; c < kSeqBlock / kWave; ++c) {
        const uint64_t s = s0 + static_cast<uint64_t>(c) * kWave + lane;
        x[c] = (s < n_sel && (flags[s] & kLocusDefault)) ? table[s * stride + amax + 1 + k] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < kSeqBlock / kWave; ++c)
        for (int j = 0; j < kWave; ++j) running += __shfl(x[c], j, kWave);      // + 0.0 where no default locus: exact
    }
  }
  if (lane == 0) out[k] = running;
  if (threadIdx.x == 0) out[4] = 0.0;
}

// processHallME's update: F <- expectation_sum / N (N = all classified loci, _calc.cpp:283).  walked > 0: the sums
// come from k_inbreed_eval_lut<1> (sum over homozygous cells of 1/den, plus 1 for every other locus slot walked).
static __global__ void __launch_bounds__(kBlock)
k_hall_update(const double* __restrict__ expectation_sum, const unsigned long long* __restrict__ counts, uint64_t n,
              unsigned long long walked, double* __restrict__ f) {
  for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n;
       g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    double sum = expectation_sum[g];
    if (walked) {
      const unsigned long long homozygous = counts[g * 6 + 0] + counts[g * 6 + 2];
      sum = f[g] > 0.0 ? f[g] * (sum - static_cast<double>(walked - homozygous)) : 0.0;
    }
    f[g] = sum / static_cast<double>(counts[g * 6 + 4]);
  }
}

// Golden-section maximiser state per genome: bracket [a,b], interior points c<d with values fc, fd.
// phase 0: f_eval holds value at c -> store, next eval at d; phase 1: value at d -> store, start shrinking;
// phase 2: f_eval is the value at the newly placed point.
struct GoldenState { double a, b, c, d, fc, fd; int last_was_c; int pad; };

static __global__ void __launch_bounds__(kBlock)
k_golden_step(GoldenState* __restrict__ st, const double* __restrict__ f_eval, uint64_t n, int phase, double* __restrict__ f_next) {
  const double inv_phi = 0.6180339887498949;
  for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n;
       g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    GoldenState s = st[g];
    if (phase == 0) {
      s.fc = f_eval[g];
      f_next[g] = s.d;
    } else {
      if (phase == 1) s.fd = f_eval[g];
      else if (s.last_was_c) s.fc = f_eval[g];
      else s.fd = f_eval[g];
      if (s.fc > s.fd) {          // maximum in [a, d]
        s.b = s.d; s.d = s.c; s.fd = s.fc;
        s.c = s.b - inv_phi * (s.b - s.a);
        s.last_was_c = 1;
        f_next[g] = s.c;
      } else {                    // maximum in [c, b]
        s.a = s.c; s.c = s.d; s.fc = s.fd;
        s.d = s.a + inv_phi * (s.b - s.a);
        s.last_was_c = 0;
        f_next[g] = s.d;
      }
    }
    st[g] = s;
  }
}

// Brent's method (golden-section steps with a parabolic step whenever the fit through the three best points is
// acceptable; "Algorithms for Minimization without Derivatives", ch. 5) on -loglikelihood over [-1, 1], one state per
// genome, one objective evaluation (= one sweep over the genotype bytes) per step for every genome that has not
// converged.  mode 0: f_eval is the value at the start point; 1: at the point proposed last; 2: no evaluation,
// just publish the best point.  still_running counts the genomes that proposed a new point.
struct BrentState { double a, b, x, w, v, fx, fw, fv, d, e, u, a0, b0; int done; int widened; };

// One step of the search: take the objective value fu (= -loglikelihood) at the point proposed last (first: at the
// start point), update the bracket and the three best points, and either finish (s.done) or propose the next point s.u.
__device__ __forceinline__ void brent_advance(BrentState& s, double fu, bool first) {
  // Absolute tolerance: the search stops with the best point within 2 * tol1 = 5e-7 of the maximiser; the reference
  // stops its Nelder-Mead at an absolute parameter change of 1e-6 (_calc.cpp:139).  Asking for much less runs into the
  // rounding noise of the objective (a sum of ~1e6 logs), where parabolic steps stop working.
  constexpr double kGold = 0.3819660112501051, kTol = 0.0, kZeps = 2.5e-7;
  if (first) {
    s.fx = s.fw = s.fv = fu;
  } else {
    const double u = s.u;
    if (fu <= s.fx) {
      if (u >= s.x) s.a = s.x; else s.b = s.x;
      s.v = s.w; s.w = s.x; s.x = u;
      s.fv = s.fw; s.fw = s.fx; s.fx = fu;
    } else {
      if (u < s.x) s.a = u; else s.b = u;
      if (fu <= s.fw || s.w == s.x) { s.v = s.w; s.w = u; s.fv = s.fw; s.fw = fu; }
      else if (fu <= s.fv || s.v == s.x || s.v == s.w) { s.v = u; s.fv = fu; }
    }
  }
  double xm = 0.5 * (s.a + s.b);
  const double tol1 = kTol * fabs(s.x) + kZeps, tol2 = 2.0 * tol1;
  if (fabs(s.x - xm) <= (tol2 - 0.5 * (s.b - s.a))) {
    // Converged inside the bracket.  The search starts in a window around the Simple estimate (brent_start); a best point
    // sitting on an edge of that window means the maximum lies beyond it: open the bracket to [-1, 1] once and go on
    // (the best points kept so far stay valid).
    const bool on_edge = (s.a0 > -1.0 && s.x - s.a0 < 1.0e-5) || (s.b0 < 1.0 && s.b0 - s.x < 1.0e-5);
    if (!on_edge || s.widened) {
      s.done = 1;
      return;
    }
    s.widened = 1;
    s.a = -1.0; s.b = 1.0; s.a0 = -1.0; s.b0 = 1.0;
    s.e = 0.0;
    xm = 0.0;
  }
  bool golden = true;
  if (fabs(s.e) > tol1) {
    const double r = (s.x - s.w) * (s.fx - s.fv);
    double q = (s.x - s.v) * (s.fx - s.fw);
    double p = (s.x - s.v) * q - (s.x - s.w) * r;
    q = 2.0 * (q - r);
    if (q > 0.0) p = -p;
    q = fabs(q);
    const double etemp = s.e;
    s.e = s.d;
    if (!(fabs(p) >= fabs(0.5 * q * etemp) || p <= q * (s.a - s.x) || p >= q * (s.b - s.x))) {
      s.d = p / q;
      const double u = s.x + s.d;
      if (u - s.a < tol2 || s.b - u < tol2) s.d = copysign(tol1, xm - s.x);
      golden = false;
    }
  }
  if (golden) {
    s.e = s.x >= xm ? s.a - s.x : s.b - s.x;
    s.d = kGold * s.e;
  }
  s.u = fabs(s.d) >= tol1 ? s.x + s.d : s.x + copysign(tol1, s.d);
}

// Start of the search for one genome: a +-0.25 window around the Simple estimate (obsHom - expHom) / (N - expHom)
// (processSimple, _calc.cpp:318-365; counts and class-frequency sums of the frequency sweep), which is within a few
// hundredths of the likelihood maximum on any real population; brent_advance opens the window to [-1, 1] if the
// maximum turns out to lie outside.  No usable estimate (no loci): the whole interval from its golden point.
__device__ __forceinline__ BrentState brent_start(const unsigned long long* __restrict__ counts, const double* __restrict__ sums) {
  BrentState s{};
  s.a = s.a0 = -1.0; s.b = s.b0 = 1.0;
  s.x = s.a + 0.3819660112501051 * (s.b - s.a);
  if (counts && sums) {
    const double observed = static_cast<double>(counts[0] + counts[2]), expected = sums[0] + sums[2];
    const double simple = (observed - expected) / (static_cast<double>(counts[4]) - expected);
    if (simple == simple && fabs(simple) <= 2.0) {
      const double x0 = simple < -0.9 ? -0.9 : (simple > 0.9 ? 0.9 : simple);
      s.a = s.a0 = x0 - 0.25 < -1.0 ? -1.0 : x0 - 0.25;
      s.b = s.b0 = x0 + 0.25 > 1.0 ? 1.0 : x0 + 0.25;
      s.x = x0;
    }
  }
  s.w = s.v = s.u = s.x;
  return s;
}

// The reference's own optimiser for this objective, step for step: nlopt's LN_NELDERMEAD in one dimension (the simplex
// method after Nelder & Mead / Box: reflection 1, expansion 2, contraction 1/2) on [-1, 1], stopped at an absolute
// simplex width of 1e-6 or 500 evaluations (createLogLikelihoodOptimizer, _calc.cpp:131-144), from the start point x0
// handed in (the reference draws it from (-0.5, 0.5], _calc.cpp:166,180) with nlopt's default first step, a quarter of
// the box, turned inward where it would leave the box.  The clamped
// objective has several local maxima where F < 0 (every homozygous cell whose probability falls under the 1e-10 floor
// stops pulling); which one a search ends on depends on its path, so the path is the reference optimiser's (as
// restated in oracle/kgo_inbreed.cpp:neldermead1D, which this follows expression for expression): the same maximum,
// to the rounding of the objective.  One evaluation per call; the state lives in a BrentState:
//   a = best point, fx = its value;  b = worst point, fw = its value;  v = reflected point, fv = its value;
//   u = the point to evaluate next;  widened = phase;  e = evaluations so far;  x = the result once done.
enum : int { kNmFirst = 0, kNmSecond, kNmReflect, kNmExpand, kNmOutside, kNmInside };
constexpr int kSearchBrent = 0, kSearchNelderMead = 1;
__device__ __forceinline__ void nm_first_simplex(double x0, double& xa, double& xb) {
  auto clampx = [](double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); };
  const double step = (1.0 - -1.0) * 0.25;
  xa = clampx(x0);
  xb = xa + step;
  if (xb > 1.0) xb = xa - step;
  xb = clampx(xb);
}
__device__ __forceinline__ BrentState nm_start(double x0) {
  BrentState s{};
  nm_first_simplex(x0, s.a, s.b);
  s.x = s.u = s.a;
  s.widened = kNmFirst;
  return s;
}
__device__ __forceinline__ void nm_advance(BrentState& s, double f) {      // f: the objective (maximised) at s.u
  auto clampx = [](double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); };
  s.e += 1.0;
  switch (s.widened) {
    case kNmFirst:
      s.fx = f;
      s.u = s.b;
      s.widened = kNmSecond;
      return;
    case kNmSecond:
      s.fw = f;
      break;
    case kNmReflect:
      s.v = s.u; s.fv = f;
      if (f > s.fx) { s.u = clampx(s.a + 2.0 * (s.a - s.b)); s.widened = kNmExpand; }
      else if (f > s.fw) { s.u = clampx(s.a + 0.5 * (s.v - s.a)); s.widened = kNmOutside; }
      else { s.u = s.a + 0.5 * (s.b - s.a); s.widened = kNmInside; }
      return;
    case kNmExpand:
      if (f > s.fv) { s.b = s.u; s.fw = f; } else { s.b = s.v; s.fw = s.fv; }
      break;
    case kNmOutside:
      if (f >= s.fv) { s.b = s.u; s.fw = f; } else { s.b = s.v; s.fw = s.fv; }
      break;
    default:                                                   // inside contraction == shrink in one dimension
      s.b = s.u; s.fw = f;
      break;
  }
  if (s.fw > s.fx) {                                           // a best, b worst
    const double x = s.a, fx = s.fx;
    s.a = s.b; s.fx = s.fw;
    s.b = x; s.fw = fx;
  }
  if (fabs(s.a - s.b) < 1e-6 || s.e >= 500.0) {
    s.x = s.a;
    s.done = 1;
    return;
  }
  s.u = clampx(s.a + (s.a - s.b));
  s.widened = kNmReflect;
}

// The same search with TWO evaluations per step (k_inbreed_eval_lut<2, ., ., true>): the two start points together, then
// for every simplex the reflected point AND the inside-contraction point -- both follow from the simplex alone, and in
// one dimension 97 % of the steps end in that contraction (once the maximum is bracketed, the reflection lands beyond the
// worse point).  When the reflection is better than the worst point after all, the expansion or the outside contraction
// is evaluated in the next step (its twin slot repeats it).  Every comparison is made on the values the one-at-a-time
// search would have seen, in its order: the same path, the same result, in about half the passes over the bytes.
//   u, d = the two points to evaluate next;  e counts the evaluations the one-at-a-time search would have made.
enum : int { kNm2Start = 0, kNm2Reflect, kNm2Expand, kNm2Outside };
constexpr int kSearchNelderMeadPair = 2;
__device__ __forceinline__ BrentState nm2_start(double x0) {
  BrentState s{};
  nm_first_simplex(x0, s.a, s.b);
  s.u = s.a; s.d = s.b; s.x = s.a;
  s.widened = kNm2Start;
  return s;
}
__device__ __forceinline__ void nm2_advance(BrentState& s, double f0, double f1) {     // the objective at s.u and at s.d
  auto clampx = [](double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); };
  switch (s.widened) {
    case kNm2Start:
      s.fx = f0; s.fw = f1;
      s.e = 2.0;
      break;
    case kNm2Reflect:
      s.v = s.u; s.fv = f0;                                    // reflected point
      if (f0 > s.fx) { s.u = s.d = clampx(s.a + 2.0 * (s.a - s.b)); s.widened = kNm2Expand; return; }
      if (f0 > s.fw) { s.u = s.d = clampx(s.a + 0.5 * (s.v - s.a)); s.widened = kNm2Outside; return; }
      s.b = s.d; s.fw = f1;                                    // inside contraction: its point was evaluated with the reflection
      s.e += 2.0;
      break;
    case kNm2Expand:
      if (f0 > s.fv) { s.b = s.u; s.fw = f0; } else { s.b = s.v; s.fw = s.fv; }
      s.e += 2.0;
      break;
    default:                                                   // outside contraction
      if (f0 >= s.fv) { s.b = s.u; s.fw = f0; } else { s.b = s.v; s.fw = s.fv; }
      s.e += 2.0;
      break;
  }
  if (s.fw > s.fx) {                                           // a best, b worst
    const double x = s.a, fx = s.fx;
    s.a = s.b; s.fx = s.fw;
    s.b = x; s.fw = fx;
  }
  if (fabs(s.a - s.b) < 1e-6 || s.e >= 500.0) {
    s.x = s.a;
    s.done = 1;
    return;
  }
  s.u = clampx(s.a + (s.a - s.b));
  s.d = s.a + 0.5 * (s.b - s.a);
  s.widened = kNm2Reflect;
}

static __global__ void __launch_bounds__(kBlock)
k_brent_init(const unsigned long long* __restrict__ counts, const double* __restrict__ sums, uint64_t n, int use_estimate, int search,
             const double* __restrict__ start, BrentState* __restrict__ st, double* __restrict__ f_next) {
  for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n;
       g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const BrentState s = search == kSearchNelderMeadPair ? nm2_start(start[g]) : search == kSearchNelderMead ? nm_start(start[g])
                         : use_estimate ? brent_start(counts + g * 6, sums + g * kParts0) : brent_start(nullptr, nullptr);
    st[g] = s;
    f_next[g] = s.x;
    if (search == kSearchNelderMeadPair) f_next[n + g] = s.d;                  // two planes of n values
  }
}

// global_of / result (both or neither): the states are a compacted subset of the call's genomes (see kgx_inbreed);
// a genome's coefficient goes to result[global_of[g]] when its search ends.
static __global__ void __launch_bounds__(kBlock)
k_brent_step(BrentState* __restrict__ st, const double* __restrict__ f_eval, uint64_t n, int mode, int search, double* __restrict__ f_next,
             unsigned int* __restrict__ still_running, const uint32_t* __restrict__ global_of, double* __restrict__ result) {
  for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n;
       g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    BrentState s = st[g];
    if (mode == 2) {
      f_next[g] = s.x;
      if (global_of) result[global_of[g]] = s.x;
      continue;
    }
    if (!s.done) {
      if (search == kSearchNelderMeadPair) nm2_advance(s, f_eval[g], f_eval[n + g]);
      else if (search == kSearchNelderMead) nm_advance(s, f_eval[g]);
      else brent_advance(s, -f_eval[g], mode == 0);
      if (!s.done) atomicAdd(still_running, 1u);
      else if (global_of) result[global_of[g]] = s.x;
      st[g] = s;
    }
    f_next[g] = s.done ? s.x : s.u;
    if (search == kSearchNelderMeadPair) f_next[n + g] = s.done ? s.x : s.d;
  }
}

// Compaction of the genomes still searching (kgx_inbreed, Loglikelihood): their genotype columns, dense in the selected
// loci, and their search states.
static __global__ void __launch_bounds__(kBlock)
k_gather_columns(const uint8_t* __restrict__ src, uint64_t src_pitch, uint64_t src_g0, const uint32_t* __restrict__ locus_index,
                 uint64_t n_sel, const uint32_t* __restrict__ columns, uint64_t n_columns, uint8_t* __restrict__ dst, uint64_t dst_pitch) {
  // a thread owns four consecutive output columns: four byte reads out of one (cached) source row, one dword store
  const uint64_t j4 = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  if (j4 >= dst_pitch) return;
  uint64_t column[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) column[k] = j4 + k < n_columns ? src_g0 + columns[j4 + k] : ~0ull;
  for (uint64_t s = blockIdx.y; s < n_sel; s += gridDim.y) {
    const uint64_t l = locus_index ? static_cast<uint64_t>(locus_index[s]) : s;
    const uint8_t* row = src + l * src_pitch;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (column[k] != ~0ull) packed |= static_cast<uint32_t>(row[column[k]]) << (8 * k);
    *reinterpret_cast<uint32_t*>(dst + s * dst_pitch + j4) = packed;      // the tail of the row is zero
  }
}

static __global__ void __launch_bounds__(kBlock)
k_gather_states(const BrentState* __restrict__ st, const double* __restrict__ f, uint64_t n_source, int planes, const uint32_t* __restrict__ columns,
                uint64_t n_columns, BrentState* __restrict__ st_out, double* __restrict__ f_out) {
  for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n_columns;
       j += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    st_out[j] = st[columns[j]];
    for (int k = 0; k < planes; ++k) f_out[k * n_columns + j] = f[k * n_source + columns[j]];      // planes of n values (the paired search: 2)
  }
}

// Window-sized calls (what the INBREED package issues: ~1000 sampled loci x one super population): the whole iteration
// of HallME (MODE 1) or Loglikelihood (MODE 2) in ONE launch.  THREADS threads own a genome -- a block (256), or a
// wave (64) where the call has genomes enough to fill the SIMDs with one wave each; thread t owns loci t, t+THREADS,
// ... and keeps what each of its cells contributes in registers (as k_inbreed_eval_lut tabulates it: y, d with the
// cell's probability or denominator y + F*d).  A pass is a few fp64 operations per cell, a DPP reduction over each row
// of 16 lanes (row_sum16) and the row sums added as a fixed tree -- the wave's four read with v_readlane, the block's
// sixteen out of LDS (one barrier per pass, the slots alternate): every thread of the genome ends with the
// bitwise-same sum, so the search's control flow is uniform over them.  No grid synchronisation, no partials in
// memory.  n_sel <= kGenomeLoci; CELLS (the host picks the smallest power of two that holds n_sel / THREADS) bounds the
// unrolled per-thread loops: with one or two waves on a SIMD a pass costs the latency of its dependent instructions,
// and a cell past the selection would cost as much as a real one (it contributes +0.0 / a factor 1.0: leaving it out
// gives the bitwise-same sums).
constexpr int kGenomeLoci = 8192;                        // a block per genome: 32 cells per thread; a wave per genome: up to kGenomeWaveLoci
constexpr int kGenomeWaveLoci = 2048;

// A double moved between the lanes of a row of 16 by a DPP control (two v_mov_b32_dpp: a few cycles, where the
// ds_bpermute pair behind __shfl_xor takes an LDS round trip -- and the passes here are nothing but such latencies).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v));
  const unsigned int lo = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(bits), CTRL, 0xf, 0xf, false));
  const unsigned int hi = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(bits >> 32), CTRL, 0xf, 0xf, false));
  return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}

// The sum over each row of 16 lanes, the bitwise-same value in all 16: every step pairs lanes symmetrically (a with b
// and b with a), and a + b == b + a.  All lanes of the wave must be active.
__device__ __forceinline__ double row_sum16(double v) {
  v += dpp_move<0xB1>(v);                                      // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_move<0x4E>(v);                                      // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_move<0x141>(v);                                     // row_half_mirror: 7 - lane within each 8 (the other quad)
  v += dpp_move<0x140>(v);                                     // row_mirror: 15 - lane (the other half)
  return v;
}

__device__ __forceinline__ double read_lane(double v, int lane) {
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v));
  const unsigned int lo = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(bits), lane));
  const unsigned int hi = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(bits >> 32), lane));
  return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}

template <int MODE, int CELLS, int THREADS>
__global__ void __launch_bounds__(kBlock)
k_inbreed_iterate_genome(const uint8_t* __restrict__ gt, uint64_t pitch, uint64_t g0, uint64_t n_genomes,
                         const uint32_t* __restrict__ locus_index, uint64_t n_sel, const double* __restrict__ table,
                         const uint8_t* __restrict__ valid, uint32_t amax, int phased, const unsigned long long* __restrict__ counts,
                         const double* __restrict__ sums, int search, const double* __restrict__ start, double* __restrict__ f_out,
                         unsigned int* __restrict__ max_evaluations) {
  static_assert(THREADS == kWave || (THREADS == kBlock && kBlock == 256), "a wave or a block of sixteen rows per genome");
  static_assert(CELLS * THREADS <= kGenomeLoci, "cells past the largest selection");
  __shared__ double row_part[2][16];
  const uint32_t t = threadIdx.x % THREADS;
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * (kBlock / THREADS) + threadIdx.x / THREADS;
  if (g >= n_genomes) return;                                 // whole waves / whole blocks only: no barrier is left waiting
  // the sum over the genome's threads of one value each, the same bits in every thread; `pass` alternates the LDS slots
  auto block_sum = [&](double v, int pass) {
    v = row_sum16(v);
    if constexpr (THREADS == kWave) {
      return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
    } else {
      if ((threadIdx.x & 15) == 0) row_part[pass & 1][threadIdx.x >> 4] = v;
      __syncthreads();
      const double* p = row_part[pass & 1];
      double pair[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pair[i] = p[2 * i] + p[2 * i + 1];
      return ((pair[0] + pair[1]) + (pair[2] + pair[3])) + ((pair[4] + pair[5]) + (pair[6] + pair[7]));
    }
  };
  const uint32_t stride = sweep_stride(amax);
  double y[CELLS], d[CELLS];
#pragma unroll
  for (int c = 0; c < CELLS; ++c) {
    y[c] = MODE == 2 ? 1.0 : 0.0;                             // contributes nothing
    d[c] = 0.0;
    const uint64_t s = static_cast<uint64_t>(c) * THREADS + t;
    if (s < n_sel && (valid[s] & kLocusValid)) {
      const uint64_t l = locus_index ? static_cast<uint64_t>(locus_index[s]) : s;
      double f1 = 0.0, f2 = 0.0;
      const int cls = classify_cell(gt[l * pitch + g0 + g], table + s * stride, amax, phased != 0, f1, f2);
      if (cls == kMajorHom || cls == kMinorHom) {
        if constexpr (MODE == 2) { y[c] = f1 * f1; d[c] = f1 - y[c]; }
        else { y[c] = f1; d[c] = 1.0; }                       // d = 1 marks a homozygous cell, y its allele frequency
      } else if (cls != kClassNone) {
        if constexpr (MODE == 2) { y[c] = 2.0 * f1 * f2; d[c] = -y[c]; }
      }
    }
  }
  if constexpr (MODE == 1) {
    // processHallME (_calc.cpp:255-285): 50 expectation steps from the genome's start point (see kgx_inbreed)
    const double total = static_cast<double>(counts[g * 6 + 4]);
    double F = start[g];
    for (int it = 0; it < 50; ++it) {
      // (selects, not branches: the quotients of different cells are independent and overlap)
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < CELLS; ++c) {
        const double denominator = F + ((1.0 - F) * y[c]);
        const double quotient = F / denominator;
        sum += (d[c] != 0.0 && denominator != 0) ? quotient : 0.0;
      }
      F = block_sum(sum, it) / total;
    }
    if (t == 0) f_out[g] = F;
  } else {
    BrentState s = search == kSearchNelderMead ? nm_start(start[g])
                   : sums ? brent_start(counts + g * 6, sums + g * kParts0) : brent_start(nullptr, nullptr);
    unsigned int evaluations = 0;
    for (int it = 0; it < (search == kSearchNelderMead ? 500 : 60); ++it) {
      const double F = it == 0 ? s.x : s.u;
      double logs = 0.0, prod = 1.0;
#pragma unroll
      for (int c = 0; c < CELLS; ++c) {
        prod *= __builtin_fmin(__builtin_fmax(__builtin_fma(F, d[c], y[c]), 1e-10), 1.0);
        if ((c & 15) == 15 || c == CELLS - 1) {               // 16 factors >= 1e-10 cannot underflow
          logs += log(prod);
          prod = 1.0;
        }
      }
      const double log_sum = block_sum(logs, it);
      ++evaluations;
      if (search == kSearchNelderMead) nm_advance(s, log_sum);
      else brent_advance(s, -log_sum, it == 0);
      if (s.done) break;                                      // uniform over the genome's threads: all hold the same state
    }
    if (t == 0) {
      f_out[g] = s.x;
      atomicMax(max_evaluations, evaluations);
    }
  }
}

// LocusResults per genome (kga_analysis_inbreed_output.h:21-35) incl. the estimator's coefficient:
//   Simple  (processSimple, _calc.cpp:318-365): (obsHom - expHom) / (N - expHom)
//   Ritland (processRitlandLocus :374-431):     sum / count
//   HallME / Loglikelihood: the iterated value handed in through f.
struct LocusResultsDev {
  unsigned long long major_hetero_count; double major_hetero_freq;
  unsigned long long minor_hetero_count; double minor_hetero_freq;
  unsigned long long minor_homo_count;   double minor_homo_freq;
  unsigned long long major_homo_count;   double major_homo_freq;
  unsigned long long total_allele_count; double inbred_allele_sum;
};

static __global__ void __launch_bounds__(kBlock)
k_finish_inbreed(const unsigned long long* __restrict__ counts, const double* __restrict__ sums, uint64_t n, int algorithm,
                 const double* __restrict__ f, LocusResultsDev* __restrict__ out) {
  for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n;
       g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const unsigned long long* c = counts + g * 6;     // majorHom, majorHet, minorHom, minorHet, total, ritland count
    const double* s = sums + g * kParts0;              // majorHom, majorHet, minorHom, minorHet, ritland sum
    LocusResultsDev r;
    r.major_homo_count = c[0]; r.major_hetero_count = c[1]; r.minor_homo_count = c[2]; r.minor_hetero_count = c[3];
    r.total_allele_count = c[4];
    r.major_homo_freq = s[0]; r.major_hetero_freq = s[1]; r.minor_homo_freq = s[2]; r.minor_hetero_freq = s[3];
    double coefficient = 0.0;
    if (algorithm == 1) {
      if (r.total_allele_count > 0) {
        const double observed_homozygous = static_cast<double>(r.minor_homo_count + r.major_homo_count);
        const double expected_homozygous = r.minor_homo_freq + r.major_homo_freq;
        coefficient = (observed_homozygous - expected_homozygous) / (static_cast<double>(r.total_allele_count) - expected_homozygous);
      }
    } else if (algorithm == 0) {
      coefficient = c[5] > 0 ? s[4] / static_cast<double>(c[5]) : 0.0;
    } else {
      coefficient = f[g];
    }
    r.inbred_allele_sum = coefficient;
    out[g] = r;
  }
}

// Synthetic multi-allelic genotype bytes straight into HBM (one thread per dword = 4 genomes of one locus),
// plus the per-locus SNP allele-frequency table [n_loci][3] (NaN padded) the inbreeding sweep needs.
static __global__ void __launch_bounds__(kBlock)
k_synth_gt8(uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t n_loci, uint64_t n_genomes, uint64_t seed,
            uint64_t genome_base, uint64_t locus_base, double* __restrict__ af_table) {
  const uint64_t quads = (n_genomes + 3) / 4;
  const uint64_t total = n_loci * quads;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t l = i / quads, q = i % quads;
    const kgx_synth_locus loc = kgx_synth_make_locus(seed, locus_base + l);
    if (q == 0 && af_table) {
      double row[KGX_SYNTH_MAX_ALTS] = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
      for (int a = 0; a < loc.n_alt; ++a)
        if (!loc.is_indel[a]) row[loc.snp_index[a] - 1] = static_cast<double>(loc.af[a]);
      for (int a = 0; a < KGX_SYNTH_MAX_ALTS; ++a) af_table[l * KGX_SYNTH_MAX_ALTS + a] = row[a];
    }
    uint32_t word = 0;
    for (int j = 0; j < 4; ++j) {
      const uint64_t g = q * 4 + j;
      if (g < n_genomes) {
        int a1, a2;
        kgx_synth_multi_genotype(seed, locus_base + l, genome_base + g, loc, a1, a2);
        word |= kgx_synth_gt8_byte(loc, a1, a2) << (8 * j);
      }
    }
    gt[l * dwords_per_row + q] = word;
  }
}

// Synthetic inbred genomes for the reference's self-check (InbreedSynthetic::generateSyntheticPopulation,
// kga_analytic/kga_inbreed/kga_analysis_inbreed_syngen.cpp:20-196): genome g has F = inbreeding[g]; at every locus
// of the list an allele class is drawn from alleleClassFrequencies(F) and alleles from the select* functions
// (kga_analysis_inbreed_freq.cpp:221-420).  The reference draws from std::random_device; here Philox4x32-10 keyed by
// `seed`, counter (locus, genome_base + genome, KGX_STREAM_SELFCHECK).  One thread per (locus, genome) byte of a shard
// whose first genome is genome_base.
static __global__ void __launch_bounds__(kBlock)
k_synth_inbred(uint8_t* __restrict__ gt, uint64_t pitch, uint64_t n_loci, uint64_t n_genomes, const double* __restrict__ af_table,
               uint32_t amax, const double* __restrict__ inbreeding, uint64_t seed, uint64_t genome_base) {
  const uint64_t total = n_loci * n_genomes;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t l = i / n_genomes, g = i % n_genomes;
    const double F = inbreeding[g];
    const double* row = af_table + l * amax;
    double f[14];
    uint32_t alt_of[14];
    uint32_t n = 0;
    double sum_minor = 0.0;
    for (uint32_t a = 0; a < amax && a < 14; ++a) {
      const double x = row[a];
      if (x == x) { f[n] = clamp01(x); alt_of[n] = a + 1; sum_minor += f[n]; ++n; }
    }
    uint32_t a1 = 0, a2 = 0;
    if (n > 0) {
      const double major = (1.0 - sum_minor) > 0.0 ? (1.0 - sum_minor) : 0.0;
      const bool rescale = sum_minor > 1.0;
      double minor_hom = 0.0, minor_het = 0.0, major_het = 0.0;
      for (uint32_t a = 0; a < n; ++a) { const double p = rescale ? f[a] / sum_minor : f[a]; minor_hom += (F * p) + ((1.0 - F) * p * p); }
      for (uint32_t a = 0; a < n; ++a)
        for (uint32_t b = a + 1; b < n; ++b) {
          const double pa = rescale ? f[a] / sum_minor : f[a], pb = rescale ? f[b] / sum_minor : f[b];
          minor_het += (1.0 - F) * 2.0 * pa * pb;
        }
      double major_hom = (F * major) + ((1.0 - F) * major * major);
      for (uint32_t a = 0; a < n; ++a) { const double p = rescale ? f[a] / sum_minor : f[a]; major_het += (1.0 - F) * 2.0 * major * p; }
      minor_hom = minor_hom > 0.0 ? minor_hom : 0.0; minor_het = minor_het > 0.0 ? minor_het : 0.0;
      major_hom = major_hom > 0.0 ? major_hom : 0.0; major_het = major_het > 0.0 ? major_het : 0.0;
      const double total_freq = major_hom + major_het + minor_hom + minor_het;
      minor_hom /= total_freq; minor_het /= total_freq; major_hom /= total_freq; major_het /= total_freq;
      const kgx_u32x4 r = kgx_philox4x32_10(static_cast<uint32_t>(l), static_cast<uint32_t>(l >> 32), static_cast<uint32_t>(genome_base + g), 5u,
                                            static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
      const double u_class = kgx_u01(r.v[0]), u_allele = kgx_u01(r.v[1]);
      double cum = minor_hom;
      if (u_class <= cum) {                                   // MINOR_HOMOZYGOUS
        if (minor_hom != 0.0) {
          if (n == 1) a1 = a2 = alt_of[0];
          else {
            double s2 = 0.0;
            for (uint32_t a = 0; a < n; ++a) {                // selectMinorHomozygous uses the un-rescaled frequencies
              s2 += ((f[a] * F) + (1.0 - F) * f[a] * f[a]) / minor_hom;
              if (u_allele <= s2) { a1 = a2 = alt_of[a]; break; }
            }
          }
        }
      } else if (u_class <= (cum += minor_het)) {             // MINOR_HETEROZYGOUS
        if (n >= 2 && minor_het != 0.0) {
          if (n == 2) { a1 = alt_of[0]; a2 = alt_of[1]; }
          else {
            double s2 = 0.0;
            bool done = false;
            for (uint32_t a = 0; a < n && !done; ++a)
              for (uint32_t b = a + 1; b < n; ++b) {
                s2 += ((1.0 - F) * 2.0 * f[a] * f[b]) / minor_het;
                if (u_allele <= s2) { a1 = alt_of[a]; a2 = alt_of[b]; done = true; break; }
              }
          }
        }
      } else if (u_class <= (cum += major_hom)) {             // MAJOR_HOMOZYGOUS
      } else if (u_class <= (cum += major_het)) {             // MAJOR_HETEROZYGOUS
        if (major_het != 0.0) {
          if (n == 1) a1 = alt_of[0];
          else {
            const double major_freq = clamp01(1.0 - clamp01(sum_minor));
            double s2 = 0.0;
            for (uint32_t a = 0; a < n; ++a) {
              s2 += ((1.0 - F) * 2.0 * major_freq * f[a]) / major_het;
              if (u_allele <= s2) { a1 = alt_of[a]; break; }
            }
          }
          // the reference's RandomBoolean phase of the single carrier (_syngen.cpp:107-115) has no slot in gt8: the low
          // nibble is the first variant of the OffsetDB array whatever its phase, and no estimator reads it
        }
      }
    }
    gt[l * pitch + g] = static_cast<uint8_t>(a1 | (a2 << 4));
  }
}

// Pack caller genome-major bytes [n][n_loci] into locus-major gt8 rows (one thread per output dword).
static __global__ void __launch_bounds__(kBlock)
k_gt8_transpose(const uint8_t* __restrict__ src, uint64_t n_src, uint64_t n_loci, uint64_t g0, uint8_t* __restrict__ gt, uint64_t pitch) {
  const uint64_t total = n_src * n_loci;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t l = i % n_loci, g = i / n_loci;      // consecutive threads read consecutive loci of one genome
    gt[l * pitch + g0 + g] = src[g * n_loci + l];
  }
}

}  // namespace kgx

#endif  // KGX_KERNELS_INBREED_H
