// Loglikelihood over a large call without a pass over the genotype bytes per evaluation (driven from kgx_inbreed.hip:
// inbreed_shard, `loglik_moments`).  The reference's objective (logLikelihood, kga_analysis_inbreed_calc.cpp:94-129) is, per genome,
//     sum over homozygous cells of log clamp(F*y + (1-F)*y*y)  +  sum over heterozygous cells of log clamp(2*(1-F)*f1*f2),
// clamp = to [1e-10, 1], y the frequency of the cell's allele.  nlopt's Nelder-Mead evaluates it ~55 times per genome
// (processLogLikelihood, :153-216); as table passes that is 38 sweeps over the matrix (two values per sweep).  Here:
//
//   homozygous cells   F*y + u*y*y = y * (F + u*y), u = 1 - F.  Above the floor the log is  log y + log(A + u*d)  with the
//       cell's bin centre c (kgx_kernels_hall.h: the y axis in bins of relative half-width 2^-8), d = y - c, A = F + u*c:
//           sum = M0 * log(c*A) + sum_j (-1)^(j+1) M_j / (j c^j)  +  sum_j (-1)^(j+1) M_j (u/A)^j / j ,   j = 1..4,
//       from the SAME per-(genome, bin) moments M_j = sum d^j HallME runs on -- one pass over the bytes per class of
//       homozygous cell, whatever the number of evaluations.  The second series converges like (u*|d|/A)^j: fine for F >= 0
//       (u*|d|/A <= |d|/c <= 2^-8), but for F < 0 the cells with y < y* = -F/u have NEGATIVE probability -- they sit on the
//       1e-10 floor -- and A -> 0 for the bins next to y*.  So a bin is taken by its moments only where u*2^-8*c <= kTmax * A
//       (truncation <= kTmax^5 / 5 = 2e-7 of a term at the band's edge and falling with the fifth power of the distance; measured
//       against the direct sum: 1.5e-11 of the objective, and the search walks the identical path: scripts/proto/loglik_moments.py);
//       a bin wholly under the floor is M0 * log(1e-10); and the bins in between -- the band [y_floor, y* / (1 - 2^-8 / kTmax)),
//       about 9 bins -- are walked EXACTLY, cell by cell: the class passes leave one bit per (position of the bin-sorted
//       order, genome) for the bins a band can reach (y < 0.542: y* <= 1/2), a genome's set bits there are its cells, and each is
//       clamp(fma(F, y - y*y, y*y)) as the table pass computes it.
//   heterozygous cells   the frequency sweep's table pass carries their term (k_eval_entries<5>: log w, w = 2*f1*f2, summed
//       per genome in fp64 like RitlandLocus' term): sum = T + H * log u unless u < 1e-10 / (the call's smallest w), where the
//       floor could bind for some -- all of them at u <= 2e-10 (F = 1, where the optimiser's box ends), otherwise the genome is
//       handed to the passes (`needs_passes`), as is a genome with a cell that can meet the UPPER bound (kOddBigHet).
//
// One workgroup per genome runs the whole search -- the reference optimiser's path, nm_advance -- on the genome's moments held
// in registers; every thread computes the bitwise-same objective (block_sum), so the control flow is uniform.
#pragma once

#include <cstdint>

#include "kgx_kernels_hall.h"

namespace kgx {

constexpr int kLoglikTmaxLog2 = 4;                           // kTmax = 2^-4
constexpr double kLoglikTmax = 1.0 / (1 << kLoglikTmaxLog2);
constexpr double kLoglikBand = 1.0 / (1.0 - (1.0 / 256.0) / kLoglikTmax);   // y* to the band's upper edge: 1 / (1 - 2^-8 / kTmax) = 16/15
constexpr double kLoglikReach = 0.5 * kLoglikBand * (1.0 + 1.0 / 64.0);   // no band reaches past this y (y* <= 1/2, and a bin of margin)
constexpr int kLoglikMaxClasses = 15;                        // byte 0x00 and a | a << 4, a = 1..14
constexpr uint32_t kLoglikListCells = 2560;                  // a genome's cells of a band (or of a stretch of it), gathered in LDS: their frequencies (20 KB:
                                                             // with ~50 KB of moments two workgroups still share a CU's 160 KB)
constexpr int kLoglikWordsPerThread = 4;                     // blocks a thread looks at per round of the sparse walk: 1024 blocks a round
// from this share of a band's slots set, the band is walked slot by slot (every slot's frequency: 520 bytes a block and genome) instead of
// listing its set bits (a word a block, a frequency a cell).  Measured at C5, search kernel: 1/32 18.4 ms (58.6 GB fetched), 1/16 15.5,
// 1/8 13.4, 1/4 12.4, 1/2 12.4 (KGX_K7_LL_DENSE_PER_1024 sets it per 1024)
constexpr double kLoglikDense = 1.0 / 4.0;
constexpr double kLogSmallProb = -23.025850929940457;        // log(1e-10)
constexpr int kLoglikBinValues = 6;                          // per bin in LDS: M0, M1, M2/2, M3/3, M4/4, the series of sum(log y / c)

struct LoglikClass {
  const double* ys;                     // the frequency of every slot of the class's blocks (k_hall_pad)
  const uint32_t* bin_block;            // [kHallBins + 1]: the first block of each bin
  const unsigned long long* words;      // hall_word_index(block, genome): bit 63 - p = "homozygous at slot 64 * block + p"
};
struct LoglikClasses { LoglikClass of[kLoglikMaxClasses]; uint32_t n; uint32_t block_bins; uint32_t plain_words; uint64_t word_blocks; double dense; };   // plain_words: [genome][block of all classes, word_blocks of them] (k_hall_mfma from bit rows)

// dynamic LDS of k_loglik_search for n_used bins
inline size_t loglik_search_lds(uint32_t n_used) {
  return static_cast<size_t>(n_used) * (kLoglikBinValues * sizeof(double) + sizeof(uint32_t)) + 8;
}

// mode 0: the search, f_out[g] = the maximiser; mode 1 (a diagnostic: kgx_inbreed_objective): the objective at start[g].
// needs_passes[g] != 0: this genome's objective cannot be had from the statistics (see above; the value says why); its f_out is NaN.
// Dynamic LDS: loglik_search_lds(*n_used_ptr) bytes -- the genome's moments, bin-major.
//
// The band's cells.  A genome's cells in the bins [key_lo, key_hi) are the set bits of its words there.  Sparse (the usual
// case: a few per cent of a band's slots): the bits of up to 1024 blocks a round are listed -- positions from a scan over the
// threads, so the list's order, and with it every rounding, is the slots' own -- and their frequencies gathered into LDS; dense:
// slot by slot, a lane a slot.  Successive evaluations of a search look at overlapping bands (the simplex shrinks), so when
// the cells of a WIDER stretch of bins -- what the whole simplex can reach -- fit the list, that stretch is
// gathered once and kept: an evaluation whose band lies inside it walks the kept frequencies (each knows its bin) and touches no
// memory but LDS.
__global__ void __launch_bounds__(kBlock)
k_loglik_search(const double* __restrict__ bins, const uint32_t* __restrict__ used, const uint32_t* __restrict__ n_used_ptr,
                const unsigned long long* __restrict__ counts, const double* __restrict__ sums,
                const unsigned long long* __restrict__ smallest_het, uint64_t n_genomes, uint64_t words_per_block, LoglikClasses classes,
                const double* __restrict__ start, int mode, double* __restrict__ f_out, uint32_t* __restrict__ needs_passes,
                uint32_t* __restrict__ handed_over, unsigned int* __restrict__ max_evaluations, unsigned long long* __restrict__ stats = nullptr) {
  // stats (KGX_K7_TRACE; may be null): evaluations, those with cells in their band, served by the kept cells, gathers kept,
  // gathers not kept, dense walks, cells listed, blocks looked at
  extern __shared__ double bin_values[];                     // [kLoglikBinValues][n_used], then the bins' keys
  __shared__ double row_part[2][3][16];
  __shared__ double cell_y[kLoglikListCells];                // the listed cells' frequencies
  __shared__ uint32_t wave_total[2][kBlock / kWave];
  const uint64_t g = blockIdx.x;
  if (g >= n_genomes) return;
  const uint32_t n_used = *n_used_ptr;
  uint32_t* const bin_keys = reinterpret_cast<uint32_t*>(bin_values + static_cast<size_t>(kLoglikBinValues) * n_used);
  for (uint32_t i = threadIdx.x; i < n_used; i += kBlock) {
    const uint32_t bin = used[i];
    const double* m = bins + static_cast<uint64_t>(bin) * kHallMoments * n_genomes + g;
    double v[kLoglikBinValues] = {m[0], 0.0, 0.0, 0.0, 0.0, 0.0};
    if (bin != 0u) {                                         // (bin 0: y = 0, probability 0: under the floor whatever F)
      const double r = 1.0 / hall_centre(bin);
      v[1] = m[n_genomes]; v[2] = m[2 * n_genomes] * 0.5; v[3] = m[3 * n_genomes] * (1.0 / 3.0); v[4] = m[4 * n_genomes] * 0.25;
      v[5] = r * (v[1] - r * (v[2] - r * (v[3] - r * v[4])));
    }
#pragma unroll
    for (int j = 0; j < kLoglikBinValues; ++j) bin_values[static_cast<size_t>(j) * n_used + i] = v[j];
    bin_keys[i] = bin;
  }
  __syncthreads();
  const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const uint64_t lanes = words_per_block / 8;
  int pass = 0, scans = 0;
  // the sums over the workgroup of three values a thread, the same bits in every thread; `pass` alternates the LDS slots
  auto block_sum3 = [&](double& a, double& b, double& c) {
    a = row_sum16(a);
    b = row_sum16(b);
    c = row_sum16(c);
    double (*slot)[16] = row_part[pass & 1];
    if ((threadIdx.x & 15) == 0) { slot[0][threadIdx.x >> 4] = a; slot[1][threadIdx.x >> 4] = b; slot[2][threadIdx.x >> 4] = c; }
    __syncthreads();
    double out[3];
#pragma unroll
    for (int which = 0; which < 3; ++which) {
      const double* p = slot[which];
      double pair[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pair[i] = p[2 * i] + p[2 * i + 1];
      out[which] = ((pair[0] + pair[1]) + (pair[2] + pair[3])) + ((pair[4] + pair[5]) + (pair[6] + pair[7]));
    }
    a = out[0]; b = out[1]; c = out[2];
    ++pass;
  };
  // exclusive prefix of `mine` over the workgroup's threads (thread order), and the total: positions that do not depend on timing
  auto block_scan = [&](uint32_t mine, uint32_t& total) {
    uint32_t inclusive = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const uint32_t up = __shfl_up(inclusive, off);
      if (lane >= static_cast<uint32_t>(off)) inclusive += up;
    }
    uint32_t* totals = wave_total[scans & 1];
    if (lane == kWave - 1) totals[wave] = inclusive;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / kWave; ++w) {
      if (w < wave) before += totals[w];
      total += totals[w];
    }
    ++scans;
    return before + inclusive - mine;
  };
  const unsigned long long sixth = counts[g * 6 + 5], total_cells = counts[g * 6 + 4];
  const double het_big = static_cast<double>(sixth >> 32);
  const double het_tiny = static_cast<double>(total_cells - (sixth & 0xFFFFFFFFull));
  const double het_cells = static_cast<double>(counts[g * 6 + 1] + counts[g * 6 + 3]) - het_tiny;        // those with a term
  const double het_term = sums[g * kParts0 + 4];
  const double w_smallest = __longlong_as_double(static_cast<long long>(*smallest_het));                    // (1.0 where the call has none)
  // why the genome goes to the passes: 1 a cell that can meet the upper bound, 2 a band past the kept blocks, 3 the floor among the
  // heterozygous cells, 4 the hits' bits and the moments disagree (cannot happen)
  uint32_t hand_over = het_big != 0.0 ? 1u : 0u;

  // bins: [0, key_lo) wholly under the floor; [key_lo, key_hi) the band, walked exactly; from key_hi by their moments
  auto band_of = [&](double F, uint32_t& key_lo, uint32_t& key_hi) {
    const double u = 1.0 - F;
    // y_floor: the cells with F*y + u*y*y < 1e-10 are those with y below the positive root (F >= 0: next to 0; F < 0: next to y*)
    const double disc = sqrt(__builtin_fma(F, F, 4.0 * u * 1e-10));
    const double y_floor = u > 0.0 ? (F <= 0.0 ? (disc - F) / (2.0 * u) : 2e-10 / (F + disc)) : 1e-10;
    const double y_under = y_floor * (1.0 - 1e-9), y_over = y_floor * (1.0 + 1e-9);      // (the root's own rounding stays inside)
    key_lo = y_under < 9.5367431640625e-07 ? 1u : hall_key(y_under);                     // (2^-20: the first bin's lower edge)
    key_hi = y_over < 9.5367431640625e-07 ? 1u : hall_key(y_over) + 1u;
    const double edge = F < 0.0 ? -F / u * kLoglikBand : 0.0;                              // y* to the band's edge (<= 8/15)
    if (edge >= 9.5367431640625e-07) {                                                     // (below the first bin: nothing to keep from the moments)
      const uint32_t k_edge = hall_key(edge);
      const uint32_t from = hall_centre(k_edge) < edge ? k_edge + 1u : k_edge;
      key_hi = from > key_hi ? from : key_hi;
    }
  };
  // what is kept in cell_y: every cell of the genome in the bins [kept_lo, kept_hi), kept_n of them (kept_hi == 0: nothing)
  uint32_t kept_lo = 0, kept_hi = 0, kept_n = 0;

  // F: the point; [reach_lo, reach_hi]: the stretch of F the search may ask about next (the simplex and its reflection)
  auto objective = [&](double F, double reach_lo, double reach_hi) -> double {
    const double u = 1.0 - F;
    uint32_t key_lo, key_hi, wide_lo, wide_hi, unused;
    band_of(F, key_lo, key_hi);
    band_of(reach_hi, wide_lo, unused);                                                    // (the floor falls as F grows ...
    band_of(reach_lo, unused, wide_hi);                                                    //  ... and y* with it)
    wide_lo = wide_lo < key_lo ? wide_lo : key_lo;
    wide_hi = wide_hi > key_hi ? wide_hi : key_hi;
    double sum = 0.0, band_cells = 0.0, wide_cells = 0.0;
    for (uint32_t i = threadIdx.x; i < n_used; i += kBlock) {
      const uint32_t key = bin_keys[i];
      const double m0 = bin_values[i];
      if (key >= key_hi) {
        const double c = hall_centre(key);
        const double A = __builtin_fma(u, c, F), t = u / A;
        const double m1 = bin_values[n_used + i], m2 = bin_values[2 * n_used + i], m3 = bin_values[3 * n_used + i], m4 = bin_values[4 * n_used + i];
        sum += __builtin_fma(m0, log(c * A), bin_values[5 * n_used + i]) + t * (m1 - t * (m2 - t * (m3 - t * m4)));
      } else if (key < key_lo) {
        sum += m0 * kLogSmallProb;
      } else {
        band_cells += m0;
      }
      if (key >= wide_lo && key < wide_hi) wide_cells += m0;
    }
    block_sum3(sum, band_cells, wide_cells);
    double value = sum;
    if (stats && threadIdx.x == 0) { atomicAdd(stats + 0, 1ull); if (band_cells > 0.0) atomicAdd(stats + 1, 1ull); }
    // Each cell of the band is clamp(fma(F, y - y*y, y*y)), the table pass's own arithmetic (k_eval_entries<2>,
    // k_inbreed_eval_lut<2>), multiplied into a running product whose exponent is peeled off as it goes: one log per thread.
    if (band_cells > 0.0) {
      if (key_hi > classes.block_bins || wide_hi > classes.block_bins) hand_over = 2u;      // (cannot happen: kLoglikReach)
      double prod = 1.0;
      int expo = 0;
      auto exact_cell = [&](double y) {
        const double yy = y * y, d = y - yy;
        const double p = __builtin_fmin(__builtin_fmax(__builtin_fma(F, d, yy), 0.0), 1.0);
        prod *= __builtin_fmax(p, 1e-10);
      };
      auto peel = [&]() {                                                                  // (after at most 8 factors >= 1e-10)
        expo += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
      };
      // the kept cells, where they cover the band: LDS alone
      const bool kept_covers = kept_hi != 0u && kept_lo <= key_lo && key_hi <= kept_hi;
      // else gather: the wider stretch if its cells fit the list (then it is kept), otherwise the band itself, a listful at a time
      const bool keep = !kept_covers && wide_cells <= static_cast<double>(kLoglikListCells);
      const uint32_t from = keep ? wide_lo : key_lo, to = keep ? wide_hi : key_hi;
      uint32_t band_blocks = 0;
      if (!kept_covers)
        for (uint32_t k = 0; k < classes.n; ++k) band_blocks += classes.of[k].bin_block[to] - classes.of[k].bin_block[from];
      const bool dense = !kept_covers && !keep && band_cells >= classes.dense * static_cast<double>(band_blocks) * kHallBlockLoci;
      auto walk_listed = [&](uint32_t n_listed) {                                          // the list's cells that lie in the band
        for (uint32_t i0 = 0; i0 < n_listed; i0 += 8 * kBlock) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const uint32_t i = i0 + static_cast<uint32_t>(q) * kBlock + threadIdx.x;
            if (i < n_listed) {
              const double y = cell_y[i];
              const uint32_t key = hall_key(y);
              if (key >= key_lo && key < key_hi) exact_cell(y);
            }
          }
          peel();
        }
      };
      if (stats && threadIdx.x == 0) {
        atomicAdd(stats + (kept_covers ? 2 : dense ? 5 : keep ? 3 : 4), 1ull);
        if (!kept_covers) atomicAdd(stats + 7, static_cast<unsigned long long>(band_blocks));
      }
      if (kept_covers) {
        walk_listed(kept_n);
      } else if (dense) {
        for (uint32_t k = 0; k < classes.n && !hand_over; ++k) {
          const LoglikClass& cl = classes.of[k];
          const uint32_t b0 = cl.bin_block[key_lo], b1 = cl.bin_block[key_hi];
          // slot by slot: a wave takes every fourth stretch of four blocks, a lane a slot of each
          for (uint32_t b = b0 + 4u * wave; b < b1; b += 16u) {
            unsigned long long word[4];
            double y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bool in = b + q < b1;
              word[q] = in ? cl.words[classes.plain_words ? g * classes.word_blocks + (b + q) : hall_word_index(b + q, g, lanes)] : 0ull;
              y[q] = cl.ys[static_cast<uint64_t>(in ? b + q : b0) * kHallBlockLoci + lane];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if ((word[q] >> (63u - lane)) & 1ull) exact_cell(y[q]);
            peel();
          }
        }
      } else {
        // sparse: rounds of 1024 blocks; a round's set bits are dealt list positions by a scan, their frequencies gathered there
        uint32_t listed = 0;
        kept_hi = 0u;                                                                      // (the list is being rewritten)
        // A class's rounds list SLOTS (in the low half of the list's entries); the slots' frequencies take their place -- all threads at
        // once, an entry each -- when the class is through or the list is full: one wait for the frequencies a class, not one a round
        // (and none per cell: a thread that fetched the frequencies of its own words' bits one after the other waited out a load each).
        // The next round's words are under way while this round's bits are counted and listed.
        uint32_t* const cell_slot = reinterpret_cast<uint32_t*>(cell_y);
        for (uint32_t k = 0; k < classes.n && !hand_over; ++k) {
          const double* const class_ys = classes.of[k].ys;                                   // (the class's members once, as scalars)
          const unsigned long long* const class_words = classes.of[k].words;
          const uint32_t b0 = classes.of[k].bin_block[from], b1 = classes.of[k].bin_block[to];
          uint32_t converted = listed;                                                     // entries below hold frequencies, [converted, listed) slots
          auto fetch_frequencies = [&]() {                                                  // (after a barrier: the slots are all written)
            for (uint32_t i = converted + threadIdx.x; i < listed; i += kBlock) cell_y[i] = class_ys[cell_slot[2u * i]];
            converted = listed;
          };
          auto load_words = [&](uint32_t base, unsigned long long (&word)[kLoglikWordsPerThread]) {
#pragma unroll
            for (int w = 0; w < kLoglikWordsPerThread; ++w) {
              const uint32_t b = base + static_cast<uint32_t>(w) * kBlock + threadIdx.x;
              word[w] = b < b1 ? class_words[classes.plain_words ? g * classes.word_blocks + b : hall_word_index(b, g, lanes)] : 0ull;
            }
          };
          unsigned long long next_word[kLoglikWordsPerThread];
          if (b0 < b1) load_words(b0, next_word);
          for (uint32_t base = b0; base < b1; base += kBlock * kLoglikWordsPerThread) {
            unsigned long long word[kLoglikWordsPerThread];
            uint32_t mine = 0;
#pragma unroll
            for (int w = 0; w < kLoglikWordsPerThread; ++w) word[w] = next_word[w];
            if (base + kBlock * kLoglikWordsPerThread < b1) load_words(base + kBlock * kLoglikWordsPerThread, next_word);
#pragma unroll
            for (int w = 0; w < kLoglikWordsPerThread; ++w) mine += static_cast<uint32_t>(__popcll(word[w]));
            uint32_t round_total = 0;
            uint32_t at = listed + block_scan(mine, round_total);
            bool to_list = true;
            if (listed + round_total > kLoglikListCells) {
              // the list is full (a band gathered a listful at a time, nothing kept): walk what it holds, start it again
              if (keep) { hand_over = 4u; break; }                                          // (a kept stretch fits by the moments' own count)
              __syncthreads();
              fetch_frequencies();
              __syncthreads();
              walk_listed(listed);
              __syncthreads();
              at -= listed;
              listed = 0;
              converted = 0;
              to_list = round_total <= kLoglikListCells;       // a round denser than the list holds: every thread walks its own blocks' cells
            }
#pragma unroll
            for (int w = 0; w < kLoglikWordsPerThread; ++w) {
              const uint32_t first = (base + static_cast<uint32_t>(w) * kBlock + threadIdx.x) * kHallBlockLoci;
              unsigned long long bits = word[w];
              int walked = 0;
              while (bits != 0ull) {
                const int p = __clzll(static_cast<long long>(bits));
                bits &= ~(0x8000000000000000ull >> p);
                if (to_list) cell_slot[2u * at++] = first + static_cast<uint32_t>(p);
                else { exact_cell(class_ys[first + static_cast<uint32_t>(p)]); if ((++walked & 7) == 0) peel(); }
              }
              if (!to_list) peel();
            }
            if (to_list) listed += round_total;
          }
          __syncthreads();                                                                 // the class's slots are listed
          fetch_frequencies();
        }
        __syncthreads();                                                                   // the list is complete
        if (stats && threadIdx.x == 0) atomicAdd(stats + 6, static_cast<unsigned long long>(listed));
        if (keep && !hand_over) {
          if (static_cast<double>(listed) != wide_cells) hand_over = 4u;                   // the bits and the moments count the same cells
          kept_lo = from; kept_hi = to; kept_n = listed;
        }
        walk_listed(listed);
        if (!keep) __syncthreads();                                                        // (a list not kept may be rewritten by the next evaluation at once)
      }
      double exact = log(prod) + static_cast<double>(expo) * 0.6931471805599453, nothing = 0.0, none = 0.0;
      block_sum3(exact, nothing, none);
      value += exact;
    }
    // heterozygous cells
    value += het_tiny * kLogSmallProb;
    if (het_cells > 0.0) {
      if (u * w_smallest >= 1.0000001e-10) value += het_term + het_cells * log(u);
      else if (u <= 2e-10) value += het_cells * kLogSmallProb;                             // w <= 1/2: every one under the floor
      else hand_over = 3u;
    }
    return value;
  };

  if (mode == 1) {
    const double v = objective(start[g], start[g], start[g]);
    if (threadIdx.x == 0) {
      f_out[g] = hand_over ? __builtin_nan("") : v;
      needs_passes[g] = hand_over;
      if (hand_over) atomicAdd(handed_over, 1u);
    }
    return;
  }
  BrentState s = nm_start(start[g]);
  unsigned int evaluations = 0;
  for (int it = 0; it < 500 && !hand_over; ++it) {
    const double F = it == 0 ? s.x : s.u;
    // what the search will mostly ask about next: the simplex (best a, worst b) -- 97 % of its steps end in a contraction into it
    // (a reflection past it finds the kept cells too narrow and gathers again)
    const double lo = s.a < s.b ? s.a : s.b, hi = s.a < s.b ? s.b : s.a;
    const double value = objective(F, lo < F ? lo : F, hi > F ? hi : F);
    ++evaluations;
    if (hand_over) break;
    nm_advance(s, value);
    if (s.done) break;                                        // uniform over the genome's threads: all hold the same state
  }
  if (threadIdx.x == 0) {
    f_out[g] = hand_over ? __builtin_nan("") : s.x;
    needs_passes[g] = hand_over;
    if (hand_over) atomicAdd(handed_over, 1u);
    atomicMax(max_evaluations, evaluations);
  }
}

// After the search on the moments, before the passes for the genomes it handed over: every other genome's state says
// "done" with the coefficient it has (k_brent_init has just written all states and start points).
__global__ void __launch_bounds__(kBlock)
k_loglik_keep(const uint32_t* __restrict__ needs_passes, const double* __restrict__ kept, uint64_t n, int planes, BrentState* __restrict__ st,
              double* __restrict__ f_next) {
  for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n; g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    if (needs_passes[g]) continue;
    BrentState s = st[g];
    s.done = 1;
    s.x = kept[g];
    st[g] = s;
    for (int k = 0; k < planes; ++k) f_next[k * n + g] = kept[g];
  }
}

}  // namespace kgx
