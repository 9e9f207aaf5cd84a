// The INBREED package's own regime -- a window of ~1000 sampled loci x the genomes of one super population, thousands of
// times per contig (kga_analysis_inbreed_diploid.cpp:53-75,98-166) -- as ONE launch for MANY (window, super population)
// tasks (driven from kgx_inbreed.hip: kgx_inbreed_batch).  A task is what InbreedingAnalysis::processResults fans out
// over the pool: a genome range, its locus list, its allele-frequency rows.  THREADS threads own a (task, genome): thread t
// loads the cells of loci t, t + THREADS, ..., classifies each with the generic kernel's own classify_cell
// (generateFrequencies, _freq.cpp:425-583), keeps what the estimator needs of it in registers, and the genome's class counts,
// class-frequency sums, RitlandLocus terms and -- HallME, Loglikelihood -- the whole iteration / search (as in
// k_inbreed_iterate_genome) come out of block-wide sums whose bits are the same in every thread.  One kernel (after
// k_locus_tables over the tasks' concatenated frequency rows) where a kgx_inbreed call makes six to ten.
#pragma once

#include <cstdint>

#include "../../include/kgx.h"
#include "kgx_kernels_inbreed.h"

namespace kgx {

// One (window, super population): genomes [g0, g0 + n_genomes) of the shard, selected loci [locus_base, locus_base + n_sel) of
// the batch's concatenated index / table, results and start points from genome_base on in the batch's concatenated arrays.
struct WindowTask { uint64_t g0; uint32_t n_genomes, n_sel, locus_base, genome_base; };

// ALGO: KGX_ALGO_* (0 RitlandLocus, 1 Simple, 2 HallME, 3 Loglikelihood).  Grid: x = ceil(largest task's genomes / (kBlock /
// THREADS)), y = task.  CELLS * THREADS >= the largest n_sel of the batch.
template <int ALGO, int CELLS, int THREADS>
__global__ void __launch_bounds__(kBlock)
k_inbreed_window(const uint8_t* __restrict__ gt, uint64_t pitch, const WindowTask* __restrict__ tasks, const uint32_t* __restrict__ locus_index,
                 const double* __restrict__ table, const uint8_t* __restrict__ valid, uint32_t amax, int phased, int search,
                 const double* __restrict__ start, LocusResultsDev* __restrict__ out, unsigned int* __restrict__ max_evaluations) {
  static_assert(THREADS == kWave || (THREADS == kBlock && kBlock == 256), "a wave or a block of sixteen rows per genome");
  constexpr int kSums = 5;                                     // majorHom, majorHet, minorHom, minorHet class-frequency sums, Ritland sum
  __shared__ double row_part[2][kSums][16];
  __shared__ unsigned long long row_counts[2][16];
  const WindowTask task = tasks[blockIdx.y];
  const uint32_t t = threadIdx.x % THREADS;
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * (kBlock / THREADS) + threadIdx.x / THREADS;
  if (g >= task.n_genomes) return;                            // whole waves / whole blocks only: no barrier is left waiting
  // the sums over the genome's threads, the same bits in every thread; `pass` alternates the LDS slots
  auto block_sum = [&](double v, int pass) {
    v = row_sum16(v);
    if constexpr (THREADS == kWave) {
      return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
    } else {
      if ((threadIdx.x & 15) == 0) row_part[pass & 1][0][threadIdx.x >> 4] = v;
      __syncthreads();
      const double* p = row_part[pass & 1][0];
      double pair[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pair[i] = p[2 * i] + p[2 * i + 1];
      return ((pair[0] + pair[1]) + (pair[2] + pair[3])) + ((pair[4] + pair[5]) + (pair[6] + pair[7]));
    }
  };
  const uint32_t stride = sweep_stride(amax);
  const uint32_t* __restrict__ index = locus_index + task.locus_base;
  const double* __restrict__ rows = table + static_cast<uint64_t>(task.locus_base) * stride;
  const uint8_t* __restrict__ flags = valid + task.locus_base;
  const uint8_t* __restrict__ column = gt + task.g0 + g;

  // generateFrequencies (_freq.cpp:425-583): the cell's class, its share of the five sums; the iterative estimators keep (y, d)
  double y[(ALGO >= 2) ? CELLS : 1], d[(ALGO >= 2) ? CELLS : 1];
  double sums[kSums] = {0.0, 0.0, 0.0, 0.0, 0.0};
  unsigned long long counted = 0;                             // majorHom | majorHet << 14 | minorHom << 28 | minorHet << 42 | Ritland terms << 56 (<= 8192 each; the last in 8 bits: see below)
  uint32_t ritland_terms = 0;
#pragma unroll
  for (int c = 0; c < CELLS; ++c) {
    if constexpr (ALGO >= 2) { y[c] = ALGO == 3 ? 1.0 : 0.0; d[c] = 0.0; }       // contributes nothing
    const uint32_t s = static_cast<uint32_t>(c) * THREADS + t;
    if (s < task.n_sel && (flags[s] & kLocusValid)) {
      const double* row = rows + static_cast<uint64_t>(s) * stride;
      double f1 = 0.0, f2 = 0.0;
      const int cls = classify_cell(column[static_cast<uint64_t>(index[s]) * pitch], row, amax, phased != 0, f1, f2);
      if (cls != kClassNone) {
        counted += 1ull << (14 * (cls - 1));
        sums[0] += row[amax + 1]; sums[1] += row[amax + 2]; sums[2] += row[amax + 3]; sums[3] += row[amax + 4];
        const bool homozygous = cls == kMajorHom || cls == kMinorHom;
        if constexpr (ALGO == KGX_ALGO_RITLAND_LOCUS) {
          if (homozygous) {
            if (f1 > 0.001) { sums[4] += 1.0 / f1; sums[4] -= 1.0; ++ritland_terms; }     // minimum_frequency (_calc.cpp:380,396)
          } else {
            sums[4] -= 1.0;
            ++ritland_terms;
          }
        } else if constexpr (ALGO == KGX_ALGO_HALL_ME) {
          if (homozygous) { y[c] = f1; d[c] = 1.0; }            // d = 1 marks a homozygous cell, y its allele frequency
        } else if constexpr (ALGO == KGX_ALGO_LOGLIKELIHOOD) {
          if (homozygous) { y[c] = f1 * f1; d[c] = f1 - y[c]; }
          else { y[c] = 2.0 * f1 * f2; d[c] = -y[c]; }
        }
      }
    }
  }
  // the genome's counts and sums: integers exact in any order; the fp64 sums a fixed tree (the reference adds them locus by
  // locus: at <= 8192 loci the two differ by a few 1e-16 of the sum)
  unsigned long long counts_all = counted;
  uint32_t ritland_all = ritland_terms;
  {
    // integer sums over the genome's threads: DPP-free, through the lanes' shuffles (once per genome)
    for (int off = 1; off < kWave; off <<= 1) {
      counts_all += __shfl_xor(counts_all, off);
      ritland_all += __shfl_xor(ritland_all, off);
    }
    if constexpr (THREADS == kBlock) {
      if ((threadIdx.x & (kWave - 1)) == 0) row_counts[0][threadIdx.x >> 6] = counts_all, row_counts[1][threadIdx.x >> 6] = ritland_all;
      __syncthreads();
      counts_all = (row_counts[0][0] + row_counts[0][1]) + (row_counts[0][2] + row_counts[0][3]);
      ritland_all = static_cast<uint32_t>((row_counts[1][0] + row_counts[1][1]) + (row_counts[1][2] + row_counts[1][3]));
    }
  }
  double total_sums[kSums];
  if constexpr (THREADS == kWave) {
#pragma unroll
    for (int k = 0; k < kSums; ++k) total_sums[k] = (ALGO == KGX_ALGO_RITLAND_LOCUS || k < 4) ? block_sum(sums[k], 0) : 0.0;
  } else {
#pragma unroll
    for (int k = 0; k < kSums; ++k) {
      const double v = row_sum16(sums[k]);
      if ((threadIdx.x & 15) == 0) row_part[1][k][threadIdx.x >> 4] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSums; ++k) {
      const double* p = row_part[1][k];
      double pair[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pair[i] = p[2 * i] + p[2 * i + 1];
      total_sums[k] = ((pair[0] + pair[1]) + (pair[2] + pair[3])) + ((pair[4] + pair[5]) + (pair[6] + pair[7]));
    }
    __syncthreads();                                           // (row_part[1] is the iteration's second slot)
  }
  const unsigned long long major_hom = counts_all & 0x3FFFull, major_het = (counts_all >> 14) & 0x3FFFull,
                           minor_hom = (counts_all >> 28) & 0x3FFFull, minor_het = (counts_all >> 42) & 0x3FFFull;
  const unsigned long long total = major_hom + major_het + minor_hom + minor_het;

  double coefficient = 0.0;
  unsigned int evaluations = 0;
  if constexpr (ALGO == KGX_ALGO_SIMPLE) {
    // processSimple (_calc.cpp:318-365)
    if (total > 0) {
      const double observed_homozygous = static_cast<double>(minor_hom + major_hom);
      const double expected_homozygous = total_sums[2] + total_sums[0];
      coefficient = (observed_homozygous - expected_homozygous) / (static_cast<double>(total) - expected_homozygous);
    }
  } else if constexpr (ALGO == KGX_ALGO_RITLAND_LOCUS) {
    coefficient = ritland_all > 0 ? total_sums[4] / static_cast<double>(ritland_all) : 0.0;
  } else if constexpr (ALGO == KGX_ALGO_HALL_ME) {
    // processHallME (_calc.cpp:255-285): 50 expectation steps from the genome's start point (see kgx_inbreed)
    const double n_cells = static_cast<double>(total);
    double F = start[task.genome_base + g];
    for (int it = 0; it < 50; ++it) {
      // (selects, not branches: the quotients of different cells are independent and overlap)
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < CELLS; ++c) {
        const double denominator = F + ((1.0 - F) * y[c]);
        // F / denominator by a reciprocal and two Newton steps (half the instructions of the IEEE division, within an ulp of it:
        // a batch of windows spends its time here -- 50 steps x 16 cells a lane)
        double r = __builtin_amdgcn_rcp(denominator);
        r = __builtin_fma(r, __builtin_fma(-denominator, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-denominator, r, 1.0), r);
        sum += (d[c] != 0.0 && denominator != 0) ? F * r : 0.0;
      }
      F = block_sum(sum, it) / n_cells;
    }
    coefficient = F;
  } else {
    BrentState s = search == kSearchNelderMead ? nm_start(start[task.genome_base + g]) : brent_start(nullptr, nullptr);
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
    coefficient = s.x;
  }
  if (t == 0) {
    LocusResultsDev r;
    r.major_homo_count = major_hom; r.major_hetero_count = major_het; r.minor_homo_count = minor_hom; r.minor_hetero_count = minor_het;
    r.total_allele_count = total;
    r.major_homo_freq = total_sums[0]; r.major_hetero_freq = total_sums[1]; r.minor_homo_freq = total_sums[2]; r.minor_hetero_freq = total_sums[3];
    r.inbred_allele_sum = coefficient;
    out[task.genome_base + g] = r;
    if constexpr (ALGO == KGX_ALGO_LOGLIKELIHOOD) atomicMax(max_evaluations, evaluations);
  }
}

}  // namespace kgx
