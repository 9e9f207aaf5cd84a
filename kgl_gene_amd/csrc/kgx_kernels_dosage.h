// HIP kernels for gfx950 (MI355X, CDNA4): the 2-bit dosage sweeps (K2/K3/K4/K8).  Integer counting over 2-bit
// dosage rows: HBM-bound, no MFMA.  Included only by kgx_dosage.hip.
//
// HBM layout ("dosage2"): variant-major rows, row v at rows + v*pitch, pitch = multiple of 16 B.
// Genome g of the shard sits in bits 2*(g%4) of byte g/4; bits past n_genomes are zero.  One
// 16-byte chunk = 64 genomes = one lane-load; a wave64 load instruction moves 1 KiB of one or more
// adjacent rows, fully coalesced.
#ifndef KGX_KERNELS_DOSAGE_H
#define KGX_KERNELS_DOSAGE_H

#include "kgx_kernels_common.h"
#include "kgx_synth.h"

namespace kgx {
// ---------------------------------------------------------------------------------------------
// DPP lane-group sums.  After group_sum<W>(x) the LAST lane of every aligned W-lane group holds
// the sum over that group (for W <= 16 every lane of the group does).
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x) {
  // x + lane-permuted x; lanes with no source (bound_ctrl) add 0.
  return x + static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xF, true));
}

template <int W>
__device__ __forceinline__ uint32_t group_sum(uint32_t x) {
  if constexpr (W >= 2) x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]  : xor 1
  if constexpr (W >= 4) x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]  : xor 2
  if constexpr (W >= 8) x = dpp_add<0x141>(x);   // row_half_mirror      : 8-lane sums
  if constexpr (W >= 16) x = dpp_add<0x140>(x);  // row_mirror           : 16-lane (row) sums
  if constexpr (W >= 32) x = dpp_add<0x142, 0xA>(x);  // row_bcast15 into rows 1,3 : 32-lane sums in rows 1,3
  if constexpr (W >= 64) x = dpp_add<0x143, 0xC>(x);  // row_bcast31 into rows 2,3 : wave sum in row 3
  return x;
}

// Three running popcounts of one 16-byte chunk (64 genomes):
//   a += #(code & 1)  = het + nondiploid,  b += #(code & 2) = hom + nondiploid,  c += #(code == 3).
__device__ __forceinline__ void count_chunk(const kgx_v4u x, uint32_t& a, uint32_t& b, uint32_t& c) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = x[i];
    a += __builtin_popcount(w & 0x55555555u);
    b += __builtin_popcount(w & 0xAAAAAAAAu);
    c += __builtin_popcount(w & (w >> 1) & 0x55555555u);
  }
}

// ---------------------------------------------------------------------------------------------
// K2  allele_count_by_locus: VariantDBVariant::summaryByVariant for every variant row
// (kgl_variant_db_variant.cpp:126-178).  W lanes cooperate on one row, 64/W rows per wave, U such
// row sets in flight per wave (U independent 16-B loads per lane per step).
// out[v] = { refHom, het, minorHom, nonDiploid }.
// ---------------------------------------------------------------------------------------------
// MASKED: `keep` is one row in the rows' layout, 0b11 where a genome takes part (a genome filter of the reference's
// kind, PopulationDB::viewFilter(GenomeListFilter)); the genomes left out count as absent and n_genomes is the number kept.
template <int W, int U, bool NT = true, bool MASKED = false>
__global__ void __launch_bounds__(kBlock)
k_allele_count(const kgx_v4u* __restrict__ rows, uint32_t chunks_per_row, uint64_t n_rows,
               uint32_t n_genomes, kgx_v4u* __restrict__ out, const kgx_v4u* __restrict__ keep = nullptr) {
  constexpr int kRowsPerSet = kWave / W;
  constexpr int kRowsPerIter = kRowsPerSet * U;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t sub = lane & (W - 1);
  const uint32_t grp = lane / W;
  const uint64_t waves_per_block = blockDim.x / kWave;
  const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * waves_per_block + threadIdx.x / kWave;
  const uint64_t n_waves = static_cast<uint64_t>(gridDim.x) * waves_per_block;

  for (uint64_t base = wave * kRowsPerIter; base < n_rows; base += n_waves * kRowsPerIter) {
    uint32_t a[U], b[U], c[U];
    const kgx_v4u* rp[U];
    uint64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = b[u] = c[u] = 0;
      row[u] = base + static_cast<uint64_t>(u) * kRowsPerSet + grp;
      const uint64_t r = row[u] < n_rows ? row[u] : n_rows - 1;   // clamp: tail lanes re-read the last row
      rp[u] = rows + r * chunks_per_row;
    }
    for (uint32_t k = sub; k < chunks_per_row; k += W) {
      kgx_v4u x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(rp[u] + k) : rp[u][k];
      if constexpr (MASKED) {
        const kgx_v4u m = keep[k];                                  // one row for every row set: stays in cache
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] &= m;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) count_chunk(x[u], a[u], b[u], c[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t sa = group_sum<W>(a[u]);
      const uint32_t sb = group_sum<W>(b[u]);
      const uint32_t sc = group_sum<W>(c[u]);
      if (sub == W - 1 && row[u] < n_rows) {
        kgx_v4u r;
        r[0] = n_genomes - sa - sb + sc;   // reference homozygous (implicit in the reference's store)
        r[1] = sa - sc;                    // minor heterozygous
        r[2] = sb - sc;                    // minor homozygous
        r[3] = sc;                         // non-diploid, not counted by the reference
        out[row[u]] = r;
      }
    }
  }
}

// af[v] = (het + 2*hom) / (2*total_genomes), fp64 (IEEE divide: bit-identical to the host).
__global__ void __launch_bounds__(kBlock)
k_allele_frequency(const kgx_v4u* __restrict__ counts, uint64_t n, uint64_t total_genomes,
                   double* __restrict__ af) {
  const double denom = 2.0 * static_cast<double>(total_genomes);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const kgx_v4u c = counts[i];
    af[i] = static_cast<double>(static_cast<uint64_t>(c[1]) + 2u * static_cast<uint64_t>(c[2])) / denom;
  }
}

// Under a genome mask a variant nobody kept carries is not in the population: its row takes no part in the by-genome
// sweep (bin 0xFF).  counts = K2's output under the same mask.
__global__ void __launch_bounds__(kBlock)
k_drop_absent_rows(const kgx_v4u* __restrict__ counts, uint64_t n, uint8_t* __restrict__ bin_of_variant) {
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const kgx_v4u c = counts[i];
    if ((c[1] | c[2] | c[3]) == 0) bin_of_variant[i] = 0xFF;
  }
}

// K4  populationSummary (kgl_variant_db_variant.cpp:234-279) = column sums of the K2 output.
__global__ void __launch_bounds__(kBlock)
k_sum_counts(const kgx_v4u* __restrict__ counts, uint64_t n, unsigned long long* __restrict__ total4) {
  unsigned long long s[4] = {0, 0, 0, 0};
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const kgx_v4u c = counts[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += c[j];
  }
  __shared__ unsigned long long sh[4][kBlock / kWave];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned long long v = s[j];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sh[j][threadIdx.x / kWave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long v = 0;
    for (int w = 0; w < kBlock / kWave; ++w) v += sh[threadIdx.x][w];
    atomicAdd(&total4[threadIdx.x], v);
  }
}

// ---------------------------------------------------------------------------------------------
// K3  genotype_count_by_genome: VariantDBVariant::summaryByGenome for every genome
// (kgl_variant_db_variant.cpp:180-231) over a selected set of variant rows (the FWS allele-frequency
// bins, kga_analysis_PfEMP_FWS.cpp:15-38,72-101).
//
// Same rows, other reduction axis.  A lane owns one 16-byte chunk column (64 genomes) and walks
// down the selected rows; each of the 128 bits of the chunk is a 1-bit stream to be counted over
// rows ("positional popcount").  Streams are counted bit-sliced: a carry-save adder tree folds 8
// rows into weight-1/2/4 planes and ripples one weight-8 carry word into HI planes, ~6 VALU ops per
// input word instead of one add per genome.  Planes are unpacked into per-workgroup LDS counters
// every 510 blocks, and LDS goes to the accumulators with integer atomics once per (workgroup, bin): exact in any order.
//
// The accumulators keep the LDS image's own layout -- acc[bin][column group][counter][lane], uint32 (a count is at most
// the number of rows, < 2^32) -- so a wave's 64 atomics fall on 256 consecutive bytes instead of 64 lines 17 KB apart
// (genome-major uint64 accumulators: 500 MB of write traffic per C3 sweep for 3.5 MB of results).  A work item is a
// stretch of the bin-grouped row list for one column group and may run across bin boundaries (it flushes at each), so
// the list is cut into equal stretches, about one per resident workgroup, whatever the bins' sizes.
// Raw stream counts { #(code&1), #(code&2), #(code==3) }; k_finish_by_genome turns them into the reference's
// { refHom, het, minorHom, nonDiploid } per genome and bin.
// ---------------------------------------------------------------------------------------------
struct GenomeWork {
  uint64_t begin;       // first position in the selected-row list (rows grouped by bin)
  uint64_t end;         // one past the last position
  uint32_t col_group;   // chunk columns [col_group * cg_width, +cg_width)
  uint32_t pad;
};
constexpr int kAccCounters = 192;            // per (bin, column group): 128 stream counters + 64 non-diploid ones, x kLdsStride lanes

constexpr int kHiPlanes = 8;                 // weights 16 .. 2048
constexpr int kBlocksPerFlush = 510;         // blocks of 8 rows: 510*8 + 15 < 16 * 2^kHiPlanes
constexpr int kLdsStride = 64;               // lanes per counter row in LDS

__device__ __forceinline__ void csa(uint32_t& carry, uint32_t& sum, uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t u = a ^ b;
  carry = (a & b) | (u & c);
  sum = u ^ c;
}

template <int NW>
struct SlicedCounters {
  // bit planes of weight 1, 2, 4, 8 and 16 << p; pending = the weight-8 carry of an odd block, waiting for its partner
  uint32_t ones[NW], twos[NW], fours[NW], eights[NW], hi[NW][kHiPlanes], pending[NW];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      ones[i] = twos[i] = fours[i] = eights[i] = pending[i] = 0;
#pragma unroll
      for (int p = 0; p < kHiPlanes; ++p) hi[i][p] = 0;
    }
  }
  // Fold 8 input words of stream-word i; returns the carry of weight 8.
  __device__ __forceinline__ uint32_t add8(int i, const uint32_t x[8]) {
    uint32_t t2a, t2b, f4a, f4b, e8;
    csa(t2a, ones[i], ones[i], x[0], x[1]);
    csa(t2b, ones[i], ones[i], x[2], x[3]);
    csa(f4a, twos[i], twos[i], t2a, t2b);
    csa(t2a, ones[i], ones[i], x[4], x[5]);
    csa(t2b, ones[i], ones[i], x[6], x[7]);
    csa(f4b, twos[i], twos[i], t2a, t2b);
    csa(e8, fours[i], fours[i], f4a, f4b);
    return e8;
  }
  // Two weight-8 carries (of two blocks of 8 rows) into the weight-8 plane; its carry ripples through the planes above --
  // once per 16 rows instead of once per 8.
  __device__ __forceinline__ void add_eights(int i, uint32_t a, uint32_t b) {
    uint32_t c16;
    csa(c16, eights[i], eights[i], a, b);
#pragma unroll
    for (int p = 0; p < kHiPlanes; ++p) {
      const uint32_t t = hi[i][p] & c16;
      hi[i][p] ^= c16;
      c16 = t;
    }
  }
  // Integer count of bit position b of stream-word i.
  __device__ __forceinline__ uint32_t value(int i, int b) const {
    uint32_t v = ((ones[i] >> b) & 1u) | (((twos[i] >> b) & 1u) << 1) | (((fours[i] >> b) & 1u) << 2) | (((eights[i] >> b) & 1u) << 3);
#pragma unroll
    for (int p = 0; p < kHiPlanes; ++p) v |= ((hi[i][p] >> b) & 1u) << (4 + p);
    return v;
  }
};

// MODE 0: stream words are the 4 chunk dwords (bit 2j = code&1, bit 2j+1 = code&2 of genome j).
// MODE 1: two words of (code==3) indicators: word0 = t(dw0) | t(dw1)<<1, word1 = t(dw2) | t(dw3)<<1.
template <int W, int MODE>
__device__ __forceinline__ void by_genome_pass(const kgx_v4u* __restrict__ rows, uint32_t chunks_per_row,
                                               const uint32_t* __restrict__ row_index, uint32_t cg_width,
                                               const GenomeWork wk, uint32_t* lds, uint32_t& saw_nondiploid) {
  constexpr int NW = MODE == 0 ? 4 : 2;
  constexpr int kRowSlots = (kWave / W) * (kBlock / kWave);   // rows visited per step by the workgroup
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t sub = lane & (W - 1);
  const uint32_t slot = (threadIdx.x / kWave) * (kWave / W) + lane / W;
  const uint32_t col = wk.col_group * cg_width + sub;
  const bool col_ok = sub < cg_width && col < chunks_per_row;

  SlicedCounters<NW> cnt;
  cnt.clear();
  int blocks = 0;
  uint32_t seen = 0;

  auto flush = [&]() {
    if (blocks & 1) {                       // an odd block's carries still wait for a partner
#pragma unroll
      for (int i = 0; i < NW; ++i) cnt.add_eights(i, cnt.pending[i], 0u);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
#pragma unroll
      for (int b = 0; b < 32; ++b) {
        const uint32_t v = cnt.value(i, b);
        if (v) atomicAdd(&lds[(i * 32 + b) * kLdsStride + sub], v);
      }
    }
    cnt.clear();
    blocks = 0;
  };

  // Gathered rows (the bins) with one row per wave-step slot (W == 64): a wave takes 8 CONSECUTIVE list positions per
  // step, so their row numbers are one wave-uniform 32-byte scalar load -- on the scalar memory counter, where it does
  // not queue behind the row loads already in flight as a per-lane index load would (vector memory returns in order).
  // row_index is padded by 8 entries past the list.
  const bool scalar_index = (W == kWave) && row_index != nullptr;
  const uint32_t wave_slot = __builtin_amdgcn_readfirstlane(slot);
  for (uint64_t p0 = wk.begin + slot; p0 - slot < wk.end; p0 += static_cast<uint64_t>(kRowSlots) * 8) {
    kgx_v4u x[8];
    if (scalar_index) {
      const uint64_t base = (p0 - slot) + static_cast<uint64_t>(wave_slot) * 8;     // wave-uniform
      uint32_t r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = row_index[base + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kgx_v4u v = {0u, 0u, 0u, 0u};
        if (base + j < wk.end && col_ok) v = __builtin_nontemporal_load(rows + static_cast<uint64_t>(r[j]) * chunks_per_row + col);
        x[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint64_t p = p0 + static_cast<uint64_t>(j) * kRowSlots;
        kgx_v4u v = {0u, 0u, 0u, 0u};
        if (p < wk.end && col_ok) {
          const uint64_t r = row_index ? static_cast<uint64_t>(row_index[p]) : p;
          v = __builtin_nontemporal_load(rows + r * chunks_per_row + col);
        }
        x[j] = v;
      }
    }
    uint32_t e8[NW];
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          w[j] = x[j][i];
          seen |= w[j] & (w[j] >> 1);       // code 3 = both bits of a field; the odd positions are masked out at the end
        }
        e8[i] = cnt.add8(i, w);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t d0 = x[j][2 * i], d1 = x[j][2 * i + 1];
          w[j] = (d0 & (d0 >> 1) & 0x55555555u) | ((d1 & (d1 >> 1) & 0x55555555u) << 1);
        }
        e8[i] = cnt.add8(i, w);
      }
    }
    if (blocks & 1) {                       // (block-uniform) the second block of a pair: both carries go up together
#pragma unroll
      for (int i = 0; i < NW; ++i) cnt.add_eights(i, cnt.pending[i], e8[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NW; ++i) cnt.pending[i] = e8[i];
    }
    if (++blocks == kBlocksPerFlush) flush();
  }
  flush();
  saw_nondiploid |= seen & 0x55555555u;
}

// PLANE: the rows are a 1-bit-per-genome matrix (the phase plane): every bit is a stream of its own, there is no
// "non-diploid" second pass.
template <int W, bool PLANE = false>
__global__ void __launch_bounds__(kBlock)
k_count_by_genome(const kgx_v4u* __restrict__ rows, uint32_t chunks_per_row, uint32_t cg_width,
                  const uint32_t* __restrict__ row_index, const GenomeWork* __restrict__ work,
                  const unsigned long long* __restrict__ bin_offset /* [n_bins + 1] */, uint32_t n_bins, uint32_t n_cg,
                  uint32_t* __restrict__ acc /* [n_bins][n_cg][kAccCounters][kLdsStride] */) {
  __shared__ uint32_t lds[128 * kLdsStride];
  const GenomeWork item = work[blockIdx.x];
  uint32_t bin = 0;
  for (uint64_t p = item.begin; p < item.end;) {
    while (bin + 1 < n_bins && bin_offset[bin + 1] <= p) ++bin;            // the bin of position p (empty bins skipped)
    const uint64_t stop = item.end < bin_offset[bin + 1] ? item.end : bin_offset[bin + 1];
    GenomeWork wk = item;
    wk.begin = p;
    wk.end = stop;
    p = stop;
    uint32_t* mine = acc + (static_cast<uint64_t>(bin) * n_cg + wk.col_group) * (kAccCounters * kLdsStride);

    for (int i = threadIdx.x; i < 128 * kLdsStride; i += kBlock) lds[i] = 0;
    __syncthreads();
    uint32_t seen = 0;
    by_genome_pass<W, 0>(rows, chunks_per_row, row_index, cg_width, wk, lds, seen);
    __syncthreads();
    // LDS counter (i*32+b, sub): genome = (col_group * cg_width + sub)*64 + i*16 + b/2, stream = b&1 (k_finish_by_genome)
    for (int idx = threadIdx.x; idx < 128 * kLdsStride; idx += kBlock) {
      const uint32_t v = lds[idx];
      if (v) atomicAdd(&mine[idx], v);
    }
    // Non-diploid codes are exceptional: count them in a second pass only if this workgroup saw one.
    if (!PLANE && __syncthreads_or(seen != 0)) {
      for (int i = threadIdx.x; i < 64 * kLdsStride; i += kBlock) lds[i] = 0;
      __syncthreads();
      uint32_t unused = 0;
      by_genome_pass<W, 1>(rows, chunks_per_row, row_index, cg_width, wk, lds, unused);
      __syncthreads();
      // LDS counter (i*32+b, sub): chunk dword = 2*i + (b&1), genome within dword = b/2.
      for (int idx = threadIdx.x; idx < 64 * kLdsStride; idx += kBlock) {
        const uint32_t v = lds[idx];
        if (v) atomicAdd(&mine[128 * kLdsStride + idx], v);
      }
    }
    __syncthreads();                                                        // the LDS counters are cleared for the next bin
  }
}

// acc {a,b,c} -> { refHom = n_rows(bin) - a - b + c, het = a - c, minorHom = b - c, nonDiploid = c }.
__global__ void __launch_bounds__(kBlock)
k_finish_by_genome(const uint32_t* __restrict__ acc, const unsigned long long* __restrict__ rows_in_bin,
                   uint64_t n_genomes, uint32_t n_bins, uint32_t n_cg, uint32_t cg_width, unsigned long long* __restrict__ out) {
  const uint64_t total = n_genomes * n_bins;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t g = i / n_bins;
    const uint32_t bin = static_cast<uint32_t>(i % n_bins);
    const uint32_t chunk = static_cast<uint32_t>(g / 64), j = static_cast<uint32_t>(g % 64);   // the genome's 16-byte chunk, its place in it
    const uint32_t cg = chunk / cg_width, sub = chunk % cg_width;
    const uint32_t dw = j / 16, jj = j % 16;                               // the chunk's dword, the 2-bit field in it
    const uint32_t* mine = acc + (static_cast<uint64_t>(bin) * n_cg + cg) * (kAccCounters * kLdsStride);
    const unsigned long long a = mine[(dw * 32 + 2 * jj) * kLdsStride + sub];
    const unsigned long long b = mine[(dw * 32 + 2 * jj + 1) * kLdsStride + sub];
    const unsigned long long c = mine[(128 + (dw / 2) * 32 + 2 * jj + (dw & 1u)) * kLdsStride + sub];
    const unsigned long long n = rows_in_bin[bin];
    out[i * 4 + 0] = n - a - b + c;
    out[i * 4 + 1] = a - c;
    out[i * 4 + 2] = b - c;
    out[i * 4 + 3] = c;
  }
}

// The same accumulators read as a 1-bit-per-genome matrix (the phase plane): k_count_by_genome counts every bit position of
// a 16-byte chunk over the rows, whatever the bits mean -- here bit p of a chunk is genome 128 * chunk + p, counter
// (p / 32) * 32 + p % 32 = p of the chunk's lane.  out[g][bin] += the count.
__global__ void __launch_bounds__(kBlock)
k_finish_plane_by_genome(const uint32_t* __restrict__ acc, uint64_t n_genomes, uint32_t n_bins, uint32_t n_cg, uint32_t cg_width,
                         unsigned long long* __restrict__ out) {
  const uint64_t total = n_genomes * n_bins;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t g = i / n_bins;
    const uint32_t bin = static_cast<uint32_t>(i % n_bins);
    const uint32_t chunk = static_cast<uint32_t>(g / 128), p = static_cast<uint32_t>(g % 128);
    const uint32_t cg = chunk / cg_width, sub = chunk % cg_width;
    const uint32_t* mine = acc + (static_cast<uint64_t>(bin) * n_cg + cg) * (kAccCounters * kLdsStride);
    out[i] = mine[p * kLdsStride + sub];
  }
}

// Group the selected rows by bin on the device (counting sort in two passes).  A chunk of kBinChunk consecutive rows
// goes to one workgroup: pass 1 counts rows per bin per chunk; after an exclusive scan over (bin, chunk) on one
// small kernel, pass 2 scatters row numbers into their bin's range.  Within a chunk the order is arbitrary (LDS
// atomics) — integer sums do not care — while chunks keep their order, so gathered rows stay nearly sequential.
constexpr int kBinChunk = 4096;
constexpr int kMaxBins = 256;

// The bin of every row from its allele frequency, on the device: the P7FrequencyFilter pair of CalcFWS
// (kga_analysis_PfEMP_FWS.cpp:15-38; kgl_variant_filter_Pf7.cpp:20-66: accepted iff AF >= cutoff) -- a row is in bin b
// iff it passes the lower filter, af >= edges[b], and fails the upper one, !(af >= edges[b + 1]); the comparisons are
// the reference's, in double.  NaN = no AF value: it passes both filters, so NOT(upper) puts it in no bin.
__global__ void __launch_bounds__(kBlock)
k_af_bins(const float* __restrict__ af, uint64_t n_rows, const double* __restrict__ edges, uint32_t n_bins, uint8_t* __restrict__ bin_of_row) {
  for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < n_rows; v += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const float f = af[v];
    uint8_t bin = 0xFF;
    if (f == f) {
      const double a = static_cast<double>(f);
      for (uint32_t b = 0; b < n_bins; ++b)
        if (a >= edges[b] && !(a >= edges[b + 1])) { bin = static_cast<uint8_t>(b); break; }
    }
    bin_of_row[v] = bin;
  }
}

__global__ void __launch_bounds__(kBlock)
k_bin_count(const uint8_t* __restrict__ bin_of_row, uint64_t n_rows, uint32_t n_bins, uint32_t* __restrict__ chunk_counts) {
  __shared__ uint32_t cnt[kMaxBins];
  for (uint32_t b = threadIdx.x; b < n_bins; b += kBlock) cnt[b] = 0;
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kBinChunk;
  for (uint32_t i = threadIdx.x; i < kBinChunk; i += kBlock) {
    const uint64_t r = base + i;
    if (r < n_rows) {
      const uint32_t b = bin_of_row[r];
      if (b < n_bins) atomicAdd(&cnt[b], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_bins; b += kBlock) chunk_counts[static_cast<uint64_t>(b) * gridDim.x + blockIdx.x] = cnt[b];
}

// Exclusive scan of chunk_counts in (bin-major, chunk) order; rows_in_bin[b] and bin_offset[b] as by-products.
// Two small launches of one workgroup per bin: totals first, then every workgroup chains the bins before it (n_bins
// adds) and scans its own chunks, a contiguous run of chunks per thread with the runs' sums scanned in LDS.
__global__ void __launch_bounds__(kBlock)
k_bin_totals(const uint32_t* __restrict__ chunk_counts, uint32_t n_chunks, unsigned long long* __restrict__ rows_in_bin) {
  __shared__ unsigned long long part[kBlock / kWave];
  const uint32_t b = blockIdx.x;
  unsigned long long t = 0;
  for (uint32_t c = threadIdx.x; c < n_chunks; c += kBlock) t += chunk_counts[static_cast<uint64_t>(b) * n_chunks + c];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long total = 0;
    for (int w = 0; w < kBlock / kWave; ++w) total += part[w];
    rows_in_bin[b] = total;
  }
}

__global__ void __launch_bounds__(kBlock)
k_bin_scan(uint32_t* __restrict__ chunk_counts, uint32_t n_chunks, uint32_t n_bins, const unsigned long long* __restrict__ rows_in_bin,
           unsigned long long* __restrict__ bin_offset) {
  __shared__ unsigned long long run_sum[kBlock];
  const uint32_t b = blockIdx.x;
  unsigned long long base = 0;
  for (uint32_t k = 0; k < b; ++k) base += rows_in_bin[k];
  if (threadIdx.x == 0) {
    bin_offset[b] = base;
    if (b + 1 == n_bins) bin_offset[n_bins] = base + rows_in_bin[b];
  }
  const uint32_t per_thread = (n_chunks + kBlock - 1) / kBlock;
  const uint32_t c0 = threadIdx.x * per_thread;
  const uint32_t c1 = c0 + per_thread < n_chunks ? c0 + per_thread : n_chunks;
  uint32_t* mine = chunk_counts + static_cast<uint64_t>(b) * n_chunks;
  unsigned long long t = 0;
  for (uint32_t c = c0; c < c1; ++c) t += mine[c];
  run_sum[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = base;
    for (int i = 0; i < kBlock; ++i) { const unsigned long long v = run_sum[i]; run_sum[i] = run; run += v; }
  }
  __syncthreads();
  unsigned long long run = run_sum[threadIdx.x];
  for (uint32_t c = c0; c < c1; ++c) {
    const uint32_t v = mine[c];
    mine[c] = static_cast<uint32_t>(run);   // < 2^32 rows in total
    run += v;
  }
}

__global__ void __launch_bounds__(kBlock)
k_bin_scatter(const uint8_t* __restrict__ bin_of_row, uint64_t n_rows, uint32_t n_bins, const uint32_t* __restrict__ chunk_offsets,
              uint32_t* __restrict__ index) {
  __shared__ uint32_t cursor[kMaxBins];
  for (uint32_t b = threadIdx.x; b < n_bins; b += kBlock) cursor[b] = chunk_offsets[static_cast<uint64_t>(b) * gridDim.x + blockIdx.x];
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kBinChunk;
  for (uint32_t i = threadIdx.x; i < kBinChunk; i += kBlock) {
    const uint64_t r = base + i;
    if (r < n_rows) {
      const uint32_t b = bin_of_row[r];
      if (b < n_bins) index[atomicAdd(&cursor[b], 1u)] = static_cast<uint32_t>(r);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K8  compound offsets of HeteroHomoZygous::updateVariantAnalysisType
// (kga_analytic/kga_PfEMP/kga_analysis_PfEMP_heterozygous.cpp:61-105).  At a contig offset where the
// population holds k >= 2 distinct variants (adjacent rows), a genome with n variant copies there counts
//   n == 1 : heterozygous_reference_minor_alleles_ += 1
//   n >= 2 : homozygous_minor_alleles_ += #distinct variants carried   (UniqueUnphasedFilter; the
//            reference's quirk: every distinct alt counts, not only homozygous ones)
//            heterozygous_minor_alleles_ += #variants carried exactly once (HeterozygousFilter)
// Offsets with a single row need no kernel: they follow from the by-genome sweep (K3).
// A lane owns 16 genomes (one dword per row) and keeps their counters in registers; a workgroup walks a
// slice of the groups; results are added to acc[g][bin][3] = {het_ref_minor, hom_minor, het_minor} with
// integer atomics.  Reads only the rows of compound offsets.
// ---------------------------------------------------------------------------------------------
struct OffsetGroup {
  uint32_t first_row;  // first row of the group -- or, with a row list, the group's first position in that list
  uint32_t n_rows;
  uint32_t bin;        // output bin (contig index)
  uint32_t pad;
};

__global__ void __launch_bounds__(kBlock)
k_compound_offsets(const uint32_t* __restrict__ rows, uint64_t dwords_per_row, uint64_t n_genomes,
                   const OffsetGroup* __restrict__ groups, uint64_t n_groups, uint64_t groups_per_slice,
                   const uint32_t* __restrict__ row_list, uint32_t n_bins, unsigned long long* __restrict__ acc) {
  const uint64_t col = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // dword column = 16 genomes
  if (col * 16 >= n_genomes) return;
  const uint64_t g_begin = static_cast<uint64_t>(blockIdx.y) * groups_per_slice;
  const uint64_t g_end = g_begin + groups_per_slice < n_groups ? g_begin + groups_per_slice : n_groups;
  uint32_t het_ref[16], hom_minor[16], het_minor[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) het_ref[j] = hom_minor[j] = het_minor[j] = 0;
  uint32_t current_bin = g_begin < g_end ? groups[g_begin].bin : 0;

  auto flush = [&](uint32_t bin) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t g = col * 16 + j;
      if (g < n_genomes) {
        unsigned long long* a = acc + (g * n_bins + bin) * 3;
        if (het_ref[j]) atomicAdd(a + 0, static_cast<unsigned long long>(het_ref[j]));
        if (hom_minor[j]) atomicAdd(a + 1, static_cast<unsigned long long>(hom_minor[j]));
        if (het_minor[j]) atomicAdd(a + 2, static_cast<unsigned long long>(het_minor[j]));
      }
      het_ref[j] = hom_minor[j] = het_minor[j] = 0;
    }
  };

  for (uint64_t gi = g_begin; gi < g_end; ++gi) {
    const OffsetGroup grp = groups[gi];
    if (grp.bin != current_bin) {
      flush(current_bin);
      current_bin = grp.bin;
    }
    // Field-wise sums in 2-bit fields would overflow for k > 3, so split even/odd genomes into 4-bit fields: good for 15
    // rows.  A wider group (the reference has no cap: kga_analysis_PfEMP_heterozygous.cpp:61-105) is walked 15 rows at a
    // time, the fields drained into per-genome counts in between.
    uint32_t present_e = 0, present_o = 0, single_e = 0, single_o = 0, ge2 = 0;
    const bool wide = grp.n_rows > 15;
    uint32_t np_wide[16], ns_wide[16];
    if (wide) {
#pragma unroll
      for (int j = 0; j < 16; ++j) np_wide[j] = ns_wide[j] = 0;
    }
    for (uint32_t r = 0; r < grp.n_rows; ++r) {
      const uint64_t row = row_list ? static_cast<uint64_t>(row_list[grp.first_row + r]) : static_cast<uint64_t>(grp.first_row) + r;   // group-uniform
      const uint32_t w = rows[row * dwords_per_row + col];
      const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
      const uint32_t present = lo | hi, single = lo & ~hi;
      present_e += present & 0x11111111u;
      present_o += (present >> 2) & 0x11111111u;
      single_e += single & 0x11111111u;
      single_o += (single >> 2) & 0x11111111u;
      ge2 |= hi;
      if (wide && (r % 15 == 14 || r + 1 == grp.n_rows)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          np_wide[j] += ((j & 1) ? present_o : present_e) >> (4 * (j >> 1)) & 0xFu;
          ns_wide[j] += ((j & 1) ? single_o : single_e) >> (4 * (j >> 1)) & 0xFu;
        }
        present_e = present_o = single_e = single_o = 0;
      }
    }
    if (!wide && (present_e | present_o) == 0) continue;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // genome j of the dword sits in bits 2j..2j+1: even j -> nibble j/2 of *_e, odd j -> nibble j/2 of *_o
      const uint32_t np = wide ? np_wide[j] : ((j & 1) ? present_o : present_e) >> (4 * (j >> 1)) & 0xFu;
      const uint32_t ns = wide ? ns_wide[j] : ((j & 1) ? single_o : single_e) >> (4 * (j >> 1)) & 0xFu;
      const uint32_t two = (ge2 >> (2 * j)) & 1u;
      const bool n_ge2 = two || np >= 2;
      het_ref[j] += (!n_ge2 && np == 1) ? 1u : 0u;
      hom_minor[j] += n_ge2 ? np : 0u;
      het_minor[j] += n_ge2 ? ns : 0u;
    }
  }
  flush(current_bin);
}

// The offset filters of kgl_genomics/kgl_variant_filter/kgl_variant_filter_db_offset.cpp as counting predicates: how many
// Variant objects each leaves of a genome's offset, summed per (genome, bin) -- PopulationDB::viewFilter(F) followed by
// variantCount().  At an offset a genome holds ns variants once, n2 twice, n3 more often (np = ns + n2 + n3 distinct):
//   HomozygousFilter     (:17-58)    two objects in all, of one variant:  n3 == 0 && n2 == 1 && ns == 0  -> 2
//   HeterozygousFilter   (:66-101)   the variants held exactly once:                                      -> ns
//   DiploidFilter        (:110-129)  everything if at most two objects: n3 == 0 && ns + 2 n2 <= 2        -> ns + 2 n2
//   UniqueUnphasedFilter (:137-156)  one object per distinct variant:                                     -> np
// Same walk as k_compound_offsets (offsets holding >= 2 distinct variants; the offsets of a single row follow from the
// by-genome sweep); acc[g][bin][4] in the order above.
__global__ void __launch_bounds__(kBlock)
k_offset_filters(const uint32_t* __restrict__ rows, uint64_t dwords_per_row, uint64_t n_genomes,
                 const OffsetGroup* __restrict__ groups, uint64_t n_groups, uint64_t groups_per_slice,
                 const uint32_t* __restrict__ row_list, uint32_t n_bins, unsigned long long* __restrict__ acc) {
  const uint64_t col = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // dword column = 16 genomes
  if (col * 16 >= n_genomes) return;
  const uint64_t g_begin = static_cast<uint64_t>(blockIdx.y) * groups_per_slice;
  const uint64_t g_end = g_begin + groups_per_slice < n_groups ? g_begin + groups_per_slice : n_groups;
  uint32_t homozygous[16], heterozygous[16], diploid[16], unique[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) homozygous[j] = heterozygous[j] = diploid[j] = unique[j] = 0;
  uint32_t current_bin = g_begin < g_end ? groups[g_begin].bin : 0;

  auto flush = [&](uint32_t bin) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t g = col * 16 + j;
      if (g < n_genomes) {
        unsigned long long* a = acc + (g * n_bins + bin) * 4;
        if (homozygous[j]) atomicAdd(a + 0, static_cast<unsigned long long>(homozygous[j]));
        if (heterozygous[j]) atomicAdd(a + 1, static_cast<unsigned long long>(heterozygous[j]));
        if (diploid[j]) atomicAdd(a + 2, static_cast<unsigned long long>(diploid[j]));
        if (unique[j]) atomicAdd(a + 3, static_cast<unsigned long long>(unique[j]));
      }
      homozygous[j] = heterozygous[j] = diploid[j] = unique[j] = 0;
    }
  };

  for (uint64_t gi = g_begin; gi < g_end; ++gi) {
    const OffsetGroup grp = groups[gi];
    if (grp.bin != current_bin) {
      flush(current_bin);
      current_bin = grp.bin;
    }
    // per-genome row counts in 4-bit fields, even and odd genomes apart: good for 15 rows; a wider group is walked 15 rows
    // at a time, the fields drained into per-genome counts in between (as in k_compound_offsets)
    uint32_t present_e = 0, present_o = 0, single_e = 0, single_o = 0, twice_e = 0, twice_o = 0, more = 0;
    const bool wide = grp.n_rows > 15;
    uint32_t np_wide[16], ns_wide[16], n2_wide[16];
    if (wide) {
#pragma unroll
      for (int j = 0; j < 16; ++j) np_wide[j] = ns_wide[j] = n2_wide[j] = 0;
    }
    for (uint32_t r = 0; r < grp.n_rows; ++r) {
      const uint64_t row = row_list ? static_cast<uint64_t>(row_list[grp.first_row + r]) : static_cast<uint64_t>(grp.first_row) + r;   // group-uniform
      const uint32_t w = rows[row * dwords_per_row + col];
      const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
      const uint32_t present = lo | hi, single = lo & ~hi, twice = hi & ~lo;
      present_e += present & 0x11111111u;
      present_o += (present >> 2) & 0x11111111u;
      single_e += single & 0x11111111u;
      single_o += (single >> 2) & 0x11111111u;
      twice_e += twice & 0x11111111u;
      twice_o += (twice >> 2) & 0x11111111u;
      more |= lo & hi;
      if (wide && (r % 15 == 14 || r + 1 == grp.n_rows)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int shift = 4 * (j >> 1);
          np_wide[j] += ((j & 1) ? present_o : present_e) >> shift & 0xFu;
          ns_wide[j] += ((j & 1) ? single_o : single_e) >> shift & 0xFu;
          n2_wide[j] += ((j & 1) ? twice_o : twice_e) >> shift & 0xFu;
        }
        present_e = present_o = single_e = single_o = twice_e = twice_o = 0;
      }
    }
    if (!wide && (present_e | present_o) == 0) continue;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int shift = 4 * (j >> 1);
      const uint32_t np = wide ? np_wide[j] : ((j & 1) ? present_o : present_e) >> shift & 0xFu;
      const uint32_t ns = wide ? ns_wide[j] : ((j & 1) ? single_o : single_e) >> shift & 0xFu;
      const uint32_t n2 = wide ? n2_wide[j] : ((j & 1) ? twice_o : twice_e) >> shift & 0xFu;
      const bool n3 = (more >> (2 * j)) & 1u;
      const uint32_t objects = ns + 2 * n2;                       // exact when n3 is false
      homozygous[j] += (!n3 && n2 == 1 && ns == 0) ? 2u : 0u;
      heterozygous[j] += ns;
      diploid[j] += (!n3 && objects <= 2) ? objects : 0u;
      unique[j] += np;
    }
  }
  flush(current_bin);
}

// ---------------------------------------------------------------------------------------------
// Genome-major row lists out of the variant-major rows: for every genome the rows it carries (code != 0), ascending --
// what GenomeDB::processAll hands a per-genome indexer (VariantSort::variantGenomeIndexMT, kgl_genomics/
// kgl_variant_analysis/kgl_variant_sort.cpp:234-306, one pool task per genome there).  A sparse transpose of the bit
// matrix: a wave owns four adjacent chunk columns (one 64-byte line, 256 genomes) and a slice of the rows; lane i loads
// the line of row r0 + i, a six-stage butterfly per chunk turns the 64 x 64 tile around (lane j ends up with the 64-row
// mask of its genome), and the set bits become row numbers at the genome's cursor.  Two launches of the same walk: FILL = false counts
// per (slice, genome), k_row_list_offsets turns the counts into cursors, FILL = true writes.
// ---------------------------------------------------------------------------------------------
constexpr int kListChunks = 4;             // chunk columns per wave: one 64-byte line of a row
struct __attribute__((packed, aligned(4))) RowQuad { uint32_t a, b, c, d; };   // four list entries, dword aligned

// One bit per genome out of the chunk's 2-bit codes: bit j set where genome j's code is not 0.
__device__ __forceinline__ unsigned long long carried_bits(kgx_v4u x) {
  uint32_t half[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t t = (x[d] | (x[d] >> 1)) & 0x55555555u;      // bit 2j: genome 16 d + j carries the row
    t = (t | (t >> 1)) & 0x33333333u;                     // the even bits drawn together, 16 of them in the end
    t = (t | (t >> 2)) & 0x0F0F0F0Fu;
    t = (t | (t >> 4)) & 0x00FF00FFu;
    half[d] = (t | (t >> 8)) & 0x0000FFFFu;
  }
  return static_cast<unsigned long long>(half[0] | (half[1] << 16)) | (static_cast<unsigned long long>(half[2] | (half[3] << 16)) << 32);
}

template <int S, uint32_t LOW>
__device__ __forceinline__ void transpose_stage(uint32_t& lo, uint32_t& hi, uint32_t lane) {
  const uint32_t other_lo = __shfl_xor(lo, S), other_hi = __shfl_xor(hi, S);
  if (lane & static_cast<uint32_t>(S)) {
    lo = ((other_lo >> S) & LOW) | (lo & ~LOW);
    hi = ((other_hi >> S) & LOW) | (hi & ~LOW);
  } else {
    lo = (lo & LOW) | ((other_lo & LOW) << S);
    hi = (hi & LOW) | ((other_hi & LOW) << S);
  }
}

// 64 x 64 bit transpose across a wave: lane i comes in with row i (bit j = column j) and leaves with column i (bit j = row
// j).  Six exchange stages, blocks of 32, 16, ... 1: at block size s the lanes i and i ^ s swap their off-diagonal s x s
// blocks -- the lower lane gives its high-s columns for the upper lane's low-s columns.
__device__ __forceinline__ unsigned long long transpose_tile(unsigned long long row, uint32_t lane) {
  uint32_t lo = static_cast<uint32_t>(row), hi = static_cast<uint32_t>(row >> 32);
  {                                                         // s = 32: whole dwords change hands
    const uint32_t other_lo = __shfl_xor(lo, 32), other_hi = __shfl_xor(hi, 32);
    if (lane & 32u) lo = other_hi; else hi = other_lo;
  }
  transpose_stage<16, 0x0000FFFFu>(lo, hi, lane);         // within each dword: columns j with (j & s) == 0 are "low"
  transpose_stage<8, 0x00FF00FFu>(lo, hi, lane);
  transpose_stage<4, 0x0F0F0F0Fu>(lo, hi, lane);
  transpose_stage<2, 0x33333333u>(lo, hi, lane);
  transpose_stage<1, 0x55555555u>(lo, hi, lane);
  return static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32);
}

template <bool FILL>
__global__ void __launch_bounds__(kBlock)
k_genome_row_lists(const kgx_v4u* __restrict__ rows, uint32_t chunks_per_row, uint64_t n_rows, uint64_t n_genomes,
                   uint64_t g_lo, uint64_t g_hi /* only genomes [g_lo, g_hi) are listed */,
                   const kgx_v4u* __restrict__ keep /* the genome mask's row, or null */, const uint8_t* __restrict__ row_selected, uint64_t rows_per_slice,
                   uint32_t* __restrict__ counts /* [slices][genomes padded to 256] */, uint64_t genomes_padded,
                   const unsigned long long* __restrict__ cursors /* same shape: where each (slice, genome) starts */,
                   uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint64_t wave = g_lo / (64u * kListChunks) + static_cast<uint64_t>(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;   // group of four chunk columns
  const uint32_t c0 = static_cast<uint32_t>(wave) * kListChunks;
  if (c0 >= chunks_per_row || static_cast<uint64_t>(c0) * 64u >= g_hi) return;
  const uint64_t slice = blockIdx.y;
  const uint64_t r_begin = slice * rows_per_slice;
  const uint64_t r_end = r_begin + rows_per_slice < n_rows ? r_begin + rows_per_slice : n_rows;
  unsigned long long cursor[kListChunks];
  uint32_t count[kListChunks];
  uint32_t queued[kListChunks][3], pending[kListChunks];     // FILL: row numbers waiting for a fourth
  bool listed[kListChunks];                  // is this lane's genome of chunk k in the range (and kept by the mask)?
#pragma unroll
  for (int k = 0; k < kListChunks; ++k) {
    count[k] = 0;
    pending[k] = 0;
    queued[k][0] = queued[k][1] = queued[k][2] = 0;
    const uint64_t genome = (static_cast<uint64_t>(c0) + k) * 64u + lane;
    listed[k] = genome >= g_lo && genome < g_hi;
    if (keep != nullptr && c0 + k < chunks_per_row) {
      const kgx_v4u m = keep[c0 + k];
      listed[k] = listed[k] && ((m[lane >> 4] >> (2 * (lane & 15))) & 1u);
    }
    if constexpr (FILL) cursor[k] = cursors[slice * genomes_padded + (static_cast<uint64_t>(c0) + k) * 64u + lane];
  }
  for (uint64_t r0 = r_begin; r0 < r_end; r0 += kWave) {
    const uint64_t r = r0 + lane;
    const bool live = r < r_end && (row_selected == nullptr || row_selected[r] != 0);
    kgx_v4u x[kListChunks];
#pragma unroll
    for (int k = 0; k < kListChunks; ++k) {
      x[k] = kgx_v4u{0u, 0u, 0u, 0u};
      if (live && c0 + k < chunks_per_row) x[k] = rows[r * chunks_per_row + c0 + k];
    }
#pragma unroll
    for (int k = 0; k < kListChunks; ++k) {
      // the row's 64 "carried" bits (bit j: genome j of the chunk), then the 64 x 64 tile turned around across the wave
      unsigned long long mine = transpose_tile(carried_bits(x[k]), lane);
      if (!listed[k]) mine = 0;
      if constexpr (FILL) {
        // four row numbers per store: a lane appends to its own list, so every store instruction touches 64 different
        // lines whatever its width -- 16 bytes each instead of 4 (the lists are only dword aligned)
        while (mine) {
          const int b = __builtin_ctzll(mine);
          mine &= mine - 1;
          const uint32_t row = static_cast<uint32_t>(r0 + b);
          if (pending[k] == 3) {
            *reinterpret_cast<RowQuad*>(out + cursor[k]) = RowQuad{queued[k][0], queued[k][1], queued[k][2], row};
            cursor[k] += 4;
            pending[k] = 0;
          } else {
            if (pending[k] == 0) queued[k][0] = row; else if (pending[k] == 1) queued[k][1] = row; else queued[k][2] = row;
            ++pending[k];
          }
        }
      } else {
        count[k] += static_cast<uint32_t>(__builtin_popcountll(mine));
      }
    }
  }
  if constexpr (FILL) {
#pragma unroll
    for (int k = 0; k < kListChunks; ++k) {
      if (pending[k] > 0) out[cursor[k]] = queued[k][0];
      if (pending[k] > 1) out[cursor[k] + 1] = queued[k][1];
      if (pending[k] > 2) out[cursor[k] + 2] = queued[k][2];
    }
  } else {
#pragma unroll
    for (int k = 0; k < kListChunks; ++k)
      counts[slice * genomes_padded + (static_cast<uint64_t>(c0) + k) * 64u + lane] = count[k];
  }
}

// counts[slice][genome] -> cursors[slice][genome] = begin[genome] + the genome's entries in earlier slices, in three small
// steps: every genome's total (one thread each), the exclusive scan of the totals into begin[] (one workgroup; the grand
// total lands in begin[n_genomes]), every genome's per-slice cursors.
constexpr int kTotalsSlices = 16;
__global__ void __launch_bounds__(kBlock)
k_row_list_totals(const uint32_t* __restrict__ counts, uint64_t n_slices, uint64_t genomes_padded, uint64_t n_genomes,
                  unsigned long long* __restrict__ totals) {
  // blockIdx.y: a run of kTotalsSlices slices; totals[] starts at zero
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_genomes) return;
  const uint64_t s_begin = static_cast<uint64_t>(blockIdx.y) * kTotalsSlices;
  const uint64_t s_end = s_begin + kTotalsSlices < n_slices ? s_begin + kTotalsSlices : n_slices;
  unsigned long long total = 0;
  for (uint64_t s = s_begin; s < s_end; ++s) total += counts[s * genomes_padded + g];
  if (total) atomicAdd(&totals[g], total);
}

__global__ void __launch_bounds__(kBlock)
k_row_list_scan(const unsigned long long* __restrict__ totals, uint64_t n_genomes, unsigned long long* __restrict__ begin) {
  __shared__ unsigned long long piece[kBlock];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n_genomes; base += kBlock) {
    const uint64_t g = base + threadIdx.x;
    const unsigned long long total = g < n_genomes ? totals[g] : 0ull;
    piece[threadIdx.x] = total;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {                       // inclusive scan of the piece (Hillis-Steele)
      const unsigned long long v = threadIdx.x >= static_cast<uint32_t>(off) ? piece[threadIdx.x - off] : 0ull;
      __syncthreads();
      piece[threadIdx.x] += v;
      __syncthreads();
    }
    if (g < n_genomes) begin[g] = carry + piece[threadIdx.x] - total;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry += piece[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) begin[n_genomes] = carry;
}

__global__ void __launch_bounds__(kBlock)
k_row_list_cursors(const uint32_t* __restrict__ counts, uint64_t n_slices, uint64_t genomes_padded, uint64_t n_genomes,
                   const unsigned long long* __restrict__ begin, unsigned long long* __restrict__ cursors) {
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_genomes) return;
  unsigned long long at = begin[g];
  for (uint64_t s = 0; s < n_slices; ++s) {
    cursors[s * genomes_padded + g] = at;
    at += counts[s * genomes_padded + g];
  }
}

// ---------------------------------------------------------------------------------------------
// Flattening helpers.
// ---------------------------------------------------------------------------------------------

// Pack the reference's VariantDBGenomeData rows (one uint8 dosage vector per genome,
// kgl_variant_db_variant.h:49-51) into dosage2 rows.  One thread per output byte (4 genomes).
// src: [n_src_genomes][n_variants] staged in device memory; genome (g0 + j) of the shard.
__global__ void __launch_bounds__(kBlock)
k_pack_dosage_u8(const uint8_t* __restrict__ src, uint64_t n_src_genomes, uint64_t n_variants,
                 uint64_t g0, uint8_t* __restrict__ rows, uint64_t pitch) {
  // g0 is a multiple of 4 (checked on the host); quads of source genomes map to whole bytes.
  const uint64_t n_quads = (n_src_genomes + 3) / 4;
  const uint64_t total = n_quads * n_variants;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t v = i % n_variants;          // consecutive threads read consecutive variants
    const uint64_t q = i / n_variants;
    uint32_t byte = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t g = q * 4 + j;
      if (g < n_src_genomes) {
        uint32_t d = src[g * n_variants + v];
        d = d > 2u ? 3u : d;
        byte |= d << (2 * j);
      }
    }
    rows[v * pitch + g0 / 4 + q] = static_cast<uint8_t>(byte);
  }
}

// Zero the bit pairs at and past n_genomes in rows [v0,v1) (after a host upload).
__global__ void __launch_bounds__(kBlock)
k_mask_row_tail(uint8_t* __restrict__ rows, uint64_t pitch, uint64_t n_genomes, uint64_t v0, uint64_t v1) {
  const uint64_t used = (n_genomes + 3) / 4;
  const uint32_t rem = static_cast<uint32_t>(n_genomes & 3u);
  const uint64_t tail = pitch - used + (rem ? 1 : 0);   // bytes to touch per row
  const uint64_t first = rem ? used - 1 : used;
  const uint64_t total = (v1 - v0) * tail;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t v = v0 + i / tail;
    const uint64_t k = first + i % tail;
    uint8_t* p = rows + v * pitch + k;
    if (rem && k == used - 1) *p &= static_cast<uint8_t>((1u << (2 * rem)) - 1u);
    else *p = 0;
  }
}

// Synthetic biallelic population straight into HBM.  One thread per 16-byte chunk.
__global__ void __launch_bounds__(kBlock)
k_synth_biallelic(kgx_v4u* __restrict__ rows, uint32_t chunks_per_row, uint64_t n_rows,
                  uint64_t n_genomes, uint64_t seed, uint64_t genome_base, uint64_t variant_base,
                  float* __restrict__ af_out) {
  const uint64_t total = n_rows * chunks_per_row;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint64_t r = i / chunks_per_row;
    const uint32_t k = static_cast<uint32_t>(i % chunks_per_row);
    const uint64_t v = variant_base + r;
    const float af = kgx_synth_af(seed, v);
    if (k == 0 && af_out) af_out[r] = af;
    const double p = static_cast<double>(af);
    kgx_v4u x;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      uint32_t word = 0;
      for (int q = 0; q < 4; ++q) {
        const uint64_t g_local = static_cast<uint64_t>(k) * 64 + w * 16 + q * 4;   // first genome of the quad
        uint32_t byte = 0;
        if (g_local + 3 < n_genomes && ((genome_base & 3u) == 0)) {
          byte = kgx_synth_quad(seed, v, (genome_base + g_local) >> 2, p);
        } else {
          for (int j = 0; j < 4; ++j)
            if (g_local + j < n_genomes)
              byte |= kgx_synth_dosage(seed, v, genome_base + g_local + j, p) << (2 * j);
        }
        word |= byte << (8 * q);
      }
      x[w] = word;
    }
    rows[i] = x;
  }
}

}  // namespace kgx

#endif  // KGX_KERNELS_DOSAGE_H
