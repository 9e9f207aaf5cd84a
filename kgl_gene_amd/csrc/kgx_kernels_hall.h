// HallME over a large call without 50 passes over the genotype bytes (driven from kgx_inbreed.hip: inbreed_shard, `hall_moments`).
//
// processHallME's step (_calc.cpp:255-285) is  F <- F * S(F) / N  with  S(F) = sum over the genome's homozygous cells of
// 1 / (F + (1-F)*y),  y = the frequency of the cell's allele at its locus (classify_cell: f1).  The cells enter S only
// through their y, and 1 / (F + (1-F)*y) is analytic in y with its pole at -F/(1-F) <= 0: around a centre c > 0
//     1 / (F + u*(c + d)) = q * sum_j (-u*q*d)^j,   u = 1 - F,  q = 1 / (F + u*c),   |u*q*d| <= |d| / c .
// So the y axis is cut into bins of relative half-width 2^-8 (the double's exponent and its top kHallKeyMantissa mantissa
// bits; y = 0 has a bin of its own, d = 0), and per (genome, bin) the moments  M_j = sum over its cells of d^j,  j = 0..4,
// are all a step needs: the truncation after j = 4 is below (2^-8)^5 = 9e-13 of a term, every term and every F.  The moments
// cost ONE pass over the bytes per class of homozygous cell (byte 0x00: the major allele, y = p_major; byte a | a << 4 of a
// phased population: alt a, y = its frequency), over the loci in bin order so that a workgroup's accumulators stay in
// registers: 1 + amax passes instead of 50, and the 50 steps then run on ~10^3 numbers per genome (k_hall_iterate).
//
// Deterministic by construction: the loci of a class are radix-sorted by bin (stable), a bin's stretch is cut into
// items of at most kHallItemLoci loci, an item's moments go to its own slot, a bin's slots are added in slot order
// (k_hall_merge), the classes one after the other; no atomics on floating point anywhere.
#pragma once

#include <cstdint>

#include "kgx_kernels_inbreed.h"

namespace kgx {

constexpr int kHallMoments = 5;                          // M0 (a count) .. M4
constexpr int kHallKeyMantissa = 7;
constexpr int kHallMinExponent = -20;                    // bins reach down to y = 2^-20; below (and above 1) the call takes the 50 passes
constexpr uint32_t kHallBins = 1u + static_cast<uint32_t>(-kHallMinExponent) * (1u << kHallKeyMantissa) + 1u;   // {0}, [2^-20, 1), {1 ..}
constexpr uint32_t kHallNoKey = 0xFFFu;                  // sorts behind every bin (12-bit keys)
constexpr uint32_t kHallItemLoci = 1024;
constexpr int kHallBinsPerThread = (kHallBins + kBlock - 1) / kBlock;
static_assert(kHallBins < kHallNoKey, "12-bit sort keys");

struct HallRecord { uint32_t row; uint32_t bin; double delta; };          // one locus of a class, in bin order: the row it reads, its y = centre(bin) + delta
struct HallItem { uint32_t begin, end, bin, pad; };                       // positions [begin, end) of the sorted order
// Every item's loci are also laid out in BLOCKS of kHallBlockLoci slots (k_hall_pad: item i's records from slot
// 64 * item_block_base[i], its last block filled up with slots that match nothing), which is what the class pass walks.
// Loglikelihood on the same moments (kgx_kernels_loglik.h) also wants, for the bins its 1e-10 floor can come near, WHICH of
// a bin's loci a genome is homozygous at: the class pass can leave one bit per (slot, genome), a block's 64 slots to a
// word (hall_word_index).  Those bins are the lowest, so their blocks are the first.
constexpr uint32_t kHallBlockLoci = 64;
// where genome g's word of block b sits: a lane of the class pass owns 8 genomes and stores their words of a block as four
// 16-byte pairs, pair q of all lanes together (1 KB a wave-store): [block][pair][lane][2].  lanes = words_per_block / 8.
__host__ __device__ inline uint64_t hall_word_index(uint64_t block, uint64_t g, uint64_t lanes) {
  return ((block * 4 + ((g >> 1) & 3)) * lanes + (g >> 3)) * 2 + (g & 1);
}

// bin of y, or kHallNoKey where the expansion has no bin for it (the caller falls back to the passes)
__device__ __forceinline__ uint32_t hall_key(double y) {
  if (y == 0.0) return 0u;
  const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(y));
  const int exponent = static_cast<int>((bits >> 52) & 0x7FFu) - 1023;           // y = 1.m * 2^exponent (negative y: sign bit set, exponent garbage, caught below)
  if (!(y > 0.0) || exponent < kHallMinExponent || y > 1.0) return kHallNoKey;
  return 1u + static_cast<uint32_t>(exponent - kHallMinExponent) * (1u << kHallKeyMantissa) +
         static_cast<uint32_t>((bits >> (52 - kHallKeyMantissa)) & ((1u << kHallKeyMantissa) - 1u));
}

// the same on the host (the Loglikelihood path asks which bins its exact walk can reach: kgx_inbreed.hip)
inline uint32_t hall_key_host(double y) {
  if (y == 0.0) return 0u;
  uint64_t bits;
  static_assert(sizeof(bits) == sizeof(y), "double");
  __builtin_memcpy(&bits, &y, sizeof(bits));
  const int exponent = static_cast<int>((bits >> 52) & 0x7FFu) - 1023;
  if (!(y > 0.0) || exponent < kHallMinExponent || y > 1.0) return kHallNoKey;
  return 1u + static_cast<uint32_t>(exponent - kHallMinExponent) * (1u << kHallKeyMantissa) +
         static_cast<uint32_t>((bits >> (52 - kHallKeyMantissa)) & ((1u << kHallKeyMantissa) - 1u));
}

// the middle of a bin (exactly representable: one more mantissa bit)
__device__ __forceinline__ double hall_centre(uint32_t key) {
  if (key == 0u) return 0.0;
  const uint32_t k = key - 1u;
  const uint64_t exponent = static_cast<uint64_t>(static_cast<int>(k >> kHallKeyMantissa) + kHallMinExponent + 1023);
  const uint64_t bits = (exponent << 52) | (static_cast<uint64_t>(k & ((1u << kHallKeyMantissa) - 1u)) << (52 - kHallKeyMantissa)) |
                        (1ull << (52 - kHallKeyMantissa - 1));
  return __longlong_as_double(static_cast<long long>(bits));
}

// Class `k` (0: byte 0x00; a: byte a | a << 4) at every selected locus: its bin as a sort key, kHallNoKey where the locus
// has no homozygous cell of that class; *unsupported is raised where such a cell exists but its y has no bin.
__global__ void __launch_bounds__(kBlock)
k_hall_keys(const double* __restrict__ table, const uint8_t* __restrict__ valid, uint64_t n_sel, uint32_t amax, int phased, uint32_t k,
            uint32_t* __restrict__ keys, uint32_t* __restrict__ slots, unsigned int* __restrict__ unsupported) {
  const uint32_t stride = sweep_stride(amax);
  for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < n_sel; s += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    uint32_t key = kHallNoKey;
    if (valid[s] & kLocusValid) {
      double f1 = 0.0, f2 = 0.0;
      const int cls = classify_cell(k | (k << 4), table + s * stride, amax, phased != 0, f1, f2);
      if (cls == kMajorHom || cls == kMinorHom) {
        key = hall_key(f1);
        if (key == kHallNoKey) atomicOr(unsupported, 1u);
      }
    }
    keys[s] = key;
    slots[s] = static_cast<uint32_t>(s);
  }
}

// In bin order: the row each locus reads and its y's distance from the bin's centre; where each bin's stretch begins and ends.
__global__ void __launch_bounds__(kBlock)
k_hall_records(const uint32_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_slots, uint64_t n_sel,
               const double* __restrict__ table, uint32_t amax, uint32_t k, const uint32_t* __restrict__ locus_index,
               HallRecord* __restrict__ records, uint32_t* __restrict__ bin_begin, uint32_t* __restrict__ bin_end) {
  const uint32_t stride = sweep_stride(amax);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_sel; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint32_t key = sorted_keys[i];
    const uint32_t before = i ? sorted_keys[i - 1] : kHallNoKey;
    if (key != before) {
      if (key != kHallNoKey) bin_begin[key] = static_cast<uint32_t>(i);
      if (i) bin_end[before] = static_cast<uint32_t>(i);                  // (before is a bin: keys ascend, the no-key stretch is last)
    }
    if (key == kHallNoKey) continue;
    if (i + 1 == n_sel) bin_end[key] = static_cast<uint32_t>(n_sel);
    const uint32_t s = sorted_slots[i];
    const double* row = table + static_cast<uint64_t>(s) * stride;
    const double y = k == 0u ? row[amax] : row[k - 1u];                   // classify_cell's f1 of the class
    HallRecord r;
    r.row = locus_index ? locus_index[s] : s;
    r.bin = key;
    r.delta = y - hall_centre(key);
    records[i] = r;
  }
}

// One workgroup: every bin's stretch cut into items of at most kHallItemLoci loci, slots dealt in bin order.
// item_base[bin] .. item_base[bin + 1] are the bin's items; *n_items their number.  Thread t owns the bins
// t * kHallBinsPerThread ..: its own running count, then one pass of thread 0 over the 256 thread totals.
// item_block_base[n_items + 1]: where each item's blocks of kHallBlockLoci slots begin; n_blocks[0] = their number,
// n_blocks[1] = the blocks of the bins below block_bins (the first ones: the bins an exact walk can reach).
__global__ void __launch_bounds__(kBlock)
k_hall_items(const uint32_t* __restrict__ bin_begin, const uint32_t* __restrict__ bin_end, uint32_t* __restrict__ item_base,
             HallItem* __restrict__ items, uint32_t* __restrict__ n_items, uint32_t block_bins, uint32_t* __restrict__ item_block_base,
             uint32_t* __restrict__ n_blocks) {
  __shared__ uint32_t thread_base[kBlock + 1];
  __shared__ uint32_t thread_blocks[kBlock + 1];
  __shared__ uint32_t thread_low_blocks[kBlock];
  const uint32_t first_bin = threadIdx.x * kHallBinsPerThread;
  uint32_t begin[kHallBinsPerThread], loci[kHallBinsPerThread], count[kHallBinsPerThread];
  uint32_t mine = 0, my_blocks = 0, my_low_blocks = 0;
  auto share_begin = [](uint32_t q, uint32_t n, uint32_t loci_of_bin) { return static_cast<uint32_t>(static_cast<uint64_t>(q) * loci_of_bin / n); };
#pragma unroll
  for (int i = 0; i < kHallBinsPerThread; ++i) {
    const uint32_t b = first_bin + i;
    begin[i] = b < kHallBins ? bin_begin[b] : 0u;
    loci[i] = b < kHallBins ? bin_end[b] - begin[i] : 0u;
    count[i] = (loci[i] + kHallItemLoci - 1) / kHallItemLoci;
    mine += count[i];
    uint32_t blocks = 0;
    for (uint32_t q = 0; q < count[i]; ++q)
      blocks += (share_begin(q + 1, count[i], loci[i]) - share_begin(q, count[i], loci[i]) + kHallBlockLoci - 1) / kHallBlockLoci;
    my_blocks += blocks;
    if (b < block_bins) my_low_blocks += blocks;
  }
  thread_base[threadIdx.x + 1] = mine;
  thread_blocks[threadIdx.x + 1] = my_blocks;
  thread_low_blocks[threadIdx.x] = my_low_blocks;
  __syncthreads();
  if (threadIdx.x == 0) {
    thread_base[0] = 0;
    thread_blocks[0] = 0;
    uint32_t low = 0;
    for (uint32_t t = 1; t <= kBlock; ++t) { thread_base[t] += thread_base[t - 1]; thread_blocks[t] += thread_blocks[t - 1]; low += thread_low_blocks[t - 1]; }
    *n_items = thread_base[kBlock];
    item_base[kHallBins] = thread_base[kBlock];
    item_block_base[thread_base[kBlock]] = thread_blocks[kBlock];
    n_blocks[0] = thread_blocks[kBlock];
    n_blocks[1] = low;
  }
  __syncthreads();
  uint32_t first = thread_base[threadIdx.x], block = thread_blocks[threadIdx.x];
#pragma unroll
  for (int i = 0; i < kHallBinsPerThread; ++i) {
    const uint32_t b = first_bin + i;
    if (b >= kHallBins) break;
    item_base[b] = first;
    const uint32_t n = count[i];
    for (uint32_t q = 0; q < n; ++q) {
      HallItem it;
      // equal shares of the stretch (whole loci): item q takes [q * loci / n, (q + 1) * loci / n)
      it.begin = begin[i] + share_begin(q, n, loci[i]);
      it.end = begin[i] + share_begin(q + 1, n, loci[i]);
      it.bin = b;
      it.pad = 0u;
      items[first + q] = it;
      item_block_base[first + q] = block;
      block += (it.end - it.begin + kHallBlockLoci - 1) / kHallBlockLoci;
    }
    first += n;
  }
}

// The items' records in blocks: item i's from slot 64 * item_block_base[i] on, the rest of its last block repeating its last
// record (the pass gives such a slot a byte no cell has).  One workgroup per item (at most 1024 loci: four per thread).
// ys (may be null): every slot's frequency y = centre(bin) + delta (the Loglikelihood walk reads these alone).
__global__ void __launch_bounds__(kBlock)
k_hall_pad(const HallRecord* __restrict__ records, const HallItem* __restrict__ items, const uint32_t* __restrict__ n_items,
           const uint32_t* __restrict__ item_block_base, HallRecord* __restrict__ padded, double* __restrict__ ys) {
  const uint32_t n = *n_items;
  for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
    const HallItem it = items[item];
    const uint32_t len = it.end - it.begin, slots = (item_block_base[item + 1] - item_block_base[item]) * kHallBlockLoci;
    const uint64_t first = static_cast<uint64_t>(item_block_base[item]) * kHallBlockLoci;
    const double centre = hall_centre(it.bin);
    for (uint32_t t = threadIdx.x; t < slots; t += blockDim.x) {
      const HallRecord r = records[it.begin + (t < len ? t : len - 1u)];
      padded[first + t] = r;
      if (ys) ys[first + t] = centre + r.delta;
    }
  }
}

// bin_block[bin] = the first block of the bin's first item, bin = 0 .. kHallBins (the last: the number of blocks).
__global__ void __launch_bounds__(kBlock)
k_hall_bin_blocks(const uint32_t* __restrict__ item_base, const uint32_t* __restrict__ item_block_base, uint32_t* __restrict__ bin_block) {
  for (uint32_t bin = blockIdx.x * blockDim.x + threadIdx.x; bin <= kHallBins; bin += gridDim.x * blockDim.x) bin_block[bin] = item_block_base[item_base[bin]];
}

// The pass over the bytes of one class: workgroup (item, genome chunk), GPL genomes per lane as in k_inbreed_eval_lut.
// A cell counts when its byte is the class's; it adds 1, d, d^2, d^3, d^4 of its locus to its genome's moments.
// moments[((item * kHallMoments + j) * n_genomes) + g].  `padded`: the class's records in blocks (k_hall_pad).
// EMIT: the items of the bins below block_bins also leave the hits themselves, one bit per (slot, genome): bit 63 - p of
// word hall_word_index(block, g) = slot p of the block.  The bits are gathered four genomes at a time (a zero-byte test on
// the dword xor the class's byte in every lane, shifted into one accumulator byte per genome), not from the compares'
// lane masks -- left to the compiler those stayed in scalar registers and came back through v_readlane, a pass twice as long.
template <int GPL, bool EMIT>
__global__ void __launch_bounds__(kBlock)
k_hall_sweep(const uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t g0, uint64_t n_genomes,
             const HallRecord* __restrict__ padded, const HallItem* __restrict__ items, const uint32_t* __restrict__ n_items,
             uint32_t n_chunks, uint32_t code, double* __restrict__ moments, const uint32_t* __restrict__ item_block_base,
             uint32_t block_bins, uint64_t words_per_block, unsigned long long* __restrict__ words) {
  constexpr int DW = GPL / 4;
  constexpr int kBatch = 8;
  static_assert(!EMIT || GPL == 8, "the hits' word layout is a lane of eight genomes'");
  const uint32_t item = blockIdx.x / n_chunks;
  if (item >= *n_items) return;
  const HallItem it = items[item];
  const uint32_t len = it.end - it.begin, first_block = item_block_base[item];
  const HallRecord* __restrict__ records = padded + static_cast<uint64_t>(first_block) * kHallBlockLoci;   // slots 0 .. len of the item (+ padding)
  const uint64_t lane = static_cast<uint64_t>(blockIdx.x % n_chunks) * blockDim.x + threadIdx.x;     // genomes g0 + GPL * lane ..
  const bool active = lane * GPL < n_genomes;
  const uint64_t col = (g0 >> 2) + (active ? lane * DW : 0);                // g0 is a multiple of GPL; idle lanes re-read the first column
  uint32_t count[GPL];
  double m1[GPL], m2[GPL], m3[GPL], m4[GPL];
  uint32_t hits[EMIT ? DW : 1];                                             // EMIT: a byte per genome, the batch's slots' hits, first slot in bit 7
#pragma unroll
  for (int j = 0; j < GPL; ++j) { count[j] = 0u; m1[j] = m2[j] = m3[j] = m4[j] = 0.0; }
  const uint32_t code4 = code * 0x01010101u;
  auto add_locus = [&](auto emit_c, const uint32_t (&w)[DW], double d1, uint32_t match) {
    const double d2 = d1 * d1, d3 = d2 * d1, d4 = d2 * d2;
    if constexpr (decltype(emit_c)::value) {
      const uint32_t inside = match == code ? 0x01010101u : 0u;            // (a slot past the item: no hit)
#pragma unroll
      for (int k = 0; k < DW; ++k) {
        const uint32_t x = w[k] ^ code4;                                     // a zero byte = a hit
        const uint32_t nonzero = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) >> 7;   // bit 0 of each byte: the byte is not zero
        hits[k] = (hits[k] << 1) | (~nonzero & inside);
      }
    }
#pragma unroll
    for (int j = 0; j < GPL; ++j) {
      const bool hit = ((w[j / 4] >> (8 * (j % 4))) & 0xFFu) == match;
      // adds under the lanes' mask, not multiplications by 0 / 1: an instruction costs what its ACTIVE lanes cost, and the
      // alt classes' cells are few (measured at C5: the major class 11.3 -> 10.6 ms, alt 1 11.8 -> 9.8 ms; the call then runs
      // at ~1060 W and 2.28 GHz, under the 1400 W cap the table passes sit on: profiles/r03_power_cap.md)
      if (hit) {
        count[j] += 1u;
        m1[j] += d1;
        m2[j] += d2;
        m3[j] += d3;
        m4[j] += d4;
      }
    }
  };
  // A batch of kBatch slots: their records by scalar loads, their bytes by kBatch vector loads issued together.  A slot
  // past the item matches no byte (0x100); the blocks' padding and the slack behind the last block keep its row readable.
  struct Batch { uint32_t w[kBatch][DW]; double delta[kBatch]; uint32_t match[kBatch]; };
  auto load_batch = [&](Batch& batch, uint32_t first) {
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const HallRecord r = records[first + b];
      batch.delta[b] = r.delta;
      batch.match[b] = first + b < len ? code : 0x100u;
      const uint32_t* p = gt + static_cast<uint64_t>(r.row) * dwords_per_row + col;
#pragma unroll
      for (int k = 0; k < DW; ++k) batch.w[b][k] = __builtin_nontemporal_load(p + k);
    }
  };
  auto add_batch = [&](auto emit_c, const Batch& batch) {
    __builtin_amdgcn_sched_barrier(0);                                      // the loads issued above stay above
#pragma unroll
    for (int b = 0; b < kBatch; ++b) add_locus(emit_c, batch.w[b], batch.delta[b], batch.match[b]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // Two batches in turn, the next one's loads in flight while this one is counted; no branch between a load and its use
  // (a branch makes the compiler wait for every outstanding load), so the last turn may load and count slots past the item.
  Batch even, odd;
  load_batch(even, 0u);
  bool emit = false;
  if constexpr (EMIT) emit = it.bin < block_bins;                           // (the same for the whole workgroup)
  if (!emit) {
    for (uint32_t i = 0; i < len; i += 2 * kBatch) {
      load_batch(odd, i + kBatch);
      add_batch(std::false_type{}, even);
      load_batch(even, i + 2 * kBatch);
      add_batch(std::false_type{}, odd);
    }
  } else if constexpr (EMIT) {
    // whole blocks of 64 slots: eight batches, each leaving a byte per genome, pushed onto the genome's half word
    const uint64_t lanes = words_per_block / GPL;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u* out = reinterpret_cast<v4u*>(words) + static_cast<uint64_t>(first_block) * 4 * lanes + lane;    // pair 0 of the item's first block
    uint32_t half_word[GPL], upper[GPL];                                    // slot p of the block at bit 63 - p: the first 32 slots are the upper half
    auto push = [&]() {
#pragma unroll
      for (int j = 0; j < GPL; ++j) half_word[j] = (half_word[j] << 8) | ((hits[j / 4] >> (8 * (j % 4))) & 0xFFu);
#pragma unroll
      for (int k = 0; k < DW; ++k) hits[k] = 0u;
    };
#pragma unroll
    for (int k = 0; k < DW; ++k) hits[k] = 0u;
    for (uint32_t i = 0; i < len; i += kHallBlockLoci, out += 4 * lanes) {
#pragma nounroll
      for (uint32_t at = i; at < i + kHallBlockLoci; at += 2 * kBatch) {
        load_batch(odd, at + kBatch);
        add_batch(std::true_type{}, even);
        push();
        load_batch(even, at + 2 * kBatch);
        add_batch(std::true_type{}, odd);
        push();
        if (at + 2 * kBatch == i + kHallBlockLoci / 2) {
#pragma unroll
          for (int j = 0; j < GPL; ++j) upper[j] = half_word[j];
        }
      }
      if (active) {
#pragma unroll
        for (int q = 0; q < GPL / 2; ++q) {                                 // genomes 2q, 2q + 1: one 16-byte store, the lanes' side by side
          v4u v;
          v.x = half_word[2 * q]; v.y = upper[2 * q]; v.z = half_word[2 * q + 1]; v.w = upper[2 * q + 1];
          out[static_cast<uint64_t>(q) * lanes] = v;
        }
      }
    }
  }
  if (!active) return;
  double* out = moments + static_cast<uint64_t>(item) * kHallMoments * n_genomes;
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const uint64_t g = lane * GPL + j;
    if (g >= n_genomes) break;
    out[g] = static_cast<double>(count[j]);
    out[n_genomes + g] = m1[j];
    out[2 * n_genomes + g] = m2[j];
    out[3 * n_genomes + g] = m3[j];
    out[4 * n_genomes + g] = m4[j];
  }
}

// bins[(bin * kHallMoments + j) * n_genomes + g] += the bin's items, in slot order (classes follow one another on the stream).
__global__ void __launch_bounds__(kBlock)
k_hall_merge(const double* __restrict__ moments, const uint32_t* __restrict__ item_base, uint64_t n_genomes, double* __restrict__ bins,
             uint32_t* __restrict__ bin_used) {
  const uint32_t bin = blockIdx.y;
  const uint32_t first = item_base[bin], last = item_base[bin + 1];
  if (first == last) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) bin_used[bin] = 1u;
  const uint64_t per_bin = static_cast<uint64_t>(kHallMoments) * n_genomes;
  for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < per_bin; e += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    double sum = bins[bin * per_bin + e];
    for (uint32_t t = first; t < last; ++t) sum += moments[t * per_bin + e];
    bins[bin * per_bin + e] = sum;
  }
}

// One workgroup: the bins that hold anything, in bin order (thread t: the bins t * kHallBinsPerThread ..).
__global__ void __launch_bounds__(kBlock)
k_hall_used_bins(const uint32_t* __restrict__ bin_used, uint32_t* __restrict__ used, uint32_t* __restrict__ n_used) {
  __shared__ uint32_t thread_base[kBlock + 1];
  const uint32_t first_bin = threadIdx.x * kHallBinsPerThread;
  uint32_t flags = 0, mine = 0;
#pragma unroll
  for (int i = 0; i < kHallBinsPerThread; ++i) {
    const uint32_t b = first_bin + i;
    if (b < kHallBins && bin_used[b]) { flags |= 1u << i; ++mine; }
  }
  thread_base[threadIdx.x + 1] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    thread_base[0] = 0;
    for (uint32_t t = 1; t <= kBlock; ++t) thread_base[t] += thread_base[t - 1];
    *n_used = thread_base[kBlock];
  }
  __syncthreads();
  uint32_t at = thread_base[threadIdx.x];
#pragma unroll
  for (int i = 0; i < kHallBinsPerThread; ++i)
    if (flags & (1u << i)) used[at++] = first_bin + i;
}

// processHallME's 50 steps (_calc.cpp:255-285) on the moments: a workgroup per genome, thread t holds the bins used[t],
// used[t + 256], ... in registers; a step is one division and a Horner chain per bin and a block sum whose bits are the
// same in every thread (row_sum16 + a fixed tree out of LDS).  F <= 0 stays 0 as in k_hall_update.
__global__ void __launch_bounds__(kBlock)
k_hall_iterate(const double* __restrict__ bins, const uint32_t* __restrict__ used, const uint32_t* __restrict__ n_used_ptr,
               const unsigned long long* __restrict__ counts, uint64_t n_genomes, const double* __restrict__ start, double* __restrict__ f_out) {
  __shared__ double row_part[2][16];
  const uint64_t g = blockIdx.x;
  if (g >= n_genomes) return;
  const uint32_t n_used = *n_used_ptr;
  double centre[kHallBinsPerThread], m[kHallBinsPerThread][kHallMoments];
#pragma unroll
  for (int i = 0; i < kHallBinsPerThread; ++i) {
    const uint32_t at = threadIdx.x + static_cast<uint32_t>(i) * kBlock;
    centre[i] = 1.0;
#pragma unroll
    for (int j = 0; j < kHallMoments; ++j) m[i][j] = 0.0;                  // nothing: q * 0
    if (at < n_used) {
      const uint32_t bin = used[at];
      centre[i] = hall_centre(bin);
#pragma unroll
      for (int j = 0; j < kHallMoments; ++j) m[i][j] = bins[(static_cast<uint64_t>(bin) * kHallMoments + j) * n_genomes + g];
    }
  }
  auto block_sum = [&](double v, int pass) {
    v = row_sum16(v);
    if ((threadIdx.x & 15) == 0) row_part[pass & 1][threadIdx.x >> 4] = v;
    __syncthreads();
    const double* p = row_part[pass & 1];
    double pair[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pair[i] = p[2 * i] + p[2 * i + 1];
    return ((pair[0] + pair[1]) + (pair[2] + pair[3])) + ((pair[4] + pair[5]) + (pair[6] + pair[7]));
  };
  const double total = static_cast<double>(counts[g * 6 + 4]);
  double F = start[g];
  for (int it = 0; it < 50; ++it) {
    const bool positive = F > 0.0;
    const double Fs = positive ? F : 1.0;                                   // (F <= 0: every term F / den is 0; walked with F = 1, then zeroed)
    const double u = 1.0 - Fs;
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < kHallBinsPerThread; ++i) {
      const double q = 1.0 / __builtin_fma(u, centre[i], Fs);
      const double t = -u * q;
      double h = m[i][4];
      h = __builtin_fma(h, t, m[i][3]);
      h = __builtin_fma(h, t, m[i][2]);
      h = __builtin_fma(h, t, m[i][1]);
      h = __builtin_fma(h, t, m[i][0]);
      sum = __builtin_fma(q, h, sum);
    }
    const double S = block_sum(sum, it);
    F = positive ? (F * S) / total : 0.0 / total;
  }
  if (threadIdx.x == 0) f_out[g] = F;
}

}  // namespace kgx
