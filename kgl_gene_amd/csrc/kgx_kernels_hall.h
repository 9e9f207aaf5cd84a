// HallME over a large call without 50 passes over the genotype bytes (driven from kgx_inbreed.hip: inbreed_shard, `hall_moments`).
//
// processHallME's step (_calc.cpp:255-285) is  F <- F * S(F) / N  with  S(F) = sum over the genome's homozygous cells of
// 1 / (F + (1-F)*y),  y = the frequency of the cell's allele at its locus (classify_cell: f1).  The cells enter S only
// through their y, and 1 / (F + (1-F)*y) is analytic in y with its pole at -F/(1-F) <= 0: around a centre c > 0
//     1 / (F + u*(c + d)) = q * sum_j (-u*q*d)^j,   u = 1 - F,  q = 1 / (F + u*c),   |u*q*d| <= |d| / c .
// So the y axis is cut into bins of relative half-width 2^-8 (the double's exponent and its top kHallKeyMantissa mantissa
// bits; y = 0 has a bin of its own, d = 0), and per (genome, bin) the moments  M_j = sum over its cells of d^j,  j = 0..4,
// are all a step needs: the truncation after j = 4 is below (2^-8)^5 = 9e-13 of a term, every term and every F.  A cell
// belongs to a CLASS of homozygous cell (byte 0x00: the major allele, y = p_major; byte a | a << 4 of a phased population:
// alt a, y = its frequency); per class the loci are put in bin order, and the moments cost ONE more pass over the bytes --
// every class's hits left as rows of bits (k_class_bits) -- and a pass over those rows per class on the matrix cores
// (k_hall_mfma: the moments are an exact integer product of the hits and the powers' fixed-point digits); the 50 steps then
// run on ~10^3 numbers per genome (k_hall_iterate).  Kept beside it as checkers: a pass over the bytes per class on the matrix
// cores (KGX_K7_CLASS_BYTES=1) and the vector sweeps of round 3 (k_hall_sweep, KGX_K7_CLASS_SWEEPS=1).
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
constexpr uint32_t kHallItemLoci = 2048;                 // (an item's digit image: 64 KB of the matrix-core pass's LDS; 1024: more items to merge, 4096: one workgroup a CU)
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
  __shared__ uint32_t bin_item[kHallBins + 1], bin_block[kHallBins + 1];     // every bin's first item and first block
  // item q of a bin of `loci` loci cut into n items takes [q * loci / n, (q + 1) * loci / n): its size is loci / n or one more, the
  // first q items hold q * loci / n loci -- so the blocks of the first q items have a closed form (no walk over the items)
  auto share_begin = [](uint32_t q, uint32_t n, uint32_t loci_of_bin) { return static_cast<uint32_t>(static_cast<uint64_t>(q) * loci_of_bin / n); };
  auto blocks_before = [&](uint32_t q, uint32_t n, uint32_t loci_of_bin) {
    if (n == 0u) return 0u;
    const uint32_t small = loci_of_bin / n, bigger = share_begin(q, n, loci_of_bin) - q * small;   // of the first q items: those one locus longer
    return (q - bigger) * ((small + kHallBlockLoci - 1) / kHallBlockLoci) + bigger * ((small + kHallBlockLoci) / kHallBlockLoci);
  };
  const uint32_t first_bin = threadIdx.x * kHallBinsPerThread;
  uint32_t begin[kHallBinsPerThread], loci[kHallBinsPerThread], count[kHallBinsPerThread];
  uint32_t mine = 0, my_blocks = 0, my_low_blocks = 0;
#pragma unroll
  for (int i = 0; i < kHallBinsPerThread; ++i) {
    const uint32_t b = first_bin + i;
    begin[i] = b < kHallBins ? bin_begin[b] : 0u;
    loci[i] = b < kHallBins ? bin_end[b] - begin[i] : 0u;
    count[i] = (loci[i] + kHallItemLoci - 1) / kHallItemLoci;
    mine += count[i];
    const uint32_t blocks = blocks_before(count[i], count[i], loci[i]);
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
    bin_item[kHallBins] = thread_base[kBlock];
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
    bin_item[b] = first;
    bin_block[b] = block;
    first += count[i];
    block += blocks_before(count[i], count[i], loci[i]);
  }
  __syncthreads();
  // the items themselves, a thread an item: its bin by bisection over the bins' first items
  const uint32_t total = bin_item[kHallBins];
  for (uint32_t e = threadIdx.x; e < total; e += kBlock) {
    uint32_t lo = 0, hi = kHallBins;                                          // bin_item[lo] <= e < bin_item[hi]: ends at the bin that holds item e
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) / 2;
      if (bin_item[mid] <= e) lo = mid; else hi = mid;
    }
    const uint32_t b = lo, q = e - bin_item[b];
    const uint32_t bin_first = bin_begin[b], bin_loci = bin_end[b] - bin_first, n = (bin_loci + kHallItemLoci - 1) / kHallItemLoci;
    HallItem it;
    it.begin = bin_first + share_begin(q, n, bin_loci);
    it.end = bin_first + share_begin(q + 1, n, bin_loci);
    it.bin = b;
    it.pad = 0u;
    items[e] = it;
    item_block_base[e] = bin_block[b] + blocks_before(q, n, bin_loci);
  }
}

// The items' records in blocks: item i's from slot 64 * item_block_base[i] on, the rest of its last block repeating its last
// record (the pass gives such a slot a byte no cell has).  One workgroup per item (at most kHallItemLoci loci).
// ys (may be null): every slot's frequency y = centre(bin) + delta (the Loglikelihood walk reads these alone).
// slot_of_locus (may be null; with sorted_slots, the sort's own output: the selected locus of every position of the bin
// order): slot_of_locus[s] = the slot of selected locus s in this class's blocks (left as it is -- 0xFFFFFFFF -- where the
// locus has no cell of the class): what the one pass that leaves every class's hits as bits walks (k_class_bits).
__global__ void __launch_bounds__(kBlock)
k_hall_pad(const HallRecord* __restrict__ records, const HallItem* __restrict__ items, const uint32_t* __restrict__ n_items,
           const uint32_t* __restrict__ item_block_base, HallRecord* __restrict__ padded, double* __restrict__ ys,
           const uint32_t* __restrict__ sorted_slots, uint32_t* __restrict__ slot_of_locus) {
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
      if (slot_of_locus && t < len) slot_of_locus[sorted_slots[it.begin + t]] = static_cast<uint32_t>(first + t);
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

// ---- every class's hits as bits, in ONE pass over the bytes -------------------------------------------------------------------
// A class pass reads the rows of the loci that have a cell of its class: 2.2 reads of the matrix over the classes of C5.  The
// hits themselves are a bit a cell: ONE pass over the selected rows leaves them, for every class the locus has a cell of, as a
// row of bits in the class's own slot order -- bits[(class_row_base[k] + slot) * row_bytes + ...], a bit a genome (spans, below) -- and
// the matrix-core pass below then reads an eighth of the bytes (k_hall_mfma<.., true>).
struct HallClassRows { uint64_t base[16]; };                               // where each class's rows begin (in rows)
// A bit row is cut into SPANS of kBitsSpanGenomes genomes = 256 bytes: a wave of k_class_bits takes a span, lane l its dwords
// l, l + 64, .. l + 448 (eight coalesced loads a locus) -- genomes 4 (l + 64 d) + b for dword d, byte b -- and keeps the flag
// of (d, b) in bit 8 b + d of ONE dword: no gathering of flags inside a dword, just the eight hit masks shifted onto each
// other.  So byte o = 4 l + b of a span holds, in bit q, genome o + 256 q of the span (hall_bits_genome).
constexpr uint64_t kBitsSpanGenomes = 2048;
constexpr uint64_t kBitsSpanBytes = kBitsSpanGenomes / 8;
__host__ __device__ inline uint64_t hall_bit_row_bytes(uint64_t n_genomes) { return (n_genomes + kBitsSpanGenomes - 1) / kBitsSpanGenomes * kBitsSpanBytes; }
// the genome (of the call) in bit q of byte `at` of a bit row
__device__ __forceinline__ uint64_t hall_bits_genome(uint64_t at, uint32_t q) { return (at / kBitsSpanBytes) * kBitsSpanGenomes + at % kBitsSpanBytes + 256u * q; }

// A wave per workgroup.  grid: x = spans, y = segments of the selected loci (whole batches of eight).
// slot_of_locus[k * sel_pitch + s]: k_hall_pad's, class by class; a batch's eight slots of a class are one scalar load, the
// next class's under way while this one's bits are gathered.  last_dword: the last dword of a row that may be read.
__global__ void __launch_bounds__(kWave)
k_class_bits(const uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t g0, uint64_t last_dword, const uint32_t* __restrict__ locus_index,
             uint64_t n_sel, uint64_t loci_per_seg, const uint32_t* __restrict__ slot_of_locus, uint64_t sel_pitch, uint32_t n_classes,
             HallClassRows class_rows, uint64_t row_bytes, uint8_t* __restrict__ bits) {
  typedef uint32_t v8u __attribute__((ext_vector_type(8)));
  constexpr int kBatch = 8;
  const uint64_t span = blockIdx.x, lane = threadIdx.x;
  uint64_t col[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const uint64_t at = (g0 >> 2) + span * (kBitsSpanGenomes / 4) + lane + 64u * d;
    col[d] = at <= last_dword ? at : last_dword;                                                    // (past the row's genomes: bits nobody reads)
  }
  const uint64_t s_begin = static_cast<uint64_t>(blockIdx.y) * loci_per_seg;
  const uint64_t s_end = s_begin + loci_per_seg < n_sel ? s_begin + loci_per_seg : n_sel;
  if (s_begin >= s_end) return;
  struct Batch { uint32_t w[kBatch][8]; };
  auto load_batch = [&](Batch& batch, uint64_t first) {
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const uint64_t at = first + i < s_end ? first + i : s_end - 1;                                // (past the segment: its last locus again, not used)
      const uint64_t row = locus_index ? static_cast<uint64_t>(locus_index[at]) : at;
      const uint32_t* p = gt + row * dwords_per_row;
#pragma unroll
      for (int d = 0; d < 8; ++d) batch.w[i][d] = __builtin_nontemporal_load(p + col[d]);
    }
  };
  auto gather = [&](const Batch& batch, uint64_t first) {
    __builtin_amdgcn_sched_barrier(0);                                                              // the loads issued above stay above
    v8u of_class = *reinterpret_cast<const v8u*>(slot_of_locus + first);                            // (the same for the whole wave: scalar loads)
    for (uint32_t k = 0; k < n_classes; ++k) {
      const v8u mine = of_class;
      if (k + 1 < n_classes) of_class = *reinterpret_cast<const v8u*>(slot_of_locus + (k + 1) * sel_pitch + first);
      const uint32_t code4 = (k | (k << 4)) * 0x01010101u;
      uint8_t* rows = bits + class_rows.base[k] * row_bytes + span * kBitsSpanBytes + lane * 4;
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        const uint32_t slot = mine[i];
        if (slot == 0xFFFFFFFFu || first + i >= s_end) continue;
        uint32_t word = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const uint32_t x = batch.w[i][d] ^ code4;                                                 // a zero byte = a hit
          const uint32_t nonzero = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;                           // bit 7 of each byte: the byte is not zero
          word |= (~nonzero & 0x80808080u) >> (7 - d);                                              // byte b's flag to bit 8 b + d
        }
        *reinterpret_cast<uint32_t*>(rows + static_cast<uint64_t>(slot) * row_bytes) = word;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  Batch even, odd;
  load_batch(even, s_begin);
  for (uint64_t s = s_begin; s < s_end; s += 2 * kBatch) {
    load_batch(odd, s + kBatch);
    gather(even, s);
    load_batch(even, s + 2 * kBatch);
    if (s + kBatch < s_end) gather(odd, s + kBatch);
  }
}

// ---- the class pass on the matrix cores ----------------------------------------------------------------------------------------
// k_hall_sweep above is bound by its vector instructions: a compare and five adds per cell that issue for the whole wave whenever
// any lane hits (5.95 wave-instructions per cell measured, 54 % of the HBM peak).  But an item's moments are a product of two
// matrices,
//     M[g][j] = sum over the item's slots s of  hit[g][s] * d_s^j ,      hit = 0 / 1,
// and with d_s^j written in FIXED POINT -- t = d / 2^(e - 7) in [-1/2, 1/2) for a bin of exponent e, V_j = round(t^j * 2^54),
// V_j in seven balanced base-256 digits (int8) -- it is an EXACT one in integers: v_mfma_i32_16x16x64_i8 sums 64 slots x 16
// genomes x 16 digit columns per instruction (an item holds at most 2048 slots: |sum| <= 2048 * 128 * 128 = 2^25, no overflow),
// and the digit sums go back to doubles once per item.  Quantisation: a term (d / c)^j / j of the series is off by at most
// 2^-(54 + 7 j) -- far below the double rounding of the adds it replaces; M0 is a count and exact.  What is left for the
// vector unit is the zero-byte test on a dword (four cells in four instructions) and the 4 x 4 byte transposes that turn "four
// genomes of one locus" into "four loci of one genome" (eight v_perm_b32 per 16 cells): ~2 instructions per cell instead of 6.
//
// Operands (lane l of a wave: c = l & 15, u = l >> 4; both take 16 k per lane and the same k map on either side, so the slot of
// position (u, j) only has to be the SAME in A and B: slot 64 * block + 16 * u + j):
//   A_a[row c][k]   a = 0, 1: the digit columns, from LDS (the item's whole digit image, k_hall_digits)
//   B[k][col c]     the hits of genome 8 * c + q of the wave's 128 (q = 0..7: two products each per block, A_0 and A_1)
//   C_a[row][col]   lane l holds column c (its genomes), rows 4 * u + reg: the digit columns are dealt so that lane group u
//                   holds moment u + 1 whole: register R = 4 * a + reg = digit R (R < 7); R = 7 of group 3 is M0's column of ones.
constexpr int kHallDigits = 7;
constexpr int kHallDigitBits = 54;                                         // |V_j| <= 2^(54 - j): the top digit stays small

// The digit image and the rows of every slot of a class: digits[(block * 2 + a) * 64 + l] 16 bytes each (2 KB a block: what the
// lanes of a wave load as A_0, A_1, in lane order), slot_rows[slot].  A slot past its item's loci: zeros (no column counts
// it) and its item's last row.  One workgroup per item.
__global__ void __launch_bounds__(kBlock)
k_hall_digits(const HallRecord* __restrict__ padded, const HallItem* __restrict__ items, const uint32_t* __restrict__ n_items,
              const uint32_t* __restrict__ item_block_base, int8_t* __restrict__ digits, uint32_t* __restrict__ slot_rows) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  __shared__ v4u image[128];                                                 // a block's 2 KB, as it lies in memory
  const uint32_t n = *n_items;
  const uint32_t slot_in_block = threadIdx.x & 63u, group = threadIdx.x >> 6;  // thread: a slot's power group + 1 (its seven digits; group 3: the ones too)
  for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
    const HallItem it = items[item];
    const uint32_t len = it.end - it.begin, first_block = item_block_base[item], n_blocks = item_block_base[item + 1] - first_block;
    const uint64_t first = static_cast<uint64_t>(first_block) * kHallBlockLoci;
    const int exponent = it.bin == 0u ? 0 : static_cast<int>((it.bin - 1u) >> kHallKeyMantissa) + kHallMinExponent;
    const double to_t = __longlong_as_double(static_cast<long long>(static_cast<uint64_t>(1023 + 7 - exponent) << 52));   // 2^(7 - e)
    const double to_fixed = __longlong_as_double(static_cast<long long>(static_cast<uint64_t>(1023 + kHallDigitBits) << 52));
    for (uint32_t t = threadIdx.x; t < n_blocks * kHallBlockLoci; t += blockDim.x) slot_rows[first + t] = padded[first + t].row;
    for (uint32_t block = 0; block < n_blocks; ++block) {
      const uint32_t slot = block * kHallBlockLoci + slot_in_block;
      long long v = 0;
      if (slot < len) {
        const double x = padded[first + slot].delta * to_t;                   // exact: a power of two
        const double x2 = x * x;
        const double power = group == 0u ? x : group == 1u ? x2 : group == 2u ? x2 * x : x2 * x2;
        v = __double2ll_rn(power * to_fixed);
      }
      // digit d of group g: operand a = d / 4, row 4 g + d % 4, lane = row + 16 (slot / 16), byte slot % 16
      int8_t* bytes = reinterpret_cast<int8_t*>(image);
      const uint32_t u = slot_in_block >> 4, j = slot_in_block & 15u;
#pragma unroll
      for (uint32_t d = 0; d < 8; ++d) {
        long long low = 0;
        if (d < 7u) {                                                         // balanced digits: -128 .. 127
          low = ((v + 128) & 255) - 128;
          v = (v - low) >> 8;
        } else {
          low = (group == 3u && slot < len) ? 1 : 0;                          // M0's column of ones
        }
        bytes[(((d >> 2) * 64u + (4u * group + (d & 3u)) + 16u * u) * 16u) + j] = static_cast<int8_t>(low);
      }
      __syncthreads();
      if (threadIdx.x < 128u) reinterpret_cast<v4u*>(digits)[(first_block + block) * 128ull + threadIdx.x] = image[threadIdx.x];
      __syncthreads();
    }
  }
}

// grid: x = item * n_chunks + chunk of 512 genomes, BITS: 1024 and 512 threads (a wave: 128, a lane: eight at 16 of the block's 64 slots).
// moments as k_hall_sweep's.  EMIT as there: the items of the bins below block_bins leave the hits as bits.
// BITS: the hits come from k_class_bits' rows (bit_rows: the class's first row) instead of the bytes: a lane loads 16 bytes
// (128 genomes) of ONE slot, the wave's 64 slots go through LDS, and lane (c, u) takes back byte c of its 16 slots -- bit q
// of byte j is the lane's genome q (hall_bits_genome) at slot 16 u + j: a shift and a mask per product and dword.
template <bool EMIT, bool BITS>
__global__ void __launch_bounds__(BITS ? 2 * kBlock : kBlock)
k_hall_mfma(const uint32_t* __restrict__ gt, uint64_t dwords_per_row, uint64_t g0, uint64_t n_genomes, const uint32_t* __restrict__ slot_rows,
            const uint8_t* __restrict__ bit_rows, uint64_t row_bytes, const int8_t* __restrict__ digits, const HallItem* __restrict__ items,
            const uint32_t* __restrict__ n_items, const uint32_t* __restrict__ item_block_base, uint32_t n_chunks, uint32_t code,
            double* __restrict__ moments, uint32_t block_bins, uint64_t words_per_block, unsigned long long* __restrict__ words, uint64_t word_blocks, uint32_t word_phase) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  constexpr uint32_t kItemBlocks = kHallItemLoci / kHallBlockLoci;
  constexpr uint32_t kWaves = (BITS ? 2 * kBlock : kBlock) / kWave;           // BITS: eight waves, a tile each -- 128 bytes of every row the workgroup reads: whole lines, read once
  __shared__ v4i lds_digits[kItemBlocks * 128];                              // 32 KB: the item's A operands, in lane order
  __shared__ uint32_t lds_rows[BITS ? 1 : kHallItemLoci];
  __shared__ v4u lds_bits[BITS ? kWaves * 2 * kWave : 1];                    // BITS: per wave two blocks' rows of 16 bytes
  const uint32_t item = blockIdx.x / n_chunks;
  if (item >= *n_items) return;
  const HallItem it = items[item];
  const uint32_t len = it.end - it.begin, first_block = item_block_base[item], n_blocks = item_block_base[item + 1] - first_block;
  {
    const v4i* src = reinterpret_cast<const v4i*>(digits) + static_cast<uint64_t>(first_block) * 128;
    for (uint32_t t = threadIdx.x; t < n_blocks * 128; t += blockDim.x) lds_digits[t] = src[t];
    if constexpr (!BITS) {
      const uint32_t* rows = slot_rows + static_cast<uint64_t>(first_block) * kHallBlockLoci;
      for (uint32_t t = threadIdx.x; t < n_blocks * kHallBlockLoci; t += blockDim.x) lds_rows[t] = rows[t];
    }
  }
  __syncthreads();
  const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, c = lane & 15u, u = lane >> 4;
  // (two tiles a wave, one after the other, read every line twice: the second visit came after the item's other rows had passed
  // through the L2 -- 13.2 GB fetched for 6.4 GB of rows)
  constexpr uint32_t kTiles = 1;
#pragma nounroll
  for (uint32_t tile = 0; tile < kTiles; ++tile) {
  // the wave's 128 genomes: 8 c + q of a run of 128, or (BITS) the genomes of 16 bytes of the bit rows (hall_bits_genome)
  const uint64_t wave_tile = (static_cast<uint64_t>(blockIdx.x % n_chunks) * kWaves + wave) * kTiles + tile;
  const uint64_t wave_first = BITS ? hall_bits_genome(wave_tile * 16, 0u) : wave_tile * 128;    // the first of them, of the call's genomes
  if (wave_first >= n_genomes) return;                                        // (whole waves, after the one barrier; later tiles begin later)
  const uint64_t lane_first = BITS ? wave_first + c : wave_first + 8 * c;
  constexpr uint64_t kGenomeStride = BITS ? 256 : 1;                          // between the lane's genomes q, q + 1
  const bool active = lane_first < n_genomes;
  const uint64_t col = (g0 >> 2) + ((active ? lane_first : wave_first) >> 2);   // (not BITS) idle lanes re-read the wave's first column
  const uint32_t code4 = code * 0x01010101u;
  v4i acc[8][2];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q][0] = acc[q][1] = v4i{0, 0, 0, 0};
  constexpr int kHeld = BITS ? 2 : 16;                                        // registers of a block in flight (v2u)
  auto load_block = [&](uint32_t block, v2u (&w)[kHeld]) {
    if constexpr (BITS) {
      // this lane's slot of the block: the 16 bytes of the wave's 128 genomes
      const uint32_t slot = block * kHallBlockLoci + lane;
      const v4u raw = *reinterpret_cast<const v4u*>(bit_rows + (static_cast<uint64_t>(first_block) * kHallBlockLoci + slot) * row_bytes + wave_tile * 16);
      w[0] = v2u{raw.x, raw.y};                                                // (masked where it is used: a select here would wait for the load)
      w[1] = v2u{raw.z, raw.w};
    } else {
      const v4u* rows = reinterpret_cast<const v4u*>(lds_rows + block * kHallBlockLoci + 16u * u);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const v4u four = rows[m];
        w[4 * m + 0] = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(gt + static_cast<uint64_t>(four.x) * dwords_per_row + col));
        w[4 * m + 1] = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(gt + static_cast<uint64_t>(four.y) * dwords_per_row + col));
        w[4 * m + 2] = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(gt + static_cast<uint64_t>(four.z) * dwords_per_row + col));
        w[4 * m + 3] = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(gt + static_cast<uint64_t>(four.w) * dwords_per_row + col));
      }
    }
  };
  bool emit = false;
  if constexpr (EMIT) emit = it.bin < block_bins;                             // (the same for the whole workgroup)
  uint32_t piece[8];                                                          // EMIT: per genome, this lane's 16 slots of the block as bits (slot j at bit 15 - j)
#pragma unroll
  for (int q = 0; q < 8; ++q) piece[q] = 0u;
  auto multiply_block = [&](uint32_t block, const v2u (&w)[kHeld]) {
    __builtin_amdgcn_sched_barrier(0);                                        // the loads issued above stay above
    const v4i a0 = lds_digits[block * 128 + lane], a1 = lds_digits[block * 128 + 64 + lane];
    // EMIT: the lane's slots that are the item's (a slot past it repeats the last row: no column counts it, and no bit may)
    const uint32_t first_slot = block * kHallBlockLoci + 16u * u;
    const uint32_t inside = first_slot + 16u <= len ? 0xFFFFu : first_slot >= len ? 0u : (0xFFFFu << (16u - (len - first_slot))) & 0xFFFFu;
    uint32_t octet[4];                                                        // BITS: byte b of dword m = the eight genomes' bits at slot 16 u + 4 m + b
    if constexpr (BITS) {
      v4u* mine = lds_bits + (wave * 2 + (block & 1u)) * kWave;               // (two regions in turn: the next block's write does not wait for this one's reads)
      const bool in = block * kHallBlockLoci + lane < len;                     // (a slot past the item: its row was never written)
      mine[lane] = in ? v4u{w[0].x, w[0].y, w[1].x, w[1].y} : v4u{0u, 0u, 0u, 0u};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint8_t* bytes = reinterpret_cast<const uint8_t*>(mine) + 16u * 16u * u + c;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        octet[m] = static_cast<uint32_t>(bytes[16 * (4 * m)]) | (static_cast<uint32_t>(bytes[16 * (4 * m + 1)]) << 8) |
                   (static_cast<uint32_t>(bytes[16 * (4 * m + 2)]) << 16) | (static_cast<uint32_t>(bytes[16 * (4 * m + 3)]) << 24);
    }
    if constexpr (EMIT && BITS) {
      if (emit) {
        // the pieces straight from the bits: 8 slots x 8 genomes are an 8 x 8 bit matrix (a byte a slot), transposed in three
        // swaps (a byte a genome); the slots go in last first, so slot j lands at bit 7 - j of its genome's byte
        auto transposed = [](uint32_t first_four, uint32_t next_four) {
          unsigned long long x = (static_cast<unsigned long long>(__builtin_amdgcn_perm(0u, first_four, 0x00010203u)) << 32) |
                                 __builtin_amdgcn_perm(0u, next_four, 0x00010203u);      // byte i = slot 7 - i
          unsigned long long t;
          t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x = x ^ t ^ (t << 7);
          t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
          t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
          return x;                                                           // byte q = genome q: bit 7 - j = slot j
        };
        const unsigned long long high = transposed(octet[0], octet[1]), low = transposed(octet[2], octet[3]);   // slots 0..7, 8..15
#pragma unroll
        for (int q = 0; q < 8; ++q)
          piece[q] = ((static_cast<uint32_t>(high >> (8 * q)) & 0xFFu) << 8 | (static_cast<uint32_t>(low >> (8 * q)) & 0xFFu)) & inside;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                             // the lane's genomes 0..3, then 4..7
      uint32_t t4[4][4];                                                      // [group of four loci][genome]: byte b = locus 4 m + b, 0x80 = a hit (-128: undone with the digit sums)
      if constexpr (BITS) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) t4[m][qq] = octet[m] & (0x01010101u << (4 * k + qq));   // a hit of genome q: 2^q (q = 7: -128), undone with the digit sums
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          uint32_t hit[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t x = (k ? w[4 * m + b].y : w[4 * m + b].x) ^ code4;   // a zero byte = a hit
            const uint32_t nonzero = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;     // bit 7 of each byte: the byte is not zero
            hit[b] = ~nonzero & 0x80808080u;
          }
          // 4 x 4 bytes transposed: (locus, genome) -> (genome, locus).  __builtin_amdgcn_perm(hi, lo, sel): selector 0..3 = lo's bytes, 4..7 = hi's
          const uint32_t p01_lo = __builtin_amdgcn_perm(hit[1], hit[0], 0x05010400u), p01_hi = __builtin_amdgcn_perm(hit[1], hit[0], 0x07030602u);
          const uint32_t p23_lo = __builtin_amdgcn_perm(hit[3], hit[2], 0x05010400u), p23_hi = __builtin_amdgcn_perm(hit[3], hit[2], 0x07030602u);
          t4[m][0] = __builtin_amdgcn_perm(p23_lo, p01_lo, 0x05040100u);
          t4[m][1] = __builtin_amdgcn_perm(p23_lo, p01_lo, 0x07060302u);
          t4[m][2] = __builtin_amdgcn_perm(p23_hi, p01_hi, 0x05040100u);
          t4[m][3] = __builtin_amdgcn_perm(p23_hi, p01_hi, 0x07060302u);
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const v4i b = {static_cast<int>(t4[0][qq]), static_cast<int>(t4[1][qq]), static_cast<int>(t4[2][qq]), static_cast<int>(t4[3][qq])};
        acc[4 * k + qq][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b, acc[4 * k + qq][0], 0, 0, 0);
        acc[4 * k + qq][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b, acc[4 * k + qq][1], 0, 0, 0);
        if constexpr (EMIT && !BITS) {
          if (emit) {                                                         // bit 7 of byte b of dword m = slot 4 m + b: to bit 15 - (4 m + b)
            uint32_t bits = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const uint32_t four = ((t4[m][qq] >> 7) * 0x08040201u) >> 24;   // byte b's flag to bit 3 - b (the other products fall below bit 24 or past bit 31)
              bits = (bits << 4) | (four & 0xFu);
            }
            piece[4 * k + qq] = bits & inside;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  v2u even[kHeld], odd[kHeld];
  load_block(0u, even);
  if constexpr (BITS && !EMIT) {
    // (a block in flight is four registers: one loop body, the next block's load above this block's products)
    for (uint32_t b = 0; b < n_blocks; ++b) {
      load_block(b + 1u < n_blocks ? b + 1u : b, odd);                         // (past the item: its last block again, not used)
      multiply_block(b, even);
      even[0] = odd[0];
      even[1] = odd[1];
    }
  } else if (!emit) {
    for (uint32_t b = 0; b < n_blocks; b += 2) {
      load_block(b + 1u < n_blocks ? b + 1u : b, odd);                         // (past the item: its last block again, not used)
      multiply_block(b, even);
      load_block(b + 2u < n_blocks ? b + 2u : b, even);
      if (b + 1u < n_blocks) multiply_block(b + 1u, odd);
    }
  } else if constexpr (EMIT) {
    // a genome's word of a block: slot p at bit 63 - p = the 16-bit pieces of the four lane groups u = 0..3, first to last.
    // The groups swap pieces (lane l <-> l ^ 16, then l ^ 32): every lane of a column then holds all eight genomes' words.
    uint32_t upper[8], lower[8];
    auto join = [&]() {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t beside = static_cast<uint32_t>(__shfl_xor(static_cast<int>(piece[q]), 16));
        const uint32_t half_word = (u & 1u) ? (beside << 16) | piece[q] : (piece[q] << 16) | beside;   // slots 0..31 (u < 2) or 32..63
        const uint32_t other = static_cast<uint32_t>(__shfl_xor(static_cast<int>(half_word), 32));
        upper[q] = u < 2u ? half_word : other;
        lower[q] = u < 2u ? other : half_word;
      }
    };
    if constexpr (BITS) {
      // The words lie [genome][block] -- all word_blocks kept blocks of all classes, a genome's side by side: what the search walks
      // is a genome's words of a stretch of blocks (a word per cache line cost it the line).  Group u keeps the words of its
      // lanes' genomes q = 2 u, 2 u + 1 over EIGHT consecutive blocks -- 64 bytes of each genome's run (word_blocks is a multiple of
      // sixteen; word_phase: where the class's first block sits in its 64 bytes) -- and stores them whole; an item's first and last
      // blocks that share 64 bytes with another item's go word by word.  (Eight-byte stores alone: this pass 4 x as long -- every
      // store a read-modify-write of its sector.)
      const uint64_t g_a = lane_first + (2u * u) * kGenomeStride, g_b = g_a + kGenomeStride;
      unsigned long long* const row_a = words + g_a * word_blocks + first_block;
      unsigned long long* const row_b = words + g_b * word_blocks + first_block;
      const bool has_a = g_a < n_genomes, has_b = g_b < n_genomes;
      auto mine = [&](const uint32_t (&x)[8], int odd_one) {                  // x[2 u + odd_one]: u is the lane's, select
        const uint32_t x01 = (u & 1u) ? x[2 + odd_one] : x[odd_one], x23 = (u & 1u) ? x[6 + odd_one] : x[4 + odd_one];
        return u < 2u ? x01 : x23;
      };
      constexpr uint32_t kRun = 8;                                              // blocks a lane keeps before it stores: 64 bytes of a genome's run
      uint32_t b = 0;
      while (b < n_blocks) {
        const uint32_t slot0 = (word_phase + first_block + b) & (kRun - 1u), begun = b;   // (the same for the whole workgroup)
        uint32_t a_lo[kRun], a_hi[kRun], b_lo[kRun], b_hi[kRun];
#pragma unroll
        for (uint32_t i = 0; i < kRun; ++i) {
          a_lo[i] = a_hi[i] = b_lo[i] = b_hi[i] = 0u;
          if (i >= slot0 && b < n_blocks) {
            load_block(b + 1u < n_blocks ? b + 1u : b, odd);
            multiply_block(b, even);
            join();
            a_lo[i] = mine(lower, 0); a_hi[i] = mine(upper, 0); b_lo[i] = mine(lower, 1); b_hi[i] = mine(upper, 1);
            even[0] = odd[0];
            even[1] = odd[1];
            ++b;
          }
        }
        if (slot0 == 0u && b - begun == kRun) {
#pragma unroll
          for (uint32_t i = 0; i < kRun; i += 2) {
            if (has_a) reinterpret_cast<v4u*>(row_a + begun)[i / 2] = v4u{a_lo[i], a_hi[i], a_lo[i + 1], a_hi[i + 1]};
            if (has_b) reinterpret_cast<v4u*>(row_b + begun)[i / 2] = v4u{b_lo[i], b_hi[i], b_lo[i + 1], b_hi[i + 1]};
          }
        } else {
#pragma unroll
          for (uint32_t i = 0; i < kRun; ++i) {
            if (i >= slot0 && i - slot0 < b - begun) {
              if (has_a) row_a[begun + i - slot0] = (static_cast<unsigned long long>(a_hi[i]) << 32) | a_lo[i];
              if (has_b) row_b[begun + i - slot0] = (static_cast<unsigned long long>(b_hi[i]) << 32) | b_lo[i];
            }
          }
        }
      }
    } else {
      // group u stores pair u of hall_word_index's layout (genomes 2 u, 2 u + 1: 16 bytes)
      const uint64_t lanes = words_per_block / 8;
      v4u* out = reinterpret_cast<v4u*>(words) + static_cast<uint64_t>(first_block) * 4 * lanes + (lane_first >> 3);   // pair 0 of the item's first block
      auto store_words = [&]() {
        join();
#pragma unroll
        for (uint32_t pair = 0; pair < 4; ++pair)                             // (a store per group: u is the lane's, an indexed register is a trip through scratch)
          if (active && u == pair) out[static_cast<uint64_t>(pair) * lanes] = v4u{lower[2 * pair], upper[2 * pair], lower[2 * pair + 1], upper[2 * pair + 1]};
        out += 4 * lanes;
      };
      for (uint32_t b = 0; b < n_blocks; b += 2) {
        load_block(b + 1u < n_blocks ? b + 1u : b, odd);
        multiply_block(b, even);
        store_words();
        load_block(b + 2u < n_blocks ? b + 2u : b, even);
        if (b + 1u < n_blocks) {
          multiply_block(b + 1u, odd);
          store_words();
        }
      }
    }
  }
  if (!active) continue;
  // the digit sums back to doubles: sum_d acc_d * 256^d, times -1/128 (the hits were -128), 2^-54 and 2^(j (e - 7)), j = u + 1
  const int exponent = it.bin == 0u ? 0 : static_cast<int>((it.bin - 1u) >> kHallKeyMantissa) + kHallMinExponent;
  const int scale = -7 - kHallDigitBits + static_cast<int>(u + 1u) * (exponent - 7);
  double* out = moments + static_cast<uint64_t>(item) * kHallMoments * n_genomes;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint64_t g = lane_first + q * kGenomeStride;
    if (g >= n_genomes) break;
    double sum = static_cast<double>(acc[q][1][2]);
    sum = sum * 256.0 + static_cast<double>(acc[q][1][1]);
    sum = sum * 256.0 + static_cast<double>(acc[q][1][0]);
    sum = sum * 256.0 + static_cast<double>(acc[q][0][3]);
    sum = sum * 256.0 + static_cast<double>(acc[q][0][2]);
    sum = sum * 256.0 + static_cast<double>(acc[q][0][1]);
    sum = sum * 256.0 + static_cast<double>(acc[q][0][0]);
    // (what a hit weighed: -128 from the bytes; 2^q from the bit rows, -128 for q = 7)
    const int weight = BITS ? q : 7;
    const double signed_one = weight == 7 ? -1.0 : 1.0;
    out[static_cast<uint64_t>(u + 1u) * n_genomes + g] = signed_one * ldexp(sum, scale + 7 - weight);
    if (u == 3u) out[g] = signed_one * ldexp(static_cast<double>(acc[q][1][3]), -weight);
  }
  }
}

// bins[(bin * kHallMoments + j) * n_genomes + g] += the bin's items, in slot order (classes follow one another on the stream).
// bin_used[bin]: 0 = nobody has written the bin yet -- this class's launch (`generation` = class + 1, which it leaves there) starts
// its sums from 0 instead of reading them, so the bins need no clearing (1 GB at 10 000 genomes); any other value: an earlier
// class's sums are there.  (Every workgroup of a bin sees 0 or this launch's own generation: the same answer.)
__global__ void __launch_bounds__(kBlock)
k_hall_merge(const double* __restrict__ moments, const uint32_t* __restrict__ item_base, uint64_t n_genomes, double* __restrict__ bins,
             uint32_t* __restrict__ bin_used, uint32_t generation) {
  const uint64_t per_bin = static_cast<uint64_t>(kHallMoments) * n_genomes;
  // blockIdx.y strides the bins (a workgroup per bin was 164 000 workgroups a launch, most of them for bins without an item: 40-70 us
  // of a 12 000-locus call's 2 ms, four times)
  for (uint32_t bin = blockIdx.y; bin < kHallBins; bin += gridDim.y) {
    const uint32_t first = item_base[bin], last = item_base[bin + 1];
    if (first == last) continue;
    const uint32_t before = __hip_atomic_load(bin_used + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool fresh = before == 0u || before == generation;
    if (blockIdx.x == 0 && threadIdx.x == 0 && before == 0u) __hip_atomic_store(bin_used + bin, generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < per_bin; e += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
      double sum = fresh ? 0.0 : bins[bin * per_bin + e];
      uint32_t t = first;
      for (; t + 8 <= last; t += 8) {                                         // eight loads under way, added in slot order
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = moments[(t + i) * per_bin + e];
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[i];
      }
      for (; t < last; ++t) sum += moments[t * per_bin + e];
      bins[bin * per_bin + e] = sum;
    }
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
template <int BINS>                                       // a thread's bins: the call's used bins over the workgroup's threads, rounded up
__device__ __forceinline__ double hall_iterate_genome(const double* __restrict__ bins, const uint32_t* __restrict__ used, uint32_t n_used,
                                                      uint64_t n_genomes, uint64_t g, double total, double F, double (&row_part)[2][16]) {
  double centre[BINS], m[BINS][kHallMoments];
#pragma unroll
  for (int i = 0; i < BINS; ++i) {
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
  for (int it = 0; it < 50; ++it) {
    const bool positive = F > 0.0;
    const double Fs = positive ? F : 1.0;                                   // (F <= 0: every term F / den is 0; walked with F = 1, then zeroed)
    const double u = 1.0 - Fs;
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < BINS; ++i) {
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
  return F;
}

__global__ void __launch_bounds__(kBlock)
k_hall_iterate(const double* __restrict__ bins, const uint32_t* __restrict__ used, const uint32_t* __restrict__ n_used_ptr,
               const unsigned long long* __restrict__ counts, uint64_t n_genomes, const double* __restrict__ start, double* __restrict__ f_out) {
  __shared__ double row_part[2][16];
  const uint64_t g = blockIdx.x;
  if (g >= n_genomes) return;
  const uint32_t n_used = *n_used_ptr;
  const double total = static_cast<double>(counts[g * 6 + 4]);
  // (as many bins a thread as the call uses: 943 bins at C5 are four a thread, not the eleven all 2562 would be -- a bin without
  // cells adds an exact 0 to the thread's sum, so the sums are the same numbers either way as long as the bins keep their threads:
  // bin `at` goes to thread at % 256 in every variant)
  double F;
  if (n_used <= 4u * kBlock) F = hall_iterate_genome<4>(bins, used, n_used, n_genomes, g, total, start[g], row_part);
  else if (n_used <= 8u * kBlock) F = hall_iterate_genome<8>(bins, used, n_used, n_genomes, g, total, start[g], row_part);
  else F = hall_iterate_genome<kHallBinsPerThread>(bins, used, n_used, n_genomes, g, total, start[g], row_part);
  if (threadIdx.x == 0) f_out[g] = F;
}

}  // namespace kgx
