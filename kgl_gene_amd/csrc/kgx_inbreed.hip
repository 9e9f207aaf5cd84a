// C ABI (include/kgx.h) of the inbreeding sweep: the allele-index matrix (gt8), K6 (per-locus class frequencies),
// K5 (generateFrequencies + Simple / RitlandLocus) and K7 (HallME, Loglikelihood), and the synthetic populations.
// A matrix is one shard of genomes per bound device; genomes are independent, so a call sweeps every shard it touches
// on its own device at the same time and concatenates the per-genome results: no exchange.  No CPU fallback exists.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "kgx_kernels_inbreed.h"
#include "kgx_kernels_hall.h"
#include "kgx_kernels_loglik.h"
#include <hipcub/hipcub.hpp>
#include "kgx_internal.h"

namespace kgx {
namespace {

int require_runtime(std::shared_ptr<Runtime>& rt) {
  rt = current_runtime();
  if (!rt) return fail(KGX_ENODEVICE, "kgx_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)");
  return KGX_OK;
}

int sync_shards(const kgx_gt8* h) {
  for (const auto& sh : h->shards) {
    if (int rc = use_device(*sh.dev)) return rc;
    KGX_HIP(hipStreamSynchronize(sh.dev->stream));
  }
  return use_device(*h->shards[0].dev);
}

void destroy_shards(kgx_gt8* h) {
  for (auto& sh : h->shards)
    if ((sh.d_gt || sh.d_wide || sh.d_wide_of_row) && use_device(*sh.dev) == KGX_OK) {
      if (sh.d_gt) (void)hipFree(sh.d_gt);
      if (sh.d_wide) (void)hipFree(sh.d_wide);
      if (sh.d_wide_of_row) (void)hipFree(sh.d_wide_of_row);
      sh.d_gt = nullptr; sh.d_wide = nullptr; sh.d_wide_of_row = nullptr;
    }
  if (!h->shards.empty()) (void)use_device(*h->shards[0].dev);
}

// kgx_inbreed for the genomes [g0, g1) of ONE shard (shard-local indices, g0 a multiple of 4); arguments checked by the caller.
// start: host [g1 - g0] start points of the iterative estimators, or null (the midpoints of the reference's start intervals).
// objective_method != 0 (kgx_inbreed_objective, a diagnostic): no search -- objective_out[g] = the log-likelihood of genome g at
// start[g], from the moments (1) or from a table pass (2); `out` is not written.
int inbreed_shard(kgx_gt8_shard& sh, uint64_t g0, uint64_t g1, const uint32_t* locus_index, uint64_t n_sel, const double* minor_af,
                  uint32_t amax, int phased, int algorithm, const double* start, kgx_locus_results* out, int objective_method = 0,
                  double* objective_out = nullptr) {
  Device& dev = *sh.dev;
  std::lock_guard<std::mutex> device_lock(dev.mutex);          // the arena, the compaction buffers and the timing events are the device's
  if (int rc = use_device(dev)) return rc;
  const uint64_t n = g1 - g0;
  if (n == 0) return KGX_OK;
  static_assert(sizeof(kgx_locus_results) == sizeof(LocusResultsDev), "LocusResults layout");

  const uint32_t stride = sweep_stride(amax);
  // Frequency pass flavour.  With allele indices that fit the tables (amax <= 7): ONE table pass that yields the class
  // counts and the class-frequency sums (k_inbreed_eval_lut<4>) and, for RitlandLocus, the Ritland terms with them
  // (k_inbreed_eval_lut<3>).  KGX_K5_NO_TABLE_SWEEP=1 (not for RitlandLocus): the SWAR sweeps instead -- 16 genomes per
  // lane (16-byte loads) when the group starts on a 16-genome boundary, otherwise 4 genomes per lane (amax <= 4).
  const bool ritland = algorithm == KGX_ALGO_RITLAND_LOCUS;
  const bool table_sweep = !env_int("KGX_K5_GENERIC", 0) && !env_int("KGX_K5_NO_EVAL_LUT", 0) && amax <= 7 &&
                           n_sel <= 65535ull * kRitlandSegment && (ritland || !env_int("KGX_K5_NO_TABLE_SWEEP", 0));
  const bool swar16 = !table_sweep && !env_int("KGX_K5_GENERIC", 0) && !env_int("KGX_K5_NO_SWAR16", 0) && amax <= 4 && (g0 & 15u) == 0 &&
                      !ritland;
  // The evaluation passes of HallME / Loglikelihood go through the per-batch LDS tables when the allele indices fit them.
  const bool eval_lut = !env_int("KGX_K5_GENERIC", 0) && !env_int("KGX_K5_NO_EVAL_LUT", 0) && amax <= 7;
  int eval_gpl = env_int("KGX_K5_EVAL_GPL", 8);            // genomes per lane: the widest load the group's alignment allows
  if (eval_gpl != 4 && eval_gpl != 8) eval_gpl = 8;
  while (eval_gpl > 4 && (g0 % static_cast<uint64_t>(eval_gpl)) != 0) eval_gpl /= 2;
  const uint32_t gx = static_cast<uint32_t>(((n + 3) / 4 + kBlock - 1) / kBlock);
  const uint32_t gx16 = static_cast<uint32_t>(((n + 15) / 16 + kBlock - 1) / kBlock);
  uint64_t n_seg = (static_cast<uint64_t>(dev.compute_units) * env_int("KGX_K5_BLOCKS_PER_CU", 8) + (swar16 ? gx16 : gx) - 1) / (swar16 ? gx16 : gx);
  if (n_seg > (n_sel + 63) / 64) n_seg = (n_sel + 63) / 64;
  if (n_seg < (n_sel + 65534) / 65535) n_seg = (n_sel + 65534) / 65535;   // 16-bit class counters per segment
  if (n_seg < 1) n_seg = 1;
  if (n_seg > 65535) n_seg = 65535;
  uint64_t per_seg = n_sel ? (n_sel + n_seg - 1) / n_seg : 8;
  per_seg = (per_seg + 7) / 8 * 8;                                          // whole 8-locus batches per segment
  n_seg = n_sel ? (n_sel + per_seg - 1) / per_seg : 1;
  // The table passes cut the loci for their own launch shape: 256-thread workgroups of eval_gpl genomes per lane, two or
  // three resident per CU (registers), about eight rounds of them and never a workgroup more than that (one more, nearly
  // empty round costs an eighth of the pass).
  uint64_t eval_n_seg = n_seg, eval_per_seg = per_seg;
  if (eval_lut || table_sweep) {
    const uint64_t eval_gx = ((n + eval_gpl - 1) / eval_gpl + kBlock - 1) / kBlock;
    const uint64_t resident = static_cast<uint64_t>(dev.compute_units) * static_cast<uint64_t>(env_int("KGX_K5_EVAL_RESIDENT", 2));
    uint64_t want = resident * static_cast<uint64_t>(env_int("KGX_K5_EVAL_ROUNDS", 8)) / eval_gx;
    if (want < 1) want = 1;
    if (want > 65535) want = 65535;
    eval_per_seg = n_sel ? (n_sel + want - 1) / want : kEvalBatch;
    eval_per_seg = (eval_per_seg + 7) / 8 * 8;
    if (eval_per_seg < 64) eval_per_seg = 64;
    eval_n_seg = n_sel ? (n_sel + eval_per_seg - 1) / eval_per_seg : 1;
    if (table_sweep) {                                       // the frequency sweep's own cut: 12-bit class counters per segment
      per_seg = eval_per_seg > kRitlandSegment ? kRitlandSegment : eval_per_seg;
      n_seg = n_sel ? (n_sel + per_seg - 1) / per_seg : 1;
    }
  }
  const uint64_t max_seg = eval_n_seg > n_seg ? eval_n_seg : n_seg;
  const uint64_t pass_n_seg = eval_lut ? eval_n_seg : n_seg;                  // segments of an estimator pass (modes 1, 2)

  double *d_af = nullptr, *d_table = nullptr, *d_part = nullptr, *d_sums = nullptr, *d_f = nullptr, *d_eval = nullptr, *d_segdef = nullptr;
  uint8_t* d_valid = nullptr;
  uint32_t* d_index = nullptr;
  unsigned long long* d_counts = nullptr;
  LocusResultsDev* d_out = nullptr;
  uint32_t* d_meta = nullptr;
  GoldenState* d_golden = nullptr;
  BrentState* d_brent = nullptr;
  unsigned int* d_running = nullptr;
  int rc = KGX_OK;
  auto try_hip = [&](hipError_t e, int code, const char* what) {
    if (rc == KGX_OK && e != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(code, "kgx_inbreed: %s failed: %s", what, hipGetErrorString(e));
    }
  };
  // Scratch comes out of one grow-only device arena kept by the library (a window loop calls this hundreds of times;
  // fourteen hipMalloc/hipFree pairs per call cost more than the sweep).  The call is synchronous, so reuse is safe.
  const uint64_t n_tab = n_sel ? n_sel : 1;
  ScratchPlan plan;
  const size_t o_af = plan.add(n_tab * amax * sizeof(double)), o_table = plan.add(n_tab * stride * sizeof(double));
  const size_t o_valid = plan.add(n_tab), o_meta = plan.add((n_tab + 8) * sizeof(uint32_t));
  const size_t o_part = plan.add(max_seg * n * kParts0 * sizeof(double)), o_segdef = plan.add(max_seg * kSegDefaults * sizeof(double));
  const size_t o_sums = plan.add(n * kParts0 * sizeof(double));
  // counts | F | the flag/counter word | locus index, adjacent: one memset clears them all (a window-sized call is bound
  // by its API calls, not by its kernels)
  // (two planes of n values each for F and the objective: the paired Loglikelihood search evaluates two points per pass)
  const size_t o_counts = plan.add(n * 6 * sizeof(unsigned long long)), o_f = plan.add(2 * n * sizeof(double));
  const size_t o_running = plan.add(sizeof(unsigned int));
  const size_t o_index = plan.add((n_sel + 8) * sizeof(uint32_t)), o_cleared_end = plan.total;
  const size_t o_eval = plan.add(2 * n * sizeof(double)), o_out = plan.add(n * sizeof(LocusResultsDev));
  const size_t o_golden = plan.add(n * sizeof(GoldenState));
  const size_t o_brent = plan.add(n * sizeof(BrentState));
  const size_t o_start = plan.add(n * sizeof(double)), o_keep = plan.add(n * sizeof(double));
  // The class-frequency sums of the defaults in the reference's own (sequential) summation order (k_seq_* kernels): from
  // the size at which a tree reduction and a sequential sum part by more than a tenth of the tolerance.
  const bool swar_family = table_sweep || (!env_int("KGX_K5_GENERIC", 0) && amax <= 4 && !ritland);
  const bool sequential_defaults = swar_family && n_sel >= static_cast<uint64_t>(env_int("KGX_K5_SEQUENTIAL_MIN", 1 << 16));
  const uint64_t n_seq_blocks = (n_sel + kSeqBlock - 1) / kSeqBlock;
  const size_t o_seq_sum = plan.add(n_seq_blocks * 4 * sizeof(double)), o_seq_e = plan.add(n_seq_blocks * 4 * sizeof(int));
  const size_t o_seq_n = plan.add(n_seq_blocks * 4 * sizeof(long long)), o_seq_out = plan.add(kParts0 * sizeof(double));
  // The table passes' entries (k_eval_entries): 64 / 256 / 1024 bytes per selected locus, only where such a pass will run.
  // (amax > 14: a selection with offsets of more than 14 reference alts -- their cells are in the matrix's wide rows, which the
  // generic per-cell kernels alone read: no one-launch iteration, no moments; the table passes and SWAR sweeps stop at amax 7 / 4 anyway)
  const bool wave_sized = n_sel > 0 && n_sel <= static_cast<uint64_t>(kGenomeLoci) && !env_int("KGX_K7_NO_WAVE", 0) && objective_method == 0 && amax <= 14;
  const bool table_passes = n_sel > 0 && (table_sweep || (eval_lut && (algorithm == 2 || algorithm == 3) && !wave_sized));
  const size_t o_entries = plan.add(table_passes ? (n_sel << (2u * eval_bits(amax))) * sizeof(EvalEntry) : 0);
  const size_t o_segcnt = plan.add(table_sweep ? max_seg * n * sizeof(unsigned long long) : sizeof(unsigned long long));
  // HallME and Loglikelihood over a large call: per-genome moments of the homozygous cells' frequencies instead of 50 / 38
  // passes (kgx_kernels_hall.h, kgx_kernels_loglik.h)
  // (any amax for HallME: the class passes compare bytes, no table; the generic / no-table flavours keep the passes they are there to test)
  const bool no_tables = env_int("KGX_K5_GENERIC", 0) || env_int("KGX_K5_NO_EVAL_LUT", 0);
  const int search = env_str("KGX_K7_SEARCH") == "brent" ? kSearchBrent : kSearchNelderMead;
  const bool hall_candidate = algorithm == KGX_ALGO_HALL_ME && n_sel > 0 && n_sel < (1ull << 31) && !wave_sized && !no_tables && amax <= 14 &&
                              !env_int("KGX_K7_HALL_PASSES", 0);                 // (the radix sort counts its items in an int)
  // Loglikelihood: the reference optimiser's path alone, a phased population (unphased: every alt homozygote is a heterozygous
  // cell that can meet the upper bound -- every genome would be handed to the passes), the frequency sweep as a table pass
  // (it carries the heterozygous cells' term), positions in 28 bits
  const bool loglik_candidate = algorithm == KGX_ALGO_LOGLIKELIHOOD && n_sel > 0 && n_sel < (1ull << 28) && !wave_sized && !no_tables &&
                                table_sweep && phased && (g0 & 7u) == 0 && search == kSearchNelderMead && !env_int("KGX_K7_GOLDEN", 0) &&
                                !env_int("KGX_K7_ESTIMATE_START", 0) && !env_int("KGX_K7_LL_PASSES", 0);
  const uint32_t hall_classes = 1u + (phased ? amax : 0u);                  // byte 0x00, and a | a << 4 of a phased population (classify_cell)
  const uint64_t hall_items = (hall_candidate || loglik_candidate) ? n_sel / kHallItemLoci + kHallBins + 1 : 0;    // per class, at most
  const uint64_t hall_blocks = (hall_candidate || loglik_candidate) ? n_sel / kHallBlockLoci + hall_items : 0;      // per class, at most: blocks of 64 slots
  const uint64_t words_per_block = (n + 7) / 8 * 8;                          // one word per genome, whole lanes of 8
  bool hall_moments = hall_candidate, loglik_moments = loglik_candidate;
  if (hall_moments || loglik_moments) {
    // the moments' buffers (40 B per item and genome, 100 KB of bins per genome; Loglikelihood: a bit per homozygous-capable
    // cell of the bins its floor can reach) must leave the device room to breathe: past half of what is free (counting what
    // this call may regrow) the passes, which need none of it, are made
    const uint64_t extra = hall_items * kHallMoments * n * sizeof(double) + static_cast<uint64_t>(kHallBins) * kHallMoments * n * sizeof(double) +
                           hall_classes * (hall_blocks + 1) * kHallBlockLoci * (sizeof(HallRecord) + 36) + n_sel * (sizeof(HallRecord) + 4 * sizeof(uint32_t)) +
                           (loglik_candidate ? hall_classes * hall_blocks * words_per_block * sizeof(unsigned long long) : 0) +
                           hall_classes * hall_blocks * kHallBlockLoci * hall_bit_row_bytes(n);     // (the hits' bit rows, at most)
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) { (void)hipGetLastError(); free_bytes = 0; }
    if (extra + plan.total > (static_cast<uint64_t>(free_bytes) + dev.scratch_bytes + dev.words_bytes + dev.bits_bytes) / 2) hall_moments = loglik_moments = false;
  }
  const bool by_moments_planned = hall_moments || loglik_moments;
  size_t hall_sort_bytes = 0;
  if (by_moments_planned) {
    // (a size query: no device work)
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, hall_sort_bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                           static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), static_cast<int>(n_sel), 0, 12,
                                           dev.stream) != hipSuccess)
      return fail(KGX_EHIP, "kgx_inbreed: radix sort size query failed");
  }
  const uint64_t plan_classes = by_moments_planned ? hall_classes : 0, plan_items = by_moments_planned ? hall_items : 0;
  // (two of each: the classes are put in order two at a time, on the call's stream and on the side stream)
  const size_t hall_keys_bytes = (4 * n_sel * sizeof(uint32_t) + 255) / 256 * 256, hall_records_bytes = (n_sel * sizeof(HallRecord) + 255) / 256 * 256;
  const size_t o_hall_keys = plan.add(by_moments_planned ? 2 * hall_keys_bytes : 0);                      // keys, slots, sorted keys, sorted slots
  const size_t o_hall_records = plan.add(by_moments_planned ? 2 * hall_records_bytes : 0);                // a class's loci in bin order, as sorted
  const size_t o_hall_padded = plan.add(plan_classes * (hall_blocks + 1) * kHallBlockLoci * sizeof(HallRecord));   // per class: the same in blocks (+ a block of slack)
  // per class: bin_begin | bin_end | item_base | bin_block (kHallBins + 1 words each); then, for all: bin_used | used; then the counters:
  // per class n_items, n_blocks, n_blocks of the bins a band can reach, 0; for all n_used, unsupported, handed over, 0
  constexpr size_t kHallClassWords = 4 * (kHallBins + 1);                   // (and bin_block: the first block of each bin)
  const size_t hall_word_count = plan_classes * kHallClassWords + 2 * (kHallBins + 1) + (plan_classes + 1) * 4;
  const size_t o_hall_words = plan.add(by_moments_planned ? hall_word_count * sizeof(uint32_t) : 0);
  const size_t o_hall_items = plan.add(plan_classes * plan_items * sizeof(HallItem));
  const size_t o_hall_item_blocks = plan.add(plan_classes * (plan_items + 1) * sizeof(uint32_t));
  const size_t o_hall_ys = plan.add(loglik_moments ? plan_classes * (hall_blocks + 1) * kHallBlockLoci * sizeof(double) : 0);   // per class: every slot's frequency
  // the matrix-core class pass (k_hall_mfma): per slot 32 digit bytes and its row
  const bool hall_mfma = by_moments_planned && eval_gpl == 8 && !env_int("KGX_K7_CLASS_SWEEPS", 0);
  const size_t o_hall_digits = plan.add(hall_mfma ? plan_classes * (hall_blocks + 1) * kHallBlockLoci * 32 : 0);
  const size_t o_hall_rows = plan.add(hall_mfma ? plan_classes * (hall_blocks + 1) * kHallBlockLoci * sizeof(uint32_t) : 0);
  // ... fed by ONE pass over the bytes that leaves every class's hits as bits (k_class_bits): where each selected locus sits in
  // each class's blocks, and the bit rows themselves (a grow-only buffer of their own: an eighth of the bytes the classes cover)
  const bool hall_bits_planned = hall_mfma && !env_int("KGX_K7_CLASS_BYTES", 0);
  const uint64_t sel_pitch = (n_sel + 7) / 8 * 8;
  const size_t o_slot_of_locus = plan.add(hall_bits_planned ? plan_classes * sel_pitch * sizeof(uint32_t) : 0);
  const uint64_t bit_row_bytes = hall_bit_row_bytes(n);                        // whole spans of 2048 genomes
  const size_t hall_sort_stride = (hall_sort_bytes + 255) / 256 * 256;
  const size_t o_hall_sort = plan.add(2 * hall_sort_stride);
  const size_t o_hall_moments = plan.add(plan_items * kHallMoments * n * sizeof(double));
  const size_t o_hall_bins = plan.add(by_moments_planned ? static_cast<size_t>(kHallBins) * kHallMoments * n * sizeof(double) : 0);
  const size_t o_needs_passes = plan.add(loglik_moments ? n * sizeof(uint32_t) : 0);
  const size_t o_smallest_het = plan.add(sizeof(unsigned long long)), o_search_stats = plan.add(8 * sizeof(unsigned long long));
  char* arena = nullptr;
  if (int arc = scratch_reserve(dev, plan.total, &arena)) return arc;
  d_af = reinterpret_cast<double*>(arena + o_af);
  d_table = reinterpret_cast<double*>(arena + o_table);
  d_valid = reinterpret_cast<uint8_t*>(arena + o_valid);
  d_meta = reinterpret_cast<uint32_t*>(arena + o_meta);
  d_part = reinterpret_cast<double*>(arena + o_part);
  d_segdef = reinterpret_cast<double*>(arena + o_segdef);
  d_sums = reinterpret_cast<double*>(arena + o_sums);
  d_counts = reinterpret_cast<unsigned long long*>(arena + o_counts);
  d_f = reinterpret_cast<double*>(arena + o_f);
  d_eval = reinterpret_cast<double*>(arena + o_eval);
  d_out = reinterpret_cast<LocusResultsDev*>(arena + o_out);
  d_golden = reinterpret_cast<GoldenState*>(arena + o_golden);
  d_brent = reinterpret_cast<BrentState*>(arena + o_brent);
  d_running = reinterpret_cast<unsigned int*>(arena + o_running);
  double* d_start = reinterpret_cast<double*>(arena + o_start);
  double* d_seq_sum = reinterpret_cast<double*>(arena + o_seq_sum);
  int* d_seq_e = reinterpret_cast<int*>(arena + o_seq_e);
  long long* d_seq_n = reinterpret_cast<long long*>(arena + o_seq_n);
  double* d_seq_out = reinterpret_cast<double*>(arena + o_seq_out);
  EvalEntry* d_entries = reinterpret_cast<EvalEntry*>(arena + o_entries);
  unsigned long long* d_smallest_het = reinterpret_cast<unsigned long long*>(arena + o_smallest_het);
  unsigned long long* d_segcnt = reinterpret_cast<unsigned long long*>(arena + o_segcnt);
  // (the per-locus bit masks are the SWAR sweeps' alone: k_locus_bits)
  if (!table_sweep) try_hip(hipMemsetAsync(d_meta, 0, (n_tab + 8) * sizeof(uint32_t), dev.stream), KGX_EHIP, "memset(meta)");
  if (locus_index && n_sel) d_index = reinterpret_cast<uint32_t*>(arena + o_index);
  hipStream_t st = dev.stream;
  // counts, F and (if the call has one: its 8 entries of padding stay 0) the locus index
  try_hip(hipMemsetAsync(arena + o_counts, 0, (d_index ? o_cleared_end : o_index) - o_counts, st), KGX_EHIP, "memset(counts, f, index)");
  if (rc == KGX_OK && n_sel) {
    // hipMemcpyDefault: the caller's tables may live on the host or already on this device (kgx.h)
    try_hip(hipMemcpyAsync(d_af, minor_af, n_sel * amax * sizeof(double), hipMemcpyDefault, st), KGX_EHIP, "copy(af)");
    if (d_index) try_hip(hipMemcpyAsync(d_index, locus_index, n_sel * sizeof(uint32_t), hipMemcpyDefault, st), KGX_EHIP, "copy(index)");
  }
  // (the table sweep and the 16-genome SWAR sweep pre-fill every partial with the segment defaults: k_fill_defaults)
  if (!(n_sel && (table_sweep || swar16))) try_hip(hipMemsetAsync(d_part, 0, n_seg * n * kParts0 * sizeof(double), st), KGX_EHIP, "memset(partials)");
  // Where the iterative estimators start (kgx.h): the caller's per-genome points -- the reference draws them, and its
  // fifth draw alone decides the result (kgx_inbreed_reference_starts) -- or the midpoint of the reference's start interval.
  std::vector<double> start_points;
  if (algorithm == KGX_ALGO_HALL_ME || algorithm == KGX_ALGO_LOGLIKELIHOOD) {
    if (start) start_points.assign(start, start + n);
    else start_points.assign(n, algorithm == KGX_ALGO_HALL_ME ? 0.25 : 0.0);
    try_hip(hipMemcpyAsync(d_start, start_points.data(), n * sizeof(double), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(start)");
  }

  const dim3 grid(gx, static_cast<uint32_t>(n_seg));
  const uint32_t* gt32 = reinterpret_cast<const uint32_t*>(sh.d_gt);
  const uint64_t dwords_per_row = sh.pitch / 4;
  // The SWAR sweeps guard against allele indexes 8..14 (past their 8-entry tables) only if the matrix holds any: looked
  // up once per content of the matrix, by one pass over its bytes (KGX_K5_ALWAYS_GUARD=1 skips the look and guards).
  bool guard = true;
  if (rc == KGX_OK && n_sel && (swar16 || amax <= 4 || table_passes) && env_int("KGX_K5_ALWAYS_GUARD", 0) == 0) {
    if (sh.wide_nibbles == 0) {
      unsigned int found = 0;
      const uint64_t n_chunks = sh.n_loci * (sh.pitch / 16);
      if (rc == KGX_OK) {
        hipLaunchKernelGGL(k_scan_wide_nibbles, dim3(stream_grid(dev, n_chunks, kBlock)), dim3(kBlock), 0, st, reinterpret_cast<const kgx_v4u*>(sh.d_gt),
                           n_chunks, d_running);
        try_hip(hipGetLastError(), KGX_EHIP, "k_scan_wide_nibbles launch");
        try_hip(hipMemcpyAsync(&found, d_running, sizeof(found), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(scan flag)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        if (found) try_hip(hipMemsetAsync(d_running, 0, sizeof(unsigned int), st), KGX_EHIP, "memset(scan flag)");   // cleared again for the search
      }
      if (rc == KGX_OK) sh.wide_nibbles = found ? 2 : 1;
    }
    guard = sh.wide_nibbles != 1;
  }
  // the table passes fold bit 7 of a byte onto bit 3 only where an index 8..14, or 7 as a real one, can occur
  const bool eval_fold = guard || amax > 6;
  // The table passes' 1-D grid.  KGX_K5_XCDS=8 deals the segments over the XCDs so that the genome chunks of a segment
  // share an L2 (see the kernel).  Measured at C5: the frequency sweep the same to 1 %, HallME / Loglikelihood passes 4 %
  // SLOWER than the plain order (the re-read entries come out of the memory-side cache either way, and the plain order
  // keeps the eight XCDs on one stretch of the matrix) -- so the plain order (1) is the default.
  const uint32_t eval_xcds = static_cast<uint32_t>(std::max(1, env_int("KGX_K5_XCDS", 1)));
  auto eval_grid = [&](uint64_t genomes, uint64_t per_lane, uint64_t segments) {
    const uint64_t chunks = ((genomes + per_lane - 1) / per_lane + kBlock - 1) / kBlock;
    return static_cast<uint32_t>(chunks * ((segments + eval_xcds - 1) / eval_xcds) * eval_xcds);
  };
  auto tabulate = [&](int mode, hipStream_t ts = nullptr) {
    if (!ts) ts = dev.stream;
    if (!table_passes) return;
    const uint32_t tab_grid = stream_grid(dev, n_sel << (2u * eval_bits(amax)), kBlock);
    if (mode == 1) hipLaunchKernelGGL((k_eval_entries<1>), dim3(tab_grid), dim3(kBlock), 0, ts, d_table, d_valid, n_sel, amax, phased, d_entries, nullptr);
    else if (mode == 2) hipLaunchKernelGGL((k_eval_entries<2>), dim3(tab_grid), dim3(kBlock), 0, ts, d_table, d_valid, n_sel, amax, phased, d_entries, nullptr);
    else if (mode == 3) hipLaunchKernelGGL((k_eval_entries<3>), dim3(tab_grid), dim3(kBlock), 0, ts, d_table, d_valid, n_sel, amax, phased, d_entries, nullptr);
    else if (mode == 5) hipLaunchKernelGGL((k_eval_entries<5>), dim3(tab_grid), dim3(kBlock), 0, ts, d_table, d_valid, n_sel, amax, phased, d_entries, d_smallest_het);
    else if (mode == 6) hipLaunchKernelGGL((k_eval_entries<6>), dim3(tab_grid), dim3(kBlock), 0, ts, d_table, d_valid, n_sel, amax, phased, d_entries, d_smallest_het);
    else hipLaunchKernelGGL((k_eval_entries<4>), dim3(tab_grid), dim3(kBlock), 0, ts, d_table, d_valid, n_sel, amax, phased, d_entries, nullptr);
  };
  bool ll_pair = false;                                       // Loglikelihood passes with two values of F per genome (set by its driver below)
  auto sweep = [&](int mode) {
    if (n_sel == 0) return;
    if (mode == 0 && sequential_defaults) {
      // beside the sweep, on the side stream: it reads the per-locus table only, and only the final reduction reads its sums
      const dim3 seq_grid(static_cast<uint32_t>(n_seq_blocks));
      hipStream_t side = dev.side_stream;
      try_hip(hipEventRecord(dev.side_begin, st), KGX_EHIP, "hipEventRecord");
      try_hip(hipStreamWaitEvent(side, dev.side_begin, 0), KGX_EHIP, "hipStreamWaitEvent");
      hipLaunchKernelGGL(k_seq_block_sums, seq_grid, dim3(kBlock), 0, side, d_table, d_valid, n_sel, amax, d_seq_sum);
      hipLaunchKernelGGL(k_seq_block_predict, dim3(1), dim3(kBlock), 0, side, d_seq_sum, n_seq_blocks, d_seq_e);
      hipLaunchKernelGGL(k_seq_block_quantize, seq_grid, dim3(kBlock), 0, side, d_table, d_valid, n_sel, amax, d_seq_e, d_seq_n);
      hipLaunchKernelGGL(k_seq_chain, dim3(1), dim3(kBlock), 0, side, d_table, d_valid, n_sel, amax, d_seq_e, d_seq_n, n_seq_blocks, d_seq_out);
      try_hip(hipEventRecord(dev.side_end, side), KGX_EHIP, "hipEventRecord");
    }
    if (mode == 0 && table_sweep) {
      // segment defaults into the partials, then the one table pass (below, as mode 3 or 4) corrects and counts
      hipLaunchKernelGGL(k_segment_defaults, dim3(static_cast<uint32_t>(n_seg)), dim3(kWave), 0, st, d_table, d_valid, n_sel, per_seg, amax, sequential_defaults ? 1 : 0, d_segdef);
      hipLaunchKernelGGL(k_fill_defaults, dim3(stream_grid(dev, n_seg * n, kBlock)), dim3(kBlock), 0, st, d_segdef, n_seg, n, d_part);
    } else if (mode == 0) {
      if (swar16) {
        hipLaunchKernelGGL(k_locus_bits, dim3(stream_grid(dev, n_sel, kBlock)), dim3(kBlock), 0, st, d_table, d_valid, n_sel, amax, d_meta);
        hipLaunchKernelGGL(k_segment_defaults, dim3(static_cast<uint32_t>(n_seg)), dim3(kWave), 0, st, d_table, d_valid, n_sel, per_seg, amax, sequential_defaults ? 1 : 0, d_segdef);
        hipLaunchKernelGGL(k_fill_defaults, dim3(stream_grid(dev, n_seg * n, kBlock)), dim3(kBlock), 0, st, d_segdef, n_seg, n, d_part);
        const dim3 grid16(gx16, static_cast<uint32_t>(n_seg));
        const kgx_v4u* gt128 = reinterpret_cast<const kgx_v4u*>(sh.d_gt);
#define KGX_SWAR16(INDEXED, GUARD)                                                                                                  \
  hipLaunchKernelGGL((k_inbreed_sweep_swar16<INDEXED, GUARD>), grid16, dim3(kBlock), 0, st, gt128, sh.pitch / 16, g0, n, d_index, n_sel, \
                     per_seg, d_table, d_meta, amax, phased, d_segdef, d_counts, d_part)
        if (d_index) { if (guard) KGX_SWAR16(true, true); else KGX_SWAR16(true, false); }
        else { if (guard) KGX_SWAR16(false, true); else KGX_SWAR16(false, false); }
#undef KGX_SWAR16
      } else if (env_int("KGX_K5_GENERIC", 0) || amax > 4 || algorithm == KGX_ALGO_RITLAND_LOCUS) {
        hipLaunchKernelGGL((k_inbreed_sweep<0>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel, per_seg, d_table,
                           d_valid, amax, phased, d_f, d_counts, d_part, sh.d_wide_of_row, sh.d_wide, sh.wide_pitch);
      } else {
        hipLaunchKernelGGL(k_locus_bits, dim3(stream_grid(dev, n_sel, kBlock)), dim3(kBlock), 0, st, d_table, d_valid, n_sel, amax, d_meta);
        hipLaunchKernelGGL(k_segment_defaults, dim3(static_cast<uint32_t>(n_seg)), dim3(kWave), 0, st, d_table, d_valid, n_sel, per_seg, amax, sequential_defaults ? 1 : 0, d_segdef);
#define KGX_SWAR(INDEXED, GUARD)                                                                                                   \
  hipLaunchKernelGGL((k_inbreed_sweep_swar<INDEXED, GUARD>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel,    \
                     per_seg, d_table, d_meta, amax, phased, d_segdef, d_counts, d_part)
        if (d_index) { if (guard) KGX_SWAR(true, true); else KGX_SWAR(true, false); }
        else { if (guard) KGX_SWAR(false, true); else KGX_SWAR(false, false); }
#undef KGX_SWAR
      }
    } else if (eval_lut || mode >= 3) {
      // the frequency sweeps (modes 3, 4) over their own segments, the estimator passes over theirs
      const uint64_t launch_n_seg = mode >= 3 ? n_seg : eval_n_seg, launch_per_seg = mode >= 3 ? per_seg : eval_per_seg;
#define KGX_EVAL(M, W, FOLD)                                                                                                         \
  hipLaunchKernelGGL((k_inbreed_eval_lut<M, W, FOLD>), dim3(eval_grid(n, W, launch_n_seg)),                                         \
                     dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel,                                               \
                     launch_per_seg, d_entries, d_table, d_valid, amax, d_f, d_part, d_counts, d_segcnt, eval_xcds)
#define KGX_EVAL_FOLD(M, W)                                                \
  do {                                                                     \
    if (eval_fold) KGX_EVAL(M, W, true); else KGX_EVAL(M, W, false);       \
  } while (0)
      if (mode == 1) {
        if (eval_gpl == 8) KGX_EVAL_FOLD(1, 8); else KGX_EVAL_FOLD(1, 4);
      } else if (mode == 2 && ll_pair) {
#define KGX_EVAL_PAIR(W, FOLD)                                                                                                      \
  hipLaunchKernelGGL((k_inbreed_eval_lut<2, W, FOLD, true>), dim3(eval_grid(n, W, launch_n_seg)),                                  \
                     dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel,                                               \
                     launch_per_seg, d_entries, d_table, d_valid, amax, d_f, d_part, d_counts, d_segcnt, eval_xcds)
        if (eval_gpl == 8) { if (eval_fold) KGX_EVAL_PAIR(8, true); else KGX_EVAL_PAIR(8, false); }
        else { if (eval_fold) KGX_EVAL_PAIR(4, true); else KGX_EVAL_PAIR(4, false); }
#undef KGX_EVAL_PAIR
      } else if (mode == 2) {
        if (eval_gpl == 8) KGX_EVAL_FOLD(2, 8); else KGX_EVAL_FOLD(2, 4);
      } else if (mode == 3) {
        if (eval_gpl == 8) KGX_EVAL_FOLD(3, 8); else KGX_EVAL_FOLD(3, 4);
      } else {
        if (eval_gpl == 8) KGX_EVAL_FOLD(4, 8); else KGX_EVAL_FOLD(4, 4);
      }
#undef KGX_EVAL_FOLD
#undef KGX_EVAL
    } else if (mode == 1)
      hipLaunchKernelGGL((k_inbreed_sweep<1>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel, per_seg, d_table,
                         d_valid, amax, phased, d_f, d_counts, d_part, sh.d_wide_of_row, sh.d_wide, sh.wide_pitch);
    else
      hipLaunchKernelGGL((k_inbreed_sweep<2>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel, per_seg, d_table,
                         d_valid, amax, phased, d_f, d_counts, d_part, sh.d_wide_of_row, sh.d_wide, sh.wide_pitch);
  };
  // (the moments' buffers and the ordering of the classes)
  uint32_t* const h_words = reinterpret_cast<uint32_t*>(arena + o_hall_words);
  uint32_t* const h_bin_used = h_words + plan_classes * kHallClassWords;
  uint32_t* const h_used = h_bin_used + (kHallBins + 1);
  uint32_t* const h_counters = h_used + (kHallBins + 1);                       // [class][4]: n_items, n_blocks; then n_used, unsupported, handed over
  uint32_t* const h_totals = h_counters + plan_classes * 4;
  double* const h_bins = reinterpret_cast<double*>(arena + o_hall_bins);
  std::vector<uint32_t> class_items(plan_classes, 0u), class_blocks(plan_classes, 0u);
  LoglikClasses loglik_classes{};
  bool moments_timed = false, search_timed = false;             // (events recorded: read after the call's last synchronisation)
  auto class_words = [&](uint32_t k) { return h_words + k * kHallClassWords; };
  const bool moments_emit = loglik_moments;                       // (Loglikelihood: the hits of the reachable bins leave as bits too)
  double* h_moments = reinterpret_cast<double*>(arena + o_hall_moments);
  const uint32_t hall_chunks = static_cast<uint32_t>(((n + eval_gpl - 1) / eval_gpl + kBlock - 1) / kBlock);
  // (a bin's workgroups: enough of them for the few bins that hold most loci -- the major allele's -- to fill the chip)
  const uint32_t hall_merge_blocks = static_cast<uint32_t>(std::min<uint64_t>(64, (kHallMoments * n + kBlock - 1) / kBlock));
  // the bins an exact walk can reach (kgx_kernels_loglik.h): those starting below kLoglikReach
  const uint32_t block_bins = moments_emit ? hall_key_host(kLoglikReach) + 1u : 0u;
  const size_t padded_records = static_cast<size_t>(hall_blocks + 1) * kHallBlockLoci;
  auto padded_of = [&](uint32_t k) { return reinterpret_cast<HallRecord*>(arena + o_hall_padded) + static_cast<uint64_t>(k) * padded_records; };
  auto ys_of = [&](uint32_t k) { return moments_emit ? reinterpret_cast<double*>(arena + o_hall_ys) + static_cast<uint64_t>(k) * padded_records : nullptr; };
  auto items_of = [&](uint32_t k) { return reinterpret_cast<HallItem*>(arena + o_hall_items) + static_cast<uint64_t>(k) * hall_items; };
  auto item_blocks_of = [&](uint32_t k) { return reinterpret_cast<uint32_t*>(arena + o_hall_item_blocks) + static_cast<uint64_t>(k) * (hall_items + 1); };
  auto digits_of = [&](uint32_t k) { return reinterpret_cast<int8_t*>(arena + o_hall_digits) + static_cast<uint64_t>(k) * padded_records * 32; };
  auto rows_of = [&](uint32_t k) { return reinterpret_cast<uint32_t*>(arena + o_hall_rows) + static_cast<uint64_t>(k) * padded_records; };
  uint32_t* const slot_of_locus = hall_bits_planned ? reinterpret_cast<uint32_t*>(arena + o_slot_of_locus) : nullptr;
  std::vector<uint32_t> counters;
  char* counters_pinned = nullptr;                                 // (page-locked)
  // The classes in bin order; the counters the host needs come back behind them (into counters_pinned).  Two classes at a time: the
  // even ones on stream `os`, the odd ones on the side stream with scratch of their own -- ~12 small kernels a class that fill a
  // fraction of the device each (1.9 -> 1.45 ms at C5).
  auto order_classes = [&](hipStream_t os) {
    const bool emit = moments_emit;
    try_hip(hipMemsetAsync(h_words, 0, hall_word_count * sizeof(uint32_t), os), KGX_EHIP, "memset(hall words)");
    // (k_hall_sweep reads up to two batches past an item's slots -- the next item's, or, behind the last one, this: row 0, matching
    // nothing; the matrix-core pass reads its items' blocks alone.  The bins need no clearing: k_hall_merge)
    if (!hall_mfma) try_hip(hipMemsetAsync(arena + o_hall_padded, 0, hall_classes * padded_records * sizeof(HallRecord), os), KGX_EHIP, "memset(hall blocks)");
    if (slot_of_locus) try_hip(hipMemsetAsync(slot_of_locus, 0xFF, static_cast<size_t>(hall_classes) * sel_pitch * sizeof(uint32_t), os), KGX_EHIP, "memset(slot of locus)");
    const bool two_lanes = hall_classes > 1 && os != dev.side_stream && !env_int("KGX_K7_ORDER_ONE_LANE", 0);
    if (two_lanes) {
      try_hip(hipEventRecord(dev.side_begin, os), KGX_EHIP, "hipEventRecord");              // (the tables, the clearing above)
      try_hip(hipStreamWaitEvent(dev.side_stream, dev.side_begin, 0), KGX_EHIP, "hipStreamWaitEvent");
    }
    for (uint32_t k = 0; k < hall_classes && rc == KGX_OK; ++k) {
      const uint32_t lane_of_class = two_lanes ? (k & 1u) : 0u;
      hipStream_t ks = lane_of_class ? dev.side_stream : os;
      uint32_t* h_keys = reinterpret_cast<uint32_t*>(arena + o_hall_keys + lane_of_class * hall_keys_bytes);
      uint32_t *h_slots = h_keys + n_sel, *h_sorted_keys = h_keys + 2 * n_sel, *h_sorted_slots = h_keys + 3 * n_sel;
      HallRecord* const h_records = reinterpret_cast<HallRecord*>(arena + o_hall_records + lane_of_class * hall_records_bytes);
      uint32_t *bin_begin = class_words(k), *bin_end = bin_begin + (kHallBins + 1), *item_base = bin_begin + 2 * (kHallBins + 1);
      hipLaunchKernelGGL(k_hall_keys, dim3(stream_grid(dev, n_sel, kBlock)), dim3(kBlock), 0, ks, d_table, d_valid, n_sel, amax, phased, k, h_keys,
                         h_slots, h_totals + 1);
      size_t sort_bytes = hall_sort_bytes;
      try_hip(hipcub::DeviceRadixSort::SortPairs(arena + o_hall_sort + lane_of_class * hall_sort_stride, sort_bytes, h_keys, h_sorted_keys, h_slots, h_sorted_slots,
                                                 static_cast<int>(n_sel), 0, 12, ks), KGX_EHIP, "radix sort");
      hipLaunchKernelGGL(k_hall_records, dim3(stream_grid(dev, n_sel, kBlock)), dim3(kBlock), 0, ks, h_sorted_keys, h_sorted_slots, n_sel, d_table,
                         amax, k, d_index, h_records, bin_begin, bin_end);
      hipLaunchKernelGGL(k_hall_items, dim3(1), dim3(kBlock), 0, ks, bin_begin, bin_end, item_base, items_of(k), h_counters + 4 * k, block_bins,
                         item_blocks_of(k), h_counters + 4 * k + 1);
      hipLaunchKernelGGL(k_hall_pad, dim3(static_cast<uint32_t>(std::min<uint64_t>(hall_items, 65535))), dim3(kBlock), 0, ks, h_records, items_of(k),
                         h_counters + 4 * k, item_blocks_of(k), padded_of(k), ys_of(k), h_sorted_slots, slot_of_locus ? slot_of_locus + k * sel_pitch : nullptr);
      if (hall_mfma) hipLaunchKernelGGL(k_hall_digits, dim3(static_cast<uint32_t>(std::min<uint64_t>(hall_items, 65535))), dim3(kBlock), 0, ks, padded_of(k),
                                        items_of(k), h_counters + 4 * k, item_blocks_of(k), digits_of(k), rows_of(k));
      if (emit) hipLaunchKernelGGL(k_hall_bin_blocks, dim3((kHallBins + kBlock) / kBlock), dim3(kBlock), 0, ks, item_base, item_blocks_of(k), item_base + (kHallBins + 1));
      try_hip(hipGetLastError(), KGX_EHIP, "hall order launch");
    }
    if (two_lanes) {
      try_hip(hipEventRecord(dev.side_end, dev.side_stream), KGX_EHIP, "hipEventRecord");
      try_hip(hipStreamWaitEvent(os, dev.side_end, 0), KGX_EHIP, "hipStreamWaitEvent");
    }
    if (rc == KGX_OK && pinned_reserve(dev, 1, (hall_classes + 1) * 4 * sizeof(uint32_t), &counters_pinned) != KGX_OK) rc = KGX_ENOMEM;
    if (rc == KGX_OK)
      try_hip(hipMemcpyAsync(counters_pinned, h_counters, (hall_classes + 1) * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, os), KGX_EHIP, "D2H(hall counters)");
  };
  const uint32_t lin_grid = stream_grid(dev, n, kBlock);
  auto reduce_grid = [&](uint64_t items) { return stream_grid(dev, (items + kReduceItems - 1) / kReduceItems * kBlock, kBlock); };
  if (rc == KGX_OK) {
    if (n_sel) hipLaunchKernelGGL((k_locus_tables<true>), dim3(stream_grid(dev, n_sel, kBlock)), dim3(kBlock), 0, st, d_af, n_sel, amax, 0.0, d_table, d_valid);
    // Timed from 2^24 cells up (kgx.h): below, the four event records cost the call more than its sweep takes.
    const bool timed = n_sel * n >= (1ull << 24) || env_int("KGX_TIME_SMALL_CALLS", 0);
    if (timed && rc == KGX_OK) try_hip(hipEventRecord(dev.sweep_begin, st), KGX_EHIP, "hipEventRecord");
    if (timed && !table_sweep && rc == KGX_OK) try_hip(hipEventRecord(dev.kernel_begin, st), KGX_EHIP, "hipEventRecord");
    // (Loglikelihood by moments: the same pass as RitlandLocus', its fp64 term the heterozygous cells' log 2*f1*f2 -- k_eval_entries<5> --
    // where the VALUE of the objective is asked for: kgx_inbreed_objective.  A search climbs the objective without that sum, a constant
    // of the genome: the heterozygous cells' marks alone, through the cheaper <4> pass -- 8.4 instead of 11.7 ms at C5)
    const bool het_terms = loglik_moments && (objective_method == 1 || env_int("KGX_K7_LL_HET_TERMS", 0));
    const int entries_mode = ritland ? 3 : het_terms ? 5 : loglik_moments ? 6 : 4;
    // A large call tabulates the pass's entries on the side stream, beside the segment defaults (both need the tables alone: 0.55 and
    // 0.36 ms at C5, one after the other before)
    const bool entries_beside = table_sweep && n_sel >= 65536 && rc == KGX_OK;
    if (table_sweep && loglik_moments) {
      static const double one = 1.0;                           // the smallest such product of the call, from 1 down
      try_hip(hipMemcpyAsync(d_smallest_het, &one, sizeof(one), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(smallest het)");
    }
    if (entries_beside) {
      try_hip(hipEventRecord(dev.side_begin, st), KGX_EHIP, "hipEventRecord");
      try_hip(hipStreamWaitEvent(dev.side_stream, dev.side_begin, 0), KGX_EHIP, "hipStreamWaitEvent");
      tabulate(entries_mode, dev.side_stream);
      try_hip(hipEventRecord(dev.entries_end, dev.side_stream), KGX_EHIP, "hipEventRecord");
    }
    sweep(0);
    if (table_sweep) {
      if (entries_beside) try_hip(hipStreamWaitEvent(st, dev.entries_end, 0), KGX_EHIP, "hipStreamWaitEvent");
      else tabulate(entries_mode);
      if (timed && rc == KGX_OK) try_hip(hipEventRecord(dev.kernel_begin, st), KGX_EHIP, "hipEventRecord");
      sweep(ritland || het_terms ? 3 : 4);
      if (timed && rc == KGX_OK) try_hip(hipEventRecord(dev.kernel_end, st), KGX_EHIP, "hipEventRecord");
      if (n_sel) {
        const uint32_t seg_rows = n_seg < 32 ? static_cast<uint32_t>(n_seg) : 32u;
        hipLaunchKernelGGL(k_reduce_class_counts, dim3(static_cast<uint32_t>((n + kBlock - 1) / kBlock), seg_rows), dim3(kBlock), 0, st, d_segcnt, n_seg, n, d_counts);
      }
    } else if (timed && rc == KGX_OK) {
      try_hip(hipEventRecord(dev.kernel_end, st), KGX_EHIP, "hipEventRecord");
    }
    if (timed && rc == KGX_OK) try_hip(hipEventRecord(dev.sweep_end, st), KGX_EHIP, "hipEventRecord");
    if (sequential_defaults && n_sel) try_hip(hipStreamWaitEvent(st, dev.side_end, 0), KGX_EHIP, "hipStreamWaitEvent");
    hipLaunchKernelGGL(k_reduce_parts, dim3(reduce_grid(n * kParts0)), dim3(kBlock), 0, st, d_part, n_seg, n * kParts0,
                       (sequential_defaults && n_sel) ? d_seq_out : nullptr, d_sums);
    dev.last_path = KGX_PATH_FREQUENCY_SWEEP;   // (Simple, RitlandLocus: nothing more; the iterative estimators say below what ran)
    // Window-sized calls: the whole iteration in one launch, a block or a wave per genome (k_inbreed_iterate_genome).
    const bool wave_path = (algorithm == 2 || algorithm == 3) && wave_sized;
    bool wave_evaluations = false;
    // Loglikelihood's search: the reference optimiser's own path (Nelder-Mead, see nm_advance) unless KGX_K7_SEARCH=brent
    // ... two evaluations per pass where a pass is a sweep over the matrix (the table passes): same path, half the passes
    const int pass_search = (search == kSearchNelderMead && eval_lut && !env_int("KGX_K7_NO_PAIR", 0)) ? kSearchNelderMeadPair : search;
    const uint64_t planes = pass_search == kSearchNelderMeadPair ? 2 : 1;
    // The moments both estimators run on over a large call (kgx_kernels_hall.h): per class of homozygous cell the selected loci in
    // bin order (keys, radix sort, records, items) -- first for every class, so that what the host must know comes back in ONE
    // read before any pass over the bytes: a frequency without a bin (below 2^-20, above 1: `false`, the caller makes the
    // passes instead and has lost a millisecond, not the class passes), the items of each class (the passes' grids) and, with
    // `emit` (Loglikelihood), the blocks whose hits the passes leave as bits -- then one pass over the bytes per class and the
    // merge of its items into the bins.
    auto gather_moments = [&](bool emit) -> bool {
      // (measured: on the side stream beside the frequency sweep, even at the highest priority, the ordering's ~50 small kernels
      // wait for the sweep's workgroups to leave -- one radix-sort kernel 5 ms -- and the call is no shorter)
      order_classes(st);
      try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      if (rc == KGX_OK) counters.assign(reinterpret_cast<const uint32_t*>(counters_pinned), reinterpret_cast<const uint32_t*>(counters_pinned) + (hall_classes + 1) * 4);
      if (rc != KGX_OK || counters[hall_classes * 4 + 1] != 0u) return false;
      uint64_t all_blocks = 0;
      for (uint32_t k = 0; k < hall_classes; ++k) {
        class_items[k] = counters[4 * k];
        class_blocks[k] = counters[4 * k + 2];                                  // the blocks whose hits are kept
        all_blocks += class_blocks[k];
      }
      unsigned long long* words = nullptr;
      const uint64_t word_blocks = (all_blocks + 16) / 16 * 16;                 // (a genome's run of words: whole lines)
      if (emit) {
        // the hits' words: [block][genome], the classes' blocks one after the other; kept between calls like the arena
        const size_t want = word_blocks * words_per_block * sizeof(unsigned long long);
        if (dev.words_bytes < want) {
          if (dev.words) (void)hipFree(dev.words);
          dev.words = nullptr;
          dev.words_bytes = 0;
          size_t free_bytes = 0, total_bytes = 0;
          if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < want + (4ull << 30) || hipMalloc(&dev.words, want) != hipSuccess) {
            (void)hipGetLastError();
            dev.words = nullptr;
            return false;                                                       // no room: the passes
          }
          dev.words_bytes = want;
        }
        words = reinterpret_cast<unsigned long long*>(dev.words);
      }
      // the classes' bit rows: all blocks of all classes (a slot a row)
      bool by_bits = hall_bits_planned;
      HallClassRows class_rows{};
      if (by_bits) {
        uint64_t rows = 0;
        for (uint32_t k = 0; k < hall_classes; ++k) { class_rows.base[k] = rows; rows += static_cast<uint64_t>(counters[4 * k + 1]) * kHallBlockLoci; }
        const size_t want = static_cast<size_t>(rows ? rows : 1) * bit_row_bytes + 64;
        if (dev.bits_bytes < want) {
          if (dev.bits) (void)hipFree(dev.bits);
          dev.bits = nullptr;
          dev.bits_bytes = 0;
          size_t free_bytes = 0, total_bytes = 0;
          if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < want + (4ull << 30) || hipMalloc(&dev.bits, want) != hipSuccess) {
            (void)hipGetLastError();
            dev.bits = nullptr;
            by_bits = false;                                                    // no room: a pass over the bytes per class
          } else {
            dev.bits_bytes = want;
          }
        }
      }
      uint64_t block_base = 0;
      try_hip(hipEventRecord(dev.moments_begin, st), KGX_EHIP, "hipEventRecord");
      if (by_bits && rc == KGX_OK) {
        const uint32_t bit_chunks = static_cast<uint32_t>((n + kBitsSpanGenomes - 1) / kBitsSpanGenomes);   // a wave a span
        uint64_t segments = std::max<uint64_t>(1, static_cast<uint64_t>(dev.compute_units) * 32 / bit_chunks);   // (waves: four to a workgroup's worth)
        uint64_t loci_per_seg = std::max<uint64_t>(64, (n_sel + segments - 1) / segments);
        loci_per_seg = (loci_per_seg + 7) / 8 * 8;
        segments = std::min<uint64_t>(65535, (n_sel + loci_per_seg - 1) / loci_per_seg);
        loci_per_seg = ((n_sel + segments - 1) / segments + 7) / 8 * 8;
        segments = (n_sel + loci_per_seg - 1) / loci_per_seg;
        hipLaunchKernelGGL(k_class_bits, dim3(bit_chunks, static_cast<uint32_t>(segments)), dim3(kWave), 0, st, gt32, dwords_per_row, g0, (g0 + n - 1) >> 2, d_index, n_sel,
                           loci_per_seg, slot_of_locus, sel_pitch, hall_classes, class_rows, bit_row_bytes, reinterpret_cast<uint8_t*>(dev.bits));
        try_hip(hipGetLastError(), KGX_EHIP, "class bits launch");
      }
      for (uint32_t k = 0; k < hall_classes && rc == KGX_OK; ++k) {
        uint32_t* item_base = class_words(k) + 2 * (kHallBins + 1);
        // (the words of a class: behind those of the classes before it -- block-major, or, from the bit rows, within every genome's run)
        unsigned long long* class_out = emit ? words + block_base * (by_bits && hall_mfma ? 1 : words_per_block) : nullptr;
        const uint64_t block_base_of_class = block_base;
        if (emit) {
          loglik_classes.of[k] = LoglikClass{ys_of(k), item_base + (kHallBins + 1), class_out};
          block_base += class_blocks[k];
        }
        if (class_items[k] == 0u) continue;                                     // no locus has a homozygote of this class
        const dim3 sweep_grid(class_items[k] * hall_chunks);
        const uint32_t code = k | (k << 4);
#define KGX_HALL_SWEEP(GPL, EMIT)                                                                                                      \
  hipLaunchKernelGGL((k_hall_sweep<GPL, EMIT>), sweep_grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, padded_of(k), items_of(k), \
                     h_counters + 4 * k, hall_chunks, code, h_moments, item_blocks_of(k), block_bins, words_per_block, class_out)
        if (hall_mfma) {
          // the class pass on the matrix cores: a workgroup = an item x 512 genomes
          const uint32_t mfma_chunks = static_cast<uint32_t>(by_bits ? bit_row_bytes / 128 : (n + 511) / 512);   // (bit rows: 8 tiles of 16 bytes a workgroup)
          const dim3 mfma_grid(class_items[k] * mfma_chunks);
          const uint8_t* bit_rows = by_bits ? reinterpret_cast<const uint8_t*>(dev.bits) + class_rows.base[k] * bit_row_bytes : nullptr;
#define KGX_HALL_MFMA(EMIT, BITS)                                                                                                      \
  hipLaunchKernelGGL((k_hall_mfma<EMIT, BITS>), mfma_grid, dim3(BITS ? 2 * kBlock : kBlock), 0, st, gt32, dwords_per_row, g0, n, rows_of(k), bit_rows, bit_row_bytes, \
                     digits_of(k), items_of(k), h_counters + 4 * k, item_blocks_of(k), mfma_chunks, code, h_moments, block_bins,        \
                     words_per_block, class_out, word_blocks, static_cast<uint32_t>(block_base_of_class & 7u))
          if (by_bits) { if (emit) KGX_HALL_MFMA(true, true); else KGX_HALL_MFMA(false, true); }
          else { if (emit) KGX_HALL_MFMA(true, false); else KGX_HALL_MFMA(false, false); }
#undef KGX_HALL_MFMA
        } else if (eval_gpl == 8) { if (emit) KGX_HALL_SWEEP(8, true); else KGX_HALL_SWEEP(8, false); }
        else KGX_HALL_SWEEP(4, false);                                          // (the hits' words are a lane of eight genomes': loglik_candidate)
#undef KGX_HALL_SWEEP
        hipLaunchKernelGGL(k_hall_merge, dim3(hall_merge_blocks, 256), dim3(kBlock), 0, st, h_moments, item_base, n, h_bins, h_bin_used, k + 1u);
        try_hip(hipGetLastError(), KGX_EHIP, "hall moments launch");
      }
      loglik_classes.n = emit ? hall_classes : 0u;
      loglik_classes.block_bins = block_bins;
      loglik_classes.plain_words = (by_bits && hall_mfma) ? 1u : 0u;
      loglik_classes.word_blocks = word_blocks;
      loglik_classes.dense = env_int("KGX_K7_LL_DENSE_PER_1024", 0) > 0 ? env_int("KGX_K7_LL_DENSE_PER_1024", 0) / 1024.0 : kLoglikDense;
      try_hip(hipEventRecord(dev.moments_end, st), KGX_EHIP, "hipEventRecord");
      moments_timed = true;
      hipLaunchKernelGGL(k_hall_used_bins, dim3(1), dim3(kBlock), 0, st, h_bin_used, h_used, h_totals);
      try_hip(hipGetLastError(), KGX_EHIP, "hall bins launch");
      return rc == KGX_OK;
    };
    if (wave_path) {
      // A block per genome while that leaves SIMDs idle or with a wave or two (latency-bound: four waves share a
      // genome's cells); a wave per genome from KGX_K7_WAVE_GENOMES genomes (throughput-bound: a block would repeat the
      // log, the search step and the reduction in all four waves).  Measured at 1000 loci (scripts/exp_window_threads.sh,
      // ms per call, block / wave): HallME 0.127 / 0.162 at 256 genomes, 0.144 / 0.154 at 512, 0.173 / 0.170 at 1024,
      // 0.335 / 0.259 at 2504; Loglikelihood 0.150 / 0.161, 0.170 / 0.157, 0.207 / 0.176, 0.315 / 0.231.
      // ... and only up to KGX_K7_WAVE_LOCI selected loci (default 1024: 16 cells per lane): with 32 cells per lane the
      // kernel holds 218 registers -- two waves per SIMD, each walking 32 dependent cells per pass -- and a 2000-locus call
      // took 2.7 ms at 2504 genomes where the block per genome takes 4 x the loci in a quarter of that.
      const uint64_t wave_loci = static_cast<uint64_t>(std::min(kGenomeWaveLoci, std::max(1, env_int("KGX_K7_WAVE_LOCI", 1024))));
      const bool per_wave = n_sel <= wave_loci &&
                            n >= static_cast<uint64_t>(std::max(1, env_int("KGX_K7_WAVE_GENOMES", algorithm == 2 ? 1024 : 512)));
      const uint32_t wave_grid = static_cast<uint32_t>(per_wave ? (n + kBlock / kWave - 1) / (kBlock / kWave) : n);
      const double* estimate = algorithm == 2 ? d_sums : env_int("KGX_K7_ESTIMATE_START", 0) ? d_sums : nullptr;
#define KGX_WAVE(MODE, CELLS, THREADS)                                                                                             \
  hipLaunchKernelGGL((k_inbreed_iterate_genome<MODE, CELLS, THREADS>), dim3(wave_grid), dim3(kBlock), 0, st, sh.d_gt, sh.pitch, g0, n, d_index, \
                     n_sel, d_table, d_valid, amax, phased, d_counts, estimate, search, d_start, d_f, d_running)
      // the smallest per-thread cell count that holds the selection (see the kernel)
#define KGX_WAVE_CELLS(MODE)                                                                                  \
  do {                                                                                                        \
    if (per_wave) {                                                                                           \
      if (n_sel <= kWave * 8) KGX_WAVE(MODE, 8, kWave);                                                       \
      else if (n_sel <= kWave * 16) KGX_WAVE(MODE, 16, kWave);                                                \
      else KGX_WAVE(MODE, 32, kWave);                                                                         \
    } else {                                                                                                  \
      if (n_sel <= kBlock * 2) KGX_WAVE(MODE, 2, kBlock);                                                     \
      else if (n_sel <= kBlock * 4) KGX_WAVE(MODE, 4, kBlock);                                                \
      else if (n_sel <= kBlock * 8) KGX_WAVE(MODE, 8, kBlock);                                                \
      else if (n_sel <= kBlock * 16) KGX_WAVE(MODE, 16, kBlock);                                              \
      else KGX_WAVE(MODE, 32, kBlock);                                                                        \
    }                                                                                                         \
  } while (0)
      if (algorithm == 2) KGX_WAVE_CELLS(1); else KGX_WAVE_CELLS(2);
#undef KGX_WAVE_CELLS
#undef KGX_WAVE
      wave_evaluations = algorithm == 3;          // its count (d_running, cleared with the counts) comes back with the results
      dev.last_path = KGX_PATH_ONE_LAUNCH;
    } else if (algorithm == 2) {
      // processHallME (_calc.cpp:225-307).  The reference restarts from U(0,0.5] and, through RetryCalcResult's
      // self-comparison (_calc.cpp:45-68), always stops after 5 restarts of exactly 50 expectation steps, keeping the
      // last: 50 steps from the start point handed in (the fifth draw, or 0.25 without one).
      bool by_moments = false;
      if (hall_moments && rc == KGX_OK && gather_moments(false)) {
        // HallME on per-genome moments (kgx_kernels_hall.h): one pass over the bytes per class of homozygous cell, then
        // the 50 steps on ~10^3 numbers a genome.
        try_hip(hipEventRecord(dev.search_begin, st), KGX_EHIP, "hipEventRecord");
        hipLaunchKernelGGL(k_hall_iterate, dim3(static_cast<uint32_t>(n)), dim3(kBlock), 0, st, h_bins, h_used, h_totals, d_counts, n, d_start, d_f);
        try_hip(hipGetLastError(), KGX_EHIP, "hall iterate launch");
        try_hip(hipEventRecord(dev.search_end, st), KGX_EHIP, "hipEventRecord");
        search_timed = true;
        by_moments = rc == KGX_OK;
        if (by_moments) dev.last_path = KGX_PATH_HALL_MOMENTS;
      }
      if (!by_moments && rc == KGX_OK) dev.last_path = KGX_PATH_HALL_PASSES;
      if (!by_moments) {
      const std::vector<double>& f0 = start_points;
      // locus slots every lane of k_inbreed_eval_lut walks: whole batches of 8 in every segment
      const unsigned long long walked = eval_lut && n_sel ? (eval_n_seg - 1) * eval_per_seg + (n_sel - (eval_n_seg - 1) * eval_per_seg + 7) / 8 * 8 : 0ull;
      try_hip(hipMemcpyAsync(d_f, f0.data(), n * sizeof(double), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(f0)");
      try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      tabulate(1);
      std::vector<double> trace_prev, trace_prev2, trace_now;
      const bool trace = env_int("KGX_K7_TRACE", 0) != 0;
      if (trace) { trace_prev = f0; trace_prev2 = f0; trace_now.resize(n); }
      for (int it = 0; it < 50 && rc == KGX_OK; ++it) {
        sweep(1);
        hipLaunchKernelGGL(k_reduce_parts, dim3(reduce_grid(n)), dim3(kBlock), 0, st, d_part, pass_n_seg, n, nullptr, d_eval);
        hipLaunchKernelGGL(k_hall_update, dim3(lin_grid), dim3(kBlock), 0, st, d_eval, d_counts, n, walked, d_f);
        if (trace) {
          try_hip(hipMemcpyAsync(trace_now.data(), d_f, n * sizeof(double), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(trace)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
          uint64_t changed = 0, cycling = 0;
          double worst = 0.0;
          for (uint64_t g = 0; g < n; ++g) {
            if (trace_now[g] != trace_prev[g]) {
              ++changed;
              if (trace_now[g] == trace_prev2[g]) ++cycling;
              const double d = std::fabs(trace_now[g] - trace_prev[g]);
              worst = d > worst ? d : worst;
            }
          }
          std::fprintf(stderr, "[kgx] HallME step %d: %llu of %llu genomes moved (%llu back to the value before), largest move %.3e\n", it + 1,
                       (unsigned long long)changed, (unsigned long long)n, (unsigned long long)cycling, worst);
          trace_prev2 = trace_prev;
          trace_prev = trace_now;
        }
      }
      }
    } else if (algorithm == 3) {
      // processLogLikelihood (_calc.cpp:153-216): maximise over [-1,1] on the reference optimiser's own path (nm_advance).
      // Over a large call the objective comes from per-genome statistics (kgx_kernels_loglik.h) and the whole search of a
      // genome runs in one workgroup; the passes over the bytes (below) serve what the statistics cannot (a frequency outside
      // the bins; a genome with a heterozygous cell that can meet either bound), the other searches (KGX_K7_SEARCH=brent,
      // KGX_K7_GOLDEN=1), amax > 7, unphased populations, and the comparison (KGX_K7_LL_PASSES=1).
      bool by_moments = false;
      std::vector<uint32_t> handed_over;                              // genomes the statistics could not serve
      uint32_t* d_needs_passes = reinterpret_cast<uint32_t*>(arena + o_needs_passes);
      double* d_keep = reinterpret_cast<double*>(arena + o_keep);
      if (objective_method == 1 && !loglik_moments)
        rc = fail(KGX_ESTATE, "kgx_inbreed_objective: this call would not run on the moments (selection size, amax, phase, memory or a switch)");
      if (loglik_moments && objective_method != 2 && rc == KGX_OK && gather_moments(true)) {
        uint32_t n_used = 0;                                        // (the search keeps the genome's bins in LDS: as much of it as they need)
        try_hip(hipMemcpyAsync(&n_used, h_totals, sizeof(uint32_t), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(used bins)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        unsigned long long* d_search_stats = env_int("KGX_K7_TRACE", 0) ? reinterpret_cast<unsigned long long*>(arena + o_search_stats) : nullptr;
        if (d_search_stats) try_hip(hipMemsetAsync(d_search_stats, 0, 8 * sizeof(unsigned long long), st), KGX_EHIP, "memset(search stats)");
        try_hip(hipEventRecord(dev.search_begin, st), KGX_EHIP, "hipEventRecord");
        hipLaunchKernelGGL(k_loglik_search, dim3(static_cast<uint32_t>(n)), dim3(kBlock), loglik_search_lds(n_used), st, h_bins, h_used, h_totals, d_counts, d_sums,
                           d_smallest_het, n, words_per_block, loglik_classes, d_start, objective_method == 1 ? 1 : 0, d_f, d_needs_passes, h_totals + 2, d_running,
                           d_search_stats);
        try_hip(hipGetLastError(), KGX_EHIP, "loglik search launch");
        try_hip(hipEventRecord(dev.search_end, st), KGX_EHIP, "hipEventRecord");
        search_timed = true;
        uint32_t handed = 0;
        unsigned int evaluations = 0;
        try_hip(hipMemcpyAsync(&handed, h_totals + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(handed over)");
        try_hip(hipMemcpyAsync(&evaluations, d_running, sizeof(unsigned int), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(evaluations)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        by_moments = rc == KGX_OK;
        if (by_moments && d_search_stats) {
          unsigned long long stats[8] = {0};
          try_hip(hipMemcpy(stats, d_search_stats, sizeof(stats), hipMemcpyDeviceToHost), KGX_EHIP, "D2H(search stats)");
          std::fprintf(stderr, "kgx: Loglikelihood search over %llu genomes (%u bins used): %llu evaluations, %llu with cells in their band -- %llu served by kept cells, "
                               "%llu gathers kept, %llu gathers not kept, %llu dense walks; %llu cells listed, %llu blocks looked at\n",
                       (unsigned long long)n, n_used, stats[0], stats[1], stats[2], stats[3], stats[4], stats[5], stats[6], stats[7]);
        }
        if (by_moments) {
          dev.last_evaluations = static_cast<int>(evaluations);
          dev.last_path = KGX_PATH_LOGLIK_MOMENTS;
        }
        if (by_moments && handed) {
          std::vector<uint32_t> flags(n);
          try_hip(hipMemcpyAsync(flags.data(), d_needs_passes, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(needs passes)");
          try_hip(hipMemcpyAsync(d_keep, d_f, n * sizeof(double), hipMemcpyDeviceToDevice, st), KGX_EHIP, "copy(kept results)");
          try_hip(hipMemsetAsync(d_running, 0, sizeof(unsigned int), st), KGX_EHIP, "memset(running)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
          for (uint64_t g = 0; g < n; ++g)
            if (flags[g]) {
              handed_over.push_back(static_cast<uint32_t>(g));
              if (env_int("KGX_K7_TRACE", 0)) std::fprintf(stderr, "kgx: Loglikelihood: genome %llu handed to the passes (reason %u)\n", (unsigned long long)(g0 + g), flags[g]);
            }
          if (rc == KGX_OK) dev.last_path = KGX_PATH_LOGLIK_MOMENTS_AND_PASSES;
        }
      } else if (objective_method == 1 && rc == KGX_OK) {
        rc = fail(KGX_ESTATE, "kgx_inbreed_objective: a frequency without a bin, or no room for the hits' words: the passes would run");
      }
      if (objective_method != 0) {
        // the diagnostic: the objective at the caller's points, by the moments (above) or by ONE table pass
        if (objective_method == 2 && rc == KGX_OK) {
          try_hip(hipMemcpyAsync(d_f, d_start, n * sizeof(double), hipMemcpyDeviceToDevice, st), KGX_EHIP, "copy(points)");
          tabulate(2);
          sweep(2);
          hipLaunchKernelGGL(k_reduce_parts, dim3(reduce_grid(n)), dim3(kBlock), 0, st, d_part, pass_n_seg, n, nullptr, d_eval);
          try_hip(hipGetLastError(), KGX_EHIP, "objective pass launch");
        }
        try_hip(hipMemcpyAsync(objective_out, objective_method == 2 ? d_eval : d_f, n * sizeof(double), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(objective)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        return rc;
      }
      if (by_moments && handed_over.empty()) {
        // done: d_f holds every genome's coefficient
      } else if (!env_int("KGX_K7_GOLDEN", 0)) {
        if (!by_moments && rc == KGX_OK) dev.last_path = KGX_PATH_LOGLIK_PASSES;
        tabulate(2);
        // Start: [-1, 1] from its golden point.  KGX_K7_ESTIMATE_START=1 starts in a window around the Simple estimate
        // instead (brent_start): 11 instead of 15 evaluations on a population with F in [0, 0.1], but where the clamped
        // objective has several local maxima (F < 0) it may settle on another one than a search from the middle does --
        // the reference itself lands on one or another from its random starts -- so it is not the default.
        ll_pair = planes == 2;
        hipLaunchKernelGGL(k_brent_init, dim3(lin_grid), dim3(kBlock), 0, st, d_counts, d_sums, n, env_int("KGX_K7_ESTIMATE_START", 0), pass_search, d_start, d_brent, d_f);
        // (after the moments: only the genomes handed over search here; the others' states say "done" with their result)
        if (by_moments)
          hipLaunchKernelGGL(k_loglik_keep, dim3(lin_grid), dim3(kBlock), 0, st, d_needs_passes, d_keep, n, static_cast<int>(planes), d_brent, d_f);
        // Brent: golden section alone would need 38, and its safeguard keeps that bound; Nelder-Mead: the reference's own cap
        const int kMaxEvaluations = search == kSearchNelderMead ? 500 : 60;
        // The genomes still searching.  When at most half of them are left -- and the call is big enough for it to pay --
        // their genotype columns and states are compacted (dense in the selected loci) and the remaining passes sweep
        // only those: on populations with F of both signs the last genomes need twice the evaluations of the first.
        // A genome's sums do not depend on its neighbours, so the results are bit-identical (KGX_K7_NO_COMPACT=1 to compare).
        uint64_t n_act = n, act_g0 = g0, act_dwords_per_row = dwords_per_row;
        const uint32_t* act_gt = gt32;
        const uint32_t* act_index = d_index;
        BrentState* act_brent = d_brent;
        double* act_f = d_f;
        uint32_t* act_global = nullptr;
        std::vector<uint32_t> global_of(n);
        for (uint64_t g = 0; g < n; ++g) global_of[g] = static_cast<uint32_t>(g);
        std::vector<BrentState> host_states;
        auto evaluate = [&]() {
          if (act_gt == gt32) { sweep(2); return; }
#define KGX_EVAL2(FOLD, PAIR)                                                                                                        \
  hipLaunchKernelGGL((k_inbreed_eval_lut<2, 8, FOLD, PAIR>), dim3(eval_grid(n_act, 8, eval_n_seg)),                                 \
                     dim3(kBlock), 0, st, act_gt, act_dwords_per_row, act_g0, n_act, act_index,                                      \
                     n_sel, eval_per_seg, d_entries, d_table, d_valid, amax, act_f, d_part, d_counts, d_segcnt, eval_xcds)
          if (ll_pair) { if (eval_fold) KGX_EVAL2(true, true); else KGX_EVAL2(false, true); }
          else { if (eval_fold) KGX_EVAL2(true, false); else KGX_EVAL2(false, false); }
#undef KGX_EVAL2
        };
        const bool may_compact = eval_lut && !env_int("KGX_K7_NO_COMPACT", 0);
        bool may_compact_now = may_compact;
        int compaction_level = 0;
        // `running` of the n_act genomes are still searching: gather their columns and states if that pays
        // (whatever the size: the genomes handed over by the moments, a few of many)
        auto compact = [&](unsigned int running, bool whatever_the_size) {
          const uint64_t new_pitch = (static_cast<uint64_t>(running) + 127) / 128 * 128;
          // worth it from ~64 M cells left (a gather costs about one pass); the two knobs are for the tests
          const uint64_t min_genomes = static_cast<uint64_t>(env_int("KGX_K7_COMPACT_MIN_GENOMES", 2048));
          const uint64_t min_cells = static_cast<uint64_t>(env_int("KGX_K7_COMPACT_MIN_CELLS", 1 << 26));
          if (!may_compact_now || running == 0 || static_cast<uint64_t>(running) * 2 > n_act) return;
          if (!whatever_the_size && (n_act < min_genomes || n_sel * static_cast<uint64_t>(running) < min_cells)) return;
          host_states.resize(n_act);
          try_hip(hipMemcpyAsync(host_states.data(), act_brent, n_act * sizeof(BrentState), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(states)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
          if (rc != KGX_OK) return;
          std::vector<uint32_t> columns, new_global;
          for (uint64_t g = 0; g < n_act; ++g)
            if (!host_states[g].done) { columns.push_back(static_cast<uint32_t>(g)); new_global.push_back(global_of[g]); }
          const uint64_t n_new = columns.size();
          // Level k lives in the library's ping-pong buffer k & 1 (what that buffer held, level k - 2, is no longer read);
          // the buffers stay allocated between calls like the scratch arena (kgx_release_scratch frees them).
          ScratchPlan level;
          const size_t o_gt = level.add(n_sel * new_pitch), o_st = level.add(n_new * sizeof(BrentState)), o_nf = level.add(planes * n_new * sizeof(double));
          const size_t o_col = level.add(n_new * sizeof(uint32_t)), o_glob = level.add(n_new * sizeof(uint32_t));
          const int slot = compaction_level & 1;
          if (dev.compact_bytes[slot] < level.total) {
            if (dev.compact[slot]) (void)hipFree(dev.compact[slot]);
            dev.compact[slot] = nullptr;
            dev.compact_bytes[slot] = 0;
            size_t free_bytes = 0, total_bytes = 0;
            if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < level.total + (4ull << 30) ||
                hipMalloc(&dev.compact[slot], level.total) != hipSuccess) {
              (void)hipGetLastError();
              dev.compact[slot] = nullptr;
              may_compact_now = false;                                     // no room: carry on as is
              return;
            }
            dev.compact_bytes[slot] = level.total;
          }
          ++compaction_level;
          char* base = dev.compact[slot];
          uint8_t* new_gt = reinterpret_cast<uint8_t*>(base + o_gt);
          BrentState* new_brent = reinterpret_cast<BrentState*>(base + o_st);
          double* new_f = reinterpret_cast<double*>(base + o_nf);
          uint32_t* new_columns = reinterpret_cast<uint32_t*>(base + o_col);
          uint32_t* d_new_global = reinterpret_cast<uint32_t*>(base + o_glob);
          try_hip(hipMemcpyAsync(new_columns, columns.data(), n_new * sizeof(uint32_t), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(columns)");
          try_hip(hipMemcpyAsync(d_new_global, new_global.data(), n_new * sizeof(uint32_t), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(global)");
          const dim3 gather_grid(static_cast<uint32_t>((new_pitch / 4 + kBlock - 1) / kBlock), static_cast<uint32_t>(std::min<uint64_t>(n_sel, 8192)));
          hipLaunchKernelGGL(k_gather_columns, gather_grid, dim3(kBlock), 0, st, reinterpret_cast<const uint8_t*>(act_gt), act_dwords_per_row * 4,
                             act_g0, act_index, n_sel, new_columns, n_new, new_gt, new_pitch);
          hipLaunchKernelGGL(k_gather_states, dim3(stream_grid(dev, n_new, kBlock)), dim3(kBlock), 0, st, act_brent, act_f, n_act, static_cast<int>(planes),
                             new_columns, n_new, new_brent, new_f);
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");           // columns / new_global leave scope
          act_gt = reinterpret_cast<const uint32_t*>(new_gt);
          act_dwords_per_row = new_pitch / 4;
          act_g0 = 0;
          act_index = nullptr;
          act_brent = new_brent;
          act_f = new_f;
          act_global = d_new_global;
          n_act = n_new;
          global_of.swap(new_global);
        };
        if (by_moments) compact(static_cast<unsigned int>(handed_over.size()), true);
        const int evaluations_by_moments = by_moments ? dev.last_evaluations.load() : 0;
        for (int it = 0; it < kMaxEvaluations && rc == KGX_OK; ++it) {
          evaluate();
          const uint32_t act_grid = stream_grid(dev, n_act, kBlock);
          hipLaunchKernelGGL(k_reduce_parts, dim3(reduce_grid(planes * n_act)), dim3(kBlock), 0, st, d_part, pass_n_seg, planes * n_act, nullptr, d_eval);
          try_hip(hipMemsetAsync(d_running, 0, sizeof(unsigned int), st), KGX_EHIP, "memset(running)");
          hipLaunchKernelGGL(k_brent_step, dim3(act_grid), dim3(kBlock), 0, st, act_brent, d_eval, n_act, it == 0 ? 0 : 1, pass_search, act_f, d_running,
                             act_global, d_f);
          unsigned int running = 0;
          try_hip(hipMemcpyAsync(&running, d_running, sizeof(unsigned int), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(running)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
          dev.last_evaluations = std::max(it + 1, evaluations_by_moments);
          if (env_int("KGX_K7_TRACE", 0)) std::fprintf(stderr, "kgx: Loglikelihood evaluation %d: %u of %llu genomes still searching\n", it + 1, running, (unsigned long long)n_act);
          if (running == 0) break;
          compact(running, false);
        }
        hipLaunchKernelGGL(k_brent_step, dim3(stream_grid(dev, n_act, kBlock)), dim3(kBlock), 0, st, act_brent, d_eval, n_act, 2, pass_search, act_f, d_running, act_global, d_f);
        ll_pair = false;
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      } else {
      if (rc == KGX_OK) dev.last_path = KGX_PATH_LOGLIK_PASSES;
      tabulate(2);
      const double inv_phi = 0.6180339887498949;
      constexpr int kGoldenSteps = 38;     // bracket 2 * 0.618^36 = 6e-8 after the two start-up evaluations
      GoldenState init;
      init.a = -1.0; init.b = 1.0;
      init.c = init.b - inv_phi * (init.b - init.a);
      init.d = init.a + inv_phi * (init.b - init.a);
      init.fc = init.fd = 0.0; init.last_was_c = 0; init.pad = 0;
      std::vector<GoldenState> gs(n, init);
      std::vector<double> f0(n, init.c);
      try_hip(hipMemcpyAsync(d_golden, gs.data(), n * sizeof(GoldenState), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(golden)");
      try_hip(hipMemcpyAsync(d_f, f0.data(), n * sizeof(double), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(f0)");
      try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      for (int it = 0; it < kGoldenSteps && rc == KGX_OK; ++it) {
        sweep(2);
        hipLaunchKernelGGL(k_reduce_parts, dim3(reduce_grid(n)), dim3(kBlock), 0, st, d_part, pass_n_seg, n, nullptr, d_eval);
        hipLaunchKernelGGL(k_golden_step, dim3(lin_grid), dim3(kBlock), 0, st, d_golden, d_eval, n, it < 2 ? it : 2, d_f);
      }
      // coefficient = the better interior point of the final bracket
      if (rc == KGX_OK) {
        try_hip(hipMemcpyAsync(gs.data(), d_golden, n * sizeof(GoldenState), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(golden)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        for (uint64_t g = 0; g < n; ++g) f0[g] = 0.5 * (gs[g].a + gs[g].b);
        try_hip(hipMemcpyAsync(d_f, f0.data(), n * sizeof(double), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(f)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      }
      dev.last_evaluations = kGoldenSteps;
      }
    }
    hipLaunchKernelGGL(k_finish_inbreed, dim3(lin_grid), dim3(kBlock), 0, st, d_counts, d_sums, n, algorithm, d_f, d_out);
    try_hip(hipGetLastError(), KGX_EHIP, "kernel launch");
    try_hip(hipMemcpyAsync(out, d_out, n * sizeof(LocusResultsDev), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(out)");
    unsigned int evaluations = 0;
    if (wave_evaluations) try_hip(hipMemcpyAsync(&evaluations, d_running, sizeof(unsigned int), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(evaluations)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "stream synchronize");
    if (wave_evaluations && rc == KGX_OK) dev.last_evaluations = static_cast<int>(evaluations);
    if (rc == KGX_OK && timed) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, dev.sweep_begin, dev.sweep_end) == hipSuccess) dev.last_sweep_ms = ms;
      if (hipEventElapsedTime(&ms, dev.kernel_begin, dev.kernel_end) == hipSuccess) dev.last_kernel_ms = ms;
    } else if (rc == KGX_OK) {
      dev.last_sweep_ms = dev.last_kernel_ms = 0.0;
    }
    // What a call on moments leaves in the grow-only buffers (the arena: moments, bins, the classes' blocks; the hits' words) stays for
    // the next call like it -- up to a quarter of the device's memory (KGX_KEEP_SCRATCH_GB); past that it is given back now, so that
    // one very large call does not starve what the process allocates next (another matrix, the compaction levels, the caller's own).
    {
      const uint64_t keep = env_int("KGX_KEEP_SCRATCH_GB", -1) >= 0 ? static_cast<uint64_t>(env_int("KGX_KEEP_SCRATCH_GB", 0)) << 30 : dev.hbm_bytes / 4;
      if (static_cast<uint64_t>(dev.scratch_bytes) + dev.words_bytes + dev.bits_bytes > keep) {
        (void)hipStreamSynchronize(st);
        if (dev.bits) (void)hipFree(dev.bits);
        dev.bits = nullptr;
        dev.bits_bytes = 0;
        if (dev.words) (void)hipFree(dev.words);
        dev.words = nullptr;
        dev.words_bytes = 0;
        if (dev.scratch) (void)hipFree(dev.scratch);
        dev.scratch = nullptr;
        dev.scratch_bytes = 0;
        arena = nullptr;                                           // (nothing below touches it)
      }
    }
    if (rc == KGX_OK) {
      float ms = 0.f;
      dev.last_moments_ms = moments_timed && hipEventElapsedTime(&ms, dev.moments_begin, dev.moments_end) == hipSuccess ? ms : 0.0;
      dev.last_search_ms = search_timed && hipEventElapsedTime(&ms, dev.search_begin, dev.search_end) == hipSuccess ? ms : 0.0;
    }
  }
  return rc;
}


}  // namespace
}  // namespace kgx

using namespace kgx;

extern "C" {

kgx_gt8* kgx_gt8_create(uint64_t n_genomes, uint64_t n_loci) {
  std::shared_ptr<Runtime> rt;
  if (require_runtime(rt)) return nullptr;
  if (n_genomes == 0) { fail(KGX_EINVAL, "n_genomes must be > 0"); return nullptr; }
  if (n_loci > 0xFFFFFFFFull) { fail(KGX_EINVAL, "n_loci exceeds the 32-bit locus index"); return nullptr; }
  kgx_gt8* h = new (std::nothrow) kgx_gt8();
  if (!h) { fail(KGX_ENOMEM, "host allocation failed"); return nullptr; }
  h->rt = rt;
  h->n_genomes = n_genomes;
  h->n_loci = n_loci;
  // Contiguous genome shards of whole 128-genome units (a row's pitch; every sweep's widest lane load divides it).
  const uint64_t n_slots = rt->devs.size();
  const uint64_t units = (n_genomes + 127) / 128, per = units / n_slots, extra = units % n_slots;
  uint64_t base = 0;
  try {
    h->shards.reserve(n_slots);
  } catch (const std::exception&) {
    delete h;
    fail(KGX_ENOMEM, "host allocation failed");
    return nullptr;
  }
  for (uint64_t s = 0; s < n_slots; ++s) {
    kgx_gt8_shard sh;
    sh.dev = rt->devs[s].get();
    sh.genome_base = base;
    const uint64_t want = (per + (s < extra ? 1 : 0)) * 128;
    sh.n_genomes = want < n_genomes - base ? want : n_genomes - base;
    sh.n_loci = n_loci;
    sh.pitch = (sh.n_genomes + 127) / 128 * 128;
    base += sh.n_genomes;
    h->shards.push_back(sh);
  }
  for (auto& sh : h->shards) {
    const uint64_t bytes = sh.pitch * n_loci;
    if (!bytes) continue;
    if (use_device(*sh.dev) != KGX_OK || hipMalloc(&sh.d_gt, bytes) != hipSuccess) {
      (void)hipGetLastError();
      fail(KGX_ENOMEM, "hipMalloc of %llu bytes for the %llu x %llu genotype matrix on device %d failed", (unsigned long long)bytes,
           (unsigned long long)n_loci, (unsigned long long)sh.n_genomes, sh.dev->id);
      destroy_shards(h);
      delete h;
      return nullptr;
    }
    if (hipMemsetAsync(sh.d_gt, 0, bytes, sh.dev->stream) != hipSuccess) {
      (void)hipGetLastError();
      fail(KGX_EHIP, "memset of the genotype matrix failed");
      destroy_shards(h);
      delete h;
      return nullptr;
    }
  }
  if (sync_shards(h) != KGX_OK) {
    destroy_shards(h);
    delete h;
    return nullptr;
  }
  return h;
}

void kgx_gt8_destroy(kgx_gt8* h) {
  if (!h) return;
  destroy_shards(h);
  delete h;
}

uint64_t kgx_gt8_genomes(const kgx_gt8* h) { return h ? h->n_genomes : 0; }
uint64_t kgx_gt8_loci(const kgx_gt8* h) { return h ? h->n_loci : 0; }
uint64_t kgx_gt8_sweep_bytes(uint64_t n_genomes, uint64_t n_selected, uint32_t amax) {
  return n_genomes * n_selected + 8ull * amax * n_selected + 80ull * n_genomes;
}
uint32_t kgx_gt8_shards(const kgx_gt8* h) { return h ? static_cast<uint32_t>(h->shards.size()) : 0; }
int kgx_gt8_shard_info(const kgx_gt8* h, uint32_t shard, int* slot, uint64_t* genome_base, uint64_t* n_genomes) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || shard >= h->shards.size()) return fail(KGX_EINVAL, "no such shard");
    const auto& sh = h->shards[shard];
    if (slot) *slot = sh.dev->slot;
    if (genome_base) *genome_base = sh.genome_base;
    if (n_genomes) *n_genomes = sh.n_genomes;
    return KGX_OK;
  });
}

int kgx_gt8_load(kgx_gt8* h, const uint8_t* src, uint64_t g0, uint64_t g1) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || !src) return fail(KGX_EINVAL, "null handle or source");
    if (g0 > g1 || g1 > h->n_genomes) return fail(KGX_EINVAL, "genome range out of bounds");
    if (g0 == g1 || h->n_loci == 0) return KGX_OK;
    const uint64_t L = h->n_loci;
    const int rc = for_each_parallel(h->shards.size(), [&](size_t s) -> int {
      kgx_gt8_shard& sh = h->shards[s];
      const uint64_t lo = g0 > sh.genome_base ? g0 : sh.genome_base;
      const uint64_t hi = g1 < sh.genome_base + sh.n_genomes ? g1 : sh.genome_base + sh.n_genomes;
      if (lo >= hi) return KGX_OK;
      if (int e = use_device(*sh.dev)) return e;
      sh.wide_nibbles = 0;              // the bytes change: look again (kgx_inbreed)
      uint64_t slab = (1ull << 30) / L;
      if (slab < 1) slab = 1;
      const uint64_t max_rows = (hi - lo) < slab ? (hi - lo) : slab;
      uint8_t* d_stage = nullptr;
      KGX_HIP_MEM(hipMalloc(&d_stage, max_rows * L));
      int r = KGX_OK;
      for (uint64_t g = lo; g < hi && r == KGX_OK; g += slab) {
        const uint64_t n = (hi - g) < slab ? (hi - g) : slab;
        if (hipMemcpyAsync(d_stage, src + (g - g0) * L, n * L, hipMemcpyHostToDevice, sh.dev->stream) != hipSuccess) {
          r = fail(KGX_EHIP, "H2D copy of genotype bytes failed");
          break;
        }
        hipLaunchKernelGGL(k_gt8_transpose, dim3(stream_grid(*sh.dev, n * L, kBlock)), dim3(kBlock), 0, sh.dev->stream, d_stage, n, L,
                           g - sh.genome_base, sh.d_gt, sh.pitch);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sh.dev->stream) != hipSuccess)
          r = fail(KGX_EHIP, "genotype transpose kernel failed");
      }
      (void)hipFree(d_stage);
      return r;
    });
    (void)use_device(*h->shards[0].dev);
    return rc;
  });
}

int kgx_gt8_load_rows(kgx_gt8* h, const uint8_t* src, uint64_t src_pitch, uint64_t l0, uint64_t l1) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || !src) return fail(KGX_EINVAL, "null handle or source");
    if (l0 > l1 || l1 > h->n_loci || src_pitch < h->n_genomes) return fail(KGX_EINVAL, "bad locus range or pitch");
    if (l0 == l1) return KGX_OK;
    for (auto& sh : h->shards) {
      if (sh.n_genomes == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      sh.wide_nibbles = 0;
      KGX_HIP(hipMemcpy2DAsync(sh.d_gt + l0 * sh.pitch, sh.pitch, src + sh.genome_base, src_pitch, sh.n_genomes, l1 - l0,
                               hipMemcpyHostToDevice, sh.dev->stream));
    }
    return sync_shards(h);
  });
}

int kgx_gt8_set_wide_rows(kgx_gt8* h, uint64_t n_wide, const uint32_t* locus, const uint16_t* cells, uint64_t cells_pitch) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || (n_wide && (!locus || !cells))) return fail(KGX_EINVAL, "null argument");
    if (n_wide && cells_pitch < h->n_genomes) return fail(KGX_EINVAL, "cells_pitch below the genome count");
    for (uint64_t i = 0; i < n_wide; ++i)
      if (locus[i] >= h->n_loci || (i && locus[i] <= locus[i - 1])) return fail(KGX_EINVAL, "wide rows must name rows of the matrix, ascending");
    std::vector<uint32_t> wide_of_row;
    if (n_wide) {
      wide_of_row.assign(h->n_loci, 0xFFFFFFFFu);
      for (uint64_t i = 0; i < n_wide; ++i) wide_of_row[locus[i]] = static_cast<uint32_t>(i);
    }
    for (auto& sh : h->shards) {
      if (int rc = use_device(*sh.dev)) return rc;
      KGX_HIP(hipStreamSynchronize(sh.dev->stream));
      if (sh.d_wide) (void)hipFree(sh.d_wide);
      if (sh.d_wide_of_row) (void)hipFree(sh.d_wide_of_row);
      sh.d_wide = nullptr; sh.d_wide_of_row = nullptr; sh.n_wide = 0; sh.wide_pitch = 0;
      if (n_wide == 0 || sh.n_genomes == 0) continue;
      sh.wide_pitch = (sh.n_genomes + 7) / 8 * 8;
      KGX_HIP_MEM(hipMalloc(&sh.d_wide_of_row, h->n_loci * sizeof(uint32_t)));
      KGX_HIP_MEM(hipMalloc(&sh.d_wide, n_wide * sh.wide_pitch * sizeof(uint16_t)));
      KGX_HIP(hipMemsetAsync(sh.d_wide, 0, n_wide * sh.wide_pitch * sizeof(uint16_t), sh.dev->stream));
      KGX_HIP(hipMemcpyAsync(sh.d_wide_of_row, wide_of_row.data(), h->n_loci * sizeof(uint32_t), hipMemcpyHostToDevice, sh.dev->stream));
      KGX_HIP(hipMemcpy2DAsync(sh.d_wide, sh.wide_pitch * sizeof(uint16_t), cells + sh.genome_base, cells_pitch * sizeof(uint16_t),
                               sh.n_genomes * sizeof(uint16_t), n_wide, hipMemcpyHostToDevice, sh.dev->stream));
      sh.n_wide = n_wide;
    }
    return sync_shards(h);
  });
}

int kgx_gt8_read_rows(const kgx_gt8* h, uint8_t* dst, uint64_t dst_pitch, uint64_t l0, uint64_t l1) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || !dst) return fail(KGX_EINVAL, "null handle or destination");
    if (l0 > l1 || l1 > h->n_loci || dst_pitch < h->n_genomes) return fail(KGX_EINVAL, "bad locus range or pitch");
    if (l0 == l1) return KGX_OK;
    for (const auto& sh : h->shards) {
      if (sh.n_genomes == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      KGX_HIP(hipMemcpy2DAsync(dst + sh.genome_base, dst_pitch, sh.d_gt + l0 * sh.pitch, sh.pitch, sh.n_genomes, l1 - l0,
                               hipMemcpyDeviceToHost, sh.dev->stream));
    }
    return sync_shards(h);
  });
}

int kgx_locus_class_frequencies(const double* minor_af, uint64_t n_loci, uint32_t amax, double inbreeding, double* out, uint8_t* valid) {
  return guarded([&]() -> int {
    std::shared_ptr<Runtime> rt;
    if (int rc = require_runtime(rt)) return rc;
    if (!minor_af || !out || amax == 0 || amax > 254) return fail(KGX_EINVAL, "bad arguments (amax must be 1..254)");
    if (n_loci == 0) return KGX_OK;
    Device& dev = *rt->devs[0];
    if (int rc = use_device(dev)) return rc;
    const uint32_t stride = amax + kTableExtra;
    double *d_in = nullptr, *d_table = nullptr;
    uint8_t* d_valid = nullptr;
    int rc = KGX_OK;
    if (hipMalloc(&d_in, n_loci * amax * sizeof(double)) != hipSuccess || hipMalloc(&d_table, n_loci * stride * sizeof(double)) != hipSuccess ||
        hipMalloc(&d_valid, n_loci) != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(KGX_ENOMEM, "locus_class_frequencies: hipMalloc failed");
    }
    if (rc == KGX_OK) {
      std::vector<double> table(n_loci * stride);
      std::vector<uint8_t> v(n_loci);
      if (hipMemcpyAsync(d_in, minor_af, n_loci * amax * sizeof(double), hipMemcpyHostToDevice, dev.stream) != hipSuccess) rc = fail(KGX_EHIP, "H2D failed");
      if (rc == KGX_OK) {
        hipLaunchKernelGGL((k_locus_tables<false>), dim3(stream_grid(dev, n_loci, kBlock)), dim3(kBlock), 0, dev.stream, d_in, n_loci, amax, inbreeding, d_table, d_valid);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(table.data(), d_table, table.size() * sizeof(double), hipMemcpyDeviceToHost, dev.stream) != hipSuccess ||
            hipMemcpyAsync(v.data(), d_valid, n_loci, hipMemcpyDeviceToHost, dev.stream) != hipSuccess ||
            hipStreamSynchronize(dev.stream) != hipSuccess)
          rc = fail(KGX_EHIP, "locus table kernel failed");
      }
      if (rc == KGX_OK) {
        for (uint64_t l = 0; l < n_loci; ++l) {
          for (int k = 0; k < 5; ++k) out[l * 5 + k] = table[l * stride + amax + k];
          if (valid) valid[l] = v[l] ? 1 : 0;
        }
      }
    }
    if (d_in) (void)hipFree(d_in);
    if (d_table) (void)hipFree(d_table);
    if (d_valid) (void)hipFree(d_valid);
    return rc;
  });
}

int kgx_inbreed(kgx_gt8* h, uint64_t g0, uint64_t g1, const uint32_t* locus_index, uint64_t n_sel, const double* minor_af,
                uint32_t amax, int phased, int algorithm, const double* start, kgx_locus_results* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || !out || (n_sel && !minor_af)) return fail(KGX_EINVAL, "null argument");
    if (g0 > g1 || g1 > h->n_genomes || (g0 & 3u)) return fail(KGX_EINVAL, "genome range must lie in the matrix and start on a multiple of 4");
    if (amax == 0 || amax > 254) return fail(KGX_EINVAL, "amax %u outside [1,254] (8-bit allele indices of the wide rows; 14 for the matrix bytes)", amax);
    if (algorithm < 0 || algorithm > 3) return fail(KGX_EINVAL, "unknown algorithm %d", algorithm);
    if (start && (algorithm == KGX_ALGO_HALL_ME || algorithm == KGX_ALGO_LOGLIKELIHOOD))
      for (uint64_t g = 0; g < g1 - g0; ++g) {
        // HallME: the reference's draws lie in (0, 0.5]; Loglikelihood: the optimiser's box [-1, 1]
        const bool ok = algorithm == KGX_ALGO_HALL_ME ? (start[g] > 0.0 && start[g] <= 1.0) : (start[g] >= -1.0 && start[g] <= 1.0);
        if (!ok) return fail(KGX_EINVAL, "start[%llu] = %g outside the estimator's interval", (unsigned long long)g, start[g]);
      }
    if (!locus_index && n_sel > h->n_loci) return fail(KGX_EINVAL, "n_sel exceeds the locus count");
    if (locus_index) {
      hipPointerAttribute_t attr;                                  // a device-resident index cannot be range-checked from here
      const bool on_device = hipPointerGetAttributes(&attr, locus_index) == hipSuccess && attr.type == hipMemoryTypeDevice;
      if (!on_device) {
        (void)hipGetLastError();
        for (uint64_t i = 0; i < n_sel; ++i)
          if (locus_index[i] >= h->n_loci) return fail(KGX_EINVAL, "locus_index[%llu] out of range", (unsigned long long)i);
      }
    }
    if (g0 == g1) return KGX_OK;
    const int rc = for_each_parallel(h->shards.size(), [&](size_t s) -> int {
      kgx_gt8_shard& sh = h->shards[s];
      const uint64_t lo = g0 > sh.genome_base ? g0 : sh.genome_base;
      const uint64_t hi = g1 < sh.genome_base + sh.n_genomes ? g1 : sh.genome_base + sh.n_genomes;
      if (lo >= hi) return KGX_OK;
      return inbreed_shard(sh, lo - sh.genome_base, hi - sh.genome_base, locus_index, n_sel, minor_af, amax, phased, algorithm,
                           start ? start + (lo - g0) : nullptr, out + (lo - g0));
    });
    (void)use_device(*h->shards[0].dev);
    return rc;
  });
}

int kgx_inbreed_reference_starts(int algorithm, uint64_t seed, uint64_t first_stream, uint64_t n, double* out) {
  return guarded([&]() -> int {
    if (!out && n) return fail(KGX_EINVAL, "null argument");
    if (algorithm != KGX_ALGO_HALL_ME && algorithm != KGX_ALGO_LOGLIKELIHOOD) return fail(KGX_EINVAL, "algorithm %d draws no start points", algorithm);
    // processHallME: UniformRealDistribution(INIT_UPPER_, 0) (_calc.cpp:237); processLogLikelihood: (INIT_UPPER_, INIT_LOWER_)
    // (:166) = std::uniform_real_distribution<>(0.5, 0 | -0.5) on a std::mt19937_64 (kel_math/kel_distribution.h:25-43, 90-108).
    // RetryCalcResult(FINAL_ACCURACY_, MIN_RETRIES_ = 5, MAX_RETRIES_) ends the restarts at the fifth: checkTolerance
    // (:45-68) compares every entry with itself.  One draw per restart, so the fifth draw is the start that counts.
    constexpr int kRestarts = 5;
    const double lower = algorithm == KGX_ALGO_HALL_ME ? 0.0 : -0.5;
    if (seed == 0) {
      // RandomEntropySource: the reference seeds one twister per genome task from std::random_device.  Independent uniform
      // draws are independent uniform draws whichever freshly seeded twister makes them, so ONE is seeded here per call
      // (seeding 2,504 of them costs more than a window-sized sweep) and every genome takes its five draws from it.
      std::random_device rd;
      std::mt19937_64 entropy_mt(rd());
      std::uniform_real_distribution<> initialize_distribution(0.5, lower);
      for (uint64_t i = 0; i < n; ++i) {
        double drawn = 0.0;
        for (int restart = 0; restart < kRestarts; ++restart) drawn = initialize_distribution(entropy_mt);
        out[i] = drawn;
      }
      return KGX_OK;
    }
    for (uint64_t i = 0; i < n; ++i) {
      std::mt19937_64 entropy_mt(seed + first_stream + i);
      std::uniform_real_distribution<> initialize_distribution(0.5, lower);
      double drawn = 0.0;
      for (int restart = 0; restart < kRestarts; ++restart) drawn = initialize_distribution(entropy_mt);
      out[i] = drawn;
    }
    return KGX_OK;
  });
}

int kgx_inbreed_objective(kgx_gt8* h, uint64_t g0, uint64_t g1, const uint32_t* locus_index, uint64_t n_sel, const double* minor_af,
                          uint32_t amax, int phased, const double* at, int by_passes, double* value) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || !at || !value || (n_sel && !minor_af)) return fail(KGX_EINVAL, "null argument");
    if (g0 > g1 || g1 > h->n_genomes || (g0 & 3u)) return fail(KGX_EINVAL, "genome range must lie in the matrix and start on a multiple of 4");
    if (amax == 0 || amax > 14) return fail(KGX_EINVAL, "amax %u outside [1,14] (4-bit allele indices)", amax);
    if (n_sel == 0) return fail(KGX_EINVAL, "no loci selected");
    for (uint64_t g = 0; g < g1 - g0; ++g)
      if (!(at[g] >= -1.0 && at[g] <= 1.0)) return fail(KGX_EINVAL, "at[%llu] = %g outside [-1, 1]", (unsigned long long)g, at[g]);
    if (!locus_index && n_sel > h->n_loci) return fail(KGX_EINVAL, "n_sel exceeds the locus count");
    if (locus_index)
      for (uint64_t i = 0; i < n_sel; ++i)
        if (locus_index[i] >= h->n_loci) return fail(KGX_EINVAL, "locus_index[%llu] out of range", (unsigned long long)i);
    if (g0 == g1) return KGX_OK;
    std::vector<kgx_locus_results> unused(1);
    const int rc = for_each_parallel(h->shards.size(), [&](size_t s) -> int {
      kgx_gt8_shard& sh = h->shards[s];
      const uint64_t lo = g0 > sh.genome_base ? g0 : sh.genome_base;
      const uint64_t hi = g1 < sh.genome_base + sh.n_genomes ? g1 : sh.genome_base + sh.n_genomes;
      if (lo >= hi) return KGX_OK;
      return inbreed_shard(sh, lo - sh.genome_base, hi - sh.genome_base, locus_index, n_sel, minor_af, amax, phased, KGX_ALGO_LOGLIKELIHOOD,
                           at + (lo - g0), unused.data(), by_passes ? 2 : 1, value + (lo - g0));
    });
    (void)use_device(*h->shards[0].dev);
    return rc;
  });
}

int kgx_inbreed_last_path(void) {
  const auto rt = current_runtime();
  int path = KGX_PATH_NONE;
  if (rt)
    for (const auto& dev : rt->devs) { const int p = dev->last_path.load(); path = p > path ? p : path; }
  return path;
}

double kgx_inbreed_last_sweep_ms(void) {
  const auto rt = current_runtime();
  double worst = 0.0;
  if (rt)
    for (const auto& dev : rt->devs) { const double ms = dev->last_sweep_ms.load(); worst = ms > worst ? ms : worst; }
  return worst;
}

double kgx_inbreed_last_kernel_ms(void) {
  const auto rt = current_runtime();
  double worst = 0.0;
  if (rt)
    for (const auto& dev : rt->devs) { const double ms = dev->last_kernel_ms.load(); worst = ms > worst ? ms : worst; }
  return worst;
}

double kgx_inbreed_last_moments_ms(void) {
  const auto rt = current_runtime();
  double worst = 0.0;
  if (rt)
    for (const auto& dev : rt->devs) { const double ms = dev->last_moments_ms.load(); worst = ms > worst ? ms : worst; }
  return worst;
}

double kgx_inbreed_last_search_ms(void) {
  const auto rt = current_runtime();
  double worst = 0.0;
  if (rt)
    for (const auto& dev : rt->devs) { const double ms = dev->last_search_ms.load(); worst = ms > worst ? ms : worst; }
  return worst;
}

int kgx_inbreed_last_evaluations(void) {
  const auto rt = current_runtime();
  int most = 0;
  if (rt)
    for (const auto& dev : rt->devs) { const int n = dev->last_evaluations.load(); most = n > most ? n : most; }
  return most;
}

int kgx_gt8_synth_multiallelic(kgx_gt8* h, uint64_t seed, uint64_t genome_base, uint64_t locus_base, double* af_table) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h) return fail(KGX_EINVAL, "null handle");
    if (h->n_loci == 0) return KGX_OK;
    const int rc = for_each_parallel(h->shards.size(), [&](size_t s) -> int {
      kgx_gt8_shard& sh = h->shards[s];
      if (sh.n_genomes == 0) return KGX_OK;
      if (int e = use_device(*sh.dev)) return e;
      sh.wide_nibbles = 0;
      const bool want_table = af_table && s == 0;               // the table does not depend on the genomes: the first shard writes it
      double* d_table = nullptr;
      if (want_table) KGX_HIP_MEM(hipMalloc(&d_table, sh.n_loci * KGX_SYNTH_MAX_ALTS * sizeof(double)));
      const uint64_t work = sh.n_loci * ((sh.n_genomes + 3) / 4);
      hipLaunchKernelGGL(k_synth_gt8, dim3(stream_grid(*sh.dev, work, kBlock)), dim3(kBlock), 0, sh.dev->stream,
                         reinterpret_cast<uint32_t*>(sh.d_gt), sh.pitch / 4, sh.n_loci, sh.n_genomes, seed, genome_base + sh.genome_base, locus_base, d_table);
      int r = KGX_OK;
      if (hipGetLastError() != hipSuccess) r = fail(KGX_EHIP, "synthetic genotype kernel launch failed");
      if (r == KGX_OK && want_table &&
          hipMemcpyAsync(af_table, d_table, sh.n_loci * KGX_SYNTH_MAX_ALTS * sizeof(double), hipMemcpyDeviceToHost, sh.dev->stream) != hipSuccess)
        r = fail(KGX_EHIP, "D2H of the allele-frequency table failed");
      if (r == KGX_OK && hipStreamSynchronize(sh.dev->stream) != hipSuccess) r = fail(KGX_EHIP, "synthetic genotype kernel failed");
      if (d_table) (void)hipFree(d_table);
      return r;
    });
    (void)use_device(*h->shards[0].dev);
    return rc;
  });
}

int kgx_synth_multiallelic_host(uint64_t seed, uint64_t genome_base, uint64_t n_genomes, uint64_t l0, uint64_t l1, uint8_t* gt8,
                                uint64_t pitch, double* af_table, uint8_t* alleles) {
  return guarded([&]() -> int {
    if (l0 > l1 || (gt8 && pitch < n_genomes)) return fail(KGX_EINVAL, "bad range or pitch");
    for (uint64_t l = l0; l < l1; ++l) {
      const kgx_synth_locus loc = kgx_synth_make_locus(seed, l);
      if (af_table) {
        double* row = af_table + (l - l0) * KGX_SYNTH_MAX_ALTS;
        for (int a = 0; a < KGX_SYNTH_MAX_ALTS; ++a) row[a] = std::nan("");
        for (int a = 0; a < loc.n_alt; ++a)
          if (!loc.is_indel[a]) row[loc.snp_index[a] - 1] = static_cast<double>(loc.af[a]);
      }
      for (uint64_t g = 0; g < n_genomes; ++g) {
        int a1, a2;
        kgx_synth_multi_genotype(seed, l, genome_base + g, loc, a1, a2);
        if (gt8) gt8[(l - l0) * pitch + g] = static_cast<uint8_t>(kgx_synth_gt8_byte(loc, a1, a2));
        if (alleles) {
          alleles[((l - l0) * n_genomes + g) * 2 + 0] = static_cast<uint8_t>(a1);
          alleles[((l - l0) * n_genomes + g) * 2 + 1] = static_cast<uint8_t>(a2);
        }
      }
    }
    return KGX_OK;
  });
}

int kgx_synth_locus_host(uint64_t seed, uint64_t l, int* n_alt, float af[3], int is_indel[3]) {
  return guarded([&]() -> int {
    if (!n_alt || !af || !is_indel) return fail(KGX_EINVAL, "null argument");
    const kgx_synth_locus loc = kgx_synth_make_locus(seed, l);
    *n_alt = loc.n_alt;
    for (int a = 0; a < KGX_SYNTH_MAX_ALTS; ++a) { af[a] = loc.af[a]; is_indel[a] = loc.is_indel[a]; }
    return KGX_OK;
  });
}

int kgx_synth_loci_host(uint64_t seed, uint64_t l0, uint64_t l1, uint8_t* n_alt, float* af, uint8_t* is_indel) {
  return guarded([&]() -> int {
    if (!n_alt || !af || !is_indel || l0 > l1) return fail(KGX_EINVAL, "bad argument");
    for (uint64_t l = l0; l < l1; ++l) {
      const kgx_synth_locus loc = kgx_synth_make_locus(seed, l);
      n_alt[l - l0] = static_cast<uint8_t>(loc.n_alt);
      for (int a = 0; a < KGX_SYNTH_MAX_ALTS; ++a) {
        af[(l - l0) * KGX_SYNTH_MAX_ALTS + a] = loc.af[a];
        is_indel[(l - l0) * KGX_SYNTH_MAX_ALTS + a] = static_cast<uint8_t>(loc.is_indel[a]);
      }
    }
    return KGX_OK;
  });
}

int kgx_gt8_synth_inbred(kgx_gt8* h, const double* minor_af, uint32_t amax, const double* inbreeding, uint64_t seed) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || !minor_af || !inbreeding) return fail(KGX_EINVAL, "null argument");
    if (amax == 0 || amax > 14) return fail(KGX_EINVAL, "amax %u outside [1,14]", amax);
    if (h->n_loci == 0) return KGX_OK;
    const int rc = for_each_parallel(h->shards.size(), [&](size_t s) -> int {
      kgx_gt8_shard& sh = h->shards[s];
      if (sh.n_genomes == 0) return KGX_OK;
      if (int e = use_device(*sh.dev)) return e;
      sh.wide_nibbles = 0;
      double *d_table = nullptr, *d_f = nullptr;
      KGX_HIP_MEM(hipMalloc(&d_table, sh.n_loci * amax * sizeof(double)));
      if (hipMalloc(&d_f, sh.n_genomes * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_table); return fail(KGX_ENOMEM, "hipMalloc failed"); }
      int r = KGX_OK;
      if (hipMemcpyAsync(d_table, minor_af, sh.n_loci * amax * sizeof(double), hipMemcpyHostToDevice, sh.dev->stream) != hipSuccess ||
          hipMemcpyAsync(d_f, inbreeding + sh.genome_base, sh.n_genomes * sizeof(double), hipMemcpyHostToDevice, sh.dev->stream) != hipSuccess)
        r = fail(KGX_EHIP, "H2D of the allele-frequency table failed");
      if (r == KGX_OK) {
        hipLaunchKernelGGL(k_synth_inbred, dim3(stream_grid(*sh.dev, sh.n_loci * sh.n_genomes, kBlock)), dim3(kBlock), 0, sh.dev->stream, sh.d_gt,
                           sh.pitch, sh.n_loci, sh.n_genomes, d_table, amax, d_f, seed, sh.genome_base);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sh.dev->stream) != hipSuccess) r = fail(KGX_EHIP, "synthetic inbred genome kernel failed");
      }
      (void)hipFree(d_table);
      (void)hipFree(d_f);
      return r;
    });
    (void)use_device(*h->shards[0].dev);
    return rc;
  });
}

}  // extern "C"
