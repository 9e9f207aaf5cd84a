// C ABI (include/kgx.h) of the 2-bit dosage population: K2 (per-variant counts), K3 (per-genome counts, binned),
// K4 (population summary), K8 (compound offsets), the loaders and the synthetic biallelic population.
// A population is one shard of genomes per bound device; per-variant counts are summed over the shards by the
// exchange step (kgx_runtime.hip), per-genome results are concatenated.  No CPU fallback exists.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kgx_kernels_dosage.h"
#include "kgx_internal.h"

namespace kgx {
namespace {

int require_runtime(std::shared_ptr<Runtime>& rt) {
  rt = current_runtime();
  if (!rt) return fail(KGX_ENODEVICE, "kgx_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)");
  return KGX_OK;
}

// Lanes cooperating on one row: smallest power of two covering the row's 16-byte chunks, <= 64.
int lanes_per_row(uint32_t chunks_per_row) {
  int w = 1;
  while (w < 64 && static_cast<uint32_t>(w) < chunks_per_row) w <<= 1;
  return w;
}

template <int W, int U, bool NT>
void launch_count(const kgx_pop_shard& sh, kgx_v4u* d_out, hipStream_t stream) {
  const uint64_t rows_per_iter = static_cast<uint64_t>(kWave / W) * U;
  const uint64_t waves = (sh.n_variants + rows_per_iter - 1) / rows_per_iter;
  const uint64_t want = (waves + (kBlock / kWave) - 1) / (kBlock / kWave);
  const uint64_t cap = static_cast<uint64_t>(sh.dev->compute_units) * env_int("KGX_K2_BLOCKS_PER_CU", 32);
  const uint32_t grid = static_cast<uint32_t>(want < cap ? (want ? want : 1) : cap);
  if (sh.d_keep) {
    if constexpr (NT)     // the masked sweep comes in the default load flavour only
      hipLaunchKernelGGL((k_allele_count<W, U, true, true>), dim3(grid), dim3(kBlock), 0, stream,
                         reinterpret_cast<const kgx_v4u*>(sh.d_rows), sh.chunks_per_row, sh.n_variants, static_cast<uint32_t>(sh.n_kept), d_out,
                         reinterpret_cast<const kgx_v4u*>(sh.d_keep));
    return;
  }
  hipLaunchKernelGGL((k_allele_count<W, U, NT>), dim3(grid), dim3(kBlock), 0, stream,
                     reinterpret_cast<const kgx_v4u*>(sh.d_rows), sh.chunks_per_row,
                     sh.n_variants, static_cast<uint32_t>(sh.n_genomes), d_out, nullptr);
}

template <int W>
void launch_count_w(const kgx_pop_shard& sh, kgx_v4u* out, hipStream_t stream) {
  const int U = env_int("KGX_K2_U", 8);
  const bool nt = sh.d_keep != nullptr || env_int("KGX_K2_NT", 1) != 0;
  if (U <= 1)      nt ? launch_count<W, 1, true>(sh, out, stream) : launch_count<W, 1, false>(sh, out, stream);
  else if (U == 2) nt ? launch_count<W, 2, true>(sh, out, stream) : launch_count<W, 2, false>(sh, out, stream);
  else if (U <= 4) nt ? launch_count<W, 4, true>(sh, out, stream) : launch_count<W, 4, false>(sh, out, stream);
  else             nt ? launch_count<W, 8, true>(sh, out, stream) : launch_count<W, 8, false>(sh, out, stream);
}

// K2 of one shard into d_out (device memory of that shard's device), asynchronous on `stream`.
int launch_allele_count(const kgx_pop_shard& sh, void* d_out, hipStream_t stream) {
  if (sh.n_variants == 0) return KGX_OK;
  if (int rc = use_device(*sh.dev)) return rc;
  if (sh.n_genomes == 0) {                                   // a slot beyond the data: its counts are zero
    KGX_HIP(hipMemsetAsync(d_out, 0, sh.n_variants * 16u, stream));
    return KGX_OK;
  }
  kgx_v4u* out = static_cast<kgx_v4u*>(d_out);
  int lanes = lanes_per_row(sh.chunks_per_row);
  const int forced = env_int("KGX_K2_W", 0);             // tuning: fewer lanes per row than the covering power of two
  if (forced > 0 && forced <= lanes && (forced & (forced - 1)) == 0) lanes = forced;
  switch (lanes) {
    case 1:  launch_count_w<1>(sh, out, stream); break;
    case 2:  launch_count_w<2>(sh, out, stream); break;
    case 4:  launch_count_w<4>(sh, out, stream); break;
    case 8:  launch_count_w<8>(sh, out, stream); break;
    case 16: launch_count_w<16>(sh, out, stream); break;
    case 32: launch_count_w<32>(sh, out, stream); break;
    default: launch_count_w<64>(sh, out, stream); break;
  }
  KGX_HIP(hipGetLastError());
  return KGX_OK;
}

int ensure_counts(kgx_pop_shard& sh) {
  if (!sh.d_counts && sh.n_variants) {
    if (int rc = use_device(*sh.dev)) return rc;
    KGX_HIP_MEM(hipMalloc(&sh.d_counts, sh.n_variants * 16u));
  }
  return KGX_OK;
}

int ensure_af(kgx_pop_shard& sh) {
  if (!sh.d_af && sh.n_variants) {
    if (int rc = use_device(*sh.dev)) return rc;
    KGX_HIP_MEM(hipMalloc(&sh.d_af, sh.n_variants * sizeof(float)));
  }
  return KGX_OK;
}

// Every shard's K2 (shard 0 into d_out0 on stream0 when given, the others into their own scratch on their own
// streams), then the exchange: afterwards every slot's buffer holds the population's counts on that slot's stream.
int sweep_and_exchange(kgx_pop* pop, void* d_out0, hipStream_t stream0, bool own_stream0) {
  const size_t n = pop->shards.size();
  std::vector<void*> buffers(n);
  std::vector<hipStream_t> streams(n);
  for (size_t s = 0; s < n; ++s) {
    kgx_pop_shard& sh = pop->shards[s];
    if (s == 0 && d_out0) {
      buffers[s] = d_out0;
      streams[s] = own_stream0 ? sh.dev->stream : stream0;
    } else {
      if (int rc = ensure_counts(sh)) return rc;
      buffers[s] = sh.d_counts;
      streams[s] = sh.dev->stream;
    }
    if (int rc = launch_allele_count(sh, buffers[s], streams[s])) return rc;
  }
  if (n > 1 || pop->rt->exchange != Exchange::None)
    if (int rc = exchange_counts(*pop->rt, buffers, pop->n_variants * 4u, streams)) return rc;
  return use_device(*pop->shards[0].dev);
}

// ---- K3 host glue -----------------------------------------------------------------------------

struct ByGenomeShape { uint32_t n_cg, cg_width; };

// The chunk columns of a row cut into column groups of at most 64 chunks (a lane owns one), whole 128-byte lines each
// and as equal as that allows: 160 chunks (10,000 genomes) = 56 + 56 + 48 rather than 64 + 64 + 32.
ByGenomeShape by_genome_shape(uint32_t chunks_per_row) {
  ByGenomeShape shape;
  shape.n_cg = (chunks_per_row + 63) / 64;
  if (shape.n_cg <= 1) { shape.n_cg = 1; shape.cg_width = 64; return shape; }
  const uint32_t lines = (chunks_per_row + 7) / 8;
  shape.cg_width = (lines + shape.n_cg - 1) / shape.n_cg * 8;
  return shape;
}

// The rows a by-genome sweep walks: the shard's 2-bit dosage rows, or its 1-bit phase plane.
struct ByGenomeRows { const uint8_t* rows; uint32_t chunks_per_row; bool plane; };

template <int W>
int launch_by_genome(const kgx_pop_shard& sh, ByGenomeRows source, ByGenomeShape shape, const uint32_t* d_index, const GenomeWork* d_work, uint32_t n_work,
                     const unsigned long long* d_binoff, uint32_t n_bins, uint32_t* d_acc, int* resident_per_cu) {
  if (resident_per_cu) {                                     // how many of these workgroups a CU holds at once (registers, LDS)
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_count_by_genome<W, false>, kBlock, 0) != hipSuccess || blocks <= 0) {
      (void)hipGetLastError();
      blocks = 3;
    }
    *resident_per_cu = blocks;
    return KGX_OK;
  }
  if (source.plane)
    hipLaunchKernelGGL((k_count_by_genome<W, true>), dim3(n_work), dim3(kBlock), 0, sh.dev->stream,
                       reinterpret_cast<const kgx_v4u*>(source.rows), source.chunks_per_row, shape.cg_width,
                       d_index, d_work, d_binoff, n_bins, shape.n_cg, d_acc);
  else
    hipLaunchKernelGGL((k_count_by_genome<W, false>), dim3(n_work), dim3(kBlock), 0, sh.dev->stream,
                       reinterpret_cast<const kgx_v4u*>(source.rows), source.chunks_per_row, shape.cg_width,
                       d_index, d_work, d_binoff, n_bins, shape.n_cg, d_acc);
  return KGX_OK;
}

template <typename... Args>
int dispatch_by_genome(int W, Args... args) {
  switch (W) {
    case 1:  return launch_by_genome<1>(args...);
    case 2:  return launch_by_genome<2>(args...);
    case 4:  return launch_by_genome<4>(args...);
    case 8:  return launch_by_genome<8>(args...);
    case 16: return launch_by_genome<16>(args...);
    case 32: return launch_by_genome<32>(args...);
    default: return launch_by_genome<64>(args...);
  }
}

// One shard's by-genome sweep; out = the shard's block [n_genomes][n_bins][4] of the caller's array.
// bin_edges (host, n_bins + 1 doubles; bin_of_variant null then): the bins are evaluated on the device from the AF column.
// plane: the sweep walks the shard's phase plane instead of its dosage rows; out = [n_genomes][n_bins] set-bit counts.
int count_by_genome_shard(kgx_pop_shard& sh, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out, const double* bin_edges = nullptr,
                          bool plane = false) {
  const uint64_t V = sh.n_variants, G = sh.n_genomes;
  if (G == 0) return KGX_OK;
  if (V > 0xFFFFFFFFull) return fail(KGX_EINVAL, "n_variants exceeds the 32-bit row index of the by-genome sweep");
  Device& dev = *sh.dev;
  std::lock_guard<std::mutex> device_lock(dev.mutex);          // the scratch arena and the timing events are the device's
  if (int rc = use_device(dev)) return rc;
  const uint64_t cells = G * n_bins;
  const bool masked = sh.d_keep != nullptr;                   // rows without a kept carrier drop out (d_counts: K2 under the mask)
  const bool identity = bin_of_variant == nullptr && bin_edges == nullptr && !masked;
  hipStream_t st = dev.stream;
  if (plane && !sh.d_phase) return fail(KGX_ESTATE, "no phase plane was loaded (kgx_population_load_phase_plane)");
  const ByGenomeRows source = plane ? ByGenomeRows{sh.d_phase, static_cast<uint32_t>(sh.phase_pitch / 16), true}
                                    : ByGenomeRows{sh.d_rows, sh.chunks_per_row, false};

  unsigned long long *d_out = nullptr, *d_nbin = nullptr, *d_binoff = nullptr;
  uint32_t *d_index = nullptr, *d_chunks = nullptr, *d_acc = nullptr;
  uint8_t* d_bins = nullptr;
  GenomeWork* d_work = nullptr;
  int rc = KGX_OK;
  auto try_hip = [&](hipError_t e, int code, const char* what) {
    if (rc == KGX_OK && e != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(code, "count_by_genome: %s failed: %s", what, hipGetErrorString(e));
    }
  };
  // every buffer of the call out of the device's arena; the work list's size is known before the bins are: about `target`
  // equal stretches of the row list (they may cross bin boundaries), one item per column group each
  const uint32_t n_chunks = static_cast<uint32_t>((V + kBinChunk - 1) / kBinChunk);
  const ByGenomeShape shape = by_genome_shape(source.chunks_per_row);
  const uint32_t n_cg = shape.n_cg;
  const int W = lanes_per_row(source.chunks_per_row);
  // Work items: equal stretches of the bin-grouped row list, one column group each, KGX_K3_ROUNDS (default 1) per workgroup
  // the device holds at once -- every item runs from the start of the kernel to its end, none is left for a thin last round.
  int resident_per_cu = 0;
  (void)dispatch_by_genome(W, sh, source, shape, static_cast<const uint32_t*>(nullptr), static_cast<const GenomeWork*>(nullptr), 0u,
                           static_cast<const unsigned long long*>(nullptr), n_bins, static_cast<uint32_t*>(nullptr), &resident_per_cu);
  const int rounds = std::max(1, env_int("KGX_K3_ROUNDS", 1));
  const uint64_t target = static_cast<uint64_t>(dev.compute_units) * static_cast<uint64_t>(resident_per_cu) * static_cast<uint64_t>(rounds);
  const uint64_t max_work = target + 2ull * n_cg + 64;
  const uint64_t acc_words = static_cast<uint64_t>(n_bins) * n_cg * kAccCounters * kLdsStride;
  ScratchPlan plan;
  const size_t o_acc = plan.add(acc_words * sizeof(uint32_t)), o_out = plan.add((cells ? cells : 1) * 4 * sizeof(unsigned long long));
  const size_t o_nbin = plan.add(n_bins * sizeof(unsigned long long)), o_binoff = plan.add((n_bins + 1) * sizeof(unsigned long long));
  const size_t o_bins = plan.add(identity ? 0 : V), o_index = plan.add(identity ? 0 : (V + 8) * sizeof(uint32_t));
  const size_t o_chunks = plan.add(identity ? 0 : static_cast<uint64_t>(n_chunks) * n_bins * sizeof(uint32_t)), o_work = plan.add(max_work * sizeof(GenomeWork));
  char* arena = nullptr;
  if (int arc = scratch_reserve(dev, plan.total, &arena)) return arc;
  d_acc = reinterpret_cast<uint32_t*>(arena + o_acc);
  d_out = reinterpret_cast<unsigned long long*>(arena + o_out);
  d_nbin = reinterpret_cast<unsigned long long*>(arena + o_nbin);
  d_binoff = reinterpret_cast<unsigned long long*>(arena + o_binoff);
  d_work = reinterpret_cast<GenomeWork*>(arena + o_work);
  try_hip(hipMemsetAsync(d_acc, 0, acc_words * sizeof(uint32_t), st), KGX_EHIP, "memset(acc)");

  // Rows grouped by bin, so that a workgroup only ever touches one bin: on the device (k_bin_count / _scan / _scatter).
  std::vector<unsigned long long> bin_offset(n_bins + 1, 0);
  if (identity) {
    bin_offset[1] = V;
    const unsigned long long v = V;
    try_hip(hipMemcpyAsync(d_nbin, &v, sizeof(v), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(rows_in_bin)");
    try_hip(hipMemcpyAsync(d_binoff, bin_offset.data(), 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(bin offsets)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
  } else if (V > 0) {
    d_bins = reinterpret_cast<uint8_t*>(arena + o_bins);
    d_index = reinterpret_cast<uint32_t*>(arena + o_index);                                       // + 8: whole 8-entry scalar fetches
    d_chunks = reinterpret_cast<uint32_t*>(arena + o_chunks);
    try_hip(hipMemsetAsync(d_index + V, 0, 8 * sizeof(uint32_t), st), KGX_EHIP, "memset(index pad)");
    if (bin_edges) {
      // d_binoff is not read before k_bin_scan writes it: the edges borrow it on their way in
      static_assert(sizeof(double) == sizeof(unsigned long long), "edge buffer");
      try_hip(hipMemcpyAsync(d_binoff, bin_edges, (n_bins + 1) * sizeof(double), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(bin edges)");
      if (rc == KGX_OK)
        hipLaunchKernelGGL(k_af_bins, dim3(stream_grid(dev, V, kBlock)), dim3(kBlock), 0, st, sh.d_af, V, reinterpret_cast<const double*>(d_binoff), n_bins, d_bins);
    } else if (bin_of_variant) {
      try_hip(hipMemcpyAsync(d_bins, bin_of_variant, V, hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(bins)");
    } else {
      try_hip(hipMemsetAsync(d_bins, 0, V, st), KGX_EHIP, "memset(bins)");
    }
    if (rc == KGX_OK && masked)
      hipLaunchKernelGGL(k_drop_absent_rows, dim3(stream_grid(dev, V, kBlock)), dim3(kBlock), 0, st, static_cast<const kgx_v4u*>(sh.d_counts), V, d_bins);
    if (rc == KGX_OK) {
      hipLaunchKernelGGL(k_bin_count, dim3(n_chunks), dim3(kBlock), 0, st, d_bins, V, n_bins, d_chunks);
      hipLaunchKernelGGL(k_bin_totals, dim3(n_bins), dim3(kBlock), 0, st, d_chunks, n_chunks, d_nbin);
      hipLaunchKernelGGL(k_bin_scan, dim3(n_bins), dim3(kBlock), 0, st, d_chunks, n_chunks, n_bins, d_nbin, d_binoff);
      hipLaunchKernelGGL(k_bin_scatter, dim3(n_chunks), dim3(kBlock), 0, st, d_bins, V, n_bins, d_chunks, d_index);
      try_hip(hipGetLastError(), KGX_EHIP, "bin grouping kernels");
      try_hip(hipMemcpyAsync(bin_offset.data(), d_binoff, (n_bins + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(bin offsets)");
      try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
    }
  } else {
    try_hip(hipMemsetAsync(d_nbin, 0, n_bins * sizeof(unsigned long long), st), KGX_EHIP, "memset(rows_in_bin)");
  }
  const uint64_t selected = bin_offset[n_bins];
  bool timed = false;

  if (rc == KGX_OK && selected > 0) {
    const uint64_t gran = static_cast<uint64_t>(64 / W) * (kBlock / kWave) * 8;
    uint64_t pieces = target / n_cg;                          // stretches of the row list; each becomes n_cg items
    if (pieces < 1) pieces = 1;
    uint64_t per_wg = (selected + pieces - 1) / pieces;
    per_wg = (per_wg + gran - 1) / gran * gran;
    if (per_wg < gran * 4) per_wg = gran * 4;
    std::vector<GenomeWork> work;
    for (uint64_t p = 0; p < selected; p += per_wg)
      for (uint32_t cg = 0; cg < n_cg; ++cg) {                // neighbours in the launch order read the same rows
        GenomeWork w;
        w.begin = p;
        w.end = p + per_wg < selected ? p + per_wg : selected;
        w.col_group = cg;
        w.pad = 0;
        work.push_back(w);
      }
    if (rc == KGX_OK && work.size() > max_work) rc = fail(KGX_ESTATE, "count_by_genome: %zu work items exceed the %llu reserved", work.size(), (unsigned long long)max_work);
    try_hip(hipMemcpyAsync(d_work, work.data(), work.size() * sizeof(GenomeWork), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(work)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");     // `work` is pageable host memory
    if (rc == KGX_OK) {
      const uint32_t n_work = static_cast<uint32_t>(work.size());
      try_hip(hipEventRecord(dev.by_genome_begin, st), KGX_EHIP, "hipEventRecord");
      (void)dispatch_by_genome(W, sh, source, shape, static_cast<const uint32_t*>(d_index), static_cast<const GenomeWork*>(d_work), n_work,
                               static_cast<const unsigned long long*>(d_binoff), n_bins, d_acc, static_cast<int*>(nullptr));
      try_hip(hipGetLastError(), KGX_EHIP, "k_count_by_genome launch");
      try_hip(hipEventRecord(dev.by_genome_end, st), KGX_EHIP, "hipEventRecord");
      timed = rc == KGX_OK;
    }
  }
  if (rc == KGX_OK && cells > 0) {
    if (plane) hipLaunchKernelGGL(k_finish_plane_by_genome, dim3(stream_grid(dev, cells, kBlock)), dim3(kBlock), 0, st, d_acc, G, n_bins, n_cg, shape.cg_width, d_out);
    else hipLaunchKernelGGL(k_finish_by_genome, dim3(stream_grid(dev, cells, kBlock)), dim3(kBlock), 0, st, d_acc, d_nbin, G, n_bins, n_cg, shape.cg_width, d_out);
    try_hip(hipGetLastError(), KGX_EHIP, "k_finish_by_genome launch");
    try_hip(hipMemcpyAsync(out, d_out, cells * (plane ? 1 : 4) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(out)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "stream synchronize");
  }
  dev.last_by_genome_ms = 0.0;
  if (rc == KGX_OK && timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, dev.by_genome_begin, dev.by_genome_end) == hipSuccess) dev.last_by_genome_ms = ms;
    else (void)hipGetLastError();
  }
  return rc;
}

int sync_shards(const kgx_pop* pop);

// The genomes a mask leaves out take no part: their blocks of a per-genome result read as zero.
void zero_masked_genomes(const kgx_pop* pop, uint64_t* out, uint64_t words_per_genome) {
  for (uint64_t g = 0; g < pop->keep.size(); ++g)
    if (!pop->keep[g]) std::memset(out + g * words_per_genome, 0, words_per_genome * sizeof(uint64_t));
}

int count_by_genome_impl(kgx_pop* pop, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out, const double* bin_edges = nullptr,
                         bool plane = false) {
  if (!pop->keep.empty() && !pop->counts_current && pop->n_variants) {
    // which rows still have a carrier: the population's K2 counts under the mask, on every shard's device
    if (int rc = sweep_and_exchange(pop, nullptr, nullptr, true)) return rc;
    if (int rc = sync_shards(pop)) return rc;
    pop->counts_current = true;
  }
  const int rc = for_each_parallel(pop->shards.size(), [&](size_t s) {
    kgx_pop_shard& sh = pop->shards[s];
    return count_by_genome_shard(sh, bin_of_variant, n_bins, out + sh.genome_base * n_bins * (plane ? 1 : 4), bin_edges, plane);
  });
  if (rc == KGX_OK) zero_masked_genomes(pop, out, static_cast<uint64_t>(n_bins) * (plane ? 1 : 4));
  return rc;
}

// row_list (may be empty): the groups' member rows; a group's first_row is then its first position in the list.
// fields: 3 = k_compound_offsets' counters, 4 = k_offset_filters'.
int compound_offsets_shard(kgx_pop_shard& sh, const std::vector<OffsetGroup>& groups, const std::vector<uint32_t>& row_list, uint32_t n_bins, uint64_t* out,
                           uint32_t fields = 3) {
  const uint64_t G = sh.n_genomes, n_groups = groups.size();
  if (G == 0) return KGX_OK;
  if (int rc = use_device(*sh.dev)) return rc;
  const Device& dev = *sh.dev;
  const uint64_t cells = G * n_bins * fields;
  OffsetGroup* d_groups = nullptr;
  uint32_t* d_list = nullptr;
  unsigned long long* d_acc = nullptr;
  int rc = KGX_OK;
  if (hipMalloc(&d_groups, n_groups * sizeof(OffsetGroup)) != hipSuccess || hipMalloc(&d_acc, cells * sizeof(unsigned long long)) != hipSuccess ||
      (!row_list.empty() && hipMalloc(&d_list, row_list.size() * sizeof(uint32_t)) != hipSuccess)) {
    (void)hipGetLastError();
    rc = fail(KGX_ENOMEM, "compound_offsets: hipMalloc failed");
  }
  if (rc == KGX_OK) {
    const uint64_t cols = (G + 15) / 16;
    const uint32_t gx = static_cast<uint32_t>((cols + kBlock - 1) / kBlock);
    uint64_t slices = (static_cast<uint64_t>(dev.compute_units) * 8 + gx - 1) / gx;
    if (slices > n_groups) slices = n_groups;
    if (slices > 65535) slices = 65535;
    const uint64_t per_slice = (n_groups + slices - 1) / slices;
    const uint32_t gy = static_cast<uint32_t>((n_groups + per_slice - 1) / per_slice);
    if (hipMemsetAsync(d_acc, 0, cells * sizeof(unsigned long long), dev.stream) != hipSuccess ||
        hipMemcpyAsync(d_groups, groups.data(), n_groups * sizeof(OffsetGroup), hipMemcpyHostToDevice, dev.stream) != hipSuccess ||
        (d_list && hipMemcpyAsync(d_list, row_list.data(), row_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, dev.stream) != hipSuccess)) {
      rc = fail(KGX_EHIP, "compound_offsets: upload failed");
    } else {
      if (fields == 4)
        hipLaunchKernelGGL(k_offset_filters, dim3(gx, gy), dim3(kBlock), 0, dev.stream,
                           reinterpret_cast<const uint32_t*>(sh.d_rows), sh.pitch / 4, G, d_groups, n_groups, per_slice,
                           d_list, n_bins, d_acc);
      else
        hipLaunchKernelGGL(k_compound_offsets, dim3(gx, gy), dim3(kBlock), 0, dev.stream,
                           reinterpret_cast<const uint32_t*>(sh.d_rows), sh.pitch / 4, G, d_groups, n_groups, per_slice,
                           d_list, n_bins, d_acc);
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(out, d_acc, cells * sizeof(unsigned long long), hipMemcpyDeviceToHost, dev.stream) != hipSuccess ||
          hipStreamSynchronize(dev.stream) != hipSuccess)
        rc = fail(KGX_EHIP, "compound_offsets: kernel or readback failed");
    }
  }
  if (d_groups) (void)hipFree(d_groups);
  if (d_list) (void)hipFree(d_list);
  if (d_acc) (void)hipFree(d_acc);
  return rc;
}

// One shard's genome-major row lists for its genomes [g_lo, g_hi) (shard-local numbers): begin (host, g_hi - g_lo + 1
// entries, starting at 0) and, when rows_out is given, the row numbers.
int genome_row_lists_shard(kgx_pop_shard& sh, uint64_t g_lo, uint64_t g_hi, const uint8_t* row_selected, std::vector<unsigned long long>& begin,
                           std::vector<uint32_t>* rows_out) {
  const uint64_t V = sh.n_variants, n = g_hi - g_lo;
  begin.assign(n + 1, 0);
  if (n == 0 || V == 0) return KGX_OK;
  if (V > 0xFFFFFFFFull) return fail(KGX_EINVAL, "n_variants exceeds the 32-bit row numbers of the row lists");
  Device& dev = *sh.dev;
  if (int rc = use_device(dev)) return rc;
  hipStream_t st = dev.stream;
  const uint64_t genomes_padded = (static_cast<uint64_t>(sh.chunks_per_row) + kListChunks - 1) / kListChunks * kListChunks * 64u;
  const uint64_t first_group = g_lo / (64u * kListChunks), last_group = (g_hi - 1) / (64u * kListChunks);
  const uint32_t gx = static_cast<uint32_t>((last_group - first_group + 1 + (kBlock / kWave) - 1) / (kBlock / kWave));
  // slices of whole 64-row tiles: enough workgroups to fill the device, at most 65535
  uint64_t slices = (static_cast<uint64_t>(dev.compute_units) * 16 + gx - 1) / gx;
  const uint64_t tiles = (V + kWave - 1) / kWave;
  if (slices > tiles) slices = tiles;
  if (slices > 65535) slices = 65535;
  const uint64_t rows_per_slice = (tiles + slices - 1) / slices * kWave;
  slices = (V + rows_per_slice - 1) / rows_per_slice;
  uint8_t* d_sel = nullptr;
  uint32_t *d_counts = nullptr, *d_out = nullptr;
  unsigned long long *d_cursors = nullptr, *d_begin = nullptr, *d_totals = nullptr;
  int rc = KGX_OK;
  auto try_hip = [&](hipError_t e, int code, const char* what) {
    if (rc == KGX_OK && e != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(code, "genome_row_lists: %s failed: %s", what, hipGetErrorString(e));
    }
  };
  try_hip(hipMalloc(&d_totals, sh.n_genomes * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(totals)");
  try_hip(hipMalloc(&d_counts, slices * genomes_padded * sizeof(uint32_t)), KGX_ENOMEM, "hipMalloc(counts)");
  try_hip(hipMalloc(&d_cursors, slices * genomes_padded * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(cursors)");
  try_hip(hipMalloc(&d_begin, (sh.n_genomes + 1) * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(begin)");
  if (row_selected) {
    try_hip(hipMalloc(&d_sel, V), KGX_ENOMEM, "hipMalloc(selection)");
    try_hip(hipMemcpyAsync(d_sel, row_selected, V, hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(selection)");
  }
  try_hip(hipMemsetAsync(d_counts, 0, slices * genomes_padded * sizeof(uint32_t), st), KGX_EHIP, "memset(counts)");
  std::vector<unsigned long long> shard_begin(sh.n_genomes + 1, 0);
  if (rc == KGX_OK) {
    hipLaunchKernelGGL((k_genome_row_lists<false>), dim3(gx, static_cast<uint32_t>(slices)), dim3(kBlock), 0, st,
                       reinterpret_cast<const kgx_v4u*>(sh.d_rows), sh.chunks_per_row, V, sh.n_genomes, g_lo, g_hi,
                       reinterpret_cast<const kgx_v4u*>(sh.d_keep), d_sel, rows_per_slice, d_counts, genomes_padded, nullptr, nullptr);
    try_hip(hipMemsetAsync(d_totals, 0, sh.n_genomes * sizeof(unsigned long long), st), KGX_EHIP, "memset(totals)");
  }
  if (rc == KGX_OK) {                                       // only over cleared totals
    const uint32_t by_genome = static_cast<uint32_t>((sh.n_genomes + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_row_list_totals, dim3(by_genome, static_cast<uint32_t>((slices + kTotalsSlices - 1) / kTotalsSlices)), dim3(kBlock), 0, st,
                       d_counts, slices, genomes_padded, sh.n_genomes, d_totals);
    hipLaunchKernelGGL(k_row_list_scan, dim3(1), dim3(kBlock), 0, st, d_totals, sh.n_genomes, d_begin);
    hipLaunchKernelGGL(k_row_list_cursors, dim3(by_genome), dim3(kBlock), 0, st, d_counts, slices, genomes_padded, sh.n_genomes, d_begin, d_cursors);
    try_hip(hipGetLastError(), KGX_EHIP, "count / offset kernels");
    try_hip(hipMemcpyAsync(shard_begin.data(), d_begin, (sh.n_genomes + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(begin)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
  }
  if (rc == KGX_OK) {
    // genomes outside [g_lo, g_hi) list nothing, so the scan's values at g_lo .. g_hi are the range's own, starting at 0
    for (uint64_t g = 0; g <= n; ++g) begin[g] = shard_begin[g_lo + g];
    const unsigned long long total = begin[n];
    if (rows_out) {
      rows_out->assign(total, 0u);
      if (total) {
        try_hip(hipMalloc(&d_out, total * sizeof(uint32_t)), KGX_ENOMEM, "hipMalloc(rows)");
        if (rc == KGX_OK) {
          hipLaunchKernelGGL((k_genome_row_lists<true>), dim3(gx, static_cast<uint32_t>(slices)), dim3(kBlock), 0, st,
                             reinterpret_cast<const kgx_v4u*>(sh.d_rows), sh.chunks_per_row, V, sh.n_genomes, g_lo, g_hi,
                             reinterpret_cast<const kgx_v4u*>(sh.d_keep), d_sel, rows_per_slice, nullptr, genomes_padded, d_cursors, d_out);
          try_hip(hipGetLastError(), KGX_EHIP, "fill kernel");
          try_hip(hipMemcpyAsync(rows_out->data(), d_out, total * sizeof(uint32_t), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(rows)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        }
      }
    }
  }
  for (void* p : {static_cast<void*>(d_sel), static_cast<void*>(d_counts), static_cast<void*>(d_cursors), static_cast<void*>(d_begin), static_cast<void*>(d_totals), static_cast<void*>(d_out)})
    if (p) (void)hipFree(p);
  return rc;
}

int sync_shards(const kgx_pop* pop) {
  for (const auto& sh : pop->shards) {
    if (int rc = use_device(*sh.dev)) return rc;
    KGX_HIP(hipStreamSynchronize(sh.dev->stream));
  }
  return use_device(*pop->shards[0].dev);
}

void destroy_shards(kgx_pop* pop) {
  for (auto& sh : pop->shards) {
    if (use_device(*sh.dev) != KGX_OK) continue;
    if (sh.d_alloc) (void)hipFree(sh.d_alloc);
    if (sh.d_af) (void)hipFree(sh.d_af);
    if (sh.d_counts) (void)hipFree(sh.d_counts);
    if (sh.d_keep) (void)hipFree(sh.d_keep);
    if (sh.d_phase) (void)hipFree(sh.d_phase);
  }
  if (!pop->shards.empty()) (void)use_device(*pop->shards[0].dev);
}

}  // namespace
}  // namespace kgx

using namespace kgx;

extern "C" {

kgx_pop* kgx_population_create(uint64_t n_genomes, uint64_t n_variants) {
  std::shared_ptr<Runtime> rt;
  if (require_runtime(rt)) return nullptr;
  if (n_genomes == 0 || n_genomes > (1ull << 31)) {
    fail(KGX_EINVAL, "n_genomes %llu outside (0, 2^31] (uint32 per-variant counts)", (unsigned long long)n_genomes);
    return nullptr;
  }
  kgx_pop* pop = new (std::nothrow) kgx_pop();
  if (!pop) { fail(KGX_ENOMEM, "host allocation failed"); return nullptr; }
  pop->rt = rt;
  pop->n_genomes = n_genomes;
  pop->n_variants = n_variants;
  // Contiguous genome shards of whole 64-genome chunks (one 16-byte lane load), sizes within one chunk of each other.
  const uint64_t n_slots = rt->devs.size();
  const uint64_t units = (n_genomes + 63) / 64, per = units / n_slots, extra = units % n_slots;
  uint64_t base = 0;
  try {
    pop->shards.reserve(n_slots);
  } catch (const std::exception&) {
    delete pop;
    fail(KGX_ENOMEM, "host allocation failed");
    return nullptr;
  }
  for (uint64_t s = 0; s < n_slots; ++s) {
    kgx_pop_shard sh;
    sh.dev = rt->devs[s].get();
    sh.genome_base = base;
    const uint64_t want = (per + (s < extra ? 1 : 0)) * 64;
    sh.n_genomes = want < n_genomes - base ? want : n_genomes - base;
    sh.n_variants = n_variants;
    sh.n_kept = sh.n_genomes;
    sh.row_bytes = (sh.n_genomes + 3) / 4;
    // Rows longer than half a wave-load start on a 128-byte line so that every 1 KiB wave load covers
    // whole lines (measured +6 % on 2500-byte rows); short rows stay densely packed.
    int align = env_int("KGX_PITCH_ALIGN", sh.row_bytes > 512 ? 128 : 16);
    if (align < 16 || (align & (align - 1))) align = 16;
    sh.pitch = (sh.row_bytes + align - 1) / align * align;
    sh.chunks_per_row = static_cast<uint32_t>(sh.pitch / 16);
    base += sh.n_genomes;
    pop->shards.push_back(sh);
  }
  for (auto& sh : pop->shards) {
    const uint64_t bytes = sh.pitch * n_variants;
    if (!bytes) continue;
    if (use_device(*sh.dev) != KGX_OK || hipMalloc(&sh.d_alloc, bytes) != hipSuccess) {
      (void)hipGetLastError();
      fail(KGX_ENOMEM, "hipMalloc of %llu bytes for %llu x %llu dosage rows on device %d failed",
           (unsigned long long)bytes, (unsigned long long)n_variants, (unsigned long long)sh.n_genomes, sh.dev->id);
      destroy_shards(pop);
      delete pop;
      return nullptr;
    }
    sh.d_rows = sh.d_alloc;
    sh.capacity = n_variants;
    if (hipMemsetAsync(sh.d_rows, 0, bytes, sh.dev->stream) != hipSuccess) {
      (void)hipGetLastError();
      fail(KGX_EHIP, "hipMemset of dosage rows failed");
      destroy_shards(pop);
      delete pop;
      return nullptr;
    }
  }
  if (sync_shards(pop) != KGX_OK) {
    destroy_shards(pop);
    delete pop;
    return nullptr;
  }
  return pop;
}

void kgx_population_destroy(kgx_pop* pop) {
  if (!pop) return;
  destroy_shards(pop);
  delete pop;
}

uint64_t kgx_population_genomes(const kgx_pop* pop) { return pop ? pop->n_genomes : 0; }
uint64_t kgx_population_variants(const kgx_pop* pop) { return pop ? pop->n_variants : 0; }
uint64_t kgx_population_row_pitch(const kgx_pop* pop) { return pop ? pop->shards[0].pitch : 0; }
uint64_t kgx_population_sweep_bytes(const kgx_pop* pop) {
  if (!pop) return 0;
  uint64_t bytes = 0;
  for (const auto& sh : pop->shards)
    if (sh.n_genomes) bytes += pop->n_variants * sh.row_bytes + 16u * pop->n_variants;
  return bytes;
}
uint32_t kgx_population_shards(const kgx_pop* pop) { return pop ? static_cast<uint32_t>(pop->shards.size()) : 0; }
int kgx_population_shard_info(const kgx_pop* pop, uint32_t shard, int* slot, uint64_t* genome_base, uint64_t* n_genomes) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || shard >= pop->shards.size()) return fail(KGX_EINVAL, "no such shard");
    const auto& sh = pop->shards[shard];
    if (slot) *slot = sh.dev->slot;
    if (genome_base) *genome_base = sh.genome_base;
    if (n_genomes) *n_genomes = sh.n_genomes;
    return KGX_OK;
  });
}

int kgx_population_load_dosage2(kgx_pop* pop, const uint8_t* src, uint64_t src_pitch, uint64_t v0, uint64_t v1) {
  return guarded([&]() -> int {
    if (pop) pop->counts_current = false;
    if (int bound = require_bound()) return bound;
    if (!pop || !src) return fail(KGX_EINVAL, "null population or source");
    if (v0 > v1 || v1 > pop->n_variants) return fail(KGX_EINVAL, "variant range [%llu,%llu) outside [0,%llu)",
        (unsigned long long)v0, (unsigned long long)v1, (unsigned long long)pop->n_variants);
    if (src_pitch < (pop->n_genomes + 3) / 4) return fail(KGX_EINVAL, "src_pitch %llu < row bytes %llu",
        (unsigned long long)src_pitch, (unsigned long long)((pop->n_genomes + 3) / 4));
    if (v0 == v1) return KGX_OK;
    for (auto& sh : pop->shards) {
      if (sh.n_genomes == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      // a shard starts on a 64-genome boundary: its bytes of a source row are whole bytes from genome_base / 4 on
      KGX_HIP(hipMemcpy2DAsync(sh.d_rows + v0 * sh.pitch, sh.pitch, src + sh.genome_base / 4, src_pitch, sh.row_bytes,
                               v1 - v0, hipMemcpyHostToDevice, sh.dev->stream));
      const uint64_t touched = (v1 - v0) * (sh.pitch - sh.row_bytes + 1);
      hipLaunchKernelGGL(k_mask_row_tail, dim3(stream_grid(*sh.dev, touched, kBlock)), dim3(kBlock), 0, sh.dev->stream,
                         sh.d_rows, sh.pitch, sh.n_genomes, v0, v1);
      KGX_HIP(hipGetLastError());
    }
    return sync_shards(pop);
  });
}

int kgx_population_load_dosage_u8(kgx_pop* pop, const uint8_t* src, uint64_t g0, uint64_t g1) {
  return guarded([&]() -> int {
    if (pop) pop->counts_current = false;
    if (int bound = require_bound()) return bound;
    if (!pop || !src) return fail(KGX_EINVAL, "null population or source");
    if (g0 > g1 || g1 > pop->n_genomes) return fail(KGX_EINVAL, "genome range [%llu,%llu) outside [0,%llu)",
        (unsigned long long)g0, (unsigned long long)g1, (unsigned long long)pop->n_genomes);
    if (g0 & 3u) return fail(KGX_EINVAL, "g0 must be a multiple of 4 (whole packed bytes)");
    if ((g1 & 3u) && g1 != pop->n_genomes) return fail(KGX_EINVAL, "g1 must be a multiple of 4 or n_genomes");
    if (g0 == g1 || pop->n_variants == 0) return KGX_OK;
    const uint64_t V = pop->n_variants;
    return for_each_parallel(pop->shards.size(), [&](size_t s) -> int {
      kgx_pop_shard& sh = pop->shards[s];
      const uint64_t lo = g0 > sh.genome_base ? g0 : sh.genome_base;
      const uint64_t hi = g1 < sh.genome_base + sh.n_genomes ? g1 : sh.genome_base + sh.n_genomes;
      if (lo >= hi) return KGX_OK;
      if (int rc = use_device(*sh.dev)) return rc;
      // Stage in slabs of genomes so the staging buffer stays bounded (<= 1 GiB).
      uint64_t slab = (1ull << 30) / V;
      slab = slab / 4 * 4;
      if (slab < 4) slab = 4;
      uint8_t* d_stage = nullptr;
      const uint64_t max_rows = (hi - lo) < slab ? (hi - lo) : slab;
      KGX_HIP_MEM(hipMalloc(&d_stage, max_rows * V));
      int rc = KGX_OK;
      for (uint64_t g = lo; g < hi && rc == KGX_OK; g += slab) {
        const uint64_t n = (hi - g) < slab ? (hi - g) : slab;
        if (hipMemcpyAsync(d_stage, src + (g - g0) * V, n * V, hipMemcpyHostToDevice, sh.dev->stream) != hipSuccess) {
          rc = fail(KGX_EHIP, "H2D copy of dosage rows failed");
          break;
        }
        const uint64_t work = (n + 3) / 4 * V;
        hipLaunchKernelGGL(k_pack_dosage_u8, dim3(stream_grid(*sh.dev, work, kBlock)), dim3(kBlock), 0, sh.dev->stream,
                           d_stage, n, V, g - sh.genome_base, sh.d_rows, sh.pitch);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sh.dev->stream) != hipSuccess)
          rc = fail(KGX_EHIP, "dosage pack kernel failed");
      }
      (void)hipFree(d_stage);
      return rc;
    });
  });
}

int kgx_population_read_dosage2(const kgx_pop* pop, uint8_t* dst, uint64_t dst_pitch, uint64_t v0, uint64_t v1) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !dst) return fail(KGX_EINVAL, "null population or destination");
    if (v0 > v1 || v1 > pop->n_variants) return fail(KGX_EINVAL, "variant range out of bounds");
    if (dst_pitch < (pop->n_genomes + 3) / 4) return fail(KGX_EINVAL, "dst_pitch too small");
    if (v0 == v1) return KGX_OK;
    for (const auto& sh : pop->shards) {
      if (sh.n_genomes == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      KGX_HIP(hipMemcpy2DAsync(dst + sh.genome_base / 4, dst_pitch, sh.d_rows + v0 * sh.pitch, sh.pitch, sh.row_bytes,
                               v1 - v0, hipMemcpyDeviceToHost, sh.dev->stream));
    }
    return sync_shards(pop);
  });
}

int kgx_population_load_phase_plane(kgx_pop* pop, const uint8_t* src, uint64_t src_pitch, uint64_t v0, uint64_t v1) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !src) return fail(KGX_EINVAL, "null population or source");
    if (v0 > v1 || v1 > pop->n_variants) return fail(KGX_EINVAL, "variant range out of bounds");
    if (src_pitch < (pop->n_genomes + 7) / 8) return fail(KGX_EINVAL, "src_pitch too small for one bit per genome");
    for (auto& sh : pop->shards) {
      if (sh.n_genomes == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      if (!sh.d_phase) {                                       // on first use: as many rows as the dosage rows have room for
        sh.phase_pitch = ((sh.n_genomes + 7) / 8 + 15) / 16 * 16;
        const uint64_t bytes = sh.phase_pitch * (sh.capacity ? sh.capacity : 1);
        KGX_HIP_MEM(hipMalloc(&sh.d_phase, bytes));
        KGX_HIP(hipMemsetAsync(sh.d_phase, 0, bytes, sh.dev->stream));
      }
      if (v0 == v1) continue;
      // genome_base is a multiple of 64: the shard's bits start on a byte of the source row
      KGX_HIP(hipMemcpy2DAsync(sh.d_phase + v0 * sh.phase_pitch, sh.phase_pitch, src + sh.genome_base / 8, src_pitch, (sh.n_genomes + 7) / 8, v1 - v0,
                               hipMemcpyHostToDevice, sh.dev->stream));
      // (every shard but the last holds whole 64-genome units: no byte is shared between shards)
    }
    return sync_shards(pop);
  });
}

int kgx_unique_phased_counts(kgx_pop* pop, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
    if (n_bins == 0 || n_bins > 254) return fail(KGX_EINVAL, "n_bins %u outside [1,254]", n_bins);
    if (!bin_of_variant && n_bins != 1) return fail(KGX_EINVAL, "without bin_of_variant every row is counted in ONE bin: n_bins must be 1");
    // UniquePhasedFilter keeps one Variant object per distinct (HGVS, phase): a variant a genome carries counts once, and
    // once more where its copies sit on both phases (the plane's bit)
    std::vector<uint64_t> present(pop->n_genomes * n_bins * 4), both(pop->n_genomes * n_bins);
    int rc = count_by_genome_impl(pop, bin_of_variant, n_bins, present.data());
    if (rc == KGX_OK) rc = count_by_genome_impl(pop, bin_of_variant, n_bins, both.data(), nullptr, true);
    (void)use_device(*pop->shards[0].dev);
    if (rc != KGX_OK) return rc;
    for (uint64_t i = 0; i < pop->n_genomes * n_bins; ++i) out[i] = present[i * 4 + 1] + present[i * 4 + 2] + present[i * 4 + 3] + both[i];
    return KGX_OK;
  });
}

int kgx_population_resize(kgx_pop* pop, uint64_t n_variants) {
  return guarded([&]() -> int {
    if (pop) pop->counts_current = false;
    if (int bound = require_bound()) return bound;
    if (!pop) return fail(KGX_EINVAL, "null population");
    // Two phases, so that a failure leaves every shard as it was: first every allocation, copy and fill that can fail
    // (into blocks the shards do not own yet), then the pointers and the row count of all shards together.
    struct Grown { uint8_t* block = nullptr; uint64_t capacity = 0; uint8_t* plane = nullptr; };
    std::vector<Grown> grown(pop->shards.size());
    int rc = KGX_OK;
    for (size_t s = 0; s < pop->shards.size() && rc == KGX_OK; ++s) {
      kgx_pop_shard& sh = pop->shards[s];
      if ((rc = use_device(*sh.dev)) != KGX_OK) break;
      if (n_variants > sh.capacity) {
        // grow by at least half, so that a population filled piece by piece is copied a bounded number of times
        uint64_t capacity = sh.capacity + sh.capacity / 2;
        if (capacity < n_variants) capacity = n_variants;
        grown[s].capacity = capacity;
        if (sh.pitch * capacity) {
          if (hipMalloc(&grown[s].block, sh.pitch * capacity) != hipSuccess) {
            (void)hipGetLastError();
            grown[s].block = nullptr;
            rc = fail(KGX_ENOMEM, "hipMalloc of %llu bytes for %llu dosage rows on device %d failed", (unsigned long long)(sh.pitch * capacity),
                      (unsigned long long)capacity, sh.dev->id);
            break;
          }
          const uint64_t held = sh.pitch * sh.n_variants;            // the rows in use; what a shrink left behind them is not carried over
          if ((held && hipMemcpyAsync(grown[s].block, sh.d_rows, held, hipMemcpyDeviceToDevice, sh.dev->stream) != hipSuccess) ||
              hipMemsetAsync(grown[s].block + held, 0, sh.pitch * capacity - held, sh.dev->stream) != hipSuccess ||
              hipStreamSynchronize(sh.dev->stream) != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(KGX_EHIP, "copying the dosage rows into the grown allocation failed");
          }
          if (rc == KGX_OK && sh.d_phase) {                       // a loaded phase plane grows with the rows
            const uint64_t held_plane = sh.phase_pitch * sh.n_variants;
            if (hipMalloc(&grown[s].plane, sh.phase_pitch * capacity) != hipSuccess) {
              (void)hipGetLastError();
              grown[s].plane = nullptr;
              rc = fail(KGX_ENOMEM, "hipMalloc of the grown phase plane on device %d failed", sh.dev->id);
            } else if ((held_plane && hipMemcpyAsync(grown[s].plane, sh.d_phase, held_plane, hipMemcpyDeviceToDevice, sh.dev->stream) != hipSuccess) ||
                       hipMemsetAsync(grown[s].plane + held_plane, 0, sh.phase_pitch * capacity - held_plane, sh.dev->stream) != hipSuccess ||
                       hipStreamSynchronize(sh.dev->stream) != hipSuccess) {
              (void)hipGetLastError();
              rc = fail(KGX_EHIP, "copying the phase plane into the grown allocation failed");
            }
          }
        }
      } else if (n_variants > sh.n_variants && sh.pitch) {
        // within the allocation: rows a shrink left behind must read as empty again (rows past n_variants: nobody reads them yet)
        if (hipMemsetAsync(sh.d_rows + sh.pitch * sh.n_variants, 0, sh.pitch * (n_variants - sh.n_variants), sh.dev->stream) != hipSuccess ||
            hipStreamSynchronize(sh.dev->stream) != hipSuccess) {
          (void)hipGetLastError();
          rc = fail(KGX_EHIP, "clearing the rows behind the population failed");
        }
        if (rc == KGX_OK && sh.d_phase &&
            (hipMemsetAsync(sh.d_phase + sh.phase_pitch * sh.n_variants, 0, sh.phase_pitch * (n_variants - sh.n_variants), sh.dev->stream) != hipSuccess ||
             hipStreamSynchronize(sh.dev->stream) != hipSuccess)) {
          (void)hipGetLastError();
          rc = fail(KGX_EHIP, "clearing the phase plane behind the population failed");
        }
      }
    }
    if (rc != KGX_OK) {
      for (size_t s = 0; s < pop->shards.size(); ++s)
        if ((grown[s].block || grown[s].plane) && use_device(*pop->shards[s].dev) == KGX_OK) {
          if (grown[s].block) (void)hipFree(grown[s].block);
          if (grown[s].plane) (void)hipFree(grown[s].plane);
        }
      (void)use_device(*pop->shards[0].dev);
      return rc;
    }
    for (size_t s = 0; s < pop->shards.size(); ++s) {
      kgx_pop_shard& sh = pop->shards[s];
      (void)use_device(*sh.dev);
      if (grown[s].capacity) {
        if (sh.d_alloc) (void)hipFree(sh.d_alloc);
        sh.d_alloc = sh.d_rows = grown[s].block;
        sh.capacity = grown[s].capacity;
        if (grown[s].plane) {
          (void)hipFree(sh.d_phase);
          sh.d_phase = grown[s].plane;
        }
      }
      if (n_variants != sh.n_variants) {                       // the per-variant columns follow the row count: re-created on demand
        if (sh.d_af) { (void)hipFree(sh.d_af); sh.d_af = nullptr; }
        if (sh.d_counts) { (void)hipFree(sh.d_counts); sh.d_counts = nullptr; }
      }
      sh.n_variants = n_variants;
    }
    if (n_variants != pop->n_variants) pop->has_af = false;
    pop->n_variants = n_variants;
    return use_device(*pop->shards[0].dev);
  });
}

int kgx_population_set_genome_mask(kgx_pop* pop, const uint8_t* keep) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop) return fail(KGX_EINVAL, "null population");
    pop->counts_current = false;
    if (!keep) {
      pop->keep.clear();
      for (auto& sh : pop->shards) {
        if (sh.d_keep) {
          if (int rc = use_device(*sh.dev)) return rc;
          KGX_HIP(hipStreamSynchronize(sh.dev->stream));
          (void)hipFree(sh.d_keep);
          sh.d_keep = nullptr;
        }
        sh.n_kept = sh.n_genomes;
      }
      return use_device(*pop->shards[0].dev);
    }
    pop->keep.assign(keep, keep + pop->n_genomes);
    for (auto& sh : pop->shards) {
      if (sh.n_genomes == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      std::vector<uint8_t> row(sh.pitch, 0);
      sh.n_kept = 0;
      for (uint64_t g = 0; g < sh.n_genomes; ++g)
        if (keep[sh.genome_base + g]) {
          row[g >> 2] |= static_cast<uint8_t>(3u << (2 * (g & 3u)));
          ++sh.n_kept;
        }
      if (!sh.d_keep) KGX_HIP_MEM(hipMalloc(&sh.d_keep, sh.pitch));
      KGX_HIP(hipMemcpyAsync(sh.d_keep, row.data(), sh.pitch, hipMemcpyHostToDevice, sh.dev->stream));
      KGX_HIP(hipStreamSynchronize(sh.dev->stream));             // `row` is pageable host memory
    }
    return use_device(*pop->shards[0].dev);
  });
}

int kgx_population_set_af(kgx_pop* pop, const float* af) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !af) return fail(KGX_EINVAL, "null population or af");
    for (auto& sh : pop->shards) {                         // per-variant columns are replicated on every shard
      if (int rc = ensure_af(sh)) return rc;
      if (pop->n_variants) {
        if (int rc = use_device(*sh.dev)) return rc;
        KGX_HIP(hipMemcpyAsync(sh.d_af, af, pop->n_variants * sizeof(float), hipMemcpyHostToDevice, sh.dev->stream));
      }
    }
    if (int rc = sync_shards(pop)) return rc;
    pop->has_af = true;
    return KGX_OK;
  });
}

int kgx_population_get_af(const kgx_pop* pop, float* af) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !af) return fail(KGX_EINVAL, "null population or af");
    if (!pop->has_af) return fail(KGX_ESTATE, "allele frequencies were never set");
    if (pop->n_variants) {
      const auto& sh = pop->shards[0];
      if (int rc = use_device(*sh.dev)) return rc;
      KGX_HIP(hipMemcpyAsync(af, sh.d_af, pop->n_variants * sizeof(float), hipMemcpyDeviceToHost, sh.dev->stream));
      KGX_HIP(hipStreamSynchronize(sh.dev->stream));
    }
    return KGX_OK;
  });
}

int kgx_population_synth_biallelic(kgx_pop* pop, uint64_t seed, uint64_t genome_base, uint64_t variant_base) {
  return guarded([&]() -> int {
    if (pop) pop->counts_current = false;
    if (int bound = require_bound()) return bound;
    if (!pop) return fail(KGX_EINVAL, "null population");
    for (auto& sh : pop->shards) {
      if (int rc = ensure_af(sh)) return rc;
      if (pop->n_variants == 0) continue;
      if (int rc = use_device(*sh.dev)) return rc;
      // a slot beyond the data still gets the allele-frequency column: one chunk per row writes it
      const uint64_t chunks = pop->n_variants * (sh.chunks_per_row ? sh.chunks_per_row : 1);
      if (sh.chunks_per_row == 0) {
        std::vector<float> af(pop->n_variants);
        for (uint64_t v = 0; v < pop->n_variants; ++v) af[v] = kgx_synth_af(seed, variant_base + v);
        KGX_HIP(hipMemcpyAsync(sh.d_af, af.data(), af.size() * sizeof(float), hipMemcpyHostToDevice, sh.dev->stream));
        KGX_HIP(hipStreamSynchronize(sh.dev->stream));
        continue;
      }
      hipLaunchKernelGGL(k_synth_biallelic, dim3(stream_grid(*sh.dev, chunks, kBlock)), dim3(kBlock), 0, sh.dev->stream,
                         reinterpret_cast<kgx_v4u*>(sh.d_rows), sh.chunks_per_row, pop->n_variants,
                         sh.n_genomes, seed, genome_base + sh.genome_base, variant_base, sh.d_af);
      KGX_HIP(hipGetLastError());
    }
    if (int rc = sync_shards(pop)) return rc;
    pop->has_af = true;
    return KGX_OK;
  });
}

int kgx_synth_biallelic_host(uint64_t seed, uint64_t genome_base, uint64_t n_genomes, uint64_t v0,
                             uint64_t v1, uint8_t* dst, uint64_t dst_pitch, float* af_out) {
  return guarded([&]() -> int {
    if (!dst) return fail(KGX_EINVAL, "null destination");
    const uint64_t row_bytes = (n_genomes + 3) / 4;
    if (v0 > v1 || dst_pitch < row_bytes) return fail(KGX_EINVAL, "bad range or pitch");
    for (uint64_t v = v0; v < v1; ++v) {
      const float af = kgx_synth_af(seed, v);
      if (af_out) af_out[v - v0] = af;
      const double p = static_cast<double>(af);
      uint8_t* row = dst + (v - v0) * dst_pitch;
      std::memset(row, 0, dst_pitch);
      for (uint64_t g = 0; g < n_genomes; ++g)
        row[g >> 2] |= static_cast<uint8_t>(kgx_synth_dosage(seed, v, genome_base + g, p) << (2 * (g & 3u)));
    }
    return KGX_OK;
  });
}

int kgx_allele_count_by_locus_dev(kgx_pop* pop, void* d_out, void* stream) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !d_out) return fail(KGX_EINVAL, "null population or output");
    if (pop->n_variants == 0) return KGX_OK;
    return sweep_and_exchange(pop, d_out, static_cast<hipStream_t>(stream), false);
  });
}

int kgx_allele_count_by_locus(kgx_pop* pop, uint32_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
    if (pop->n_variants == 0) return KGX_OK;
    if (int rc = sweep_and_exchange(pop, nullptr, nullptr, true)) return rc;
    kgx_pop_shard& sh = pop->shards[0];
    KGX_HIP(hipMemcpyAsync(out, sh.d_counts, pop->n_variants * 16u, hipMemcpyDeviceToHost, sh.dev->stream));
    const int rc = sync_shards(pop);
    pop->counts_current = rc == KGX_OK;      // every shard's d_counts now holds the population's counts
    return rc;
  });
}

int kgx_allele_frequency_dev(const void* d_counts, uint64_t n_variants, uint64_t total_genomes, void* d_af, void* stream) {
  return guarded([&]() -> int {
    std::shared_ptr<Runtime> rt;
    if (int rc = require_runtime(rt)) return rc;
    if (!d_counts || !d_af) return fail(KGX_EINVAL, "null device pointer");
    if (total_genomes == 0) return fail(KGX_EINVAL, "total_genomes must be > 0");
    if (n_variants == 0) return KGX_OK;
    // the buffers say which device this runs on (a caller's tensors on the first slot's device, normally)
    hipPointerAttribute_t attr;
    const Device* dev = rt->devs[0].get();
    if (hipPointerGetAttributes(&attr, d_counts) == hipSuccess) {
      for (const auto& d : rt->devs)
        if (d->id == attr.device) { dev = d.get(); break; }
    } else {
      (void)hipGetLastError();
    }
    if (int rc = use_device(*dev)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_allele_frequency, dim3(stream_grid(*dev, n_variants, kBlock)), dim3(kBlock), 0, s,
                       static_cast<const kgx_v4u*>(d_counts), n_variants, total_genomes, static_cast<double*>(d_af));
    KGX_HIP(hipGetLastError());
    return KGX_OK;
  });
}

int kgx_allele_count_timed(kgx_pop* pop, void* d_out, void* stream, int warmup, int iters, float* ms_each) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !d_out || !ms_each || iters <= 0 || warmup < 0) return fail(KGX_EINVAL, "bad arguments");
    pop->counts_current = false;             // the shards' d_counts take LOCAL counts below (no exchange)
    const size_t n = pop->shards.size();
    std::vector<void*> buffers(n);
    std::vector<hipStream_t> streams(n);
    for (size_t s = 0; s < n; ++s) {
      kgx_pop_shard& sh = pop->shards[s];
      if (s == 0) { buffers[s] = d_out; streams[s] = static_cast<hipStream_t>(stream); continue; }
      if (int rc = ensure_counts(sh)) return rc;
      buffers[s] = sh.d_counts;
      streams[s] = sh.dev->stream;
    }
    for (int i = 0; i < warmup; ++i)
      for (size_t s = 0; s < n; ++s)
        if (int rc = launch_allele_count(pop->shards[s], buffers[s], streams[s])) return rc;
    std::vector<hipEvent_t> ev(2 * static_cast<size_t>(iters) * n, nullptr);
    int rc = KGX_OK;
    for (size_t s = 0; s < n && rc == KGX_OK; ++s) {
      rc = use_device(*pop->shards[s].dev);
      for (int i = 0; i < 2 * iters && rc == KGX_OK; ++i)
        if (hipEventCreate(&ev[s * 2 * iters + i]) != hipSuccess) rc = fail(KGX_EHIP, "hipEventCreate failed");
    }
    for (int i = 0; i < iters && rc == KGX_OK; ++i)
      for (size_t s = 0; s < n && rc == KGX_OK; ++s) {
        rc = use_device(*pop->shards[s].dev);
        hipEvent_t* e = &ev[s * 2 * iters + 2 * i];
        if (rc == KGX_OK && hipEventRecord(e[0], streams[s]) != hipSuccess) rc = fail(KGX_EHIP, "hipEventRecord failed");
        if (rc == KGX_OK) rc = launch_allele_count(pop->shards[s], buffers[s], streams[s]);
        if (rc == KGX_OK && hipEventRecord(e[1], streams[s]) != hipSuccess) rc = fail(KGX_EHIP, "hipEventRecord failed");
      }
    for (size_t s = 0; s < n && rc == KGX_OK; ++s) {
      rc = use_device(*pop->shards[s].dev);
      if (rc == KGX_OK && hipStreamSynchronize(streams[s]) != hipSuccess) rc = fail(KGX_EHIP, "stream synchronize failed");
    }
    for (int i = 0; i < iters && rc == KGX_OK; ++i) {
      float worst = 0.f;
      for (size_t s = 0; s < n && rc == KGX_OK; ++s) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[s * 2 * iters + 2 * i], ev[s * 2 * iters + 2 * i + 1]) != hipSuccess)
          rc = fail(KGX_EHIP, "hipEventElapsedTime failed");
        worst = ms > worst ? ms : worst;
      }
      ms_each[i] = worst;
    }
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

int kgx_population_summary(kgx_pop* pop, uint64_t out[4]) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (pop->n_variants == 0) return KGX_OK;
    pop->counts_current = false;             // every shard's d_counts takes its LOCAL counts below: k_drop_absent_rows must not read them
    std::vector<unsigned long long> totals(pop->shards.size() * 4, 0);
    const int rc = for_each_parallel(pop->shards.size(), [&](size_t s) -> int {
      kgx_pop_shard& sh = pop->shards[s];
      if (sh.n_genomes == 0) return KGX_OK;
      if (int e = ensure_counts(sh)) return e;
      if (int e = launch_allele_count(sh, sh.d_counts, sh.dev->stream)) return e;
      unsigned long long* d_total = nullptr;
      KGX_HIP_MEM(hipMalloc(&d_total, 4 * sizeof(unsigned long long)));
      int r = KGX_OK;
      if (hipMemsetAsync(d_total, 0, 4 * sizeof(unsigned long long), sh.dev->stream) != hipSuccess) r = fail(KGX_EHIP, "memset failed");
      if (r == KGX_OK) {
        hipLaunchKernelGGL(k_sum_counts, dim3(stream_grid(*sh.dev, sh.n_variants, kBlock)), dim3(kBlock), 0, sh.dev->stream,
                           static_cast<const kgx_v4u*>(sh.d_counts), sh.n_variants, d_total);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(&totals[s * 4], d_total, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, sh.dev->stream) != hipSuccess ||
            hipStreamSynchronize(sh.dev->stream) != hipSuccess)
          r = fail(KGX_EHIP, "population summary reduction failed");
      }
      (void)hipFree(d_total);
      return r;
    });
    if (rc != KGX_OK) return rc;
    for (size_t s = 0; s < pop->shards.size(); ++s)
      for (int j = 0; j < 4; ++j) out[j] += totals[s * 4 + j];
    return use_device(*pop->shards[0].dev);
  });
}

int kgx_count_by_genome(kgx_pop* pop, const uint8_t* variant_mask, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
    int rc;
    if (!variant_mask) {
      rc = count_by_genome_impl(pop, nullptr, 1, out);
    } else {
      std::vector<uint8_t> bins(pop->n_variants);
      for (uint64_t v = 0; v < pop->n_variants; ++v) bins[v] = variant_mask[v] ? 0 : 0xFF;
      rc = count_by_genome_impl(pop, bins.data(), 1, out);
    }
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

int kgx_count_by_genome_binned(kgx_pop* pop, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out || !bin_of_variant) return fail(KGX_EINVAL, "null population, bins or output");
    if (n_bins == 0 || n_bins > 254) return fail(KGX_EINVAL, "n_bins %u outside [1,254]", n_bins);
    const int rc = count_by_genome_impl(pop, bin_of_variant, n_bins, out);
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

int kgx_count_by_genome_af_bins(kgx_pop* pop, const double* bin_edges, uint32_t n_bins, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out || !bin_edges) return fail(KGX_EINVAL, "null population, bin edges or output");
    if (n_bins == 0 || n_bins > 254) return fail(KGX_EINVAL, "n_bins %u outside [1,254]", n_bins);
    if (!pop->has_af) return fail(KGX_ESTATE, "allele frequencies were never set (kgx_population_set_af)");
    const int rc = count_by_genome_impl(pop, nullptr, n_bins, out, bin_edges);
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

double kgx_count_by_genome_last_ms(void) {
  const auto rt = current_runtime();
  double worst = 0.0;
  if (rt)
    for (const auto& dev : rt->devs) { const double ms = dev->last_by_genome_ms.load(); worst = ms > worst ? ms : worst; }
  return worst;
}

int kgx_compound_offsets(kgx_pop* pop, const uint32_t* first_row, const uint32_t* n_rows, const uint32_t* bin,
                         uint64_t n_groups, uint32_t n_bins, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out || (n_groups && (!first_row || !n_rows || !bin))) return fail(KGX_EINVAL, "null argument");
    if (n_bins == 0) return fail(KGX_EINVAL, "n_bins must be > 0");
    std::memset(out, 0, pop->n_genomes * n_bins * 3 * sizeof(uint64_t));
    if (n_groups == 0) return KGX_OK;
    std::vector<OffsetGroup> groups(n_groups);
    for (uint64_t i = 0; i < n_groups; ++i) {
      if (static_cast<uint64_t>(first_row[i]) + n_rows[i] > pop->n_variants || bin[i] >= n_bins)
        return fail(KGX_EINVAL, "group %llu out of range", (unsigned long long)i);
      groups[i] = OffsetGroup{first_row[i], n_rows[i], bin[i], 0};
    }
    const std::vector<uint32_t> no_list;
    const int rc = for_each_parallel(pop->shards.size(), [&](size_t s) {
      kgx_pop_shard& sh = pop->shards[s];
      return compound_offsets_shard(sh, groups, no_list, n_bins, out + sh.genome_base * n_bins * 3);
    });
    if (rc == KGX_OK) zero_masked_genomes(pop, out, static_cast<uint64_t>(n_bins) * 3);
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

int kgx_compound_offsets_listed(kgx_pop* pop, const uint32_t* member_rows, uint64_t n_members, const uint32_t* first_member, const uint32_t* n_rows,
                                const uint32_t* bin, uint64_t n_groups, uint32_t n_bins, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out || (n_groups && (!member_rows || !first_member || !n_rows || !bin))) return fail(KGX_EINVAL, "null argument");
    if (n_bins == 0) return fail(KGX_EINVAL, "n_bins must be > 0");
    if (n_members > 0xFFFFFFFFull) return fail(KGX_EINVAL, "more than 2^32 member rows");
    std::memset(out, 0, pop->n_genomes * n_bins * 3 * sizeof(uint64_t));
    if (n_groups == 0) return KGX_OK;
    std::vector<OffsetGroup> groups(n_groups);
    for (uint64_t i = 0; i < n_groups; ++i) {
      if (static_cast<uint64_t>(first_member[i]) + n_rows[i] > n_members || bin[i] >= n_bins)
        return fail(KGX_EINVAL, "group %llu out of range", (unsigned long long)i);
      groups[i] = OffsetGroup{first_member[i], n_rows[i], bin[i], 0};
    }
    const std::vector<uint32_t> row_list(member_rows, member_rows + n_members);
    for (uint32_t row : row_list)
      if (row >= pop->n_variants) return fail(KGX_EINVAL, "member row %u past the population's %llu rows", row, (unsigned long long)pop->n_variants);
    const int rc = for_each_parallel(pop->shards.size(), [&](size_t s) {
      kgx_pop_shard& sh = pop->shards[s];
      return compound_offsets_shard(sh, groups, row_list, n_bins, out + sh.genome_base * n_bins * 3);
    });
    if (rc == KGX_OK) zero_masked_genomes(pop, out, static_cast<uint64_t>(n_bins) * 3);
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

int kgx_offset_filter_counts(kgx_pop* pop, const uint8_t* single_bin, const uint32_t* member_rows, uint64_t n_members, const uint32_t* first_member,
                             const uint32_t* n_rows, const uint32_t* bin, uint64_t n_groups, uint32_t n_bins, uint64_t* out) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !out || !single_bin || (n_groups && (!member_rows || !first_member || !n_rows || !bin))) return fail(KGX_EINVAL, "null argument");
    if (n_bins == 0 || n_bins > 254) return fail(KGX_EINVAL, "n_bins %u outside [1,254]", n_bins);
    if (n_members > 0xFFFFFFFFull) return fail(KGX_EINVAL, "more than 2^32 member rows");
    const uint64_t cells = pop->n_genomes * n_bins;
    // offsets of one row: its dosage decides, per genome -- the by-genome sweep's counters, re-read per filter
    std::vector<uint64_t> single(cells * 4, 0);
    if (pop->n_variants)
      if (int rc = count_by_genome_impl(pop, single_bin, n_bins, single.data())) { (void)use_device(*pop->shards[0].dev); return rc; }
    for (uint64_t c = 0; c < cells; ++c) {
      const uint64_t het = single[c * 4 + 1], hom = single[c * 4 + 2], more = single[c * 4 + 3];
      out[c * 4 + 0] = 2 * hom;                 // HomozygousFilter: the two copies of the one variant
      out[c * 4 + 1] = het;                     // HeterozygousFilter
      out[c * 4 + 2] = het + 2 * hom;           // DiploidFilter: nothing of an offset holding more than two
      out[c * 4 + 3] = het + hom + more;        // UniqueUnphasedFilter
    }
    if (n_groups == 0) return use_device(*pop->shards[0].dev);
    std::vector<OffsetGroup> groups(n_groups);
    for (uint64_t i = 0; i < n_groups; ++i) {
      if (static_cast<uint64_t>(first_member[i]) + n_rows[i] > n_members || bin[i] >= n_bins)
        return fail(KGX_EINVAL, "group %llu out of range", (unsigned long long)i);
      groups[i] = OffsetGroup{first_member[i], n_rows[i], bin[i], 0};
    }
    const std::vector<uint32_t> row_list(member_rows, member_rows + n_members);
    for (uint32_t row : row_list) {
      if (row >= pop->n_variants) return fail(KGX_EINVAL, "member row %u past the population's %llu rows", row, (unsigned long long)pop->n_variants);
      if (single_bin[row] != 0xFF) return fail(KGX_EINVAL, "row %u is a member of a group and has a bin of its own", row);
    }
    std::vector<uint64_t> grouped(cells * 4, 0);
    const int rc = for_each_parallel(pop->shards.size(), [&](size_t s) {
      kgx_pop_shard& sh = pop->shards[s];
      return compound_offsets_shard(sh, groups, row_list, n_bins, grouped.data() + sh.genome_base * n_bins * 4, 4);
    });
    if (rc == KGX_OK) {
      zero_masked_genomes(pop, grouped.data(), static_cast<uint64_t>(n_bins) * 4);
      for (uint64_t i = 0; i < cells * 4; ++i) out[i] += grouped[i];
    }
    (void)use_device(*pop->shards[0].dev);
    return rc;
  });
}

int kgx_genome_row_lists(kgx_pop* pop, uint64_t g0, uint64_t g1, const uint8_t* row_selected, uint64_t* begin, uint32_t* rows, uint64_t capacity) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!pop || !begin) return fail(KGX_EINVAL, "null population or begin array");
    if (g0 > g1 || g1 > pop->n_genomes) return fail(KGX_EINVAL, "genome range [%llu, %llu) outside the population's %llu genomes", (unsigned long long)g0,
                                                    (unsigned long long)g1, (unsigned long long)pop->n_genomes);
    const size_t n_shards = pop->shards.size();
    std::vector<std::vector<unsigned long long>> shard_begin(n_shards);
    std::vector<std::vector<uint32_t>> shard_rows(n_shards);
    const int rc = for_each_parallel(n_shards, [&](size_t s) -> int {
      kgx_pop_shard& sh = pop->shards[s];
      const uint64_t lo = g0 > sh.genome_base ? g0 : sh.genome_base;
      const uint64_t hi = g1 < sh.genome_base + sh.n_genomes ? g1 : sh.genome_base + sh.n_genomes;
      if (lo >= hi) { shard_begin[s].assign(1, 0); return KGX_OK; }
      return genome_row_lists_shard(sh, lo - sh.genome_base, hi - sh.genome_base, row_selected, shard_begin[s], rows ? &shard_rows[s] : nullptr);
    });
    (void)use_device(*pop->shards[0].dev);
    if (rc != KGX_OK) return rc;
    // the shards hold consecutive genome ranges: their lists follow one another
    uint64_t at = 0, g = 0;
    begin[0] = 0;
    for (size_t s = 0; s < n_shards; ++s) {
      const auto& b = shard_begin[s];
      for (size_t i = 0; i + 1 < b.size(); ++i) begin[++g] = at + b[i + 1];
      const uint64_t total = b.back();
      if (rows) {
        if (at + total > capacity) return fail(KGX_EINVAL, "row lists need %llu entries, capacity is %llu", (unsigned long long)(at + total), (unsigned long long)capacity);
        if (total) std::memcpy(rows + at, shard_rows[s].data(), total * sizeof(uint32_t));
      }
      at += total;
    }
    return KGX_OK;
  });
}

}  // extern "C"
