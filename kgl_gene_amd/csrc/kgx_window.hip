// C ABI (include/kgx.h): kgx_inbreed_batch -- many window-sized (genome range, locus list) tasks of the INBREED package in one
// launch per device (kgx_kernels_window.h).  A batch whose tasks do not fit the one-launch kernel is made of kgx_inbreed calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "kgx_kernels_window.h"
#include "kgx_internal.h"

namespace kgx {
namespace {

// the part of a task that lies on one shard
struct Piece { uint32_t task; uint64_t lo, hi; };            // shard-local genomes [lo, hi), lo - first genome of the task = out / start offset

int batch_shard(kgx_gt8_shard& sh, const kgx_inbreed_task* tasks, const std::vector<Piece>& pieces, uint32_t amax, int phased, int algorithm) {
  Device& dev = *sh.dev;
  std::lock_guard<std::mutex> device_lock(dev.mutex);
  if (int rc = use_device(dev)) return rc;
  static_assert(sizeof(kgx_locus_results) == sizeof(LocusResultsDev), "LocusResults layout");
  const uint32_t stride = sweep_stride(amax);
  uint64_t n_loci = 0, n_genomes = 0, most_genomes = 0, most_loci = 0;
  for (const Piece& p : pieces) {
    n_loci += tasks[p.task].n_selected;
    n_genomes += p.hi - p.lo;
    most_genomes = std::max<uint64_t>(most_genomes, p.hi - p.lo);
    most_loci = std::max<uint64_t>(most_loci, tasks[p.task].n_selected);
  }
  if (n_loci >= (1ull << 32) || n_genomes >= (1ull << 32)) return fail(KGX_EINVAL, "kgx_inbreed_batch: batch too large (2^32 loci or genomes on one device)");
  const bool iterative = algorithm == KGX_ALGO_HALL_ME || algorithm == KGX_ALGO_LOGLIKELIHOOD;
  // One packed host image -> ONE copy in: tasks | locus index | start points | allele-frequency rows.
  ScratchPlan in;
  const size_t o_tasks = in.add(pieces.size() * sizeof(WindowTask)), o_index = in.add((n_loci + 1) * sizeof(uint32_t));
  const size_t o_start = in.add((iterative ? n_genomes : 0) * sizeof(double) + 8), o_af = in.add((n_loci ? n_loci : 1) * amax * sizeof(double));
  ScratchPlan plan = in;
  const size_t o_table = plan.add((n_loci ? n_loci : 1) * stride * sizeof(double)), o_valid = plan.add(n_loci + 1);
  const size_t o_out = plan.add(n_genomes * sizeof(LocusResultsDev)), o_evaluations = plan.add(sizeof(unsigned int));
  char* image = nullptr;
  if (int rc = pinned_reserve(dev, 0, in.total, &image)) return rc;
  WindowTask* h_tasks = reinterpret_cast<WindowTask*>(image + o_tasks);
  uint32_t* h_index = reinterpret_cast<uint32_t*>(image + o_index);
  double* h_start = reinterpret_cast<double*>(image + o_start);
  double* h_af = reinterpret_cast<double*>(image + o_af);
  uint64_t locus_at = 0, genome_at = 0;
  for (size_t i = 0; i < pieces.size(); ++i) {
    const Piece& p = pieces[i];
    const kgx_inbreed_task& t = tasks[p.task];
    h_tasks[i] = WindowTask{p.lo, static_cast<uint32_t>(p.hi - p.lo), static_cast<uint32_t>(t.n_selected), static_cast<uint32_t>(locus_at),
                            static_cast<uint32_t>(genome_at)};
    if (t.locus_index) std::memcpy(h_index + locus_at, t.locus_index, t.n_selected * sizeof(uint32_t));
    else for (uint64_t s = 0; s < t.n_selected; ++s) h_index[locus_at + s] = static_cast<uint32_t>(s);
    if (t.n_selected) std::memcpy(h_af + locus_at * amax, t.minor_af, t.n_selected * amax * sizeof(double));
    if (iterative) {
      const uint64_t first = p.lo + sh.genome_base - t.g0;                      // of the task's genomes
      for (uint64_t g = 0; g < p.hi - p.lo; ++g)
        h_start[genome_at + g] = t.start ? t.start[first + g] : (algorithm == KGX_ALGO_HALL_ME ? 0.25 : 0.0);
    }
    locus_at += t.n_selected;
    genome_at += p.hi - p.lo;
  }
  char* arena = nullptr;
  if (int rc = scratch_reserve(dev, plan.total, &arena)) return rc;
  hipStream_t st = dev.stream;
  KGX_HIP(hipMemcpyAsync(arena, image, in.total, hipMemcpyHostToDevice, st));
  unsigned int* d_evaluations = reinterpret_cast<unsigned int*>(arena + o_evaluations);
  KGX_HIP(hipMemsetAsync(d_evaluations, 0, sizeof(unsigned int), st));
  const WindowTask* d_tasks = reinterpret_cast<const WindowTask*>(arena + o_tasks);
  const uint32_t* d_index = reinterpret_cast<const uint32_t*>(arena + o_index);
  const double* d_start = reinterpret_cast<const double*>(arena + o_start);
  const double* d_af = reinterpret_cast<const double*>(arena + o_af);
  double* d_table = reinterpret_cast<double*>(arena + o_table);
  uint8_t* d_valid = reinterpret_cast<uint8_t*>(arena + o_valid);
  LocusResultsDev* d_out = reinterpret_cast<LocusResultsDev*>(arena + o_out);
  if (n_loci) hipLaunchKernelGGL((k_locus_tables<true>), dim3(stream_grid(dev, n_loci, kBlock)), dim3(kBlock), 0, st, d_af, n_loci, amax, 0.0, d_table, d_valid);
  // a wave per genome where the batch has genomes enough to fill the SIMDs with one each and a lane holds the selection in
  // <= 16 cells (as kgx_inbreed decides for one call: KGX_K7_WAVE_GENOMES, KGX_K7_WAVE_LOCI); a block per genome otherwise
  const uint64_t wave_loci = static_cast<uint64_t>(std::min(kGenomeWaveLoci, std::max(1, env_int("KGX_K7_WAVE_LOCI", 1024))));
  const bool per_wave = most_loci <= wave_loci && n_genomes >= static_cast<uint64_t>(std::max(1, env_int("KGX_K7_WAVE_GENOMES", algorithm == KGX_ALGO_HALL_ME ? 1024 : 512)));
  const dim3 grid(static_cast<uint32_t>(per_wave ? (most_genomes + kBlock / kWave - 1) / (kBlock / kWave) : most_genomes), static_cast<uint32_t>(pieces.size()));
  const int search = env_str("KGX_K7_SEARCH") == "brent" ? kSearchBrent : kSearchNelderMead;
#define KGX_WINDOW(ALGO, CELLS, THREADS)                                                                                                \
  hipLaunchKernelGGL((k_inbreed_window<ALGO, CELLS, THREADS>), grid, dim3(kBlock), 0, st, sh.d_gt, sh.pitch, d_tasks, d_index, d_table, d_valid, \
                     amax, phased, search, d_start, d_out, d_evaluations)
#define KGX_WINDOW_CELLS(ALGO)                                                                  \
  do {                                                                                          \
    if (per_wave) {                                                                             \
      if (most_loci <= kWave * 8) KGX_WINDOW(ALGO, 8, kWave);                                   \
      else if (most_loci <= kWave * 16) KGX_WINDOW(ALGO, 16, kWave);                            \
      else KGX_WINDOW(ALGO, 32, kWave);                                                         \
    } else {                                                                                    \
      if (most_loci <= kBlock * 4) KGX_WINDOW(ALGO, 4, kBlock);                                 \
      else if (most_loci <= kBlock * 8) KGX_WINDOW(ALGO, 8, kBlock);                            \
      else if (most_loci <= kBlock * 16) KGX_WINDOW(ALGO, 16, kBlock);                          \
      else KGX_WINDOW(ALGO, 32, kBlock);                                                        \
    }                                                                                           \
  } while (0)
  if (!pieces.empty() && most_genomes) {
    if (algorithm == KGX_ALGO_RITLAND_LOCUS) KGX_WINDOW_CELLS(KGX_ALGO_RITLAND_LOCUS);
    else if (algorithm == KGX_ALGO_SIMPLE) KGX_WINDOW_CELLS(KGX_ALGO_SIMPLE);
    else if (algorithm == KGX_ALGO_HALL_ME) KGX_WINDOW_CELLS(KGX_ALGO_HALL_ME);
    else KGX_WINDOW_CELLS(KGX_ALGO_LOGLIKELIHOOD);
  }
#undef KGX_WINDOW_CELLS
#undef KGX_WINDOW
  KGX_HIP(hipGetLastError());
  // (the results and the evaluation count lie side by side in the arena: ONE copy out)
  char* results = nullptr;
  if (int rc = pinned_reserve(dev, 1, o_evaluations + sizeof(unsigned int) - o_out, &results)) return rc;
  KGX_HIP(hipMemcpyAsync(results, d_out, o_evaluations + sizeof(unsigned int) - o_out, hipMemcpyDeviceToHost, st));
  KGX_HIP(hipStreamSynchronize(st));
  genome_at = 0;
  for (const Piece& p : pieces) {
    const kgx_inbreed_task& t = tasks[p.task];
    std::memcpy(t.out + (p.lo + sh.genome_base - t.g0), results + genome_at * sizeof(LocusResultsDev), (p.hi - p.lo) * sizeof(LocusResultsDev));
    genome_at += p.hi - p.lo;
  }
  unsigned int evaluations = 0;
  std::memcpy(&evaluations, results + (o_evaluations - o_out), sizeof(evaluations));
  if (algorithm == KGX_ALGO_LOGLIKELIHOOD) dev.last_evaluations = static_cast<int>(evaluations);
  dev.last_path = KGX_PATH_ONE_LAUNCH;
  dev.last_sweep_ms = dev.last_kernel_ms = dev.last_moments_ms = dev.last_search_ms = 0.0;
  return KGX_OK;
}

}  // namespace
}  // namespace kgx

using namespace kgx;

extern "C" int kgx_inbreed_batch(kgx_gt8* h, const kgx_inbreed_task* tasks, uint32_t n_tasks, uint32_t amax, int phased, int algorithm) {
  return guarded([&]() -> int {
    if (int bound = require_bound()) return bound;
    if (!h || (n_tasks && !tasks)) return fail(KGX_EINVAL, "null argument");
    if (amax == 0 || amax > 254) return fail(KGX_EINVAL, "amax %u outside [1,254]", amax);
    if (algorithm < 0 || algorithm > 3) return fail(KGX_EINVAL, "unknown algorithm %d", algorithm);
    bool one_launch = !env_int("KGX_K7_NO_WAVE", 0) && !env_int("KGX_BATCH_BY_CALLS", 0) && amax <= 14;   // (wider loci: the generic kernels, call by call)
    for (uint32_t i = 0; i < n_tasks; ++i) {
      const kgx_inbreed_task& t = tasks[i];
      if (!t.out || (t.n_selected && !t.minor_af)) return fail(KGX_EINVAL, "task %u: null argument", i);
      if (t.g0 > t.g1 || t.g1 > h->n_genomes || (t.g0 & 3u)) return fail(KGX_EINVAL, "task %u: genome range must lie in the matrix and start on a multiple of 4", i);
      if (!t.locus_index && t.n_selected > h->n_loci) return fail(KGX_EINVAL, "task %u: n_selected exceeds the locus count", i);
      if (t.locus_index)
        for (uint64_t s = 0; s < t.n_selected; ++s)
          if (t.locus_index[s] >= h->n_loci) return fail(KGX_EINVAL, "task %u: locus_index[%llu] out of range", i, (unsigned long long)s);
      if (t.start && (algorithm == KGX_ALGO_HALL_ME || algorithm == KGX_ALGO_LOGLIKELIHOOD))
        for (uint64_t g = 0; g < t.g1 - t.g0; ++g) {
          const bool ok = algorithm == KGX_ALGO_HALL_ME ? (t.start[g] > 0.0 && t.start[g] <= 1.0) : (t.start[g] >= -1.0 && t.start[g] <= 1.0);
          if (!ok) return fail(KGX_EINVAL, "task %u: start[%llu] = %g outside the estimator's interval", i, (unsigned long long)g, t.start[g]);
        }
      if (t.n_selected == 0 || t.n_selected > static_cast<uint64_t>(kGenomeLoci)) one_launch = false;
    }
    if (!one_launch) {
      // tasks the one-launch kernel does not hold (an empty or a large selection), or the comparison: call by call
      for (uint32_t i = 0; i < n_tasks; ++i) {
        const kgx_inbreed_task& t = tasks[i];
        if (int rc = kgx_inbreed(h, t.g0, t.g1, t.locus_index, t.n_selected, t.minor_af, amax, phased, algorithm, t.start, t.out)) return rc;
      }
      return KGX_OK;
    }
    std::vector<std::vector<Piece>> pieces(h->shards.size());
    for (uint32_t i = 0; i < n_tasks; ++i)
      for (size_t s = 0; s < h->shards.size(); ++s) {
        const kgx_gt8_shard& sh = h->shards[s];
        const uint64_t lo = std::max(tasks[i].g0, sh.genome_base), hi = std::min(tasks[i].g1, sh.genome_base + sh.n_genomes);
        if (lo < hi) pieces[s].push_back(Piece{i, lo - sh.genome_base, hi - sh.genome_base});
      }
    const int rc = for_each_parallel(h->shards.size(), [&](size_t s) -> int {
      if (pieces[s].empty()) return KGX_OK;
      return batch_shard(h->shards[s], tasks, pieces[s], amax, phased, algorithm);
    });
    (void)use_device(*h->shards[0].dev);
    return rc;
  });
}
