// TEST / BENCH INFRASTRUCTURE ONLY.  NOT the reference's algorithm: a tuned CPU comparator for the allele-count sweep
// (SURVEY.md §8d "optimised CPU": the same 2-bit rows the GPU sweeps, 64-bit popcounts, every host thread), reported
// next to the reference-faithful port so that the GPU is not compared only with a pointer-chasing loop.  Nothing in
// the product links or calls this.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

// rows: [n_rows] rows of row_bytes bytes at `pitch`; genome g in bits 2(g%4) of byte g/4; codes 0 ref-hom, 1 het,
// 2 minor-hom, 3 non-diploid.  out[n_rows][4] = { refHom, het, minorHom, nonDiploid }.  Returns the threads used.
__attribute__((target("popcnt")))
int kgo_fast_count_by_variant(const uint8_t* rows, uint64_t n_rows, uint64_t row_bytes, uint64_t pitch, uint64_t n_genomes,
                              uint32_t* out, int threads, int repeats, double* best_seconds) {
  if (!rows || !out || repeats < 1) return -1;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned n_threads = threads > 0 ? static_cast<unsigned>(threads) : hw;
  // Each thread owns a contiguous band of rows and works on its own copy of it, touched first by that thread, so that on
  // a multi-socket host the pages sit next to the cores that sweep them (the caller's buffer was filled by one thread).
  const uint64_t band = (n_rows + n_threads - 1) / n_threads;
  std::vector<uint8_t*> local(n_threads, nullptr);
  auto count_band = [&](unsigned t) {
    const uint64_t begin = std::min<uint64_t>(n_rows, t * band), end = std::min<uint64_t>(n_rows, begin + band);
    const uint8_t* base = local[t];
    for (uint64_t r = begin; r < end; ++r) {
      const uint8_t* p = base + (r - begin) * row_bytes;
      uint64_t het = 0, hom = 0, nd = 0, k = 0;
      for (; k + 8 <= row_bytes; k += 8) {
        uint64_t w;
        std::memcpy(&w, p + k, 8);
        const uint64_t lo = w & 0x5555555555555555ull, hi = (w >> 1) & 0x5555555555555555ull;
        het += static_cast<uint64_t>(__builtin_popcountll(lo & ~hi));
        hom += static_cast<uint64_t>(__builtin_popcountll(hi & ~lo));
        nd += static_cast<uint64_t>(__builtin_popcountll(lo & hi));
      }
      if (k < row_bytes) {
        uint64_t w = 0;
        std::memcpy(&w, p + k, row_bytes - k);
        const uint64_t lo = w & 0x5555555555555555ull, hi = (w >> 1) & 0x5555555555555555ull;
        het += static_cast<uint64_t>(__builtin_popcountll(lo & ~hi));
        hom += static_cast<uint64_t>(__builtin_popcountll(hi & ~lo));
        nd += static_cast<uint64_t>(__builtin_popcountll(lo & hi));
      }
      uint32_t* o = out + r * 4;
      o[0] = static_cast<uint32_t>(n_genomes - het - hom - nd);
      o[1] = static_cast<uint32_t>(het);
      o[2] = static_cast<uint32_t>(hom);
      o[3] = static_cast<uint32_t>(nd);
    }
  };
  auto run = [&](auto fn) {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(fn, t);
    fn(0u);
    for (auto& th : pool) th.join();
  };
  run([&](unsigned t) {                                    // first touch + copy, untimed
    const uint64_t begin = std::min<uint64_t>(n_rows, t * band), end = std::min<uint64_t>(n_rows, begin + band);
    if (end == begin) return;
    local[t] = static_cast<uint8_t*>(std::malloc((end - begin) * row_bytes));
    for (uint64_t r = begin; r < end; ++r) std::memcpy(local[t] + (r - begin) * row_bytes, rows + r * pitch, row_bytes);
  });
  double best = 1e300;
  for (int rep = 0; rep < repeats; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    run(count_band);
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  for (uint8_t* p : local) std::free(p);
  if (best_seconds) *best_seconds = best;
  return static_cast<int>(n_threads);
}

}  // extern "C"
