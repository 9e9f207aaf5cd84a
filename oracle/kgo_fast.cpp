// TEST / BENCH INFRASTRUCTURE ONLY.  NOT the reference's algorithm: a tuned CPU comparator for the allele-count sweep
// (SURVEY.md §8d "optimised CPU": the same 2-bit rows the GPU sweeps, 64-bit popcounts, every host thread), reported
// next to the reference-faithful port so that the GPU is not compared only with a pointer-chasing loop.  Nothing in
// the product links or calls this.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
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
  double best = 1e300;
  for (int rep = 0; rep < repeats; ++rep) {
    std::atomic<uint64_t> next{0};
    constexpr uint64_t kChunk = 4096;
    auto worker = [&]() {
      for (uint64_t begin = next.fetch_add(kChunk); begin < n_rows; begin = next.fetch_add(kChunk)) {
        const uint64_t end = std::min(n_rows, begin + kChunk);
        for (uint64_t r = begin; r < end; ++r) {
          const uint8_t* p = rows + r * pitch;
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
      }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  if (best_seconds) *best_seconds = best;
  return static_cast<int>(n_threads);
}

}  // extern "C"
