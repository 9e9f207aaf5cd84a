// Internal declarations shared by the C-ABI translation units (kgx_runtime.hip, kgx_dosage.hip, kgx_inbreed.hip).
//
// A Runtime is one binding of the library to a list of devices (kgx_init).  Every handle keeps the Runtime it was
// created under alive and owns one shard per device slot; nothing per-device lives in process globals -- the only
// global is the pointer to the binding new handles are created under.
#ifndef KGX_INTERNAL_H
#define KGX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kgx.h"

namespace kgx {

// One device slot of a binding: its own stream, scratch arena and timing events.
struct Device {
  int slot = 0;                  // index in Runtime::devs
  int id = -1;                   // HIP ordinal
  int compute_units = 0;
  uint64_t hbm_bytes = 0;
  hipStream_t stream = nullptr;  // the library's stream on this device (non-blocking)
  hipStream_t side_stream = nullptr;   // small per-locus work that runs beside a sweep (the sequential-sum chain)
  hipEvent_t side_begin = nullptr, side_end = nullptr;
  hipEvent_t entries_end = nullptr;    // a large call's table entries, tabulated on the side stream beside the segment defaults
  char name[128] = {0};
  char arch[64] = {0};
  hipEvent_t sweep_begin = nullptr, sweep_end = nullptr;   // bracket the frequency sweep of the last kgx_inbreed call here
  hipEvent_t kernel_begin = nullptr, kernel_end = nullptr; // ... and the one kernel of it that walks the genotype bytes
  hipEvent_t moments_begin = nullptr, moments_end = nullptr;   // ... the class passes of a call that ran on moments (sweeps + merges)
  hipEvent_t search_begin = nullptr, search_end = nullptr;     // ... and the kernel that iterates / searches on them
  hipEvent_t ready = nullptr;                              // cross-stream ordering (kgx_allele_count_by_locus_dev)
  hipEvent_t by_genome_begin = nullptr, by_genome_end = nullptr;   // bracket the K3 kernel of the last by-genome sweep here
  // what the kgx_*_last_* getters report: written by the call that owns `mutex`, read by any thread without it
  std::atomic<double> last_by_genome_ms{0.0};
  std::atomic<double> last_sweep_ms{0.0}, last_kernel_ms{0.0}, last_moments_ms{0.0}, last_search_ms{0.0};
  std::atomic<int> last_evaluations{0};                    // objective evaluations of the last Loglikelihood call here
  std::atomic<int> last_path{0};                           // KGX_PATH_*: what the last kgx_inbreed call here ran on
  char* scratch = nullptr;                                 // grow-only arena for kgx_inbreed's per-call buffers
  size_t scratch_bytes = 0;
  char* compact[2] = {nullptr, nullptr};                   // ping-pong buffers of the Loglikelihood search's compaction levels
  size_t compact_bytes[2] = {0, 0};
  char* pinned[2] = {nullptr, nullptr};                    // kgx_inbreed_batch: page-locked host images of a batch's inputs / results (grow-only)
  size_t pinned_bytes[2] = {0, 0};
  char* words = nullptr;                                   // Loglikelihood by moments: the class passes' hit bits, [block][genome] (grow-only, like the arena)
  size_t words_bytes = 0;
  char* bits = nullptr;                                    // both estimators by moments: every class's hits as bit rows, [slot][genome / 8] (k_class_bits; grow-only)
  size_t bits_bytes = 0;
  void* exchange_stage = nullptr;                          // "peer" exchange: staging for another shard's counts
  size_t exchange_stage_bytes = 0;
  std::mutex mutex;                                        // serialises the per-device state above between handles
  ~Device();
};

enum class Exchange { None, Rccl, Peer };

struct Runtime {
  std::vector<std::unique_ptr<Device>> devs;
  Exchange exchange = Exchange::None;
  std::vector<void*> comms;       // ncclComm_t per slot (Exchange::Rccl)
  ~Runtime();
};

// The binding new handles are created under (null before kgx_init).
std::shared_ptr<Runtime> current_runtime();
// KGX_ENODEVICE (with the "no CPU fallback" message) until kgx_init has succeeded: the first check of every compute entry point.
int require_bound();

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const std::string& last_error();
void set_last_error(const std::string& message);
// The KGX_* switches, from the snapshot taken at kgx_init / kgx_reload_options (never from the environment itself).
int env_int(const char* name, int dflt);
std::string env_str(const char* name);
void reload_options();
uint32_t stream_grid(const Device& dev, uint64_t work_items, uint32_t items_per_block);

// Per-call device buffers come out of ONE grow-only arena per device (a window loop calls kgx_inbreed thousands of times, the
// FWS analysis the by-genome sweep once per population: a dozen hipMalloc / hipFree pairs per call cost more than some of
// the sweeps).  The caller holds the device's mutex; kgx_release_scratch frees the arena.
struct ScratchPlan {
  size_t total = 0;
  size_t add(size_t bytes) {
    const size_t at = total;
    total += (bytes + 255u) & ~static_cast<size_t>(255u);
    return at;
  }
};
int scratch_reserve(Device& dev, size_t bytes, char** out);
// ... and a page-locked host buffer of the device (which: 0 inputs, 1 results), grown when needed; the caller holds the mutex.
int pinned_reserve(Device& dev, int which, size_t bytes, char** out);

// Makes `dev` the calling thread's current HIP device.
int use_device(const Device& dev);

// No exception crosses the C ABI (or leaves a worker thread): host containers sized from caller or device values may throw.
template <typename Fn>
int guarded(Fn&& fn) noexcept {
  try {
    return fn();
  } catch (const std::bad_alloc&) {
    return fail(KGX_ENOMEM, "host allocation failed");
  } catch (const std::length_error&) {
    return fail(KGX_ENOMEM, "host allocation failed (size beyond the container's limit)");
  } catch (const std::exception& e) {
    return fail(KGX_ESTATE, "unexpected exception: %s", e.what());
  }
}

// Run fn(i) for i in [0, n): inline when n == 1, otherwise one host thread per item (each shard drives its own device
// and blocks on its own stream).  Returns the first non-zero code; the failing worker's message becomes this thread's.
template <typename Fn>
int for_each_parallel(size_t n, Fn&& fn) {
  if (n == 0) return KGX_OK;
  if (n == 1) return guarded([&]() -> int { return fn(static_cast<size_t>(0)); });
  std::vector<int> codes(n, KGX_OK);
  std::vector<std::string> messages(n);
  std::vector<std::thread> workers;
  workers.reserve(n);
  for (size_t i = 0; i < n; ++i)
    workers.emplace_back([&, i]() {
      codes[i] = guarded([&]() -> int { return fn(i); });         // an exception leaving a thread would end the process
      if (codes[i] != KGX_OK) messages[i] = last_error();
    });
  for (auto& w : workers) w.join();
  for (size_t i = 0; i < n; ++i)
    if (codes[i] != KGX_OK) {
      set_last_error(messages[i]);
      return codes[i];
    }
  return KGX_OK;
}

// Sum d_counts[slot] (uint32 x n_words each, one buffer per slot, in place) over the slots; queued behind the work
// already on each slot's `streams[slot]`, complete on every slot's stream afterwards.
int exchange_counts(Runtime& rt, const std::vector<void*>& d_counts, uint64_t n_words, const std::vector<hipStream_t>& streams);

}  // namespace kgx

// One genome shard of a population on one device slot.
struct kgx_pop_shard {
  kgx::Device* dev = nullptr;
  uint64_t genome_base = 0;      // first genome of the shard (a multiple of 64)
  uint64_t n_genomes = 0;
  uint64_t n_variants = 0;
  uint64_t capacity = 0;         // rows allocated (>= n_variants; kgx_population_resize)
  uint64_t row_bytes = 0;        // ceil(n_genomes / 4): algorithmic bytes per row
  uint64_t pitch = 0;            // device row pitch, multiple of 16
  uint32_t chunks_per_row = 0;   // pitch / 16
  uint8_t* d_alloc = nullptr;    // the hipMalloc'd block holding d_rows
  uint8_t* d_rows = nullptr;     // [n_variants][pitch] dosage2
  float* d_af = nullptr;         // [n_variants] INFO allele frequency (float32, NaN = missing)
  void* d_counts = nullptr;      // [n_variants][4] u32 scratch for the host-returning entry points
  uint8_t* d_keep = nullptr;     // [pitch] genome mask in the rows' own layout: 0b11 where the genome takes part; null = all do
  uint64_t n_kept = 0;           // genomes of the shard taking part (n_genomes without a mask)
  // phase plane (kgx_population_load_phase_plane): one bit per (row, genome of the shard), rows of phase_pitch bytes
  // (a multiple of 16: 128-genome chunks), capacity rows like d_rows; null until a plane is loaded
  uint8_t* d_phase = nullptr;
  uint64_t phase_pitch = 0;
};

struct kgx_pop {
  std::shared_ptr<kgx::Runtime> rt;
  uint64_t n_genomes = 0;        // all shards
  uint64_t n_variants = 0;
  std::vector<kgx_pop_shard> shards;
  bool has_af = false;
  std::vector<uint8_t> keep;     // kgx_population_set_genome_mask: one byte per genome, empty = no mask
  bool counts_current = false;   // with a mask: every shard's d_counts holds the population's K2 counts under it
};

struct kgx_gt8_shard {
  kgx::Device* dev = nullptr;
  uint64_t genome_base = 0;      // first genome of the shard (a multiple of 128)
  uint64_t n_genomes = 0;
  uint64_t n_loci = 0;
  uint64_t pitch = 0;            // bytes per locus row, multiple of 128
  uint8_t* d_gt = nullptr;       // [n_loci][pitch]
  // Does any byte hold an allele index 8..14 (a nibble with bit 3 set that is not 15)?  0 = not looked at since the
  // bytes last changed, 1 = none, 2 = some.  Without them the SWAR sweeps need no guard against indexes past their
  // 8-entry tables (k_scan_wide_nibbles; every flattener output at <= 7 reference alts is such a matrix).
  int wide_nibbles = 0;
  // Offsets with more than 14 reference alts (kgx_gt8_set_wide_rows): their cells as 16-bit pairs of 8-bit indices.
  uint32_t* d_wide_of_row = nullptr;   // [n_loci]: the row's wide row, 0xFFFFFFFF = none; null = the matrix has none
  uint16_t* d_wide = nullptr;          // [n_wide][wide_pitch]
  uint64_t wide_pitch = 0, n_wide = 0;
};

struct kgx_gt8 {
  std::shared_ptr<kgx::Runtime> rt;
  uint64_t n_genomes = 0;
  uint64_t n_loci = 0;
  std::vector<kgx_gt8_shard> shards;
};

#define KGX_HIP(call)                                                                             \
  do {                                                                                            \
    hipError_t kgx_err_ = (call);                                                                 \
    if (kgx_err_ != hipSuccess) {                                                                 \
      (void)hipGetLastError();                                                                    \
      return ::kgx::fail(KGX_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(kgx_err_),  \
                         __FILE__, __LINE__);                                                     \
    }                                                                                             \
  } while (0)

#define KGX_HIP_MEM(call)                                                                         \
  do {                                                                                            \
    hipError_t kgx_err_ = (call);                                                                 \
    if (kgx_err_ != hipSuccess) {                                                                 \
      (void)hipGetLastError();                                                                    \
      return ::kgx::fail(KGX_ENOMEM, "%s failed: %s (%s:%d)", #call, hipGetErrorString(kgx_err_), \
                         __FILE__, __LINE__);                                                     \
    }                                                                                             \
  } while (0)

#endif  // KGX_INTERNAL_H
