// Internal declarations shared by the C-ABI translation units.
#ifndef KGX_INTERNAL_H
#define KGX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

struct kgx_pop {
  uint64_t n_genomes = 0;        // genomes in this shard
  uint64_t n_variants = 0;       // variant rows
  uint64_t row_bytes = 0;        // ceil(n_genomes / 4): algorithmic bytes per row
  uint64_t pitch = 0;            // device row pitch, multiple of 16
  uint32_t chunks_per_row = 0;   // pitch / 16
  uint8_t* d_alloc = nullptr;    // the hipMalloc'd block holding d_rows
  uint8_t* d_rows = nullptr;     // [n_variants][pitch] dosage2
  float* d_af = nullptr;         // [n_variants] INFO allele frequency (float32, NaN = missing)
  void* d_counts = nullptr;      // [n_variants][4] u32 scratch for the host-returning entry points
  bool has_af = false;
};

struct kgx_gt8 {
  uint64_t n_genomes = 0;
  uint64_t n_loci = 0;
  uint64_t pitch = 0;            // bytes per locus row, multiple of 128
  uint8_t* d_gt = nullptr;       // [n_loci][pitch]
  // Does any byte hold an allele index 8..14 (a nibble with bit 3 set that is not 15)?  0 = not looked at since the
  // bytes last changed, 1 = none, 2 = some.  Without them the SWAR sweeps need no guard against indexes past their
  // 8-entry tables (k_scan_wide_nibbles; every flattener output at <= 7 reference alts is such a matrix).
  int wide_nibbles = 0;
};

namespace kgx {

struct State {
  bool ready = false;
  int device = -1;
  int compute_units = 0;
  uint64_t hbm_bytes = 0;
  hipStream_t stream = nullptr;
  char name[128] = {0};
  char arch[64] = {0};
  hipEvent_t sweep_begin = nullptr, sweep_end = nullptr;   // bracket the frequency sweep of the last kgx_inbreed call
  double last_sweep_ms = 0.0;
  int last_evaluations = 0;                                // objective evaluations of the last Loglikelihood call
  char* scratch = nullptr;                                 // grow-only arena for kgx_inbreed's per-call buffers
  size_t scratch_bytes = 0;
  char* compact[2] = {nullptr, nullptr};                   // ping-pong buffers of the Loglikelihood search's compaction levels
  size_t compact_bytes[2] = {0, 0};
};

extern State g_state;

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int require_device();
uint32_t stream_grid(uint64_t work_items, uint32_t items_per_block);
int launch_allele_count(const kgx_pop* pop, void* d_out, hipStream_t stream);
int ensure_counts(kgx_pop* pop);

}  // namespace kgx

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
