// C ABI (include/kgx.h) over the gfx950 kernels.  Owns device memory and the launch logic;
// no CPU fallback exists: without a usable device every compute entry point fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/kgx.h"
#include "kgx_kernels.h"
#include "kgx_internal.h"

namespace {

thread_local std::string g_error;

}  // namespace

namespace kgx {

State g_state;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

int require_device() {
  if (!g_state.ready) return fail(KGX_ENODEVICE, "kgx_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)");
  return KGX_OK;
}

uint32_t stream_grid(uint64_t work_items, uint32_t items_per_block) {
  const uint64_t want = (work_items + items_per_block - 1) / items_per_block;
  const uint64_t cap = static_cast<uint64_t>(g_state.compute_units) * 8u;
  const uint64_t g = want < cap ? want : cap;
  return static_cast<uint32_t>(g ? g : 1);
}

// Lanes cooperating on one row: smallest power of two covering the row's 16-byte chunks, <= 64.
static int lanes_per_row(uint32_t chunks_per_row) {
  int w = 1;
  while (w < 64 && static_cast<uint32_t>(w) < chunks_per_row) w <<= 1;
  return w;
}

// Tuning knobs (environment, read per launch; defaults are the shipped configuration).
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

template <int W, int U, bool NT>
static void launch_count(const kgx_pop* pop, kgx_v4u* d_out, hipStream_t stream) {
  const uint64_t rows_per_iter = static_cast<uint64_t>(kWave / W) * U;
  const uint64_t waves = (pop->n_variants + rows_per_iter - 1) / rows_per_iter;
  const uint64_t want = (waves + (kBlock / kWave) - 1) / (kBlock / kWave);
  const uint64_t cap = static_cast<uint64_t>(g_state.compute_units) * env_int("KGX_K2_BLOCKS_PER_CU", 32);
  const uint32_t grid = static_cast<uint32_t>(want < cap ? (want ? want : 1) : cap);
  hipLaunchKernelGGL((k_allele_count<W, U, NT>), dim3(grid), dim3(kBlock), 0, stream,
                     reinterpret_cast<const kgx_v4u*>(pop->d_rows), pop->chunks_per_row,
                     pop->n_variants, static_cast<uint32_t>(pop->n_genomes), d_out);
}

template <int W>
static void launch_count_w(const kgx_pop* pop, kgx_v4u* out, hipStream_t stream) {
  const int U = env_int("KGX_K2_U", 8);
  const bool nt = env_int("KGX_K2_NT", 1) != 0;
  if (U <= 1)      nt ? launch_count<W, 1, true>(pop, out, stream) : launch_count<W, 1, false>(pop, out, stream);
  else if (U == 2) nt ? launch_count<W, 2, true>(pop, out, stream) : launch_count<W, 2, false>(pop, out, stream);
  else if (U <= 4) nt ? launch_count<W, 4, true>(pop, out, stream) : launch_count<W, 4, false>(pop, out, stream);
  else             nt ? launch_count<W, 8, true>(pop, out, stream) : launch_count<W, 8, false>(pop, out, stream);
}

int launch_allele_count(const kgx_pop* pop, void* d_out, hipStream_t stream) {
  if (pop->n_variants == 0) return KGX_OK;
  kgx_v4u* out = static_cast<kgx_v4u*>(d_out);
  int lanes = lanes_per_row(pop->chunks_per_row);
  const int forced = env_int("KGX_K2_W", 0);             // tuning: fewer lanes per row than the covering power of two
  if (forced > 0 && forced <= lanes && (forced & (forced - 1)) == 0) lanes = forced;
  switch (lanes) {
    case 1:  launch_count_w<1>(pop, out, stream); break;
    case 2:  launch_count_w<2>(pop, out, stream); break;
    case 4:  launch_count_w<4>(pop, out, stream); break;
    case 8:  launch_count_w<8>(pop, out, stream); break;
    case 16: launch_count_w<16>(pop, out, stream); break;
    case 32: launch_count_w<32>(pop, out, stream); break;
    default: launch_count_w<64>(pop, out, stream); break;
  }
  KGX_HIP(hipGetLastError());
  return KGX_OK;
}

int ensure_counts(kgx_pop* pop) {
  if (!pop->d_counts && pop->n_variants) {
    KGX_HIP_MEM(hipMalloc(&pop->d_counts, pop->n_variants * 16u));
  }
  return KGX_OK;
}


// ---- K3 host glue -----------------------------------------------------------------------------

template <int W>
static void launch_by_genome(const kgx_pop* pop, const uint32_t* d_index, const GenomeWork* d_work,
                             uint32_t n_work, uint32_t n_bins, unsigned long long* d_acc) {
  hipLaunchKernelGGL((k_count_by_genome<W>), dim3(n_work), dim3(kBlock), 0, g_state.stream,
                     reinterpret_cast<const kgx_v4u*>(pop->d_rows), pop->chunks_per_row, pop->n_genomes,
                     d_index, d_work, n_bins, d_acc);
}

static int count_by_genome_impl(kgx_pop* pop, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out) {
  const uint64_t V = pop->n_variants, G = pop->n_genomes;
  if (V > 0xFFFFFFFFull) return fail(KGX_EINVAL, "n_variants exceeds the 32-bit row index of the by-genome sweep");
  const uint64_t cells = G * n_bins;
  const bool identity = bin_of_variant == nullptr;
  hipStream_t st = g_state.stream;

  unsigned long long *d_acc = nullptr, *d_out = nullptr, *d_nbin = nullptr, *d_binoff = nullptr;
  uint32_t *d_index = nullptr, *d_chunks = nullptr;
  uint8_t* d_bins = nullptr;
  GenomeWork* d_work = nullptr;
  int rc = KGX_OK;
  auto try_hip = [&](hipError_t e, int code, const char* what) {
    if (rc == KGX_OK && e != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(code, "count_by_genome: %s failed: %s", what, hipGetErrorString(e));
    }
  };
  try_hip(hipMalloc(&d_acc, (cells ? cells : 1) * 3 * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(acc)");
  try_hip(hipMalloc(&d_out, (cells ? cells : 1) * 4 * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(out)");
  try_hip(hipMalloc(&d_nbin, n_bins * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(rows_in_bin)");
  try_hip(hipMalloc(&d_binoff, (n_bins + 1) * sizeof(unsigned long long)), KGX_ENOMEM, "hipMalloc(bin_offset)");
  try_hip(hipMemsetAsync(d_acc, 0, (cells ? cells : 1) * 3 * sizeof(unsigned long long), st), KGX_EHIP, "memset(acc)");

  // Rows grouped by bin, so that a workgroup only ever touches one bin: on the device (k_bin_count / _scan / _scatter).
  std::vector<unsigned long long> bin_offset(n_bins + 1, 0);
  if (identity) {
    bin_offset[1] = V;
    const unsigned long long v = V;
    try_hip(hipMemcpyAsync(d_nbin, &v, sizeof(v), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(rows_in_bin)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
  } else if (V > 0) {
    const uint32_t n_chunks = static_cast<uint32_t>((V + kBinChunk - 1) / kBinChunk);
    try_hip(hipMalloc(&d_bins, V), KGX_ENOMEM, "hipMalloc(bins)");
    try_hip(hipMalloc(&d_index, (V + 8) * sizeof(uint32_t)), KGX_ENOMEM, "hipMalloc(index)");      // + 8: whole 8-entry scalar fetches
    try_hip(hipMemsetAsync(d_index + V, 0, 8 * sizeof(uint32_t), st), KGX_EHIP, "memset(index pad)");
    try_hip(hipMalloc(&d_chunks, static_cast<uint64_t>(n_chunks) * n_bins * sizeof(uint32_t)), KGX_ENOMEM, "hipMalloc(chunk counts)");
    try_hip(hipMemcpyAsync(d_bins, bin_of_variant, V, hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(bins)");
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

  if (rc == KGX_OK && selected > 0) {
    const int W = lanes_per_row(pop->chunks_per_row);
    const uint32_t n_cg = (pop->chunks_per_row + 63) / 64;
    const uint64_t gran = static_cast<uint64_t>(64 / W) * (kBlock / kWave) * 8;
    const uint64_t target = static_cast<uint64_t>(g_state.compute_units) * 10u;
    uint64_t per_wg = (selected * n_cg + target - 1) / target;
    per_wg = (per_wg + gran - 1) / gran * gran;
    if (per_wg < gran * 4) per_wg = gran * 4;
    std::vector<GenomeWork> work;
    for (uint32_t b = 0; b < n_bins; ++b)
      for (uint64_t p = bin_offset[b]; p < bin_offset[b + 1]; p += per_wg)
        for (uint32_t cg = 0; cg < n_cg; ++cg) {
          GenomeWork w;
          w.begin = p;
          w.end = (p + per_wg < bin_offset[b + 1]) ? p + per_wg : bin_offset[b + 1];
          w.col_group = cg;
          w.bin = b;
          work.push_back(w);
        }
    try_hip(hipMalloc(&d_work, work.size() * sizeof(GenomeWork)), KGX_ENOMEM, "hipMalloc(work)");
    try_hip(hipMemcpyAsync(d_work, work.data(), work.size() * sizeof(GenomeWork), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(work)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");     // `work` is pageable host memory
    if (rc == KGX_OK) {
      const uint32_t n_work = static_cast<uint32_t>(work.size());
      switch (W) {
        case 1:  launch_by_genome<1>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
        case 2:  launch_by_genome<2>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
        case 4:  launch_by_genome<4>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
        case 8:  launch_by_genome<8>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
        case 16: launch_by_genome<16>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
        case 32: launch_by_genome<32>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
        default: launch_by_genome<64>(pop, d_index, d_work, n_work, n_bins, d_acc); break;
      }
      try_hip(hipGetLastError(), KGX_EHIP, "k_count_by_genome launch");
    }
  }
  if (rc == KGX_OK && cells > 0) {
    hipLaunchKernelGGL(k_finish_by_genome, dim3(stream_grid(cells, kBlock)), dim3(kBlock), 0, st, d_acc, d_nbin, G, n_bins, d_out);
    try_hip(hipGetLastError(), KGX_EHIP, "k_finish_by_genome launch");
    try_hip(hipMemcpyAsync(out, d_out, cells * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(out)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "stream synchronize");
  }
  for (void* p : {static_cast<void*>(d_acc), static_cast<void*>(d_out), static_cast<void*>(d_nbin), static_cast<void*>(d_binoff),
                  static_cast<void*>(d_index), static_cast<void*>(d_chunks), static_cast<void*>(d_bins), static_cast<void*>(d_work)})
    if (p) (void)hipFree(p);
  return rc;
}

}  // namespace kgx

using namespace kgx;

extern "C" {

const char* kgx_version(void) { return "kgx 0.1.0 (gfx950)"; }

const char* kgx_last_error(void) { return g_error.c_str(); }

int kgx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int kgx_init(int device) {
  int n = kgx_device_count();
  if (n <= 0) return fail(KGX_ENODEVICE, "no HIP device visible (there is no CPU fallback)");
  if (device < 0 || device >= n) return fail(KGX_EINVAL, "device %d out of range [0,%d)", device, n);
  KGX_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  KGX_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(KGX_ENODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  if (g_state.ready && g_state.device == device) return KGX_OK;
  if (g_state.ready && g_state.stream) {      // rebinding to another device: drop what lives on the old one
    (void)hipStreamDestroy(g_state.stream);
    g_state.stream = nullptr;
    if (g_state.sweep_begin) (void)hipEventDestroy(g_state.sweep_begin);
    if (g_state.sweep_end) (void)hipEventDestroy(g_state.sweep_end);
    g_state.sweep_begin = g_state.sweep_end = nullptr;
    if (g_state.scratch) (void)hipFree(g_state.scratch);
    g_state.scratch = nullptr;
    g_state.scratch_bytes = 0;
    for (int k = 0; k < 2; ++k) {
      if (g_state.compact[k]) (void)hipFree(g_state.compact[k]);
      g_state.compact[k] = nullptr;
      g_state.compact_bytes[k] = 0;
    }
  }
  KGX_HIP(hipStreamCreateWithFlags(&g_state.stream, hipStreamNonBlocking));
  g_state.device = device;
  g_state.compute_units = prop.multiProcessorCount;
  g_state.hbm_bytes = prop.totalGlobalMem;
  std::snprintf(g_state.name, sizeof(g_state.name), "%s", prop.name);
  std::snprintf(g_state.arch, sizeof(g_state.arch), "%s", prop.gcnArchName);
  g_state.ready = true;
  return KGX_OK;
}

int kgx_device_info(char* name, size_t name_len, char* arch, size_t arch_len, int* compute_units,
                    uint64_t* hbm_bytes) {
  if (int rc = require_device()) return rc;
  if (name && name_len) std::snprintf(name, name_len, "%s", g_state.name);
  if (arch && arch_len) std::snprintf(arch, arch_len, "%s", g_state.arch);
  if (compute_units) *compute_units = g_state.compute_units;
  if (hbm_bytes) *hbm_bytes = g_state.hbm_bytes;
  return KGX_OK;
}

void* kgx_stream(void) { return g_state.ready ? static_cast<void*>(g_state.stream) : nullptr; }

int kgx_synchronize(void) {
  if (int rc = require_device()) return rc;
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  KGX_HIP(hipDeviceSynchronize());
  return KGX_OK;
}

kgx_pop* kgx_population_create(uint64_t n_genomes, uint64_t n_variants) {
  if (require_device()) return nullptr;
  if (n_genomes == 0 || n_genomes > (1ull << 31)) {
    fail(KGX_EINVAL, "n_genomes %llu outside (0, 2^31] (uint32 per-variant counts)", (unsigned long long)n_genomes);
    return nullptr;
  }
  kgx_pop* pop = new (std::nothrow) kgx_pop();
  if (!pop) { fail(KGX_ENOMEM, "host allocation failed"); return nullptr; }
  pop->n_genomes = n_genomes;
  pop->n_variants = n_variants;
  pop->row_bytes = (n_genomes + 3) / 4;
  {
    // Rows longer than half a wave-load start on a 128-byte line so that every 1 KiB wave load covers
    // whole lines (measured +6 % on 2500-byte rows); short rows stay densely packed.
    int align = env_int("KGX_PITCH_ALIGN", pop->row_bytes > 512 ? 128 : 16);
    if (align < 16 || (align & (align - 1))) align = 16;
    pop->pitch = (pop->row_bytes + align - 1) / align * align;
  }
  pop->chunks_per_row = static_cast<uint32_t>(pop->pitch / 16);
  const uint64_t bytes = pop->pitch * n_variants;
  if (bytes) {
    if (hipMalloc(&pop->d_alloc, bytes) != hipSuccess) {
      (void)hipGetLastError();
      fail(KGX_ENOMEM, "hipMalloc of %llu bytes for %llu x %llu dosage rows failed",
           (unsigned long long)bytes, (unsigned long long)n_variants, (unsigned long long)n_genomes);
      delete pop;
      return nullptr;
    }
    pop->d_rows = pop->d_alloc;
    if (hipMemsetAsync(pop->d_rows, 0, bytes, g_state.stream) != hipSuccess ||
        hipStreamSynchronize(g_state.stream) != hipSuccess) {
      fail(KGX_EHIP, "hipMemset of dosage rows failed");
      (void)hipFree(pop->d_alloc);
      delete pop;
      return nullptr;
    }
  }
  return pop;
}

void kgx_population_destroy(kgx_pop* pop) {
  if (!pop) return;
  if (pop->d_alloc) (void)hipFree(pop->d_alloc);
  if (pop->d_af) (void)hipFree(pop->d_af);
  if (pop->d_counts) (void)hipFree(pop->d_counts);
  delete pop;
}

uint64_t kgx_population_genomes(const kgx_pop* pop) { return pop ? pop->n_genomes : 0; }
uint64_t kgx_population_variants(const kgx_pop* pop) { return pop ? pop->n_variants : 0; }
uint64_t kgx_population_row_pitch(const kgx_pop* pop) { return pop ? pop->pitch : 0; }
uint64_t kgx_population_sweep_bytes(const kgx_pop* pop) {
  return pop ? pop->n_variants * pop->row_bytes + 16u * pop->n_variants : 0;
}

int kgx_population_load_dosage2(kgx_pop* pop, const uint8_t* src, uint64_t src_pitch, uint64_t v0,
                                uint64_t v1) {
  if (int rc = require_device()) return rc;
  if (!pop || !src) return fail(KGX_EINVAL, "null population or source");
  if (v0 > v1 || v1 > pop->n_variants) return fail(KGX_EINVAL, "variant range [%llu,%llu) outside [0,%llu)",
      (unsigned long long)v0, (unsigned long long)v1, (unsigned long long)pop->n_variants);
  if (src_pitch < pop->row_bytes) return fail(KGX_EINVAL, "src_pitch %llu < row bytes %llu",
      (unsigned long long)src_pitch, (unsigned long long)pop->row_bytes);
  if (v0 == v1) return KGX_OK;
  KGX_HIP(hipMemcpy2DAsync(pop->d_rows + v0 * pop->pitch, pop->pitch, src, src_pitch, pop->row_bytes,
                           v1 - v0, hipMemcpyHostToDevice, g_state.stream));
  const uint64_t touched = (v1 - v0) * (pop->pitch - pop->row_bytes + 1);
  hipLaunchKernelGGL(k_mask_row_tail, dim3(stream_grid(touched, kBlock)), dim3(kBlock), 0, g_state.stream,
                     pop->d_rows, pop->pitch, pop->n_genomes, v0, v1);
  KGX_HIP(hipGetLastError());
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  return KGX_OK;
}

int kgx_population_load_dosage_u8(kgx_pop* pop, const uint8_t* src, uint64_t g0, uint64_t g1) {
  if (int rc = require_device()) return rc;
  if (!pop || !src) return fail(KGX_EINVAL, "null population or source");
  if (g0 > g1 || g1 > pop->n_genomes) return fail(KGX_EINVAL, "genome range [%llu,%llu) outside [0,%llu)",
      (unsigned long long)g0, (unsigned long long)g1, (unsigned long long)pop->n_genomes);
  if (g0 & 3u) return fail(KGX_EINVAL, "g0 must be a multiple of 4 (whole packed bytes)");
  if ((g1 & 3u) && g1 != pop->n_genomes) return fail(KGX_EINVAL, "g1 must be a multiple of 4 or n_genomes");
  if (g0 == g1 || pop->n_variants == 0) return KGX_OK;
  // Stage in slabs of genomes so the staging buffer stays bounded (<= 1 GiB).
  const uint64_t V = pop->n_variants;
  uint64_t slab = (1ull << 30) / V;
  slab = slab / 4 * 4;
  if (slab < 4) slab = 4;
  uint8_t* d_stage = nullptr;
  const uint64_t max_rows = (g1 - g0) < slab ? (g1 - g0) : slab;
  KGX_HIP_MEM(hipMalloc(&d_stage, max_rows * V));
  int rc = KGX_OK;
  for (uint64_t g = g0; g < g1 && rc == KGX_OK; g += slab) {
    const uint64_t n = (g1 - g) < slab ? (g1 - g) : slab;
    if (hipMemcpyAsync(d_stage, src + (g - g0) * V, n * V, hipMemcpyHostToDevice, g_state.stream) != hipSuccess) {
      rc = fail(KGX_EHIP, "H2D copy of dosage rows failed");
      break;
    }
    const uint64_t work = (n + 3) / 4 * V;
    hipLaunchKernelGGL(k_pack_dosage_u8, dim3(stream_grid(work, kBlock)), dim3(kBlock), 0, g_state.stream,
                       d_stage, n, V, g, pop->d_rows, pop->pitch);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_state.stream) != hipSuccess)
      rc = fail(KGX_EHIP, "dosage pack kernel failed");
  }
  (void)hipFree(d_stage);
  return rc;
}

int kgx_population_read_dosage2(const kgx_pop* pop, uint8_t* dst, uint64_t dst_pitch, uint64_t v0,
                                uint64_t v1) {
  if (int rc = require_device()) return rc;
  if (!pop || !dst) return fail(KGX_EINVAL, "null population or destination");
  if (v0 > v1 || v1 > pop->n_variants) return fail(KGX_EINVAL, "variant range out of bounds");
  if (dst_pitch < pop->row_bytes) return fail(KGX_EINVAL, "dst_pitch too small");
  if (v0 == v1) return KGX_OK;
  KGX_HIP(hipMemcpy2DAsync(dst, dst_pitch, pop->d_rows + v0 * pop->pitch, pop->pitch, pop->row_bytes,
                           v1 - v0, hipMemcpyDeviceToHost, g_state.stream));
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  return KGX_OK;
}

static int ensure_af(kgx_pop* pop) {
  if (!pop->d_af && pop->n_variants) KGX_HIP_MEM(hipMalloc(&pop->d_af, pop->n_variants * sizeof(float)));
  return KGX_OK;
}

int kgx_population_set_af(kgx_pop* pop, const float* af) {
  if (int rc = require_device()) return rc;
  if (!pop || !af) return fail(KGX_EINVAL, "null population or af");
  if (int rc = ensure_af(pop)) return rc;
  if (pop->n_variants) {
    KGX_HIP(hipMemcpyAsync(pop->d_af, af, pop->n_variants * sizeof(float), hipMemcpyHostToDevice, g_state.stream));
    KGX_HIP(hipStreamSynchronize(g_state.stream));
  }
  pop->has_af = true;
  return KGX_OK;
}

int kgx_population_get_af(const kgx_pop* pop, float* af) {
  if (int rc = require_device()) return rc;
  if (!pop || !af) return fail(KGX_EINVAL, "null population or af");
  if (!pop->has_af) return fail(KGX_ESTATE, "allele frequencies were never set");
  if (pop->n_variants) {
    KGX_HIP(hipMemcpyAsync(af, pop->d_af, pop->n_variants * sizeof(float), hipMemcpyDeviceToHost, g_state.stream));
    KGX_HIP(hipStreamSynchronize(g_state.stream));
  }
  return KGX_OK;
}

int kgx_population_synth_biallelic(kgx_pop* pop, uint64_t seed, uint64_t genome_base, uint64_t variant_base) {
  if (int rc = require_device()) return rc;
  if (!pop) return fail(KGX_EINVAL, "null population");
  if (int rc = ensure_af(pop)) return rc;
  if (pop->n_variants == 0) return KGX_OK;
  const uint64_t chunks = pop->n_variants * pop->chunks_per_row;
  hipLaunchKernelGGL(k_synth_biallelic, dim3(stream_grid(chunks, kBlock)), dim3(kBlock), 0, g_state.stream,
                     reinterpret_cast<kgx_v4u*>(pop->d_rows), pop->chunks_per_row, pop->n_variants,
                     pop->n_genomes, seed, genome_base, variant_base, pop->d_af);
  KGX_HIP(hipGetLastError());
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  pop->has_af = true;
  return KGX_OK;
}

int kgx_synth_biallelic_host(uint64_t seed, uint64_t genome_base, uint64_t n_genomes, uint64_t v0,
                             uint64_t v1, uint8_t* dst, uint64_t dst_pitch, float* af_out) {
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
}

int kgx_allele_count_by_locus_dev(kgx_pop* pop, void* d_out, void* stream) {
  if (int rc = require_device()) return rc;
  if (!pop || !d_out) return fail(KGX_EINVAL, "null population or output");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return launch_allele_count(pop, d_out, s);
}

int kgx_allele_count_by_locus(kgx_pop* pop, uint32_t* out) {
  if (int rc = require_device()) return rc;
  if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
  if (pop->n_variants == 0) return KGX_OK;
  if (int rc = ensure_counts(pop)) return rc;
  if (int rc = launch_allele_count(pop, pop->d_counts, g_state.stream)) return rc;
  KGX_HIP(hipMemcpyAsync(out, pop->d_counts, pop->n_variants * 16u, hipMemcpyDeviceToHost, g_state.stream));
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  return KGX_OK;
}

int kgx_allele_frequency_dev(const void* d_counts, uint64_t n_variants, uint64_t total_genomes, void* d_af,
                             void* stream) {
  if (int rc = require_device()) return rc;
  if (!d_counts || !d_af) return fail(KGX_EINVAL, "null device pointer");
  if (total_genomes == 0) return fail(KGX_EINVAL, "total_genomes must be > 0");
  if (n_variants == 0) return KGX_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_allele_frequency, dim3(stream_grid(n_variants, kBlock)), dim3(kBlock), 0, s,
                     static_cast<const kgx_v4u*>(d_counts), n_variants, total_genomes, static_cast<double*>(d_af));
  KGX_HIP(hipGetLastError());
  return KGX_OK;
}

int kgx_allele_count_timed(kgx_pop* pop, void* d_out, void* stream, int warmup, int iters, float* ms_each) {
  if (int rc = require_device()) return rc;
  if (!pop || !d_out || !ms_each || iters <= 0 || warmup < 0) return fail(KGX_EINVAL, "bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i = 0; i < warmup; ++i)
    if (int rc = launch_allele_count(pop, d_out, s)) return rc;
  std::vector<hipEvent_t> ev(2 * static_cast<size_t>(iters));
  for (auto& e : ev) KGX_HIP(hipEventCreate(&e));
  int rc = KGX_OK;
  for (int i = 0; i < iters && rc == KGX_OK; ++i) {
    if (hipEventRecord(ev[2 * i], s) != hipSuccess) { rc = fail(KGX_EHIP, "hipEventRecord failed"); break; }
    rc = launch_allele_count(pop, d_out, s);
    if (rc == KGX_OK && hipEventRecord(ev[2 * i + 1], s) != hipSuccess) rc = fail(KGX_EHIP, "hipEventRecord failed");
  }
  if (rc == KGX_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(KGX_EHIP, "stream synchronize failed");
  for (int i = 0; i < iters && rc == KGX_OK; ++i)
    if (hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess)
      rc = fail(KGX_EHIP, "hipEventElapsedTime failed");
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int kgx_population_summary(kgx_pop* pop, uint64_t out[4]) {
  if (int rc = require_device()) return rc;
  if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (pop->n_variants == 0) return KGX_OK;
  if (int rc = ensure_counts(pop)) return rc;
  if (int rc = launch_allele_count(pop, pop->d_counts, g_state.stream)) return rc;
  unsigned long long* d_total = nullptr;
  KGX_HIP_MEM(hipMalloc(&d_total, 4 * sizeof(unsigned long long)));
  int rc = KGX_OK;
  if (hipMemsetAsync(d_total, 0, 4 * sizeof(unsigned long long), g_state.stream) != hipSuccess) rc = fail(KGX_EHIP, "memset failed");
  if (rc == KGX_OK) {
    hipLaunchKernelGGL(k_sum_counts, dim3(stream_grid(pop->n_variants, kBlock)), dim3(kBlock), 0, g_state.stream,
                       static_cast<const kgx_v4u*>(pop->d_counts), pop->n_variants, d_total);
    unsigned long long h[4];
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h, d_total, sizeof(h), hipMemcpyDeviceToHost, g_state.stream) != hipSuccess ||
        hipStreamSynchronize(g_state.stream) != hipSuccess) {
      rc = fail(KGX_EHIP, "population summary reduction failed");
    } else {
      for (int j = 0; j < 4; ++j) out[j] = h[j];
    }
  }
  (void)hipFree(d_total);
  return rc;
}

int kgx_count_by_genome(kgx_pop* pop, const uint8_t* variant_mask, uint64_t* out) {
  if (int rc = require_device()) return rc;
  if (!pop || !out) return fail(KGX_EINVAL, "null population or output");
  if (!variant_mask) return count_by_genome_impl(pop, nullptr, 1, out);
  std::vector<uint8_t> bins(pop->n_variants);
  for (uint64_t v = 0; v < pop->n_variants; ++v) bins[v] = variant_mask[v] ? 0 : 0xFF;
  return count_by_genome_impl(pop, bins.data(), 1, out);
}

int kgx_count_by_genome_binned(kgx_pop* pop, const uint8_t* bin_of_variant, uint32_t n_bins, uint64_t* out) {
  if (int rc = require_device()) return rc;
  if (!pop || !out || !bin_of_variant) return fail(KGX_EINVAL, "null population, bins or output");
  if (n_bins == 0 || n_bins > 254) return fail(KGX_EINVAL, "n_bins %u outside [1,254]", n_bins);
  return count_by_genome_impl(pop, bin_of_variant, n_bins, out);
}

int kgx_compound_offsets(kgx_pop* pop, const uint32_t* first_row, const uint32_t* n_rows, const uint32_t* bin,
                         uint64_t n_groups, uint32_t n_bins, uint64_t* out) {
  if (int rc = require_device()) return rc;
  if (!pop || !out || (n_groups && (!first_row || !n_rows || !bin))) return fail(KGX_EINVAL, "null argument");
  if (n_bins == 0) return fail(KGX_EINVAL, "n_bins must be > 0");
  const uint64_t G = pop->n_genomes;
  const uint64_t cells = G * n_bins * 3;
  std::memset(out, 0, cells * sizeof(uint64_t));
  if (n_groups == 0) return KGX_OK;
  std::vector<OffsetGroup> groups(n_groups);
  for (uint64_t i = 0; i < n_groups; ++i) {
    if (n_rows[i] > 15) return fail(KGX_EINVAL, "group %llu has %u rows; at most 15 distinct variants per offset are supported",
                                    (unsigned long long)i, n_rows[i]);
    if (static_cast<uint64_t>(first_row[i]) + n_rows[i] > pop->n_variants || bin[i] >= n_bins)
      return fail(KGX_EINVAL, "group %llu out of range", (unsigned long long)i);
    groups[i] = OffsetGroup{first_row[i], n_rows[i], bin[i], 0};
  }
  OffsetGroup* d_groups = nullptr;
  unsigned long long* d_acc = nullptr;
  int rc = KGX_OK;
  if (hipMalloc(&d_groups, n_groups * sizeof(OffsetGroup)) != hipSuccess || hipMalloc(&d_acc, cells * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    rc = fail(KGX_ENOMEM, "compound_offsets: hipMalloc failed");
  }
  if (rc == KGX_OK) {
    const uint64_t cols = (G + 15) / 16;
    const uint32_t gx = static_cast<uint32_t>((cols + kBlock - 1) / kBlock);
    uint64_t slices = (static_cast<uint64_t>(g_state.compute_units) * 8 + gx - 1) / gx;
    if (slices > n_groups) slices = n_groups;
    if (slices > 65535) slices = 65535;
    const uint64_t per_slice = (n_groups + slices - 1) / slices;
    const uint32_t gy = static_cast<uint32_t>((n_groups + per_slice - 1) / per_slice);
    if (hipMemsetAsync(d_acc, 0, cells * sizeof(unsigned long long), g_state.stream) != hipSuccess ||
        hipMemcpyAsync(d_groups, groups.data(), n_groups * sizeof(OffsetGroup), hipMemcpyHostToDevice, g_state.stream) != hipSuccess) {
      rc = fail(KGX_EHIP, "compound_offsets: upload failed");
    } else {
      hipLaunchKernelGGL(k_compound_offsets, dim3(gx, gy), dim3(kBlock), 0, g_state.stream,
                         reinterpret_cast<const uint32_t*>(pop->d_rows), pop->pitch / 4, G, d_groups, n_groups, per_slice,
                         n_bins, d_acc);
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(out, d_acc, cells * sizeof(unsigned long long), hipMemcpyDeviceToHost, g_state.stream) != hipSuccess ||
          hipStreamSynchronize(g_state.stream) != hipSuccess)
        rc = fail(KGX_EHIP, "compound_offsets: kernel or readback failed");
    }
  }
  if (d_groups) (void)hipFree(d_groups);
  if (d_acc) (void)hipFree(d_acc);
  return rc;
}

// ---- gt8 + inbreeding -------------------------------------------------------------------------------

kgx_gt8* kgx_gt8_create(uint64_t n_genomes, uint64_t n_loci) {
  if (require_device()) return nullptr;
  if (n_genomes == 0) { fail(KGX_EINVAL, "n_genomes must be > 0"); return nullptr; }
  if (n_loci > 0xFFFFFFFFull) { fail(KGX_EINVAL, "n_loci exceeds the 32-bit locus index"); return nullptr; }
  kgx_gt8* h = new (std::nothrow) kgx_gt8();
  if (!h) { fail(KGX_ENOMEM, "host allocation failed"); return nullptr; }
  h->n_genomes = n_genomes;
  h->n_loci = n_loci;
  h->pitch = (n_genomes + 127) / 128 * 128;
  const uint64_t bytes = h->pitch * n_loci;
  if (bytes) {
    if (hipMalloc(&h->d_gt, bytes) != hipSuccess) {
      (void)hipGetLastError();
      fail(KGX_ENOMEM, "hipMalloc of %llu bytes for the %llu x %llu genotype matrix failed", (unsigned long long)bytes,
           (unsigned long long)n_loci, (unsigned long long)n_genomes);
      delete h;
      return nullptr;
    }
    if (hipMemsetAsync(h->d_gt, 0, bytes, g_state.stream) != hipSuccess || hipStreamSynchronize(g_state.stream) != hipSuccess) {
      fail(KGX_EHIP, "memset of the genotype matrix failed");
      (void)hipFree(h->d_gt);
      delete h;
      return nullptr;
    }
  }
  return h;
}

void kgx_gt8_destroy(kgx_gt8* h) {
  if (!h) return;
  if (h->d_gt) (void)hipFree(h->d_gt);
  delete h;
}

uint64_t kgx_gt8_genomes(const kgx_gt8* h) { return h ? h->n_genomes : 0; }
uint64_t kgx_gt8_loci(const kgx_gt8* h) { return h ? h->n_loci : 0; }
uint64_t kgx_gt8_sweep_bytes(uint64_t n_genomes, uint64_t n_selected, uint32_t amax) {
  return n_genomes * n_selected + 8ull * amax * n_selected + 80ull * n_genomes;
}

int kgx_gt8_load(kgx_gt8* h, const uint8_t* src, uint64_t g0, uint64_t g1) {
  if (int rc = require_device()) return rc;
  if (!h || !src) return fail(KGX_EINVAL, "null handle or source");
  if (g0 > g1 || g1 > h->n_genomes) return fail(KGX_EINVAL, "genome range out of bounds");
  if (g0 == g1 || h->n_loci == 0) return KGX_OK;
  h->wide_nibbles = 0;              // the bytes change: look again (kgx_inbreed)
  const uint64_t L = h->n_loci;
  uint64_t slab = (1ull << 30) / L;
  if (slab < 1) slab = 1;
  const uint64_t max_rows = (g1 - g0) < slab ? (g1 - g0) : slab;
  uint8_t* d_stage = nullptr;
  KGX_HIP_MEM(hipMalloc(&d_stage, max_rows * L));
  int rc = KGX_OK;
  for (uint64_t g = g0; g < g1 && rc == KGX_OK; g += slab) {
    const uint64_t n = (g1 - g) < slab ? (g1 - g) : slab;
    if (hipMemcpyAsync(d_stage, src + (g - g0) * L, n * L, hipMemcpyHostToDevice, g_state.stream) != hipSuccess) {
      rc = fail(KGX_EHIP, "H2D copy of genotype bytes failed");
      break;
    }
    hipLaunchKernelGGL(k_gt8_transpose, dim3(stream_grid(n * L, kBlock)), dim3(kBlock), 0, g_state.stream, d_stage, n, L, g,
                       h->d_gt, h->pitch);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_state.stream) != hipSuccess)
      rc = fail(KGX_EHIP, "genotype transpose kernel failed");
  }
  (void)hipFree(d_stage);
  return rc;
}

int kgx_gt8_load_rows(kgx_gt8* h, const uint8_t* src, uint64_t src_pitch, uint64_t l0, uint64_t l1) {
  if (int rc = require_device()) return rc;
  if (!h || !src) return fail(KGX_EINVAL, "null handle or source");
  if (l0 > l1 || l1 > h->n_loci || src_pitch < h->n_genomes) return fail(KGX_EINVAL, "bad locus range or pitch");
  if (l0 == l1) return KGX_OK;
  h->wide_nibbles = 0;
  KGX_HIP(hipMemcpy2DAsync(h->d_gt + l0 * h->pitch, h->pitch, src, src_pitch, h->n_genomes, l1 - l0, hipMemcpyHostToDevice, g_state.stream));
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  return KGX_OK;
}

int kgx_gt8_read_rows(const kgx_gt8* h, uint8_t* dst, uint64_t dst_pitch, uint64_t l0, uint64_t l1) {
  if (int rc = require_device()) return rc;
  if (!h || !dst) return fail(KGX_EINVAL, "null handle or destination");
  if (l0 > l1 || l1 > h->n_loci || dst_pitch < h->n_genomes) return fail(KGX_EINVAL, "bad locus range or pitch");
  if (l0 == l1) return KGX_OK;
  KGX_HIP(hipMemcpy2DAsync(dst, dst_pitch, h->d_gt + l0 * h->pitch, h->pitch, h->n_genomes, l1 - l0, hipMemcpyDeviceToHost, g_state.stream));
  KGX_HIP(hipStreamSynchronize(g_state.stream));
  return KGX_OK;
}

int kgx_locus_class_frequencies(const double* minor_af, uint64_t n_loci, uint32_t amax, double inbreeding, double* out, uint8_t* valid) {
  if (int rc = require_device()) return rc;
  if (!minor_af || !out || amax == 0 || amax > 14) return fail(KGX_EINVAL, "bad arguments (amax must be 1..14)");
  if (n_loci == 0) return KGX_OK;
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
    if (hipMemcpyAsync(d_in, minor_af, n_loci * amax * sizeof(double), hipMemcpyHostToDevice, g_state.stream) != hipSuccess) rc = fail(KGX_EHIP, "H2D failed");
    if (rc == KGX_OK) {
      hipLaunchKernelGGL((k_locus_tables<false>), dim3(stream_grid(n_loci, kBlock)), dim3(kBlock), 0, g_state.stream, d_in, n_loci, amax, inbreeding, d_table, d_valid);
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(table.data(), d_table, table.size() * sizeof(double), hipMemcpyDeviceToHost, g_state.stream) != hipSuccess ||
          hipMemcpyAsync(v.data(), d_valid, n_loci, hipMemcpyDeviceToHost, g_state.stream) != hipSuccess ||
          hipStreamSynchronize(g_state.stream) != hipSuccess)
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
}

namespace {
struct ScratchPlan {
  size_t total = 0;
  size_t add(size_t bytes) {
    const size_t at = total;
    total += (bytes + 255u) & ~static_cast<size_t>(255u);
    return at;
  }
};
int scratch_reserve(size_t bytes, char** out) {
  if (bytes > g_state.scratch_bytes) {
    if (g_state.scratch) (void)hipFree(g_state.scratch);
    g_state.scratch = nullptr;
    g_state.scratch_bytes = 0;
    const size_t want = bytes + bytes / 8;            // headroom: windows of a contig differ a little in locus count
    if (hipMalloc(&g_state.scratch, want) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc(&g_state.scratch, bytes) != hipSuccess) {
        (void)hipGetLastError();
        g_state.scratch = nullptr;
        return fail(KGX_ENOMEM, "hipMalloc of %llu scratch bytes failed", static_cast<unsigned long long>(bytes));
      }
      g_state.scratch_bytes = bytes;
    } else {
      g_state.scratch_bytes = want;
    }
  }
  *out = g_state.scratch;
  return KGX_OK;
}
}  // namespace

int kgx_inbreed(kgx_gt8* h, uint64_t g0, uint64_t g1, const uint32_t* locus_index, uint64_t n_sel, const double* minor_af,
                uint32_t amax, int phased, int algorithm, kgx_locus_results* out) {
  if (int rc = require_device()) return rc;
  if (!h || !out || (n_sel && !minor_af)) return fail(KGX_EINVAL, "null argument");
  if (g0 > g1 || g1 > h->n_genomes || (g0 & 3u)) return fail(KGX_EINVAL, "genome range must lie in the matrix and start on a multiple of 4");
  if (amax == 0 || amax > 14) return fail(KGX_EINVAL, "amax %u outside [1,14] (4-bit allele indices)", amax);
  if (algorithm < 0 || algorithm > 3) return fail(KGX_EINVAL, "unknown algorithm %d", algorithm);
  if (!locus_index && n_sel > h->n_loci) return fail(KGX_EINVAL, "n_sel exceeds the locus count");
  if (locus_index)
    for (uint64_t i = 0; i < n_sel; ++i)
      if (locus_index[i] >= h->n_loci) return fail(KGX_EINVAL, "locus_index[%llu] out of range", (unsigned long long)i);
  const uint64_t n = g1 - g0;
  if (n == 0) return KGX_OK;
  static_assert(sizeof(kgx_locus_results) == sizeof(LocusResultsDev), "LocusResults layout");

  const uint32_t stride = sweep_stride(amax);
  // Frequency pass flavour: 16 genomes per lane (SWAR, 16-byte loads) when the group starts on a 16-genome boundary and
  // the estimator needs no Ritland terms; otherwise 4 genomes per lane.
  // RitlandLocus with allele indices that fit the tables: the plain frequency sweep, then one table pass for its terms.
  const bool ritland_lut = algorithm == KGX_ALGO_RITLAND_LOCUS && !env_int("KGX_K5_GENERIC", 0) && !env_int("KGX_K5_NO_EVAL_LUT", 0) &&
                           amax <= 4;
  const bool swar16 = !env_int("KGX_K5_GENERIC", 0) && !env_int("KGX_K5_NO_SWAR16", 0) && amax <= 4 && (g0 & 15u) == 0 &&
                      (algorithm != KGX_ALGO_RITLAND_LOCUS || ritland_lut);
  // The evaluation passes of HallME / Loglikelihood go through the per-batch LDS tables when the allele indices fit them.
  const bool eval_lut = !env_int("KGX_K5_GENERIC", 0) && !env_int("KGX_K5_NO_EVAL_LUT", 0) && amax <= 7;
  int eval_gpl = env_int("KGX_K5_EVAL_GPL", 8);            // genomes per lane: the widest load the group's alignment allows
  if (eval_gpl != 4 && eval_gpl != 8) eval_gpl = 8;
  while (eval_gpl > 4 && (g0 % static_cast<uint64_t>(eval_gpl)) != 0) eval_gpl /= 2;
  const uint32_t gx = static_cast<uint32_t>(((n + 3) / 4 + kBlock - 1) / kBlock);
  const uint32_t gx16 = static_cast<uint32_t>(((n + 15) / 16 + kBlock - 1) / kBlock);
  uint64_t n_seg = (static_cast<uint64_t>(g_state.compute_units) * env_int("KGX_K5_BLOCKS_PER_CU", 8) + (swar16 ? gx16 : gx) - 1) / (swar16 ? gx16 : gx);
  if (n_seg > (n_sel + 63) / 64) n_seg = (n_sel + 63) / 64;
  if (n_seg < (n_sel + 65534) / 65535) n_seg = (n_sel + 65534) / 65535;   // 16-bit class counters per segment
  if (n_seg < 1) n_seg = 1;
  if (n_seg > 65535) n_seg = 65535;
  uint64_t per_seg = n_sel ? (n_sel + n_seg - 1) / n_seg : 8;
  per_seg = (per_seg + 7) / 8 * 8;                                          // whole 8-locus batches per segment
  n_seg = n_sel ? (n_sel + per_seg - 1) / per_seg : 1;

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
  const size_t o_part = plan.add(n_seg * n * kParts0 * sizeof(double)), o_segdef = plan.add(n_seg * kSegDefaults * sizeof(double));
  const size_t o_sums = plan.add(n * kParts0 * sizeof(double)), o_counts = plan.add(n * 6 * sizeof(unsigned long long));
  const size_t o_f = plan.add(n * sizeof(double)), o_eval = plan.add(n * sizeof(double)), o_out = plan.add(n * sizeof(LocusResultsDev));
  const size_t o_index = plan.add((n_sel + 8) * sizeof(uint32_t)), o_golden = plan.add(n * sizeof(GoldenState));
  const size_t o_brent = plan.add(n * sizeof(BrentState)), o_running = plan.add(sizeof(unsigned int));
  char* arena = nullptr;
  if (int arc = scratch_reserve(plan.total, &arena)) return arc;
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
  try_hip(hipMemsetAsync(d_meta, 0, (n_tab + 8) * sizeof(uint32_t), g_state.stream), KGX_EHIP, "memset(meta)");
  if (locus_index && n_sel) {
    d_index = reinterpret_cast<uint32_t*>(arena + o_index);
    try_hip(hipMemsetAsync(d_index, 0, (n_sel + 8) * sizeof(uint32_t), g_state.stream), KGX_EHIP, "memset(index)");
  }
  hipStream_t st = g_state.stream;
  if (rc == KGX_OK && n_sel) {
    // hipMemcpyDefault: the caller's tables may live on the host or already on this device (kgx.h)
    try_hip(hipMemcpyAsync(d_af, minor_af, n_sel * amax * sizeof(double), hipMemcpyDefault, st), KGX_EHIP, "copy(af)");
    if (d_index) try_hip(hipMemcpyAsync(d_index, locus_index, n_sel * sizeof(uint32_t), hipMemcpyDefault, st), KGX_EHIP, "copy(index)");
  }
  try_hip(hipMemsetAsync(d_counts, 0, n * 6 * sizeof(unsigned long long), st), KGX_EHIP, "memset(counts)");
  try_hip(hipMemsetAsync(d_part, 0, n_seg * n * kParts0 * sizeof(double), st), KGX_EHIP, "memset(partials)");
  try_hip(hipMemsetAsync(d_f, 0, n * sizeof(double), st), KGX_EHIP, "memset(f)");

  const dim3 grid(gx, static_cast<uint32_t>(n_seg));
  const uint32_t* gt32 = reinterpret_cast<const uint32_t*>(h->d_gt);
  const uint64_t dwords_per_row = h->pitch / 4;
  // The SWAR sweeps guard against allele indexes 8..14 (past their 8-entry tables) only if the matrix holds any: looked
  // up once per content of the matrix, by one pass over its bytes (KGX_K5_ALWAYS_GUARD=1 skips the look and guards).
  bool guard = true;
  if (rc == KGX_OK && n_sel && (swar16 || amax <= 4) && env_int("KGX_K5_ALWAYS_GUARD", 0) == 0) {
    if (h->wide_nibbles == 0) {
      unsigned int found = 0;
      try_hip(hipMemsetAsync(d_running, 0, sizeof(unsigned int), st), KGX_EHIP, "memset(scan flag)");
      const uint64_t n_chunks = h->n_loci * (h->pitch / 16);
      if (rc == KGX_OK) {
        hipLaunchKernelGGL(k_scan_wide_nibbles, dim3(stream_grid(n_chunks, kBlock)), dim3(kBlock), 0, st, reinterpret_cast<const kgx_v4u*>(h->d_gt),
                           n_chunks, d_running);
        try_hip(hipGetLastError(), KGX_EHIP, "k_scan_wide_nibbles launch");
        try_hip(hipMemcpyAsync(&found, d_running, sizeof(found), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(scan flag)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      }
      if (rc == KGX_OK) h->wide_nibbles = found ? 2 : 1;
    }
    guard = h->wide_nibbles != 1;
  }
  auto sweep = [&](int mode) {
    if (n_sel == 0) return;
    if (mode == 0) {
      if (swar16) {
        hipLaunchKernelGGL(k_locus_bits, dim3(stream_grid(n_sel, kBlock)), dim3(kBlock), 0, st, d_table, d_valid, n_sel, amax, d_meta);
        hipLaunchKernelGGL(k_segment_defaults, dim3(static_cast<uint32_t>(n_seg)), dim3(kWave), 0, st, d_table, d_valid, n_sel, per_seg, amax, d_segdef);
        hipLaunchKernelGGL(k_fill_defaults, dim3(stream_grid(n_seg * n, kBlock)), dim3(kBlock), 0, st, d_segdef, n_seg, n, d_part);
        const dim3 grid16(gx16, static_cast<uint32_t>(n_seg));
        const kgx_v4u* gt128 = reinterpret_cast<const kgx_v4u*>(h->d_gt);
#define KGX_SWAR16(INDEXED, GUARD)                                                                                                  \
  hipLaunchKernelGGL((k_inbreed_sweep_swar16<INDEXED, GUARD>), grid16, dim3(kBlock), 0, st, gt128, h->pitch / 16, g0, n, d_index, n_sel, \
                     per_seg, d_table, d_meta, amax, phased, d_segdef, d_counts, d_part)
        if (d_index) { if (guard) KGX_SWAR16(true, true); else KGX_SWAR16(true, false); }
        else { if (guard) KGX_SWAR16(false, true); else KGX_SWAR16(false, false); }
#undef KGX_SWAR16
      } else if (env_int("KGX_K5_GENERIC", 0) || amax > 4 || (algorithm == KGX_ALGO_RITLAND_LOCUS && !ritland_lut)) {
        hipLaunchKernelGGL((k_inbreed_sweep<0>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel, per_seg, d_table,
                           d_valid, amax, phased, d_f, d_counts, d_part);
      } else {
        hipLaunchKernelGGL(k_locus_bits, dim3(stream_grid(n_sel, kBlock)), dim3(kBlock), 0, st, d_table, d_valid, n_sel, amax, d_meta);
        hipLaunchKernelGGL(k_segment_defaults, dim3(static_cast<uint32_t>(n_seg)), dim3(kWave), 0, st, d_table, d_valid, n_sel, per_seg, amax, d_segdef);
#define KGX_SWAR(INDEXED, GUARD)                                                                                                   \
  hipLaunchKernelGGL((k_inbreed_sweep_swar<INDEXED, GUARD>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel,    \
                     per_seg, d_table, d_meta, amax, phased, d_segdef, d_counts, d_part)
        if (d_index) { if (guard) KGX_SWAR(true, true); else KGX_SWAR(true, false); }
        else { if (guard) KGX_SWAR(false, true); else KGX_SWAR(false, false); }
#undef KGX_SWAR
      }
    } else if (eval_lut || mode == 3) {
      const dim3 grid_eval(static_cast<uint32_t>(((n + eval_gpl - 1) / eval_gpl + kBlock - 1) / kBlock), static_cast<uint32_t>(n_seg));
#define KGX_EVAL(M, W, B)                                                                                                         \
  hipLaunchKernelGGL((k_inbreed_eval_lut<M, W, B>), grid_eval, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel,  \
                     per_seg, d_table, d_valid, amax, phased, d_f, d_part, d_counts)
#define KGX_EVAL_BITS(M, W)                                                                \
  do {                                                                                     \
    if (amax <= 1) KGX_EVAL(M, W, 1); else if (amax <= 3) KGX_EVAL(M, W, 2); else KGX_EVAL(M, W, 3); \
  } while (0)
      if (mode == 1) {
        if (eval_gpl == 8) KGX_EVAL_BITS(1, 8); else KGX_EVAL_BITS(1, 4);
      } else if (mode == 2) {
        if (eval_gpl == 8) KGX_EVAL_BITS(2, 8); else KGX_EVAL_BITS(2, 4);
      } else {
        if (eval_gpl == 8) KGX_EVAL_BITS(3, 8); else KGX_EVAL_BITS(3, 4);
      }
#undef KGX_EVAL_BITS
#undef KGX_EVAL
    } else if (mode == 1)
      hipLaunchKernelGGL((k_inbreed_sweep<1>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel, per_seg, d_table,
                         d_valid, amax, phased, d_f, d_counts, d_part);
    else
      hipLaunchKernelGGL((k_inbreed_sweep<2>), grid, dim3(kBlock), 0, st, gt32, dwords_per_row, g0, n, d_index, n_sel, per_seg, d_table,
                         d_valid, amax, phased, d_f, d_counts, d_part);
  };
  const uint32_t lin_grid = stream_grid(n, kBlock);
  if (rc == KGX_OK) {
    if (n_sel) hipLaunchKernelGGL((k_locus_tables<true>), dim3(stream_grid(n_sel, kBlock)), dim3(kBlock), 0, st, d_af, n_sel, amax, 0.0, d_table, d_valid);
    if (!g_state.sweep_begin) {
      try_hip(hipEventCreate(&g_state.sweep_begin), KGX_EHIP, "hipEventCreate");
      try_hip(hipEventCreate(&g_state.sweep_end), KGX_EHIP, "hipEventCreate");
    }
    if (rc == KGX_OK) try_hip(hipEventRecord(g_state.sweep_begin, st), KGX_EHIP, "hipEventRecord");
    sweep(0);
    if (ritland_lut) sweep(3);
    if (rc == KGX_OK) try_hip(hipEventRecord(g_state.sweep_end, st), KGX_EHIP, "hipEventRecord");
    hipLaunchKernelGGL(k_reduce_parts, dim3(stream_grid(n * kParts0, kBlock)), dim3(kBlock), 0, st, d_part, n_seg, n * kParts0, d_sums);
    // Window-sized calls: the whole iteration in one launch, a wave per genome (k_inbreed_iterate_wave).
    const bool wave_path = (algorithm == 2 || algorithm == 3) && n_sel > 0 && n_sel <= 64ull * kWaveCells && !env_int("KGX_K7_NO_WAVE", 0);
    if (wave_path) {
      const uint32_t wave_grid = static_cast<uint32_t>((n * kWave + kBlock - 1) / kBlock);
      try_hip(hipMemsetAsync(d_running, 0, sizeof(unsigned int), st), KGX_EHIP, "memset(evaluations)");
      if (algorithm == 2)
        hipLaunchKernelGGL((k_inbreed_iterate_wave<1>), dim3(wave_grid), dim3(kBlock), 0, st, h->d_gt, h->pitch, g0, n, d_index, n_sel, d_table, d_valid,
                           amax, phased, d_counts, d_sums, d_f, d_running);
      else
        hipLaunchKernelGGL((k_inbreed_iterate_wave<2>), dim3(wave_grid), dim3(kBlock), 0, st, h->d_gt, h->pitch, g0, n, d_index, n_sel, d_table, d_valid,
                           amax, phased, d_counts, env_int("KGX_K7_ESTIMATE_START", 0) ? d_sums : nullptr, d_f, d_running);
      if (algorithm == 3) {
        unsigned int evaluations = 0;
        try_hip(hipMemcpyAsync(&evaluations, d_running, sizeof(unsigned int), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(evaluations)");
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
        g_state.last_evaluations = static_cast<int>(evaluations);
      }
    } else if (algorithm == 2) {
      // processHallME (_calc.cpp:225-307).  The reference restarts from U(0,0.5] and, through RetryCalcResult's
      // self-comparison, always stops after 5 restarts of exactly 50 expectation steps, keeping the last; the
      // fixed start 0.25 (the mean of its start distribution) replaces the random draw.
      std::vector<double> f0(n, 0.25);
      // locus slots every lane of k_inbreed_eval_lut walks: whole batches of 8 in every segment
      const unsigned long long walked = eval_lut && n_sel ? (n_seg - 1) * per_seg + (n_sel - (n_seg - 1) * per_seg + 7) / 8 * 8 : 0ull;
      try_hip(hipMemcpyAsync(d_f, f0.data(), n * sizeof(double), hipMemcpyHostToDevice, st), KGX_EHIP, "H2D(f0)");
      try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      for (int it = 0; it < 50 && rc == KGX_OK; ++it) {
        sweep(1);
        hipLaunchKernelGGL(k_reduce_parts, dim3(lin_grid), dim3(kBlock), 0, st, d_part, n_seg, n, d_eval);
        hipLaunchKernelGGL(k_hall_update, dim3(lin_grid), dim3(kBlock), 0, st, d_eval, d_counts, n, walked, d_f);
      }
    } else if (algorithm == 3) {
      // processLogLikelihood (_calc.cpp:153-216): maximise over [-1,1].  The objective is a sum of logs of
      // clamped linear functions of F; Brent's method on that same clamped objective replaces nlopt's Nelder-Mead
      // (un-vendored, unpinned), to within 5e-7 in F where the reference stops at an absolute change of 1e-6.  KGX_K7_GOLDEN=1 runs the
      // plain golden-section search instead (38 evaluations, bracket 6e-8).
      if (!env_int("KGX_K7_GOLDEN", 0)) {
        // Start: [-1, 1] from its golden point.  KGX_K7_ESTIMATE_START=1 starts in a window around the Simple estimate
        // instead (brent_start): 11 instead of 15 evaluations on a population with F in [0, 0.1], but where the clamped
        // objective has several local maxima (F < 0) it may settle on another one than a search from the middle does --
        // the reference itself lands on one or another from its random starts -- so it is not the default.
        hipLaunchKernelGGL(k_brent_init, dim3(lin_grid), dim3(kBlock), 0, st, d_counts, d_sums, n, env_int("KGX_K7_ESTIMATE_START", 0), d_brent, d_f);
        constexpr int kMaxEvaluations = 60;        // golden section alone would need 38; Brent's safeguard keeps that bound
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
        int act_gpl = eval_gpl;
        std::vector<uint32_t> global_of(n);
        for (uint64_t g = 0; g < n; ++g) global_of[g] = static_cast<uint32_t>(g);
        std::vector<BrentState> host_states;
        auto evaluate = [&]() {
          if (act_gt == gt32) { sweep(2); return; }
          const dim3 grid_eval(static_cast<uint32_t>(((n_act + act_gpl - 1) / act_gpl + kBlock - 1) / kBlock), static_cast<uint32_t>(n_seg));
#define KGX_EVAL2(W, B)                                                                                                              \
  hipLaunchKernelGGL((k_inbreed_eval_lut<2, W, B>), grid_eval, dim3(kBlock), 0, st, act_gt, act_dwords_per_row, act_g0, n_act, act_index, \
                     n_sel, per_seg, d_table, d_valid, amax, phased, act_f, d_part, d_counts)
          if (amax <= 1) KGX_EVAL2(8, 1); else if (amax <= 3) KGX_EVAL2(8, 2); else KGX_EVAL2(8, 3);
#undef KGX_EVAL2
        };
        const bool may_compact = eval_lut && !env_int("KGX_K7_NO_COMPACT", 0);
        bool may_compact_now = may_compact;
        int compaction_level = 0;
        for (int it = 0; it < kMaxEvaluations && rc == KGX_OK; ++it) {
          evaluate();
          const uint32_t act_grid = stream_grid(n_act, kBlock);
          hipLaunchKernelGGL(k_reduce_parts, dim3(act_grid), dim3(kBlock), 0, st, d_part, n_seg, n_act, d_eval);
          try_hip(hipMemsetAsync(d_running, 0, sizeof(unsigned int), st), KGX_EHIP, "memset(running)");
          hipLaunchKernelGGL(k_brent_step, dim3(act_grid), dim3(kBlock), 0, st, act_brent, d_eval, n_act, it == 0 ? 0 : 1, act_f, d_running,
                             act_global, d_f);
          unsigned int running = 0;
          try_hip(hipMemcpyAsync(&running, d_running, sizeof(unsigned int), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(running)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
          g_state.last_evaluations = it + 1;
          if (env_int("KGX_K7_TRACE", 0)) std::fprintf(stderr, "kgx: Loglikelihood evaluation %d: %u of %llu genomes still searching\n", it + 1, running, (unsigned long long)n_act);
          if (running == 0) break;
          const uint64_t new_pitch = (static_cast<uint64_t>(running) + 127) / 128 * 128;
          // worth it from ~64 M cells left (a gather costs about one pass); the two knobs are for the tests
          const uint64_t min_genomes = static_cast<uint64_t>(env_int("KGX_K7_COMPACT_MIN_GENOMES", 2048));
          const uint64_t min_cells = static_cast<uint64_t>(env_int("KGX_K7_COMPACT_MIN_CELLS", 1 << 26));
          if (!may_compact_now || static_cast<uint64_t>(running) * 2 > n_act || n_act < min_genomes || n_sel * static_cast<uint64_t>(running) < min_cells) continue;
          host_states.resize(n_act);
          try_hip(hipMemcpyAsync(host_states.data(), act_brent, n_act * sizeof(BrentState), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(states)");
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
          if (rc != KGX_OK) break;
          std::vector<uint32_t> columns, new_global;
          for (uint64_t g = 0; g < n_act; ++g)
            if (!host_states[g].done) { columns.push_back(static_cast<uint32_t>(g)); new_global.push_back(global_of[g]); }
          const uint64_t n_new = columns.size();
          // Level k lives in the library's ping-pong buffer k & 1 (what that buffer held, level k - 2, is no longer read);
          // the buffers stay allocated between calls like the scratch arena (kgx_release_scratch frees them).
          ScratchPlan level;
          const size_t o_gt = level.add(n_sel * new_pitch), o_st = level.add(n_new * sizeof(BrentState)), o_nf = level.add(n_new * sizeof(double));
          const size_t o_col = level.add(n_new * sizeof(uint32_t)), o_glob = level.add(n_new * sizeof(uint32_t));
          const int slot = compaction_level & 1;
          if (g_state.compact_bytes[slot] < level.total) {
            if (g_state.compact[slot]) (void)hipFree(g_state.compact[slot]);
            g_state.compact[slot] = nullptr;
            g_state.compact_bytes[slot] = 0;
            size_t free_bytes = 0, total_bytes = 0;
            if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess || free_bytes < level.total + (4ull << 30) ||
                hipMalloc(&g_state.compact[slot], level.total) != hipSuccess) {
              (void)hipGetLastError();
              g_state.compact[slot] = nullptr;
              may_compact_now = false;                                     // no room: carry on as is
              continue;
            }
            g_state.compact_bytes[slot] = level.total;
          }
          ++compaction_level;
          char* base = g_state.compact[slot];
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
          hipLaunchKernelGGL(k_gather_states, dim3(stream_grid(n_new, kBlock)), dim3(kBlock), 0, st, act_brent, act_f, new_columns, n_new, new_brent, new_f);
          try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");           // columns / new_global leave scope
          act_gt = reinterpret_cast<const uint32_t*>(new_gt);
          act_dwords_per_row = new_pitch / 4;
          act_g0 = 0;
          act_index = nullptr;
          act_brent = new_brent;
          act_f = new_f;
          act_global = d_new_global;
          act_gpl = 8;
          n_act = n_new;
          global_of.swap(new_global);
        }
        hipLaunchKernelGGL(k_brent_step, dim3(stream_grid(n_act, kBlock)), dim3(kBlock), 0, st, act_brent, d_eval, n_act, 2, act_f, d_running, act_global, d_f);
        try_hip(hipStreamSynchronize(st), KGX_EHIP, "sync");
      } else {
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
        hipLaunchKernelGGL(k_reduce_parts, dim3(lin_grid), dim3(kBlock), 0, st, d_part, n_seg, n, d_eval);
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
      g_state.last_evaluations = kGoldenSteps;
      }
    }
    hipLaunchKernelGGL(k_finish_inbreed, dim3(lin_grid), dim3(kBlock), 0, st, d_counts, d_sums, n, algorithm, d_f, d_out);
    try_hip(hipGetLastError(), KGX_EHIP, "kernel launch");
    try_hip(hipMemcpyAsync(out, d_out, n * sizeof(LocusResultsDev), hipMemcpyDeviceToHost, st), KGX_EHIP, "D2H(out)");
    try_hip(hipStreamSynchronize(st), KGX_EHIP, "stream synchronize");
    if (rc == KGX_OK) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, g_state.sweep_begin, g_state.sweep_end) == hipSuccess) g_state.last_sweep_ms = ms;
    }
  }
  return rc;
}

int kgx_release_scratch(void) {
  if (g_state.scratch) {
    if (g_state.stream) (void)hipStreamSynchronize(g_state.stream);
    (void)hipFree(g_state.scratch);
  }
  g_state.scratch = nullptr;
  g_state.scratch_bytes = 0;
  for (int k = 0; k < 2; ++k) {
    if (g_state.compact[k]) (void)hipFree(g_state.compact[k]);
    g_state.compact[k] = nullptr;
    g_state.compact_bytes[k] = 0;
  }
  return KGX_OK;
}

double kgx_inbreed_last_sweep_ms(void) { return g_state.last_sweep_ms; }
int kgx_inbreed_last_evaluations(void) { return g_state.last_evaluations; }

int kgx_gt8_synth_multiallelic(kgx_gt8* h, uint64_t seed, uint64_t genome_base, uint64_t locus_base, double* af_table) {
  if (int rc = require_device()) return rc;
  if (!h) return fail(KGX_EINVAL, "null handle");
  if (h->n_loci == 0) return KGX_OK;
  h->wide_nibbles = 0;
  double* d_table = nullptr;
  if (af_table) KGX_HIP_MEM(hipMalloc(&d_table, h->n_loci * KGX_SYNTH_MAX_ALTS * sizeof(double)));
  const uint64_t work = h->n_loci * ((h->n_genomes + 3) / 4);
  hipLaunchKernelGGL(k_synth_gt8, dim3(stream_grid(work, kBlock)), dim3(kBlock), 0, g_state.stream,
                     reinterpret_cast<uint32_t*>(h->d_gt), h->pitch / 4, h->n_loci, h->n_genomes, seed, genome_base, locus_base, d_table);
  int rc = KGX_OK;
  if (hipGetLastError() != hipSuccess) rc = fail(KGX_EHIP, "synthetic genotype kernel launch failed");
  if (rc == KGX_OK && af_table &&
      hipMemcpyAsync(af_table, d_table, h->n_loci * KGX_SYNTH_MAX_ALTS * sizeof(double), hipMemcpyDeviceToHost, g_state.stream) != hipSuccess)
    rc = fail(KGX_EHIP, "D2H of the allele-frequency table failed");
  if (rc == KGX_OK && hipStreamSynchronize(g_state.stream) != hipSuccess) rc = fail(KGX_EHIP, "synthetic genotype kernel failed");
  if (d_table) (void)hipFree(d_table);
  return rc;
}

int kgx_synth_multiallelic_host(uint64_t seed, uint64_t genome_base, uint64_t n_genomes, uint64_t l0, uint64_t l1, uint8_t* gt8,
                                uint64_t pitch, double* af_table, uint8_t* alleles) {
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
}

int kgx_synth_locus_host(uint64_t seed, uint64_t l, int* n_alt, float af[3], int is_indel[3]) {
  if (!n_alt || !af || !is_indel) return fail(KGX_EINVAL, "null argument");
  const kgx_synth_locus loc = kgx_synth_make_locus(seed, l);
  *n_alt = loc.n_alt;
  for (int a = 0; a < KGX_SYNTH_MAX_ALTS; ++a) { af[a] = loc.af[a]; is_indel[a] = loc.is_indel[a]; }
  return KGX_OK;
}

int kgx_gt8_synth_inbred(kgx_gt8* h, const double* minor_af, uint32_t amax, const double* inbreeding, uint64_t seed) {
  if (int rc = require_device()) return rc;
  if (!h || !minor_af || !inbreeding) return fail(KGX_EINVAL, "null argument");
  if (amax == 0 || amax > 14) return fail(KGX_EINVAL, "amax %u outside [1,14]", amax);
  if (h->n_loci == 0) return KGX_OK;
  h->wide_nibbles = 0;
  double *d_table = nullptr, *d_f = nullptr;
  KGX_HIP_MEM(hipMalloc(&d_table, h->n_loci * amax * sizeof(double)));
  if (hipMalloc(&d_f, h->n_genomes * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_table); return fail(KGX_ENOMEM, "hipMalloc failed"); }
  int rc = KGX_OK;
  if (hipMemcpyAsync(d_table, minor_af, h->n_loci * amax * sizeof(double), hipMemcpyHostToDevice, g_state.stream) != hipSuccess ||
      hipMemcpyAsync(d_f, inbreeding, h->n_genomes * sizeof(double), hipMemcpyHostToDevice, g_state.stream) != hipSuccess)
    rc = fail(KGX_EHIP, "H2D of the allele-frequency table failed");
  if (rc == KGX_OK) {
    hipLaunchKernelGGL(k_synth_inbred, dim3(stream_grid(h->n_loci * h->n_genomes, kBlock)), dim3(kBlock), 0, g_state.stream, h->d_gt,
                       h->pitch, h->n_loci, h->n_genomes, d_table, amax, d_f, seed);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_state.stream) != hipSuccess) rc = fail(KGX_EHIP, "synthetic inbred genome kernel failed");
  }
  (void)hipFree(d_table);
  (void)hipFree(d_f);
  return rc;
}

}  // extern "C"
