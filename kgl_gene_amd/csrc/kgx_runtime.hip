// Device binding of the C ABI (include/kgx.h): kgx_init and what every other translation unit needs from it -- the
// per-slot streams and scratch, error text, and the one exchange step of the path (the sum of per-variant counts over
// the genome shards: a direct RCCL all-reduce over xGMI).  No CPU fallback exists: without a usable gfx950 device
// kgx_init fails and every compute entry point after it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>        // types and prototypes only: the library itself is loaded on demand (load_rccl)

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "kgx_internal.h"

namespace {

thread_local std::string g_error;

// RCCL is a 0.5 GB library that a one-device process never needs; a process that already holds one (torch) gets that
// same copy back from dlopen by its soname.
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return KGX_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return kgx::fail(KGX_EHIP, "cannot load RCCL (librccl.so.1): %s", dlerror());
  RcclApi api;
  api.handle = h;
  api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!api.CommInitAll || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd || !api.GetErrorString) {
    dlclose(h);
    return kgx::fail(KGX_EHIP, "librccl.so.1 lacks an expected nccl* symbol");
  }
  g_rccl = api;
  return KGX_OK;
}

// The binding new handles are created under.  Heap-held and never destroyed at exit: tearing HIP objects down from a
// static destructor races the runtime's own teardown.
std::shared_ptr<kgx::Runtime>& binding() {
  static auto* slot = new std::shared_ptr<kgx::Runtime>();
  return *slot;
}
std::mutex g_binding_mutex;

// counts[i] += other[i]: the "peer" exchange's add (two shards on one device, or RCCL switched off).
__global__ void __launch_bounds__(256)
k_add_u32(uint32_t* __restrict__ counts, const uint32_t* __restrict__ other, uint64_t n_words) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const uint64_t n4 = n_words / 4;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    v4u a = reinterpret_cast<v4u*>(counts)[i];
    const v4u b = reinterpret_cast<const v4u*>(other)[i];
    a += b;
    reinterpret_cast<v4u*>(counts)[i] = a;
  }
  for (uint64_t i = n4 * 4 + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_words; i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    counts[i] += other[i];
}

}  // namespace

namespace kgx {

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

const std::string& last_error() { return g_error; }
void set_last_error(const std::string& message) { g_error = message; }

// Tuning knobs (environment, read per launch; defaults are the shipped configuration).
// The KGX_* switches (tuning knobs, the tests' path selectors) are parsed ONCE -- at kgx_init and by kgx_reload_options -- into
// a snapshot every call reads (a window-sized kgx_inbreed call asked the environment ~40 questions).
using Options = std::unordered_map<std::string, std::string>;
std::shared_ptr<const Options> g_options;
std::mutex g_options_mutex;

std::shared_ptr<const Options> parse_options() {
  auto parsed = std::make_shared<Options>();
  for (char** e = environ; e && *e; ++e) {
    if (std::strncmp(*e, "KGX_", 4) != 0) continue;
    const char* eq = std::strchr(*e, '=');
    if (eq) (*parsed)[std::string(*e, static_cast<size_t>(eq - *e))] = std::string(eq + 1);
  }
  return parsed;
}

std::shared_ptr<const Options> options() {
  std::lock_guard<std::mutex> lock(g_options_mutex);
  if (!g_options) g_options = parse_options();
  return g_options;
}

void reload_options() {
  auto parsed = parse_options();
  std::lock_guard<std::mutex> lock(g_options_mutex);
  g_options = std::move(parsed);
}

int env_int(const char* name, int dflt) {
  const auto snapshot = options();
  const auto it = snapshot->find(name);
  return it != snapshot->end() && !it->second.empty() ? std::atoi(it->second.c_str()) : dflt;
}

std::string env_str(const char* name) {
  const auto snapshot = options();
  const auto it = snapshot->find(name);
  return it != snapshot->end() ? it->second : std::string();
}

uint32_t stream_grid(const Device& dev, uint64_t work_items, uint32_t items_per_block) {
  const uint64_t want = (work_items + items_per_block - 1) / items_per_block;
  const uint64_t cap = static_cast<uint64_t>(dev.compute_units) * 8u;
  const uint64_t g = want < cap ? want : cap;
  return static_cast<uint32_t>(g ? g : 1);
}

int scratch_reserve(Device& dev, size_t bytes, char** out) {
  if (bytes > dev.scratch_bytes) {
    if (dev.scratch) (void)hipFree(dev.scratch);
    dev.scratch = nullptr;
    dev.scratch_bytes = 0;
    const size_t want = bytes + bytes / 8;            // headroom: windows of a contig differ a little in locus count
    if (hipMalloc(&dev.scratch, want) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc(&dev.scratch, bytes) != hipSuccess) {
        (void)hipGetLastError();
        dev.scratch = nullptr;
        return fail(KGX_ENOMEM, "hipMalloc of %llu scratch bytes failed", static_cast<unsigned long long>(bytes));
      }
      dev.scratch_bytes = bytes;
    } else {
      dev.scratch_bytes = want;
    }
  }
  *out = dev.scratch;
  return KGX_OK;
}

int pinned_reserve(Device& dev, int which, size_t bytes, char** out) {
  if (bytes > dev.pinned_bytes[which]) {
    if (dev.pinned[which]) (void)hipHostFree(dev.pinned[which]);
    dev.pinned[which] = nullptr;
    dev.pinned_bytes[which] = 0;
    const size_t want = bytes + bytes / 4;
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(KGX_ENOMEM, "hipHostMalloc of %llu bytes failed", static_cast<unsigned long long>(want));
    }
    dev.pinned[which] = static_cast<char*>(p);
    dev.pinned_bytes[which] = want;
  }
  *out = dev.pinned[which];
  return KGX_OK;
}

int use_device(const Device& dev) {
  KGX_HIP(hipSetDevice(dev.id));
  return KGX_OK;
}

std::shared_ptr<Runtime> current_runtime() {
  std::lock_guard<std::mutex> lock(g_binding_mutex);
  return binding();
}

int require_bound() {
  if (!current_runtime()) return fail(KGX_ENODEVICE, "kgx_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)");
  return KGX_OK;
}

Device::~Device() {
  if (id < 0 || hipSetDevice(id) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  if (stream) (void)hipStreamSynchronize(stream);
  if (scratch) (void)hipFree(scratch);
  for (int k = 0; k < 2; ++k)
    if (compact[k]) (void)hipFree(compact[k]);
  if (words) (void)hipFree(words);
  if (bits) (void)hipFree(bits);
  for (int k = 0; k < 2; ++k)
    if (pinned[k]) (void)hipHostFree(pinned[k]);
  if (exchange_stage) (void)hipFree(exchange_stage);
  if (sweep_begin) (void)hipEventDestroy(sweep_begin);
  if (sweep_end) (void)hipEventDestroy(sweep_end);
  if (kernel_begin) (void)hipEventDestroy(kernel_begin);
  if (kernel_end) (void)hipEventDestroy(kernel_end);
  for (hipEvent_t e : {moments_begin, moments_end, search_begin, search_end})
    if (e) (void)hipEventDestroy(e);
  if (ready) (void)hipEventDestroy(ready);
  if (by_genome_begin) (void)hipEventDestroy(by_genome_begin);
  if (by_genome_end) (void)hipEventDestroy(by_genome_end);
  if (side_begin) (void)hipEventDestroy(side_begin);
  if (side_end) (void)hipEventDestroy(side_end);
  if (entries_end) (void)hipEventDestroy(entries_end);
  if (side_stream) (void)hipStreamDestroy(side_stream);
  if (stream) (void)hipStreamDestroy(stream);
}

Runtime::~Runtime() {
  if (g_rccl.handle)
    for (void* comm : comms)
      if (comm) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
}

int exchange_counts(Runtime& rt, const std::vector<void*>& d_counts, uint64_t n_words, const std::vector<hipStream_t>& streams) {
  const size_t n = rt.devs.size();
  if (rt.exchange == Exchange::None || n_words == 0) return KGX_OK;
  if (d_counts.size() != n || streams.size() != n) return fail(KGX_EINVAL, "exchange_counts: one buffer and one stream per slot");
  if (rt.exchange == Exchange::Rccl) {
    // The one exchange step of the path: ncclAllReduce(sum, uint32) of the [V][4] count vectors, one
    // rank per slot, issued from this thread as one group; every rank's call is queued on that slot's stream behind
    // the sweep that produced its counts.
    ncclResult_t rc = g_rccl.GroupStart();
    for (size_t s = 0; s < n && rc == ncclSuccess; ++s) {
      if (int e = use_device(*rt.devs[s])) { (void)g_rccl.GroupEnd(); return e; }
      rc = g_rccl.AllReduce(d_counts[s], d_counts[s], n_words, ncclUint32, ncclSum, static_cast<ncclComm_t>(rt.comms[s]), streams[s]);
    }
    const ncclResult_t rc_end = g_rccl.GroupEnd();
    if (rc == ncclSuccess) rc = rc_end;
    (void)use_device(*rt.devs[0]);
    if (rc != ncclSuccess) return fail(KGX_EHIP, "ncclAllReduce of the per-variant counts failed: %s", g_rccl.GetErrorString(rc));
    return KGX_OK;
  }
  // Exchange::Peer -- slot 0 gathers and adds, then every other slot copies the sums back.
  Device& root = *rt.devs[0];
  const size_t bytes = n_words * sizeof(uint32_t);
  if (int e = use_device(root)) return e;
  if (root.exchange_stage_bytes < bytes) {
    if (root.exchange_stage) (void)hipFree(root.exchange_stage);
    root.exchange_stage = nullptr;
    root.exchange_stage_bytes = 0;
    KGX_HIP_MEM(hipMalloc(&root.exchange_stage, bytes));
    root.exchange_stage_bytes = bytes;
  }
  for (size_t s = 1; s < n; ++s) {
    Device& dev = *rt.devs[s];
    if (int e = use_device(dev)) return e;
    KGX_HIP(hipEventRecord(dev.ready, streams[s]));
    if (int e = use_device(root)) return e;
    KGX_HIP(hipStreamWaitEvent(streams[0], dev.ready, 0));
    KGX_HIP(hipMemcpyAsync(root.exchange_stage, d_counts[s], bytes, hipMemcpyDefault, streams[0]));
    hipLaunchKernelGGL(k_add_u32, dim3(stream_grid(root, n_words / 4 + 1, 256)), dim3(256), 0, streams[0],
                       static_cast<uint32_t*>(d_counts[0]), static_cast<const uint32_t*>(root.exchange_stage), n_words);
    KGX_HIP(hipGetLastError());
  }
  KGX_HIP(hipEventRecord(root.ready, streams[0]));
  for (size_t s = 1; s < n; ++s) {
    Device& dev = *rt.devs[s];
    if (int e = use_device(dev)) return e;
    KGX_HIP(hipStreamWaitEvent(streams[s], root.ready, 0));
    KGX_HIP(hipMemcpyAsync(d_counts[s], d_counts[0], bytes, hipMemcpyDefault, streams[s]));
  }
  return use_device(root);
}

}  // namespace kgx

using namespace kgx;

extern "C" {

const char* kgx_version(void) { return "kgx 0.2.0 (gfx950)"; }

const char* kgx_last_error(void) { return g_error.c_str(); }

int kgx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int kgx_init(int device_count, const int* device_ids) {
  return guarded([&]() -> int {
    const int visible = kgx_device_count();
    if (visible <= 0) return fail(KGX_ENODEVICE, "no HIP device visible (there is no CPU fallback)");
    if (device_count < 0 || device_count > 64) return fail(KGX_EINVAL, "device_count %d outside [0,64]", device_count);
    std::vector<int> ids;
    if (device_count == 0) {
      for (int d = 0; d < visible; ++d) ids.push_back(d);
    } else {
      for (int i = 0; i < device_count; ++i) ids.push_back(device_ids ? device_ids[i] : i);
    }
    bool distinct = true;
    for (size_t i = 0; i < ids.size(); ++i) {
      if (ids[i] < 0 || ids[i] >= visible) return fail(KGX_EINVAL, "device %d out of range [0,%d)", ids[i], visible);
      for (size_t j = 0; j < i; ++j) distinct = distinct && ids[j] != ids[i];
    }
    reload_options();                                            // the KGX_* switches as they stand now (and at kgx_reload_options)
    const std::string forced_name = env_str("KGX_EXCHANGE");     // "peer" | "rccl": tests and bring-up; default by the binding
    const char* forced = forced_name.empty() ? nullptr : forced_name.c_str();
    Exchange exchange = ids.size() == 1 ? Exchange::None : (distinct ? Exchange::Rccl : Exchange::Peer);
    if (forced && std::strcmp(forced, "peer") == 0 && ids.size() > 1) exchange = Exchange::Peer;
    if (forced && std::strcmp(forced, "rccl") == 0) {
      if (!distinct) return fail(KGX_EINVAL, "KGX_EXCHANGE=rccl: RCCL cannot span a binding that lists a device twice");
      exchange = Exchange::Rccl;                                 // also with ONE slot: the all-reduce then runs over one rank
    }
    {
      std::lock_guard<std::mutex> lock(g_binding_mutex);
      const auto& now = binding();
      if (now && now->exchange == exchange && now->devs.size() == ids.size()) {
        bool same = true;
        for (size_t i = 0; i < ids.size(); ++i) same = same && now->devs[i]->id == ids[i];
        if (same) return hipSetDevice(ids[0]) == hipSuccess ? KGX_OK : fail(KGX_EHIP, "hipSetDevice(%d) failed", ids[0]);
      }
    }

    auto rt = std::make_shared<Runtime>();
    rt->exchange = exchange;
    for (size_t s = 0; s < ids.size(); ++s) {
      KGX_HIP(hipSetDevice(ids[s]));
      hipDeviceProp_t prop;
      KGX_HIP(hipGetDeviceProperties(&prop, ids[s]));
      if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KGX_ENODEVICE, "device %d is %s; this library is built for gfx950 only", ids[s], prop.gcnArchName);
      auto dev = std::make_unique<Device>();
      dev->slot = static_cast<int>(s);
      dev->id = ids[s];
      dev->compute_units = prop.multiProcessorCount;
      dev->hbm_bytes = prop.totalGlobalMem;
      std::snprintf(dev->name, sizeof(dev->name), "%s", prop.name);
      std::snprintf(dev->arch, sizeof(dev->arch), "%s", prop.gcnArchName);
      KGX_HIP(hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking));
      KGX_HIP(hipStreamCreateWithFlags(&dev->side_stream, hipStreamNonBlocking));
      KGX_HIP(hipEventCreateWithFlags(&dev->side_begin, hipEventDisableTiming));
      KGX_HIP(hipEventCreateWithFlags(&dev->side_end, hipEventDisableTiming));
      KGX_HIP(hipEventCreateWithFlags(&dev->entries_end, hipEventDisableTiming));
      KGX_HIP(hipEventCreate(&dev->sweep_begin));
      KGX_HIP(hipEventCreate(&dev->sweep_end));
      KGX_HIP(hipEventCreate(&dev->kernel_begin));
      KGX_HIP(hipEventCreate(&dev->kernel_end));
      KGX_HIP(hipEventCreate(&dev->moments_begin));
      KGX_HIP(hipEventCreate(&dev->moments_end));
      KGX_HIP(hipEventCreate(&dev->search_begin));
      KGX_HIP(hipEventCreate(&dev->search_end));
      KGX_HIP(hipEventCreateWithFlags(&dev->ready, hipEventDisableTiming));
      KGX_HIP(hipEventCreate(&dev->by_genome_begin));
      KGX_HIP(hipEventCreate(&dev->by_genome_end));
      rt->devs.push_back(std::move(dev));
    }
    if (ids.size() > 1 && distinct) {
      // direct loads / copies between the shards' devices (tables kept on one device, the peer exchange)
      for (size_t a = 0; a < ids.size(); ++a) {
        KGX_HIP(hipSetDevice(ids[a]));
        for (size_t b = 0; b < ids.size(); ++b) {
          if (a == b) continue;
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, ids[a], ids[b]) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(ids[b], 0);
            if (e != hipSuccess) (void)hipGetLastError();            // already enabled is fine
          } else {
            (void)hipGetLastError();
          }
        }
      }
    }
    if (exchange == Exchange::Rccl) {
      if (int rc = load_rccl()) return rc;
      std::vector<ncclComm_t> comms(ids.size(), nullptr);
      const ncclResult_t rc = g_rccl.CommInitAll(comms.data(), static_cast<int>(ids.size()), ids.data());
      if (rc != ncclSuccess) return fail(KGX_EHIP, "ncclCommInitAll over %zu devices failed: %s", ids.size(), g_rccl.GetErrorString(rc));
      for (ncclComm_t c : comms) rt->comms.push_back(c);
    }
    KGX_HIP(hipSetDevice(ids[0]));
    std::lock_guard<std::mutex> lock(g_binding_mutex);
    binding() = std::move(rt);
    return KGX_OK;
  });
}

int kgx_bound_devices(void) {
  const auto rt = current_runtime();
  return rt ? static_cast<int>(rt->devs.size()) : 0;
}

const char* kgx_exchange_kind(void) {
  const auto rt = current_runtime();
  if (!rt) return "none";
  return rt->exchange == Exchange::Rccl ? "rccl" : (rt->exchange == Exchange::Peer ? "peer" : "none");
}

int kgx_device_info(int slot, char* name, size_t name_len, char* arch, size_t arch_len, int* compute_units, uint64_t* hbm_bytes) {
  return guarded([&]() -> int {
    if (int rc = require_bound()) return rc;
    const auto rt = current_runtime();
    if (slot < 0 || static_cast<size_t>(slot) >= rt->devs.size()) return fail(KGX_EINVAL, "slot %d outside [0,%zu)", slot, rt->devs.size());
    const Device& dev = *rt->devs[static_cast<size_t>(slot)];
    if (name && name_len) std::snprintf(name, name_len, "%s", dev.name);
    if (arch && arch_len) std::snprintf(arch, arch_len, "%s", dev.arch);
    if (compute_units) *compute_units = dev.compute_units;
    if (hbm_bytes) *hbm_bytes = dev.hbm_bytes;
    return KGX_OK;
  });
}

void* kgx_stream(int slot) {
  const auto rt = current_runtime();
  if (!rt || slot < 0 || static_cast<size_t>(slot) >= rt->devs.size()) return nullptr;
  return static_cast<void*>(rt->devs[static_cast<size_t>(slot)]->stream);
}

int kgx_synchronize(void) {
  return guarded([&]() -> int {
    if (int rc = require_bound()) return rc;
    const auto rt = current_runtime();
    for (const auto& dev : rt->devs) {
      if (int rc = use_device(*dev)) return rc;
      KGX_HIP(hipStreamSynchronize(dev->stream));
      KGX_HIP(hipDeviceSynchronize());
    }
    return use_device(*rt->devs[0]);
  });
}

int kgx_reload_options(void) {
  return guarded([&]() -> int {
    reload_options();
    return KGX_OK;
  });
}

int kgx_release_scratch(void) {
  return guarded([&]() -> int {
    const auto rt = current_runtime();
    if (!rt) return KGX_OK;
    for (const auto& dev : rt->devs) {
      std::lock_guard<std::mutex> lock(dev->mutex);
      if (use_device(*dev) != KGX_OK) continue;
      if (dev->scratch) {
        if (dev->stream) (void)hipStreamSynchronize(dev->stream);
        (void)hipFree(dev->scratch);
      }
      dev->scratch = nullptr;
      dev->scratch_bytes = 0;
      for (int k = 0; k < 2; ++k) {
        if (dev->compact[k]) (void)hipFree(dev->compact[k]);
        dev->compact[k] = nullptr;
        dev->compact_bytes[k] = 0;
      }
      if (dev->words) (void)hipFree(dev->words);
      dev->words = nullptr;
      dev->words_bytes = 0;
      if (dev->bits) (void)hipFree(dev->bits);
      dev->bits = nullptr;
      dev->bits_bytes = 0;
      for (int k = 0; k < 2; ++k) {
        if (dev->pinned[k]) (void)hipHostFree(dev->pinned[k]);
        dev->pinned[k] = nullptr;
        dev->pinned_bytes[k] = 0;
      }
    }
    return use_device(*rt->devs[0]);
  });
}

}  // extern "C"
