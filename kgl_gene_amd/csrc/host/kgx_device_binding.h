// Device binding shared by the GPU analysis packages: which MI355X devices a package shards its genomes over.
// The reference is ONE process fanning out one task per genome over its thread pool
// (kgl_variant_db_population.cpp:386-433); here the same process fans the genomes out over the devices of the node,
// one contiguous shard each (kgx_init, include/kgx.h), with one RCCL all-reduce of the per-variant counts.
//
// Package parameters (all optional; the first one present wins, in this order):
//   DeviceList  comma-separated HIP ordinals, one genome shard per entry ("0,1,2,3"; an ordinal may repeat)
//   Devices     how many devices to shard over, ordinals 0..N-1; 0 = every visible device
//   Device      one ordinal (default 0)
#ifndef KGX_DEVICE_BINDING_H
#define KGX_DEVICE_BINDING_H

#include <charconv>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/kgx.h"

namespace kellerberrin::genome::analysis::gpu {

// Returns false (and the reason) when no device can be bound: the package is then disabled, the reference's
// convention for a failed initializeAnalysis (kgl_app/kgl_package_analysis.cpp:41-42).  There is no CPU fallback.
template <typename ParameterList>
inline bool bindDevices(const ParameterList& named_parameters, std::string& description, std::string& error) {
  std::vector<int> ordinals{0};
  bool all_visible = false;
  bool decided = false;
  for (int pass = 0; pass < 3 && !decided; ++pass)
    for (const auto& [block_name, named_vector] : named_parameters.getMap())
      for (const auto& parameter_map : named_vector.second) {
        if (decided) break;
        if (pass == 0) {
          if (auto v = parameter_map.getString("DeviceList")) {
            ordinals.clear();
            std::stringstream list(v.value().front());
            std::string item;
            while (std::getline(list, item, ',')) {
              if (item.empty()) continue;
              int ordinal = 0;
              const auto [end, errc] = std::from_chars(item.data(), item.data() + item.size(), ordinal);
              if (errc != std::errc() || end != item.data() + item.size() || ordinal < 0) {
                error = "DeviceList entry '" + item + "' is not a device ordinal";
                return false;
              }
              ordinals.push_back(ordinal);
            }
            decided = !ordinals.empty();
          }
        } else if (pass == 1) {
          if (auto v = parameter_map.getSize("Devices")) {
            const size_t n = v.value().front();
            ordinals.clear();
            all_visible = n == 0;
            for (size_t d = 0; d < n; ++d) ordinals.push_back(static_cast<int>(d));
            decided = true;
          }
        } else if (auto v = parameter_map.getSize("Device")) {
          ordinals.assign(1, static_cast<int>(v.value().front()));
          decided = true;
        }
      }
  const int rc = all_visible ? kgx_init(0, nullptr) : kgx_init(static_cast<int>(ordinals.size()), ordinals.data());
  if (rc != KGX_OK) {
    error = kgx_last_error();
    return false;
  }
  std::stringstream text;
  text << kgx_bound_devices() << " device(s), count exchange: " << kgx_exchange_kind();
  description = text.str();
  return true;
}

}  // namespace kellerberrin::genome::analysis::gpu

#endif  // KGX_DEVICE_BINDING_H
