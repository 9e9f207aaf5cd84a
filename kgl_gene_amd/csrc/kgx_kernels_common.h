// Shared by the kernel headers of the C-ABI translation units (gfx950, wave64).
#ifndef KGX_KERNELS_COMMON_H
#define KGX_KERNELS_COMMON_H

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

typedef uint32_t kgx_v4u __attribute__((ext_vector_type(4)));

namespace kgx {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves per workgroup

}  // namespace kgx

#endif  // KGX_KERNELS_COMMON_H
