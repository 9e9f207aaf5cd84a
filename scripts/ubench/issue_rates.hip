// Issue-rate microbenchmark for the instructions the sweeps lean on (gfx950): cycles per wave-instruction at 1, 2, 3, 4
// and 8 waves per SIMD, from s_memtime around a long unrolled chain of INDEPENDENT instructions.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rates scripts/ubench/issue_rates.hip && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP>
__global__ void __launch_bounds__(64) k_rate(unsigned long long* out, unsigned* sink, int iters, unsigned seed) {
  unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  double d0 = a0 * 1e-9 + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
  const unsigned k0 = seed ^ 0x07070707u, k1 = seed | 0x01010101u;
  const double e = 1.0000001, f = 1e-9;
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = i;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP16(asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_perm_b32 %3, %1, %2, %3\n\tv_perm_b32 %4, %1, %2, %4\n\tv_perm_b32 %5, %1, %2, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(k0), "v"(k1));) }
    if (OP == 1) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x6c\n\tv_bitop3_b32 %1, %1, %4, %5 bitop3:0x6c\n\tv_bitop3_b32 %2, %2, %4, %5 bitop3:0x6c\n\tv_bitop3_b32 %3, %3, %4, %5 bitop3:0x6c" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0), "v"(k1));) }
    if (OP == 2) { REP16(asm volatile("v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0));) }
    if (OP == 3) { REP16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0));) }
    if (OP == 4) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e), "v"(f));) }
    if (OP == 5) { REP16(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(f));) }
    if (OP == 6) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));) }
    if (OP == 7) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 4, %4\n\tv_lshl_add_u32 %1, %1, 4, %4\n\tv_lshl_add_u32 %2, %2, 4, %4\n\tv_lshl_add_u32 %3, %3, 4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0));) }
    if (OP == 8) { REP16(asm volatile("v_bfe_u32 %0, %0, 8, 8\n\tv_bfe_u32 %1, %1, 8, 8\n\tv_bfe_u32 %2, %2, 8, 8\n\tv_bfe_u32 %3, %3, 8, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 9) {   // ds_read_b128 with in-range pseudo-random 16-byte slots (conflicts as they fall), 4 per group
      unsigned ad0 = (a0 & 1023u) << 4, ad1 = (a1 & 1023u) << 4, ad2 = (a2 & 1023u) << 4, ad3 = (a3 & 1023u) << 4;
      REP16(asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(&d0))), "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(&d2))),
                           "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(&d4))), "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(&d6)))
                         : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3) : "memory");)
    }
    if (OP == 10) { REP16(asm volatile("v_lshrrev_b32 %0, 4, %0\n\tv_lshrrev_b32 %1, 4, %1\n\tv_lshrrev_b32 %2, 4, %2\n\tv_lshrrev_b32 %3, 4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 11) { REP16(asm volatile("v_max_f64 %0, %0, %4\n\tv_max_f64 %1, %1, %4\n\tv_max_f64 %2, %2, %4\n\tv_max_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(f));) }
    if (OP == 12) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0) : "vcc");) }
    if (OP == 13) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0), "v"(k1));) }
    if (OP == 14) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n\tv_pk_add_u16 %1, %1, %4\n\tv_pk_add_u16 %2, %2, %4\n\tv_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0));) }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ static_cast<unsigned>(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int OP>
void run(const char* name, int per_iter) {
  const int iters = 200;
  unsigned long long* d_out;
  unsigned* d_sink;
  const int max_blocks = 256 * 4 * 8;
  hipMalloc(&d_out, max_blocks * sizeof(unsigned long long));
  hipMalloc(&d_sink, max_blocks * 64 * sizeof(unsigned));
  printf("%-22s", name);
  for (int waves_per_simd : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * waves_per_simd;          // one 64-thread block per wave; fills every SIMD with that many waves
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(64), 0, 0, d_out, d_sink, iters, 12345u);
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(64), 0, 0, d_out, d_sink, iters, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += double(v);
    mean /= blocks;
    // s_memtime ticks at 100 MHz on gfx9xx? report both raw ticks per instruction and per-SIMD aggregate
    const double per_instr = mean / (double(iters) * per_iter);
    printf("  w%d: %7.3f (x%d = %6.3f)", waves_per_simd, per_instr, waves_per_simd, per_instr / waves_per_simd);
  }
  printf("\n");
  hipFree(d_out);
  hipFree(d_sink);
}

int main() {
  printf("ticks of s_memtime per wave-instruction as seen by ONE wave (and divided by the waves sharing its SIMD)\n");
  run<2>("v_and_b32", 64);
  run<3>("v_add_u32", 64);
  run<10>("v_lshrrev_b32", 64);
  run<0>("v_perm_b32", 64);
  run<1>("v_bitop3_b32", 64);
  run<7>("v_lshl_add_u32", 64);
  run<8>("v_bfe_u32", 64);
  run<13>("v_and_or_b32", 64);
  run<14>("v_pk_add_u16", 64);
  run<12>("v_add_co/addc pair(x2)", 64);
  run<4>("v_fma_f64", 64);
  run<5>("v_add_f64", 64);
  run<6>("v_mul_f64", 64);
  run<11>("v_max_f64", 64);
  run<9>("ds_read_b128", 64);
  return 0;
}
