#!/bin/bash
# AddressSanitizer + UBSan over the host-side C++ (flatteners, packages' host logic) and the oracle, CPU only:
# builds instrumented copies under /tmp and runs the CPU tests that drive them.   bash tests/tools/sanitize_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=/tmp/kgx_san
rm -rf $OUT && mkdir -p $OUT
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
cd $ROOT/kgl_gene_amd/csrc/host
g++ $SAN -std=c++20 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -pthread -shared -o $OUT/libkgx_analysis.so \
    kgx_flatten.cpp kgx_vcf_flatten.cpp kgx_variant_sort.cpp kgx_vcf_io.cpp kgx_pf7_resources.cpp kgx_host_capi.cpp kga_analysis_gpu_allele.cpp kga_analysis_gpu_location.cpp kga_analysis_gpu_inbreed.cpp \
    -L$ROOT/kgl_gene_amd/lib -lkgx -lz -Wl,-rpath,$ROOT/kgl_gene_amd/lib
cd $ROOT/oracle
g++ $SAN -std=c++20 -fPIC -Wall -pthread -ffp-contract=off -shared -o $OUT/libkgo.so kgo_core.cpp kgo_analysis.cpp kgo_inbreed.cpp kgo_inbreed_dense.cpp kgo_vcf.cpp kgo_sort.cpp kgo_pf7.cpp kgo_capi.cpp kgo_fast.cpp
cd $ROOT
export KGX_SANITIZED_HOST_LIB=$OUT/libkgx_analysis.so KGX_SANITIZED_ORACLE_LIB=$OUT/libkgo.so
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_vcf_flatten_cpu.py tests/test_variant_sort_cpu.py tests/test_oracle_pins.py tests/test_golden.py tests/test_pf7_location_cpu.py -x -q -s -m "not gpu" -p no:cacheprovider "$@"
# -s: a sanitizer report goes to stderr and the process _exit()s; under pytest's capture it would be lost
