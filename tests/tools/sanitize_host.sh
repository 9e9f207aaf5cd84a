#!/bin/bash
# Sanitizers over the host-side C++ (flatteners, VCF readers, packages' host logic) and the oracle, CPU only: builds
# instrumented copies under /tmp, runs the CPU tests that drive them and a stand-alone check of the PopulationDB
# flattener, and keeps what the runs print under docs/sanitizers/.
#   bash tests/tools/sanitize_host.sh            AddressSanitizer + UBSan
#   bash tests/tools/sanitize_host.sh thread     ThreadSanitizer (the flatteners, the bgzf reader and the variant sort are threaded)
set -e
MODE=${1:-address}
shift || true
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=/tmp/kgx_san_$MODE
rm -rf $OUT && mkdir -p $OUT $ROOT/docs/sanitizers
if [ "$MODE" = thread ]; then
  SAN="-O1 -g -fsanitize=thread -fno-omit-frame-pointer"
  PRELOAD=$(gcc -print-file-name=libtsan.so)
  export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 exitcode=66"
  TESTS="tests/test_vcf_flatten_cpu.py tests/test_variant_sort_cpu.py tests/test_pf7_location_cpu.py"
else
  SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
  PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
  export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1
  TESTS="tests/test_vcf_flatten_cpu.py tests/test_variant_sort_cpu.py tests/test_oracle_pins.py tests/test_golden.py tests/test_pf7_location_cpu.py"
fi
LOG=$ROOT/docs/sanitizers/${MODE}_sanitizer_run.txt
{
echo "# tests/tools/sanitize_host.sh $MODE -- $(g++ --version | head -1)"
cd $ROOT/kgl_gene_amd/csrc/host
g++ $SAN -std=c++20 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -pthread -shared -o $OUT/libkgx_analysis.so \
    kgx_flatten.cpp kgx_vcf_flatten.cpp kgx_variant_sort.cpp kgx_vcf_io.cpp kgx_pf7_resources.cpp kgx_host_capi.cpp kga_analysis_gpu_allele.cpp kga_analysis_gpu_location.cpp kga_analysis_gpu_inbreed.cpp \
    -L$ROOT/kgl_gene_amd/lib -lkgx -lz -Wl,-rpath,$ROOT/kgl_gene_amd/lib
# the PopulationDB flattener on its own (multi-threaded discovery walk + packing), 1 thread against many
g++ $SAN -std=c++20 -Wall -ffp-contract=off -pthread -I. -o $OUT/flatten_population_check $ROOT/tests/tools/flatten_population_check.cpp kgx_flatten.cpp
echo "## flatten_population_check"
$OUT/flatten_population_check
cd $ROOT/oracle
g++ $SAN -std=c++20 -fPIC -Wall -pthread -ffp-contract=off -shared -o $OUT/libkgo.so kgo_core.cpp kgo_analysis.cpp kgo_inbreed.cpp kgo_inbreed_dense.cpp kgo_vcf.cpp kgo_sort.cpp kgo_pf7.cpp kgo_capi.cpp kgo_fast.cpp
cd $ROOT
export KGX_SANITIZED_HOST_LIB=$OUT/libkgx_analysis.so KGX_SANITIZED_ORACLE_LIB=$OUT/libkgo.so
echo "## pytest $TESTS"
# -s: a sanitizer report goes to stderr and the process _exit()s; under pytest's capture it would be lost
LD_PRELOAD=$PRELOAD python -m pytest $TESTS -x -q -s -m "not gpu" -p no:cacheprovider "$@"
echo "## done: exit 0"
} 2>&1 | tee $LOG
exit ${PIPESTATUS[0]}
