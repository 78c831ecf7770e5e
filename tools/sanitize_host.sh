#!/bin/bash
# Sanitizers over the HOST side of the product and over the oracle, on the CPU (no GPU needed; sanitizers are not available on
# the GPU pool).  Exits 0 when every build is clean.
#
#   1. libgpvecchia_hip's host code (gpv_api.hip, the launch wrappers of every kernel TU, gpv_order.cpp): every .hip file compiled
#      --offload-host-only with -fsanitize=address,undefined, linked against tests/sanitize/mock_hip_runtime.cpp (host-memory
#      stand-in for the HIP runtime: kernels do not run) and driven through the public C ABI by tests/sanitize/host_driver.cpp;
#   2. the same under -fsanitize=thread (hash threads, staged copies, plan cache mutex, replica threads);
#   3. oracle/*.c with -fsanitize=address,undefined, driven by the non-GPU test suite's oracle tests.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${SAN_OUT:-/tmp/gpv_sanitize}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANGXX=/opt/rocm/lib/llvm/bin/clang++
CSRC=$ROOT/gpvecchia_amd/csrc
PLIST="${SAN_PLIST:-4 11 21 31 61}"          # row lengths whose launch wrappers are compiled (host side only: seconds each)
mkdir -p "$OUT"

build_and_run() {   # $1 = tag, $2 = sanitizer flags, $3 = driver args
  local tag=$1 san=$2 args=$3 B=$OUT/$1
  mkdir -p "$B"
  local CF="--offload-host-only -O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer $san -DGPV_DEVELOPER -Wno-unused-variable"
  local plx=""; for P in $PLIST; do plx="$plx X($P)"; done
  local pids=()
  for f in gpv_api gpv_aux_kernels gpv_posterior gpv_laplace gpv_sets_generic gpv_nn; do
    $HIPCC $CF "-DGPV_P_LIST(X)=$plx" -c $CSRC/$f.hip -o $B/$f.o & pids+=($!)
  done
  for P in $PLIST; do
    $HIPCC $CF "-DGPV_P_LIST(X)=$plx" -DGPV_INST_P=$P -c $CSRC/gpv_sets_inst.hip -o $B/sets_p$P.o & pids+=($!)
  done
  $HIPCC $CF -x c++ -c $CSRC/gpv_order.cpp -o $B/order.o & pids+=($!)
  $HIPCC $CF -c $ROOT/tests/sanitize/mock_hip_runtime.cpp -o $B/mock.o & pids+=($!)
  $HIPCC $CF -c $ROOT/tests/sanitize/host_driver.cpp -o $B/driver.o & pids+=($!)
  for p in "${pids[@]}"; do wait $p; done
  # the objects name their (absent) device images: one dummy symbol each
  nm -u $B/*.o | awk '/__hip_fatbin_/ {print $2}' | sort -u | awk '{print "const char " $1 "[8] = {0};"}' > $B/fatbin_stubs.c
  gcc -c $B/fatbin_stubs.c -o $B/fatbin_stubs.o
  $CLANGXX $san -g $B/*.o -o $B/host_driver -lpthread -ldl -lm
  echo "== $tag: running host_driver $args"
  ( cd $B && ASAN_OPTIONS=detect_leaks=1:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
      TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 timeout 1500 ./host_driver $args ) 2>&1 | tee $B/run.log
  local rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ] || grep -q "runtime error\|ERROR: AddressSanitizer\|WARNING: ThreadSanitizer\|ERROR: LeakSanitizer" $B/run.log; then
    echo "== $tag: FAILED (rc $rc)"; return 1
  fi
  echo "== $tag: clean"
}

build_and_run asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined" ""
build_and_run tsan "-fsanitize=thread" "--quick"

echo "== oracle: -fsanitize=address,undefined under the oracle's own tests"
OB=$OUT/oracle; mkdir -p $OB
cp $ROOT/oracle/u_nzentries_oracle.c $ROOT/oracle/sparse_chol_oracle.c $OB/
gcc -O1 -g -fopenmp -fPIC -Wall -Wextra -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -shared -o $OB/liboracle.so $OB/u_nzentries_oracle.c $OB/sparse_chol_oracle.c -lm
( cd $ROOT && GPV_ORACLE_LIB=$OB/liboracle.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 1500 python -m pytest tests/test_oracle.py tests/test_golden.py -x -q -m "not gpu" \
    -p no:cacheprovider ) 2>&1 | tee $OB/run.log | tail -5
if [ ${PIPESTATUS[0]} -ne 0 ] || grep -q "runtime error\|ERROR: AddressSanitizer" $OB/run.log; then echo "== oracle: FAILED"; exit 1; fi
echo "== oracle: clean"
echo "sanitize_host.sh: all clean"
