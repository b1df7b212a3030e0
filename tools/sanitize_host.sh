#!/bin/bash
# Usage (CPU only, from the repo root): tools/sanitize_host.sh OUTFILE
# Builds the host translation unit of the library with -fsanitize=thread and with -fsanitize=address,undefined (host code only: -Xarch_host)
# and runs tools/sanitize_driver.cpp against each: the GroupPool rendezvous (clean, injected error, time-out) and a sample of choreography
# dry runs. Any sanitizer report fails the run. ~4 minutes (two extra builds of lbm_hip.hip + the four kernel objects).
set -e
ROOT=$PWD
R=$ROOT/highperformancecomputing-latticeboltzmannmethod_amd/csrc
OUT=$ROOT/${1:?usage: tools/sanitize_host.sh OUTFILE}
W=$(mktemp -d /tmp/lbm_san.XXXXXX)
CXX=/opt/rocm/lib/llvm/bin/clang++
cd "$W"
for u in "-DLBM_COL_T=double:f64" "-DLBM_COL_T=float:f32" "-DLBM_COL_TALL=1:tc" "-DLBM_COL_TALL=0:ts"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fPIC -c ${u%%:*} "$R/lbm_col.hip" -o col_${u##*:}.o &
done
wait
: > "$OUT"
for SAN in thread address; do
  EXTRA=""; XEXTRA=""
  if [ $SAN = address ]; then EXTRA="-fsanitize=undefined"; XEXTRA="-Xarch_host -fsanitize=undefined"; fi
  hipcc --offload-arch=gfx950 -O1 -g -std=c++20 -ffp-contract=off -fPIC -pthread -Xarch_host -fsanitize=$SAN $XEXTRA -DLBM_BUILD_ID_STR=\"sanitizer-------\" -c "$R/lbm_hip.hip" -o lbm_hip_$SAN.o 2> build_$SAN.log
  $CXX -O1 -g -fsanitize=$SAN $EXTRA -c "$ROOT/tools/sanitize_driver.cpp" -o driver_$SAN.o
  $CXX -fsanitize=$SAN $EXTRA -pthread -o drv_$SAN driver_$SAN.o lbm_hip_$SAN.o col_f64.o col_f32.o col_tc.o col_ts.o -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "== host code of lbm_hip.hip built with -fsanitize=$SAN $EXTRA" >> "$OUT"
  set +e
  TSAN_OPTIONS="halt_on_error=0 exitcode=66" ASAN_OPTIONS="detect_leaks=0 exitcode=66" UBSAN_OPTIONS="halt_on_error=0 print_stacktrace=1" ./drv_$SAN >> "$OUT" 2>&1
  echo "exit code $?" >> "$OUT"
  set -e
done
cd "$ROOT"
rm -rf "$W"
cat "$OUT"
! grep -q "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:\|exit code [1-9]" "$OUT"
