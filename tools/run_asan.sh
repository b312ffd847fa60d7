#!/bin/bash
# The CPU test-suite against the host sanitizer builds (AddressSanitizer + UBSan; no GPU involved):
#   libpmk_host_asan.so  = host side of the C ABI (pmk_api.cpp, pmk_comm.cpp, pmk_bsp.cpp) + no-GPU stubs for the launchers
#   libpmk_oracle_asan.so = the CPU oracle
# Python itself is not instrumented: the runtime is preloaded, leak detection is off (the interpreter never frees its
# arenas), and any ASan / UBSan report makes the run fail.  Log: profiles/r03_asan_cpu_tests.log
set -e
cd "$(dirname "$0")/.."
make -C patchmixturekriging_amd/csrc asan >/dev/null
make -C oracle asan >/dev/null
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export PMK_LIB="$PWD/patchmixturekriging_amd/csrc/libpmk_host_asan.so"
export PMK_ORACLE_LIB="$PWD/oracle/libpmk_oracle_asan.so"
export PMK_ASAN_RUN=1
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
