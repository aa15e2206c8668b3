#!/bin/bash
# The host side (FASTX readers, the scanner, the gzip decoders, offsetter, results) under AddressSanitizer + UBSan, CPU only:
# builds sgcount_amd/libsgcount_host_asan.so from the host sources and runs the CPU host tests against it.
#   tools/asan_host_tests.sh [pytest -k expression]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/sgcount_amd
python3 -c "from sgcount_amd import build as b; b.build()" 2>/dev/null || (cd $R && python3 -c "from sgcount_amd import build as b; b.build()")
g++ -O1 -g -std=c++17 -fPIC -Wall -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared -DSGH_NO_MAIN \
    -o libsgcount_host_asan.so csrc/host/sgh.cpp csrc/host/sgh_scan.cpp csrc/host/sgh_inflate.cpp csrc/host/sgh_cli.cpp csrc/host/sgh_capi.cpp \
    -L. -lsgcount_hip -lz -Wl,-rpath,$R/sgcount_amd
cd $R
export SGH_HOST_LIB=$R/sgcount_amd/libsgcount_host_asan.so
export LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python3 -m pytest tests/test_host_cpu.py -x -q ${1:+-k "$1"}
