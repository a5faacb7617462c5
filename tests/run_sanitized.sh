#!/bin/bash
# tests/run_sanitized.sh -- the CPU test-suite against AddressSanitizer + UBSan builds of the oracle and of the host
# side of libmsnv.so (GPU sanitizers are not available on the pool; kernels are covered by the parity tests).
# Builds into /tmp, leaves the tree untouched.  Run at the end of every round (round 3: 63 passed, clean; round 4: 66 passed, clean; round 5: 68 passed, clean).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/msnv_sanitized; rm -rf $W; mkdir -p $W/metasnv_amd
cp -r $ROOT/metasnv_amd/csrc $W/metasnv_amd/csrc; cp -r $ROOT/include $W/include
make -C $W/metasnv_amd/csrc clean >/dev/null
make -C $W/metasnv_amd/csrc libmsnv.so CXXFLAGS="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -pthread" >/dev/null
# the oracle's sanitizer build goes to /tmp too and is picked up through ORC_LIBRARY (tests/orc.py): nothing in the tree is overwritten, so a
# killed run leaves no sanitizer oracle behind (round 4 swapped oracle/liborc.so and restored it by a trap)
cp -r $ROOT/oracle $W/oracle
make -C $W/oracle asan >/dev/null
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 MSNV_LIBRARY=$W/metasnv_amd/csrc/libmsnv.so ORC_LIBRARY=$W/oracle/liborc_asan.so \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi.py::test_drop_in_executables_exist_and_print_usage
