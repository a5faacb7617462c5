#!/bin/bash
# tests/run_sanitized.sh -- the CPU test-suite against AddressSanitizer + UBSan builds of the oracle and of the host
# side of libmsnv.so (GPU sanitizers are not available on the pool; kernels are covered by the parity tests).
# Builds into /tmp, leaves the tree untouched.  Run at the end of every round (round 3: 63 passed, clean; round 4: 66 passed, clean).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/msnv_sanitized; rm -rf $W; mkdir -p $W/metasnv_amd
cp -r $ROOT/metasnv_amd/csrc $W/metasnv_amd/csrc; cp -r $ROOT/include $W/include
make -C $W/metasnv_amd/csrc clean >/dev/null
make -C $W/metasnv_amd/csrc libmsnv.so CXXFLAGS="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -pthread" >/dev/null
make -C $ROOT/oracle asan >/dev/null
cp $ROOT/oracle/liborc.so $W/liborc_plain.so; cp $ROOT/oracle/liborc_asan.so $ROOT/oracle/liborc.so
trap 'cp $W/liborc_plain.so $ROOT/oracle/liborc.so; rm -f $ROOT/oracle/liborc_asan.so' EXIT
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 MSNV_LIBRARY=$W/metasnv_amd/csrc/libmsnv.so \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi.py::test_drop_in_executables_exist_and_print_usage
