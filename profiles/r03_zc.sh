#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zc; mkdir -p $O
MSNV_GUARD_ALLOC=1 timeout 200 python3 profiles/pmc_probe.py > $O/probe_guard.log 2>&1; echo "probe rc $?"; tail -n 6 $O/probe_guard.log
timeout 900 python3 -m pytest tests/test_gpu_guard.py -q -m gpu > $O/pytest_guard.log 2>&1; tail -n 25 $O/pytest_guard.log | cut -c1-300
