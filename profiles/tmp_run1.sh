timeout 1500 python3 tests/fuzz_parity.py 1500 777 2>&1 | tail -4
FUZZ_PACK=device MSNV_FRONT=careful timeout 900 python3 tests/fuzz_parity.py 600 778 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_devpack.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
