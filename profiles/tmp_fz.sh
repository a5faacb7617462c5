timeout 1700 python3 tests/fuzz_parity.py 4000 31337 > gpurun_out/t_fz1.txt 2>&1; tail -n 1 gpurun_out/t_fz1.txt
FUZZ_PACK=device MSNV_FRONT=careful timeout 1200 python3 tests/fuzz_parity.py 2000 31338 > gpurun_out/t_fz2.txt 2>&1; tail -n 1 gpurun_out/t_fz2.txt
MSNV_FUSE=1 timeout 1200 python3 tests/fuzz_parity.py 2000 31339 > gpurun_out/t_fz3.txt 2>&1; tail -n 1 gpurun_out/t_fz3.txt
FUZZ_MANY=overlap timeout 1200 python3 tests/fuzz_parity.py 1500 31340 > gpurun_out/t_fz4.txt 2>&1; tail -n 1 gpurun_out/t_fz4.txt
