export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_devpack.py -x -q 2>&1 | tail -4
timeout 900 python3 tests/fuzz_parity.py 200 7 2>&1 | tail -2
python3 profiles/pack_resident.py testdata 1 3 | python3 -c "
import json,sys; d=json.load(sys.stdin)
for r in d['reps']: print(r)"
rm -rf gpurun_out/r05d_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/r05d_prof -o p --output-format csv -- python3 profiles/pack_resident.py testdata 1 3 > gpurun_out/r05d_prof.log 2>&1
rm -f gpurun_out/r05d_prof/p_kernel_trace.csv
