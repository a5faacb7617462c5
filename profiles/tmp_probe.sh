export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_devpack.py -x -q 2>&1 | tail -2
timeout 900 python3 tests/fuzz_parity.py 300 93 2>&1 | tail -1
timeout -s ABRT 300 python3 -X faulthandler bench.py --workload config3 --scale 0.25 --no-cpu-baseline --no-annotation --steps 3 --warmup 1 2> gpurun_out/c3.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['host']['wall_s_whole_run'], d['host']['pack_on_device_rank0'])"
