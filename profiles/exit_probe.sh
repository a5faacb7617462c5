t() { python3 - "$@" <<'PY'
import subprocess, sys, time
t0 = time.perf_counter(); r = subprocess.run(sys.argv[1:], capture_output=True, text=True); print("%.3f s : %s | %s" % (time.perf_counter() - t0, " ".join(sys.argv[1:])[:110], r.stdout.strip()[-200:]))
PY
}
cd /root/repo
t python3 -c "import time; t0=time.perf_counter(); from metasnv_amd import core; t1=time.perf_counter(); print('import %.3f' % (t1-t0))"
t python3 -c "import time,os; from metasnv_amd import core; t0=time.perf_counter(); n=core.device_count(); t1=time.perf_counter(); print('device_count %.3f' % (t1-t0)); os._exit(0)"
t python3 -c "import time,os; from metasnv_amd import core; t0=time.perf_counter(); c=core.Context(0); t1=time.perf_counter(); print('ctx %.3f' % (t1-t0)); import sys; sys.stdout.flush(); os._exit(0)"
t python3 -c "import time,os; from metasnv_amd import core; t0=time.perf_counter(); c=core.Context(0); t1=time.perf_counter(); c.close(); t2=time.perf_counter(); print('ctx %.3f close %.3f' % (t1-t0, t2-t1))"
t python3 -c "import time,os,sys; import numpy as np; a=np.ones(3<<30, dtype=np.uint8); sys.stdout.flush(); os._exit(0)"
cat /sys/kernel/mm/transparent_hugepage/enabled
