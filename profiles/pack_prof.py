"""profiles/pack_prof.py [testdata|config3|config4shard] [scale] -- builds one workload's dataset through the device pack (csrc/devpack.hip)
and prints what the per-read stage cost; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from metasnv_amd import core  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "testdata"
    argv = ["--workload", wl] + (["--scale", sys.argv[2]] if len(sys.argv) > 2 else [])
    sys.argv = [sys.argv[0]] + argv
    a = bench.parse()
    kw, label = bench.workload_params(a)
    sp = core.synth_params(**kw)
    syn = core.Synth(sp)
    ctx = core.Context(0)
    out = {"workload": label, "MSNV_PACK": os.environ.get("MSNV_PACK", "device")}
    for rep in range(2):                                   # (the first build pays the allocations and the code-object load)
        core.host_timers(reset=True)
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        t0 = time.perf_counter()
        ds.add_synth_samples(sp, 0, sp.n_samples, 0)
        t1 = time.perf_counter()
        info = ds.finalize()
        t2 = time.perf_counter()
        ps = ds.pack_stats()
        st = ds.run()
        out["build_%d" % rep] = {"add_samples_wall_s": t1 - t0, "finalize_wall_s": t2 - t1, "pack": ps, "host_timers": core.host_timers(),
                                 "pileup_bases": info["n_pileup_bases"], "pass_ms": st["ms_total"], "called": st["n_called_pop"]}
        ds.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
