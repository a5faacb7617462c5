"""profiles/stage_threads.py -- wall seconds and inflate thread-seconds of reading + inflating the benchmark's 160 BAMs (msnv_dataset_stage_sample_bams) over the host thread count."""
import sys, time
sys.path.insert(0, ".")
from metasnv_amd import core
W = "/tmp/e2e_ab"
bams = open(W + "/all_samples").read().split()
for T in ([int(x) for x in sys.argv[1:]] or [8, 16, 32, 64, 128]):
    for rep in range(2):
        t0 = core.host_timers()
        ds = core.Dataset.from_files(None, bams[0], W + "/ref.fa")
        a = time.perf_counter(); ds.stage_sample_bams(bams, T); b = time.perf_counter()
        t1 = core.host_timers()
        ds.close()
        print("threads %3d: wall %.3f s, inflate %.2f thread-s (%.3f per thread), read %.2f thread-s" % (T, b - a, t1["inflate_host_s"] - t0["inflate_host_s"], (t1["inflate_host_s"] - t0["inflate_host_s"]) / T, t1["read_s"] - t0["read_s"]), flush=True)
