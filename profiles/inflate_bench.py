"""profiles/inflate_bench.py [samples] -- BGZF inflate of the benchmark shape's BAM files: host decoder (one thread) against the device
kernel (msnv_inflate_blocks: kernel time, and wall time with the transfers both ways)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from metasnv_amd import core
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sp = core.synth_params(n_species=3, contig_len=300000, n_samples=n, mean_cov=10.0, seed=1)
syn = core.Synth(sp)
os.makedirs("/tmp/ib", exist_ok=True)
paths = []
for i in range(n):
    p = "/tmp/ib/s%03d.bam" % i
    core.write_bam(p, syn.names, syn.lengths, syn.sample_records(i)); paths.append(p)
ctx = core.Context(0)
core.bgzf_inflate(paths[0], ctx)
for p in paths[:3]:
    t0 = time.perf_counter(); h, _ = core.bgzf_inflate(p); t1 = time.perf_counter()
    d, c = core.bgzf_inflate(p, ctx); t2 = time.perf_counter()
    print("%s: %.1f MB -> %.1f MB; host %.0f MB/s; device kernel %.3f ms = %.1f GB/s (%d blocks, %d refused), wall %.0f MB/s; equal %s" %
          (os.path.basename(p), os.path.getsize(p) / 1e6, h.size / 1e6, h.size / (t1 - t0) / 1e6, c["kernel_ms"], h.size / c["kernel_ms"] / 1e6,
           c["blocks"], c["host_blocks"], h.size / (t2 - t1) / 1e6, bool(np.array_equal(h, d))))
