"""profiles/cov_time.py -- qaCompute coverage kernel time on the benchmark shape (MSNV_COV_ITEM = intervals per work item)."""
import os, sys
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
sp = core.synth_params(n_species=3, contig_len=300000, n_samples=int(os.environ.get("NS", "160")), mean_cov=float(os.environ.get("COV", "10")), seed=1)
syn = core.Synth(sp); ctx = core.Context(0)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds.add_synth_samples(sp, 0, sp.n_samples, 0); info = ds.finalize()
ms = [ds.coverage_run()["ms_coverage"] for _ in range(8)]
print("MSNV_COV_ITEM=%s  coverage kernel %.4f ms (min %.4f)  intervals %d" % (os.environ.get("MSNV_COV_ITEM", "default"), sum(ms[2:]) / len(ms[2:]), min(ms), info["n_reads_pileup"]))
