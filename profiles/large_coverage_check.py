"""One-off: qaCompute outputs (cov + detail files) of the device path against the oracle for every sample of the benchmark shape
(160 samples x 3 refGenomes x 300 kb) and of a deep variant (16 samples at 200x: ~4200 intervals per (tile, sample) pair)."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
import orc
for (ns, cov) in ((160, 10.0), (16, 200.0)):
    t0 = time.time()
    sp = core.synth_params(n_species=3, contig_len=300000, n_samples=ns, mean_cov=cov, seed=5)
    syn = core.Synth(sp)
    ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    ds.add_synth_samples(sp, 0, sp.n_samples, 0)
    ds.finalize(); st = ds.coverage_run()
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for i in range(ns):
            ds.write_coverage(i, td + "/c", td + "/d")
            want = orc.qacompute(syn.names, syn.lengths, syn.sample_records(i))
            if (open(td + "/c").read(), open(td + "/d").read()) != want:
                bad += 1
    print("%d samples at %gx: coverage kernel %.3f ms, %d samples differ from the oracle (%.0f s)" % (ns, cov, st["ms_coverage"], bad, time.time() - t0))
    ds.close(); ctx.close()
