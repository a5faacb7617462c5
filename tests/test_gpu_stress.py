"""Races inside a pass: one cohort packed, run and read back dozens of times in one process -- every run must give the bytes of the
first one and of the oracle.  The cohort is the one the fuzz sweep met a doubled per-sample count on (1 case in ~10 000): noisy reads,
33 samples, small work items at the end of every contig; the staging buffer of the allele events was reserved with add / take-back,
which let a second wavefront's slots end up above the fill count (kernels.hip: narrow_pass).  About one run in eight showed it."""
import os
import tempfile

import pytest

from metasnv_amd import core
from parity import run_oracle

pytestmark = pytest.mark.gpu

KW = dict(n_species=5, contig_len=20000, n_samples=33, mean_cov=6.06, read_len=150, sigma_cov=1.0, frac_absent=0.0, snv_density=0.007, error_rate=0.02,
          frac_lowq=0.1, frac_indel_reads=0.0, frac_clip_reads=0.3, frac_flagged=0.0, lowercase_ref=1, frac_paired=0.0, seed=180524760)
PK = dict(min_coverage=4, calling_threshold=4, min_fraction=0.01, min_baseq=13, max_depth=7, count_orphans=1, flag_filter=1024)


@pytest.mark.parametrize("layout,alleles", [("dense", "events"), ("pieces", "events"), ("pieces", "planes")])
def test_repeated_runs_of_one_cohort_give_the_same_bytes(layout, alleles, monkeypatch):
    monkeypatch.setenv("MSNV_LAYOUT", layout)
    monkeypatch.setenv("MSNV_ALLELES", alleles)
    sp = core.synth_params(**KW)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    p = core.default_params(**PK)
    want = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    ctx = core.Context(0)
    for it in range(40):
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
        for s in samples:
            ds.add_sample_records(s)
        ds.finalize()
        ds.run()
        if it % 3 == 1:
            ds.run_many(3, overlap=it % 2 == 1)
        with tempfile.TemporaryDirectory() as td:
            ds.write_calls(os.path.join(td, "c"), os.path.join(td, "i"), None, None)
            pop, ind = open(os.path.join(td, "c")).read(), open(os.path.join(td, "i")).read()
        ds.close()
        assert pop == want[0] and ind == want[1], "run %d differs from the oracle" % it
    ctx.close()
