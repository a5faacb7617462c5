"""profiles/shape_sweep.py -- pileup kernel time and rate over input shapes away from the BASELINE one (same total bases
where possible): looks for performance cliffs, not for records.  Prints one line per shape."""
import os, sys
sys.path.insert(0, os.getcwd())
from metasnv_amd import core

BASE = dict(n_species=3, contig_len=300000, n_samples=160, mean_cov=10.0, seed=1)
SHAPES = [
    ("baseline", {}),
    ("lowq_30pct", dict(frac_lowq=0.30)),
    ("lowq_80pct", dict(frac_lowq=0.80)),
    ("indel_reads_50pct", dict(frac_indel_reads=0.5)),
    ("clip_reads_50pct", dict(frac_clip_reads=0.5)),
    ("flagged_50pct", dict(frac_flagged=0.5)),
    ("uneven_sigma2", dict(sigma_cov=2.0)),
    ("absent_80pct", dict(frac_absent=0.8, mean_cov=50.0)),
    ("snv_dense_5pct", dict(snv_density=0.05)),
    ("tiny_contigs_3000x300", dict(n_species=3000, contig_len=300)),
    ("small_contigs_300x3000", dict(n_species=300, contig_len=3000)),
    ("few_deep_16x100", dict(n_samples=16, mean_cov=100.0)),
    ("many_shallow_1600x1", dict(n_samples=1600, mean_cov=1.0)),
    ("one_sample_1600x", dict(n_samples=1, mean_cov=1600.0, sigma_cov=0.0, frac_absent=0.0)),
    ("sparse_500x5x_20ofN", dict(n_species=150, contig_len=2070000, n_samples=500, mean_cov=5.0, sigma_cov=0.3, contigs_per_species_max=20, species_per_sample=1, frac_absent=0.75)),
    ("config3_quarter", dict(n_species=25, contig_len=3000000, n_samples=160, mean_cov=10.0, sigma_cov=0.7, contigs_per_species_max=50, species_per_sample=3, frac_absent=0.1667)),
]
only = sys.argv[1:]
ctx = core.Context(0)
for name, kw in SHAPES:
    if only and name not in only:
        continue
    sp = core.synth_params(**{**BASE, **kw})
    syn = core.Synth(sp)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    ds.add_synth_samples(sp, 0, sp.n_samples, 0)
    info = ds.finalize()
    for _ in range(2):
        ds.run()
    sts = ds.run_many(10)
    ms = sum(s["ms_pileup"] for s in sts) / len(sts)
    tot = sum(s["ms_total"] for s in sts) / len(sts)
    b = info["n_pileup_bases"]
    print("%-26s bases %.3e pairs %7d  pileup %.3f ms  pass %.3f ms  %.2f Tbases/s (kernel)  sites %d events %d" %
          (name, b, info["n_pairs"], ms, tot, b / ms / 1e9, sts[-1]["n_sites"], sts[-1]["n_events"]), flush=True)
    ds.close()
