"""profiles/stress_case.py N [mode] -- one fuzz case N times in ONE process (device memory is recycled between datasets, buffers come back
dirty): every run's called_SNPs / indiv_called against the first run's.  mode: run | many | overlap."""
import os, sys, tempfile, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
kw = {'n_species': 5, 'contig_len': 20000, 'n_samples': 33, 'mean_cov': 6.0606060606060606, 'read_len': 150, 'sigma_cov': 1.0, 'frac_absent': 0.0, 'snv_density': 0.007, 'error_rate': 0.02, 'frac_lowq': 0.1, 'frac_indel_reads': 0.0, 'frac_clip_reads': 0.3, 'frac_flagged': 0.0, 'lowercase_ref': 1, 'frac_paired': 0.0, 'seed': 180524760}
pk = {'min_coverage': 4, 'calling_threshold': 4, 'min_fraction': 0.01, 'min_baseq': 13, 'max_depth': 7, 'min_mapq': 0, 'count_orphans': 1, 'flag_filter': 1024, 'ignore_overlaps': 0}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "run"
sp = core.synth_params(**kw); syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
p = core.default_params(**pk)
ctx = core.Context(0)
ref = None; bad = 0
for it in range(N):
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
    for s in samples: ds.add_sample_records(s)
    ds.finalize(); st = ds.run()
    if mode == "many": ds.run_many(3, overlap=False)
    elif mode == "overlap": ds.run_many(3, overlap=True)
    with tempfile.TemporaryDirectory() as td:
        ds.write_calls(td + "/c", td + "/i", None, None); pop, ind = open(td + "/c").read(), open(td + "/i").read()
    ds.close()
    if ref is None: ref = (pop, ind); print("reference events", st["n_events"], flush=True)
    elif (pop, ind) != ref:
        bad += 1
        a, b = (pop, ref[0]) if pop != ref[0] else (ind, ref[1])
        nd = 0
        for x, y in zip(a.split("\n"), b.split("\n")):
            if x != y:
                nd += 1
                fx, fy = x.split("\t"), y.split("\t")
                who = [(k, u, v) for k, (u, v) in enumerate(zip("|".join(fx[5:]).split("|"), "|".join(fy[5:]).split("|"))) if u != v]
                if nd <= 6: print("iteration", it, "events", st["n_events"], fx[0], fx[2], "fields", who[:8], flush=True)
        print("iteration", it, "differing lines", nd, "events", st["n_events"], flush=True)
print(mode, N, "iterations,", bad, "differ from the first", flush=True)
ctx.close()
