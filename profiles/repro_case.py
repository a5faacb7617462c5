"""profiles/repro_case.py -- one fuzz case (kw / params copied from a MISMATCH line of tests/fuzz_parity.py) through the product in several
pass sequences, each compared with the oracle; MSNV_LAYOUT / MSNV_LIBRARY etc. from the environment."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
from parity import run_oracle, first_diff
kw = {'n_species': 5, 'contig_len': 20000, 'n_samples': 33, 'mean_cov': 6.0606060606060606, 'read_len': 150, 'sigma_cov': 1.0, 'frac_absent': 0.0, 'snv_density': 0.007, 'error_rate': 0.02, 'frac_lowq': 0.1, 'frac_indel_reads': 0.0, 'frac_clip_reads': 0.3, 'frac_flagged': 0.0, 'lowercase_ref': 1, 'frac_paired': 0.0, 'seed': 180524760}
pk = {'min_coverage': 4, 'calling_threshold': 4, 'min_fraction': 0.01, 'min_baseq': 13, 'max_depth': 7, 'min_mapq': 0, 'count_orphans': 1, 'flag_filter': 1024, 'ignore_overlaps': 0}
sp = core.synth_params(**kw); syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
p = core.default_params(**pk)
o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p) if not os.environ.get("NO_ORACLE") else (None, None)
ctx = core.Context(0)
for seq in sys.argv[1:] or ["run", "run,many", "run,overlap", "fused,overlap"]:
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
    for s in samples: ds.add_sample_records(s)
    info = ds.finalize()
    for step in seq.split(","):
        if step == "run": st = ds.run()
        elif step == "fused": ds.fused_run()
        elif step == "many": ds.run_many(3, overlap=False)
        elif step == "overlap": ds.run_many(3, overlap=True)
    with tempfile.TemporaryDirectory() as td:
        ds.write_calls(td + "/c", td + "/i", None, None); pop, ind = open(td + "/c").read(), open(td + "/i").read()
    if o[0] is None:
        print(seq, "ran", len(pop), len(ind), flush=True); ds.close(); continue
    ok = pop == o[0] and ind == o[1]
    print(seq, "OK" if ok else "MISMATCH " + (first_diff(pop, o[0]) if pop != o[0] else first_diff(ind, o[1]))[:400], {k: info[k] for k in ("allele_planes", "n_work", "n_pairs")}, flush=True)
    ds.close()
ctx.close()
