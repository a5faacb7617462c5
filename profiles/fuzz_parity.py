"""Randomised parity sweep (HIP path vs oracle) over the generator's and the caller's parameters; run on the GPU box:
   python3 profiles/fuzz_parity.py [n_cases] [seed].  Every case is small enough for the oracle to finish in < 1 s."""
import os, random, sys, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
from parity import run_oracle, first_diff

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = core.Context(0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    read_len = rnd.choice([20, 36, 50, 75, 100, 100, 150, 250, 400])
    contig_len = rnd.choice([300, 1500, 2047, 2048, 2049, 4096, 7000, 20000])
    n_species = rnd.choice([1, 1, 2, 3, 5])
    n_samples = rnd.choice([1, 2, 3, 7, 16, 33])
    mean_cov = rnd.choice([0.5, 2, 5, 10, 30, 80, 200, 300])
    budget = 2.0e7
    if contig_len * n_species * n_samples * mean_cov > budget:
        mean_cov = max(0.5, budget / (contig_len * n_species * n_samples))
    kw = dict(n_species=n_species, contig_len=contig_len, n_samples=n_samples, mean_cov=mean_cov, read_len=min(read_len, contig_len),
              sigma_cov=rnd.choice([0.1, 0.5, 1.0]), frac_absent=rnd.choice([0.0, 0.1, 0.5]), snv_density=rnd.choice([0.0, 0.007, 0.05]),
              error_rate=rnd.choice([0.0, 0.001, 0.02]), frac_lowq=rnd.choice([0.0, 0.1, 0.5]), frac_indel_reads=rnd.choice([0.0, 0.04, 0.3]),
              frac_clip_reads=rnd.choice([0.0, 0.03, 0.3]), frac_flagged=rnd.choice([0.0, 0.03]), lowercase_ref=rnd.choice([0, 1]), seed=rnd.randrange(1 << 30))
    pk = dict(min_coverage=rnd.choice([1, 4, 4, 10]), calling_threshold=rnd.choice([1, 2, 4, 4]), min_fraction=rnd.choice([0.01, 0.01, 0.2, 0.0]),
              min_baseq=rnd.choice([0, 13, 13, 30]))
    os.environ["MSNV_LAYOUT"] = rnd.choice(["pieces", "dense"])
    sp = core.synth_params(**kw)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    p = core.default_params(**pk)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
    for s in samples:
        ds.add_sample_records(s)
    info = ds.finalize(); ds.run()
    if rnd.random() < 0.3:
        ds.run_many(3, overlap=rnd.random() < 0.5)
    with tempfile.TemporaryDirectory() as td:
        ds.write_calls(td + "/c", td + "/i"); pop, ind = open(td + "/c").read(), open(td + "/i").read()
    ds.close()
    o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    ok = pop == o[0] and ind == o[1] and info["n_pileup_bases"] == o[3]
    if not ok:
        bad += 1
        print("MISMATCH case %d layout %s kw %s params %s\n  %s" % (case, os.environ["MSNV_LAYOUT"], kw, pk, first_diff(pop, o[0]) if pop != o[0] else first_diff(ind, o[1])))
print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
