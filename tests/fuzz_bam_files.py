"""Randomised check of the file entry points (msnv_call / msnv_coverage on BGZF BAM files written by the library) against
the oracle; run on the GPU box: python3 tests/fuzz_bam_files.py [seed]."""
import os, random, sys, tempfile, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core, _lib
from parity import run_oracle, first_diff
import orc
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ctx = core.Context(0); bad = 0
for case in range(40):
    kw = dict(n_species=rnd.choice([1, 2, 4]), contig_len=rnd.choice([500, 2048, 5000, 12000]), n_samples=rnd.choice([1, 3, 9]),
              mean_cov=rnd.choice([1, 8, 40, 260]), read_len=rnd.choice([30, 100, 150, 300]), frac_absent=rnd.choice([0, 0.4]),
              frac_indel_reads=rnd.choice([0, 0.2]), frac_clip_reads=rnd.choice([0, 0.2]), lowercase_ref=rnd.choice([0, 1]), seed=rnd.randrange(1 << 30))
    if kw["read_len"] + 10 > kw["contig_len"]: kw["read_len"] = 30
    sp = core.synth_params(**kw); syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    with tempfile.TemporaryDirectory() as td:
        fa = td + "/ref.fa"; syn.write_fasta(fa); paths = []
        for i, s in enumerate(samples):
            p = td + "/s%03d.bam" % i; core.write_bam(p, syn.names, syn.lengths, s); paths.append(p)
        a = _lib.CallArgs()
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        a.bam_paths, a.n_bams, a.ref_fasta = arr, len(paths), fa.encode()
        a.out_called_path = (td + "/c").encode(); a.out_indiv_path = (td + "/i").encode(); a.host_threads = rnd.choice([0, 1, 3])
        _lib.lib.msnv_params_default(C.byref(a.params))
        _lib.check(_lib.lib.msnv_call(ctx._h, C.byref(a)))
        pop, ind = open(td + "/c").read(), open(td + "/i").read()
        # coverage through the one-call form for one random BAM with mapped reads
        k = rnd.randrange(len(paths)); cov_ok = True
        if samples[k].size:
            try:
                want = orc.qacompute(syn.names, syn.lengths, samples[k])
                ca = _lib.CovArgs(paths[k].encode(), 10, 1, (td + "/v").encode(), (td + "/d").encode())
                _lib.check(_lib.lib.msnv_coverage(ctx._h, C.byref(ca)))
                cov_ok = open(td + "/v").read() == want[0] and open(td + "/d").read() == want[1]
            except orc.OrcError:
                pass
    o = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    if pop != o[0] or ind != o[1] or not cov_ok:
        bad += 1; print("MISMATCH", case, kw, "cov_ok", cov_ok, first_diff(pop, o[0])[:300])
print("40 cases from BAM files,", bad, "mismatches")
