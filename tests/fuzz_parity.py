"""Randomised parity sweep (HIP path vs oracle) over the generator's and the caller's parameters; run on the GPU box:
   python3 tests/fuzz_parity.py [n_cases] [seed].  Every case is small enough for the oracle to finish in < 1 s."""
import os, random, sys, tempfile, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
from metasnv_amd import core
from parity import run_oracle, first_diff
import orc

def sweep(n_cases, seed, verbose=True):
    """Runs n_cases random cases; returns the number of mismatches."""
    rnd = random.Random(seed)
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
                  frac_clip_reads=rnd.choice([0.0, 0.03, 0.3]), frac_flagged=rnd.choice([0.0, 0.03]), lowercase_ref=rnd.choice([0, 1]), frac_paired=rnd.choice([0.0, 0.0, 0.5, 1.0]), seed=rnd.randrange(1 << 30))
        pk = dict(min_coverage=rnd.choice([1, 4, 4, 10]), calling_threshold=rnd.choice([1, 2, 4, 4]), min_fraction=rnd.choice([0.01, 0.01, 0.2, 0.0]),
                  min_baseq=rnd.choice([0, 13, 13, 30]), max_depth=rnd.choice([8000, 8000, 8000, 60, 7]), min_mapq=rnd.choice([0, 0, 1, 30]),
                  count_orphans=rnd.choice([0, 1]), flag_filter=rnd.choice([0x704, 0x704, 0x400, 0]), ignore_overlaps=rnd.choice([0, 0, 0, 1]))
        rt = random.Random(kw["seed"] + 17)                      # snpCall's token, shortened now and then so that shallow pileups reach the cut (oracle/orc.h: token_cap)
        pk["token_limit"] = rt.choice([10000, 10000, 10000, 10000, 300, 90, 45])
        os.environ["MSNV_LAYOUT"] = rnd.choice(["pieces", "dense"])
        # the per-read stage: kernels (csrc/devpack.hip; tiny scan segments put a seam into most records) or, now and then, the host threads
        rk = random.Random(kw["seed"])
        os.environ["MSNV_PACK"] = os.environ.get("FUZZ_PACK") or rk.choice(["device", "device", "device", "host"])
        os.environ["MSNV_SCAN_SEG_KB"] = rk.choice(["1", "8", "256"])
        os.environ["MSNV_SCAN_SUB"] = rk.choice(["64", "200", "4096", "4096"])       # (the quick record scan's sub-segments; a seam that does not hold falls back to the careful kernel)
        os.environ["MSNV_LEAN"] = random.Random(kw["seed"] + 29).choice(["1", "1", "1", "0"])      # (whole-tile items: msnv_pileup_tiles_lean, or the ordinary body)
        os.environ["MSNV_MERGED_GATHER"] = rk.choice(["", "", "block"])                 # (sparse cohorts: the merged gather with a wavefront / a workgroup per group)
        if rk.random() < 0.15: os.environ["MSNV_SCAN"] = "segments"
        else: os.environ.pop("MSNV_SCAN", None)
        # what a real aligner writes: auxiliary fields behind the qualities, now and then a record without SEQ
        kw["frac_aux"] = rk.choice([0.0, 0.3, 1.0]); kw["frac_noseq"] = rk.choice([0.0, 0.0, 0.05])
        # KNOWN DEVIATION: a base behind snpCall's token cut is marked in its read's quality byte; a SEQ-less read has none.
        # Its bases print as N and count only under -Q 0 over a reference N ("." / ","): there, behind the cut, the product counts what snpCall
        # drops.  Not reachable with metaSNV's own command line (-Q 13 by default); the sweep keeps the combination out
        if pk["token_limit"] < 10000 and pk["min_baseq"] == 0: kw["frac_noseq"] = 0.0
        # KNOWN DEVIATION (same section): a deletion element `*` is printed when the quality of the base BEHIND the deletion passes -Q.  htslib edits
        # that base when the read's mate is pushed -- maybe after the `*` was printed (how far the engine has read ahead); the product edits all
        # mates first.  snpCall skips `*`, but a `*` more or less in front of the cut moves the cut by one character.  Overlapping mates, a
        # deletion and a short token at once: the sweep runs those cases with mpileup -x
        if pk["token_limit"] < 10000 and kw["frac_paired"] > 0 and kw["frac_indel_reads"] > 0: pk["ignore_overlaps"] = 1
        deep_mode = os.environ.get("MSNV_DEEP", "split")
        sp = core.synth_params(**kw)
        syn = core.Synth(sp)
        samples = [syn.sample_records(i) for i in range(sp.n_samples)]
        p = core.default_params(**pk)
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
        bed = None
        if rnd.random() < 0.25:                                  # a best_split file: `name 1 LEN` for a subset of the contigs
            keep = [t for t in range(n_species) if rnd.random() < 0.6] or [0]
            bed = [(t, 1, syn.lengths[t]) for t in keep]
            ds.set_bed(bed)
        for s in samples:
            ds.add_sample_records(s)
        info = ds.finalize()
        cov_ok = True
        if rnd.random() < 0.4 and bed is None:                   # qaCompute half on the same resident columns
            ds.fused_run()
            with tempfile.TemporaryDirectory() as td:
                for i, s in enumerate(samples):
                    if s.size == 0:
                        continue
                    try:
                        want = orc.qacompute(syn.names, syn.lengths, s)
                    except orc.OrcError:
                        continue                                 # a sample without mapped reads: undefined in the reference
                    ds.write_coverage(i, td + "/v", td + "/d")
                    cov_ok &= open(td + "/v").read() == want[0] and open(td + "/d").read() == want[1]
        else:
            ds.run()
        force = os.environ.get("FUZZ_MANY")                      # "overlap" / "many": every case runs a batch of passes that way (hunting races)
        r1, r2 = rnd.random(), rnd.random()
        if force or r1 < 0.3:
            ds.run_many(3, overlap=(force == "overlap") if force else r2 < 0.5)
        ann = fa = None
        with tempfile.TemporaryDirectory() as td:
            if rnd.random() < 0.3:                               # random gene table (overlaps, both strands) -> device annotation
                fa, ann = td + "/ref.fa", td + "/ann.tsv"
                syn.write_fasta(fa)
                with open(ann, "w") as g:
                    g.write("gene_id\texternal_id\tsequence_id\ttype\tinfo\tlength\tstart\tend\tstrand\tsc\tstop\tgc\n")
                    k = 0
                    for t in range(n_species):
                        if rnd.random() < 0.3:
                            continue
                        for _ in range(rnd.randrange(1, 12)):
                            a = rnd.randrange(1, max(2, contig_len - 5)); b = min(contig_len - 3, a + rnd.randrange(0, max(1, contig_len // 3)))
                            g.write("%d\tg%d\t%s\tCDS\tx\t%d\t%d\t%d\t%s\tATG\tTAG\t0.4\n" % (k, k, syn.names[t], b - a + 1, a, b, rnd.choice("+-"))); k += 1
            try:
                ds.write_calls(td + "/c", td + "/i", ann, fa); pop, ind = open(td + "/c").read(), open(td + "/i").read()
                perr = None
            except core._lib.MsnvError as e:
                pop = ind = None; perr = e
            try:
                o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed, params=p, ann=ann, fasta=fa)
                oerr = None
            except orc.OrcError as e:
                o = None; oerr = e
        ds.close()
        if perr is not None or oerr is not None:                 # inputs outside the reference's domain: both sides must refuse
            ok = perr is not None and oerr is not None
            if not ok:
                print("DOMAIN DISAGREEMENT product=%s oracle=%s" % (perr, oerr)); pop, o = "", ("x", "", 0, 0); ind = ""
        else:
            # with a BED file the oracle counts only the bases mpileup prints; the library counts every shipped base of a read
            # that overlaps a region (position 1 of `name 1 LEN` splits): compare the unit-of-work count without BED only
            ok = pop == o[0] and ind == o[1] and (bed is not None or info["n_pileup_bases"] == o[3]) and cov_ok
            if not cov_ok:
                print("COVERAGE MISMATCH")
        if not ok:
            bad += 1
            why = []
            if pop is not None and o is not None:
                if pop != o[0]: why.append("called_SNPs")
                if ind != o[1]: why.append("indiv_called")
                if bed is None and info["n_pileup_bases"] != o[3]: why.append("bases %d vs %d" % (info["n_pileup_bases"], o[3]))
                if not cov_ok: why.append("coverage")
            print("MISMATCH[%s] bed=%s ann=%s case %d layout %s kw %s params %s\n  %s" % (",".join(why), bed is not None, ann is not None, case, os.environ["MSNV_LAYOUT"], kw, pk, first_diff(pop, o[0]) if pop != o[0] else first_diff(ind, o[1])))
    if verbose:
        print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
    ctx.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
