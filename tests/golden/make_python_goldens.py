#!/usr/bin/env python
"""Generates tests/golden/python_callers/* by RUNNING the reference's own Python callers of the hot path
(src/computeGenomeCoverage.py, src/collapse_coverages.py, src/createOptimumSplit.py) and its downstream
consumer metaSNV_Filtering.py from /root/reference on small inputs.  Only inputs and outputs are kept
(data, not source).  Re-run in the build container:  python tests/golden/make_python_goldens.py
"""
import os
import shutil
import subprocess
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "python_callers")


def w(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def main():
    if os.path.exists(OUT):
        shutil.rmtree(OUT)
    proj = os.path.join(OUT, "proj")
    # ---- qaCompute-style inputs for three BAMs (values chosen to exercise %f rounding and species grouping)
    contigs = [("spA.p.c1", 1000), ("spA.p.c2", 500), ("spB.q.c1", 2000), ("spC", 77), ("spB.q.c2", 333)]
    data = {
        "s1.bam": [(12.34567, 900, 800), (0.0, 0, 0), (3.5, 1500, 1000), (0.12987, 10, 0), (7.77777, 333, 300)],
        "s2.bam": [(0.00001, 1, 0), (99.99999, 500, 500), (1.0 / 3, 700, 300), (0.0, 0, 0), (2.5, 200, 100)],
        "a0.bam": [(5.0, 1000, 1000), (5.0, 500, 499), (0.0, 0, 0), (10.0, 77, 77), (0.00499, 3, 1)],
    }
    for bam, rows in data.items():
        cov = "Chromosome\tSeq_lem\tAvg_Cov\n"
        det = ""
        for (name, length), (avg, c1, c2) in zip(contigs, rows):
            cov += "%s\t%d\t%3.5f\n" % (name, length, avg)
            det += "%s\t%d\t%d\t%d\t" % (name, length, c1, c2) + "\t".join(["0"] * 8) + "\t\n"
        cov += "\nCov*X\tPercentage\tNr. of bases\n1\t50.00000\t123\n\nOther\nTotal number of reads: 10\n"
        w(os.path.join(proj, "cov", bam + ".cov"), cov)
        w(os.path.join(proj, "cov", bam + ".cov.detail"), det)
        subprocess.check_call([sys.executable, os.path.join(REF, "src/computeGenomeCoverage.py"),
                               os.path.join(proj, "cov", bam + ".cov"), os.path.join(proj, "cov", bam + ".cov.detail"),
                               os.path.join(proj, "cov", bam + ".cov.summary")])
    subprocess.check_call([sys.executable, os.path.join(REF, "src/collapse_coverages.py"), proj])
    w(os.path.join(proj, "bed_header"), "".join("%s\t1\t%d\n" % c for c in contigs))
    for n in (1, 2, 3, 5):
        os.makedirs(os.path.join(proj, "bestsplits_%d" % n), exist_ok=True)
        subprocess.check_call([sys.executable, os.path.join(REF, "src/createOptimumSplit.py"),
                               os.path.join(proj, "proj.all_cov.tab"), os.path.join(proj, "proj.all_perc.tab"),
                               os.path.join(proj, "bed_header"), str(n), os.path.join(proj, "bestsplits_%d" % n, "best_split")],
                              stdout=subprocess.DEVNULL)
    # ---- downstream contract: metaSNV_Filtering.py must read called_SNPs as we write it (SURVEY.md golden vector 5)
    f = os.path.join(OUT, "filtering", "proj")
    w(os.path.join(f, "all_samples"), "/x/y/s1.bam\n/x/y/s2.bam\n")
    w(os.path.join(f, "proj.all_cov.tab"), "\ts1.bam\ts2.bam\nTaxId\tAverage_cov\tAverage_cov\nspA\t8.230447\t8.230447\nspB\t3.500000\t3.500000\n")
    w(os.path.join(f, "proj.all_perc.tab"), "\ts1.bam\ts2.bam\nTaxId\tPercentage_1x\tPercentage_1x\nspA\t60.000000\t60.000000\nspB\t75.000000\t75.000000\n")
    w(os.path.join(f, "snpCaller", "called_SNPs"),
      "spA.p.c1\t-\t11\tC\t5|8\t4|T|.|4|0,6|G|N[GCT-GGT]|1|5\n"
      "spA.p.c1\tgeneA\t12\tG\t2|9\t4|A|.|2|2\n"
      "spB.q.c1\t-\t7\tT\t40|40\t4|C|.|4|0\n")
    w(os.path.join(f, "snpCaller", "indiv_called"), "")
    for sub in ("filtered/pop", "filtered/ind"):
        os.makedirs(os.path.join(f, sub), exist_ok=True)
    env = dict(os.environ)
    subprocess.check_call([sys.executable, os.path.join(REF, "metaSNV_Filtering.py"), f], cwd=os.path.dirname(f), env=env, stdout=subprocess.DEVNULL)
    # ---- second Filtering case (section 8 f1): three species, six samples, random counts that exercise repr() of
    # the frequencies (exponent form below 1e-4, 16-17 significant digits), zero coverages, the -c / -p gates, --ind
    import json
    import random
    import importlib.util
    rnd = random.Random(11)
    f2 = os.path.join(OUT, "filtering2", "proj")
    names = ["s%d.bam" % i for i in range(6)]
    w(os.path.join(f2, "all_samples"), "".join("/data/run/%s\n" % n for n in names))
    covtab = "\t" + "\t".join(names) + "\nTaxId\t" + "\t".join(["Average_cov"] * 6) + "\n"
    pertab = "\t" + "\t".join(names) + "\nTaxId\t" + "\t".join(["Percentage_1x"] * 6) + "\n"
    rows = {"spA": ([9.5, 0.5, 7.25, 6.0, 1.0, 12.0], [80, 5, 55.5, 41, 30, 99]),
            "spB": ([2.0, 2.0, 2.0, 2.0, 2.0, 2.0], [50, 50, 50, 50, 50, 50]),
            "spC": ([1.0, 30.0, 0.0, 25.5, 0.0, 0.0], [10, 90, 0, 70, 0, 0]),
            "spD": ([3.0, 3.0, 3.0, 3.0, 3.0, 3.0], [9, 9, 9, 9, 9, 9])}
    for sp, (c, p) in rows.items():
        covtab += sp + "\t" + "\t".join("%f" % x for x in c) + "\n"
        pertab += sp + "\t" + "\t".join("%f" % x for x in p) + "\n"
    w(os.path.join(f2, "proj.all_cov.tab"), covtab)
    w(os.path.join(f2, "proj.all_perc.tab"), pertab)

    def snp_lines(n_lines, seed):
        r = random.Random(seed)
        out = ""
        contigs = ["spA.p.c1", "spA.p.c2", "spB.q.c1", "spC", "spD.x", "spE.y"]
        for k in range(n_lines):
            ctg = contigs[min(len(contigs) - 1, k * len(contigs) // n_lines)]
            cov = [r.choice([0, 0, 1, 2, 3, 4, 5, 6, 7, 9, 13, 40, 97, 1000, 29989, 200003]) for _ in range(6)]
            ents = []
            for alt in r.sample("ACGT", r.choice([1, 1, 1, 2, 3])):
                cnt = [r.randint(0, c) if c else 0 for c in cov]
                tag = r.choice([".", ".", "S[GCT-GCC]", "N[ATG-ACG]", "N[TA-TC]"])
                ents.append("%d|%s|%s|%s" % (sum(cnt), alt, tag, "|".join(map(str, cnt))))
            out += "%s\t%s\t%d\t%s\t%s\t%s\n" % (ctg, r.choice(["-", "gene%d" % k]), 10 + 3 * k, r.choice("ACGTacgtN"),
                                                 "|".join(map(str, cov)), ",".join(ents))
        return out
    tiny = "spA.p.c2\tgeneT\t9001\tA\t200003|29989|200003|1000|97|13\t3|C|.|1|0|2|0|0|0,31|G|N[TA-TC]|7|3|1|5|9|6\n"   # 1/200003 -> repr in exponent form
    w(os.path.join(f2, "snpCaller", "called_SNPs.best_split_0"), snp_lines(120, 1) + tiny)
    w(os.path.join(f2, "snpCaller", "indiv_called.best_split_0"), snp_lines(60, 2))
    subprocess.check_call([sys.executable, os.path.join(REF, "metaSNV_Filtering.py"), f2, "-m", "2", "-d", "1", "-b", "10", "-c", "3", "-p", "0.4", "--ind"],
                          cwd=os.path.dirname(f2), env=env, stdout=subprocess.DEVNULL)
    # ---- metaSNV_DistDiv.py --dist (section 8 f3) on a copy of the filtered tables just produced + a longer synthetic
    # table that makes numpy's pairwise summation recurse (> 128 rows)
    dd = os.path.join(OUT, "distdiv", "proj")
    shutil.copytree(os.path.join(f2, "filtered", "pop"), os.path.join(dd, "filtered", "pop"))
    for t in ("proj.all_cov.tab", "proj.all_perc.tab"):
        shutil.copy(os.path.join(f2, t), os.path.join(dd, t))
    w(os.path.join(dd, "bed_header"), "spA.p.c1\t1\t1000\n")
    big = "\t" + "\t".join(names) + "\n"
    for k in range(700):
        vals = []
        for s in range(6):
            c = rnd.choice([0, 3, 7, 40, 97, 1000, 29989])
            vals.append("-1" if (c == 0 or rnd.random() < 0.15 or s == 4) else repr(rnd.randint(0, c) / c))
        big += "spZ.c:-:%d:A>T:.\t%s\n" % (k + 1, "\t".join(vals))
    w(os.path.join(dd, "filtered", "pop", "spZ.filtered.freq"), big)
    subprocess.check_call([sys.executable, os.path.join(REF, "metaSNV_DistDiv.py"), "--filt", os.path.join(dd, "filtered", "pop"), "--dist"],
                          cwd=os.path.dirname(dd), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # FILTER I known answers for a few threshold sets (relevant_taxa of the reference module, imported here only)
    spec = importlib.util.spec_from_file_location("ref_filtering", os.path.join(REF, "metaSNV_Filtering.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    answers = []
    for (b, d, m) in [(40.0, 5.0, 2), (10.0, 1.0, 2), (50.0, 2.0, 6), (0.0, 0.0, 1), (95.0, 10.0, 1)]:
        ns = type("A", (), {})()
        ns.coverage_file, ns.percentage_file, ns.b, ns.d, ns.m = os.path.join(f2, "proj.all_cov.tab"), os.path.join(f2, "proj.all_perc.tab"), b, d, m
        answers.append({"b": b, "d": d, "m": m, "SoI": mod.relevant_taxa(ns)["SoI"]})
    w(os.path.join(OUT, "filtering2", "relevant_taxa.json"), json.dumps(answers, indent=1, sort_keys=True))
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
