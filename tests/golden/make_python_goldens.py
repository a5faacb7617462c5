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
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
