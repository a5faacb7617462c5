#!/usr/bin/env python
"""Generates tests/golden/subpopr/* by RUNNING the reference's raw-SNV consumers (SURVEY.md section 8 row f4:
src/subpopr/inst/getGenotypingSNVSubset.py and src/subpopr/inst/convertSNVtoAlleleFreq.py) from /root/reference on
small inputs written here.  Only inputs and outputs are kept (data, not source).
Re-run in the build container:  python tests/golden/make_subpopr_goldens.py
"""
import glob
import json
import os
import random
import shutil
import subprocess
import sys

REF = "/root/reference/src/subpopr/inst"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "subpopr")


def w(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def snp_line(rnd, contig, gene, pos, n_samples):
    cov = [rnd.choice([0, 1, 2, 3, 4, 5, 6, 7, 9, 13, 30, 77, 255, 1000, 7999]) for _ in range(n_samples)]
    ents = []
    for base in rnd.sample("ACGT", rnd.choice([1, 1, 2, 3])):
        cnt = [rnd.randrange(0, c + 1) if c else 0 for c in cov]
        tag = "." if gene == "-" else rnd.choice(["S[GCT-GCC]", "N[GCT-GGT]"])
        ents.append("%d|%s|%s|%s" % (sum(cnt), base, tag, "|".join(map(str, cnt))))
    return "%s\t%s\t%d\t%s\t%s\t%s\n" % (contig, gene, pos, rnd.choice("ACGT"), "|".join(map(str, cov)), ",".join(ents))


def main():
    if os.path.exists(OUT):
        shutil.rmtree(OUT)
    rnd = random.Random(4)
    S = 7
    contigs = {"spA.p.c1": "-", "spA.p.c2": "geneA2", "sp_B_x.q.c1": "-", "spC": "gC"}
    # ---- case "two": two called_SNPs files, three species (one position list shares positions with another)
    case = os.path.join(OUT, "two")
    lines = {c: [snp_line(rnd, c, g, p, S) for p in sorted(rnd.sample(range(1, 5000), 40))] for c, g in contigs.items()}
    w(os.path.join(case, "metasnv", "snpCaller", "called_SNPs.best_split_1"), "".join(lines["spA.p.c1"] + lines["spC"]))
    w(os.path.join(case, "metasnv", "snpCaller", "called_SNPs.best_split_2"), "".join(lines["spA.p.c2"] + lines["sp_B_x.q.c1"]))
    w(os.path.join(case, "metasnv", "snpCaller", "indiv_called.best_split_1"), "not matched by the glob\n")

    def hap(spec, picks, extra=()):
        rows = ["\tposId\tcluster\n"]
        for k, ln in enumerate(picks):
            f = ln.split("\t")
            rows.append("%d\t%s:%s:%s:%s\t%d\n" % (k + 1, f[0], f[1], f[2], f[5].split("|")[1], 1 + k % 2))
        for k, (c, p) in enumerate(extra):
            rows.append("x%d\t%s:-:%d:A\t1\n" % (k, c, p))                     # positions that were never called
        w(os.path.join(case, "hap", spec + "_hap_positions.tab"), "".join(rows))
    pa = rnd.sample(lines["spA.p.c1"], 12) + rnd.sample(lines["spA.p.c2"], 9)
    hap("spA", pa + pa[:3], extra=[("spA.p.c1", 4999), ("nosuch", 5)])            # duplicate rows too
    hap("sp_B_x", rnd.sample(lines["sp_B_x.q.c1"], 15) + pa[:4])                   # shares four positions with spA
    hap("spC", rnd.sample(lines["spC"], 5))
    w(os.path.join(case, "hap", "notes.txt"), "ignored\n")
    hap_order = glob.glob(os.path.join(case, "hap") + "/*hap_positions.tab")
    snp_order = glob.glob(os.path.join(case, "metasnv") + "/snpCaller/called_SNPs*")
    subprocess.check_call([sys.executable, os.path.join(REF, "getGenotypingSNVSubset.py"), os.path.join(case, "hap"), os.path.join(case, "metasnv")],
                          stdout=subprocess.DEVNULL)
    # the reference walks both globs in directory order: keep the order it saw next to its outputs
    json.dump({"hap": [os.path.basename(p) for p in hap_order], "snp": [os.path.basename(p) for p in snp_order]},
              open(os.path.join(case, "glob_order.json"), "w"), indent=1)
    os.makedirs(os.path.join(case, "expected"))
    for p in glob.glob(os.path.join(case, "hap", "*.pos")):
        shutil.move(p, os.path.join(case, "expected", os.path.basename(p)))
    # ---- convertSNVtoAlleleFreq on every .pos at two depth cutoffs
    for p in sorted(glob.glob(os.path.join(case, "expected", "*.pos"))):
        for md in (5, 1):
            subprocess.check_call([sys.executable, os.path.join(REF, "convertSNVtoAlleleFreq.py"), p, str(md)])
            shutil.move(p + ".freq", p + ".minDepth%d.freq" % md)
    # ---- case "one": a single called_SNPs file (no directory-order dependence): the mirror's main() end to end
    case1 = os.path.join(OUT, "one")
    w(os.path.join(case1, "metasnv", "snpCaller", "called_SNPs"), "".join(lines["spA.p.c1"] + lines["spA.p.c2"] + lines["spC"]))
    shutil.copytree(os.path.join(case, "hap"), os.path.join(case1, "hap"))
    os.remove(os.path.join(case1, "hap", "sp_B_x_hap_positions.tab"))
    subprocess.check_call([sys.executable, os.path.join(REF, "getGenotypingSNVSubset.py"), os.path.join(case1, "hap"), os.path.join(case1, "metasnv")],
                          stdout=subprocess.DEVNULL)
    os.makedirs(os.path.join(case1, "expected"))
    for p in glob.glob(os.path.join(case1, "hap", "*.pos")):
        shutil.move(p, os.path.join(case1, "expected", os.path.basename(p)))


if __name__ == "__main__":
    main()
