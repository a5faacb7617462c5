"""Test helper: called_SNPs / indiv_called text -> site records (the inverse of the library's formatter)."""
import numpy as np

from metasnv_amd.core import SITE_DTYPE, SAMPLE_DTYPE

ALLELE = {"A": 0, "C": 1, "G": 2, "T": 3}


def parse_calls(pop_text, ind_text, names):
    tid_of = {n: i for i, n in enumerate(names)}
    recs = {}
    for text, is_pop in ((pop_text, True), (ind_text, False)):
        for line in text.splitlines():
            contig, gene, pos, ref, covs, entries = line.split("\t")
            key = (tid_of[contig], int(pos) - 1)
            cov = [int(x) for x in covs.split("|")]
            r = recs.setdefault(key, dict(ref=ref, cov=cov, n={}, pop=0, ind=0, tot={}))
            for e in entries.split(","):
                f = e.split("|")
                x = ALLELE[f[1]]
                r["tot"][x] = int(f[0])
                r["n"][x] = [int(v) for v in f[3:]]
                if is_pop:
                    r["pop"] |= 1 << x
                else:
                    r["ind"] |= 1 << x
    keys = sorted(recs)
    S = len(recs[keys[0]]["cov"]) if keys else 0
    sites = np.zeros(len(keys), SITE_DTYPE)
    samples = np.zeros((len(keys), max(S, 1)), SAMPLE_DTYPE)
    for i, k in enumerate(keys):
        r = recs[k]
        sites[i]["tid"], sites[i]["pos"] = k
        sites[i]["cov"] = sum(r["cov"])
        sites[i]["pop_mask"], sites[i]["ind_mask"] = r["pop"], r["ind"]
        sites[i]["refchar"] = ord(r["ref"])
        samples[i, :S]["cov"] = r["cov"]
        for x, per in r["n"].items():
            sites[i]["n"][x] = r["tot"][x]
            samples[i, :S]["n"][:, x] = per
    return sites, samples, S
