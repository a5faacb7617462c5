"""Shared helpers of the parity tests: run the HIP path (through the C ABI) and the oracle on
the same inputs and return both texts."""
import os
import tempfile

import numpy as np

import orc
from metasnv_amd import core


def run_product(names, lengths, seqs, samples, bed=None, params=None, ann=None, fasta=None, device=0, return_ds=False):
    ctx = core.Context(device)
    ds = core.Dataset(ctx, names, lengths, seqs, params)
    if bed:
        ds.set_bed(bed)
    for s in samples:
        ds.add_sample_records(s)
    info = ds.finalize()
    st = ds.run()
    info = dict(info, pack_stats=ds.pack_stats())        # (which route the per-read stage took: the tests of the depth cap / token limit look)
    with tempfile.TemporaryDirectory() as td:
        pp, ip = os.path.join(td, "called_SNPs"), os.path.join(td, "indiv_called")
        ds.write_calls(pp, ip, ann, fasta)
        pop, ind = open(pp).read(), open(ip).read()
    if return_ds:
        return pop, ind, info, st, ds, ctx
    ds.close(); ctx.close()
    return pop, ind, info, st


def run_oracle(names, lengths, seqs, samples, bed=None, params=None, ann=None, fasta=None):
    p = params or core.default_params()
    mp = dict(min_baseq=p.min_baseq, flag_filter=p.flag_filter, count_orphans=p.count_orphans, max_depth=p.max_depth, min_mapq=p.min_mapq,
              ignore_overlaps=p.ignore_overlaps)
    sc = dict(min_coverage=p.min_coverage, calling_threshold=p.calling_threshold, calling_min_fraction=p.min_fraction)
    if 0 < p.token_limit < 10000:
        sc["token_cap"] = p.token_limit          # (a shorter token buffer than snpCall's: oracle/orc.h, the fuzz sweep)
    elif p.token_limit != 10000:
        raise NotImplementedError("the reference's token holds 10000 characters")
    pop, ind, nl, nb = orc.call(names, lengths, seqs, samples, bed=bed, fasta=fasta if ann else None, genes=ann, mp=mp, sc=sc)
    if not p.drop_first_line:
        raise NotImplementedError("the reference always drops the first line")
    return pop, ind, nl, nb


def synth_case(**kw):
    sp = core.synth_params(**kw)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    return syn, samples


def first_diff(a, b):
    la, lb = a.split("\n"), b.split("\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got      %s\n  expected %s" % (i + 1, x[:300], y[:300])
    return "line counts differ: got %d expected %d" % (len(la), len(lb))
