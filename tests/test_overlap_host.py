"""Host stage of the product (no GPU): the qualities the pileup engine sees after the overlapping-mate tweak
(msnv_dataset_pileup_qualities, pack.cpp's closed form of htslib's cursor loop) against the oracle's literal restatement
of sam.c, differentially: oracle(original records, tweak on) must print the same text as oracle(product-edited records,
tweak off), on random proper pairs with indels, clips, odd flags and templates with three alignments."""
import random

import numpy as np

import bamtools as bt
import orc
from metasnv_amd import core


def _rand_cigar(rnd, n):
    """A CIGAR consuming n query bases: M blocks with a few I / D / N / S."""
    ops, left = [], n
    if rnd.random() < 0.15:
        k = rnd.randint(1, 5); ops.append((k, "S")); left -= k
    while left > 0:
        m = min(left, rnd.randint(3, 40))
        ops.append((m, rnd.choice("M=X") if rnd.random() < 0.1 else "M")); left -= m
        if left > 3 and rnd.random() < 0.35:
            t = rnd.choice("IDDN")
            k = rnd.randint(1, 4)
            if t == "I":
                k = min(k, left - 1); left -= k
            ops.append((k, t))
    if ops[-1][1] in "DN":
        ops.pop()
    return "".join("%d%s" % o for o in ops)


def _pairs(seed, L=600, n_pairs=150):
    rnd = random.Random(seed)
    ref = "".join(rnd.choice("ACGT") for _ in range(L))
    recs = []
    for p in range(n_pairs):
        rl = rnd.randint(20, 70)
        frag = rnd.randint(rl, 3 * rl)
        a_pos = rnd.randint(0, L - frag - 20)
        b_pos = a_pos + max(0, frag - rl) if rnd.random() < 0.9 else a_pos
        reads = []
        for who, pos in (("a", a_pos), ("b", b_pos)):
            cig = _rand_cigar(rnd, rl)
            seq = []
            rp = pos
            for n, op in bt.parse_cigar(cig):
                c = bt.CIGAR_OPS[op]
                if c in "M=X":
                    seq += [ref[rp + i] if rnd.random() > 0.15 else rnd.choice("ACGTN") for i in range(n)]
                    rp += n
                elif c in "IS":
                    seq += [rnd.choice("ACGT") for _ in range(n)]
                elif c in "DN":
                    rp += n
            qual = [rnd.choice([2, 12, 13, 15, 20, 30, 35, 40, 41, 90, 120]) for _ in range(rl)]
            reads.append((pos, cig, "".join(seq), qual))
        name = "t%d" % p
        fa, fb = 99, 147
        r = rnd.random()
        if r < 0.05:
            fa, fb = 0, 16                                           # single-end flags: never tweaked
        elif r < 0.08:
            fb = 147 | 0x400                                         # duplicate second mate: filtered, first mate keeps waiting
        elif r < 0.11:
            fa = 99 | 8                                              # "mate unmapped"
        tl = reads[1][0] + 60 - reads[0][0]
        if rnd.random() < 0.05:
            tl = 1000                                                # implausible insert size: the engine does not even look
        mpos_a = reads[1][0] if rnd.random() > 0.04 else -1          # unknown mate position still waits (PAIRED)
        recs.append((reads[0][0], bt.make_record(0, reads[0][0], reads[0][1], reads[0][2], qual=reads[0][3], flag=fa, name=name, mtid=0, mpos=mpos_a, tlen=tl)))
        recs.append((reads[1][0], bt.make_record(0, reads[1][0], reads[1][1], reads[1][2], qual=reads[1][3], flag=fb, name=name, mtid=0, mpos=reads[0][0], tlen=-tl)))
        if rnd.random() < 0.06:                                      # a third alignment of the template (supplementary)
            sp = rnd.randint(reads[0][0], reads[1][0] + 5)
            recs.append((sp, bt.make_record(0, sp, "%dM" % rl, "".join(ref[sp + i] for i in range(rl)), qual=[30] * rl, flag=99 | 0x800, name=name, mtid=0, mpos=reads[1][0], tlen=tl)))
    recs.sort(key=lambda x: x[0])                                    # stable: file order inside a position
    return ref, bt.records(*[r for _, r in recs])


def _without_deletion_quals(text):
    """mpileup text with the quality characters of '*' / '<' / '>' elements blanked.  A deletion element prints the quality
    of the base BEHIND the deletion; when that base is edited by the tweak later on, the value printed depends on how far
    the engine had read ahead at that moment.  snpCall ignores such elements (call_vC.cpp:517-521), so the comparison does too."""
    out = []
    for line in text.split("\n"):
        w = line.split("\t")
        for k in range(4, len(w), 3):
            bases, quals, i, e = w[k], list(w[k + 1]), 0, 0
            if bases == "*":
                continue
            while i < len(bases):
                c = bases[i]
                if c == "^":
                    i += 2
                    continue
                if c in "+-":
                    j = i + 1
                    while bases[j].isdigit():
                        j += 1
                    i = j + int(bases[i + 1:j])
                    continue
                if c == "$":
                    i += 1
                    continue
                if c in "*<>":
                    quals[e] = "_"
                e += 1
                i += 1
            assert e == len(quals)
            w[k + 1] = "".join(quals)
        out.append("\t".join(w))
    return "\n".join(out)


def test_product_overlap_edit_equals_the_literal_restatement_on_random_pairs():
    n_edited = 0
    for seed in range(40):
        ref, s = _pairs(seed)
        L = len(ref)
        ds = core.Dataset(None, ["c1"], [L], [ref])
        edited = ds.pileup_qualities(s)
        ds.close()
        n_edited += int((edited != s).sum())
        want = orc.mpileup_text(["c1"], [L], [ref], [s], mp=dict(min_baseq=0))
        got = orc.mpileup_text(["c1"], [L], [ref], [edited], mp=dict(min_baseq=0, ignore_overlaps=1))
        assert _without_deletion_quals(got) == _without_deletion_quals(want), seed
        assert orc.mpileup_text(["c1"], [L], [ref], [s], mp=dict(min_baseq=0, ignore_overlaps=1)) != want
    assert n_edited > 20000


def test_ignore_overlaps_parameter_leaves_qualities_alone():
    ref, s = _pairs(3)
    ds = core.Dataset(None, ["c1"], [len(ref)], [ref], core.default_params(ignore_overlaps=1))
    assert (ds.pileup_qualities(s) == s).all()
    ds.close()


def test_host_only_dataset_refuses_to_compute():
    ref, s = _pairs(1, n_pairs=5)
    ds = core.Dataset(None, ["c1"], [len(ref)], [ref])
    ds.add_sample_records(s)
    try:
        ds.finalize()
        raise AssertionError("finalize without a device context must fail")
    except core._lib.MsnvError as e:
        assert e.code == core._lib.ENODEV and "no CPU fallback" in str(e)
    ds.close()


# ---------------------------------------------------------------------------------- snpCall's 10000-character token
def _stack(seed, L=400, n_stack=5600, extra=300):
    """One sample whose base string passes 10000 characters: a stack of reads starting at ONE position (`^]` + base = 3
    characters per read start), some with insertions / deletions behind the stack position, plus ordinary reads."""
    rnd = random.Random(seed)
    ref = "".join(rnd.choice("ACGT") for _ in range(L))
    recs = []
    p0 = 100
    for i in range(n_stack):
        rl = rnd.randint(8, 40)
        seq = [ref[p0 + j] if rnd.random() > 0.3 else rnd.choice("ACGT") for j in range(rl)]
        cig = "%dM" % rl
        if rnd.random() < 0.2 and rl > 12:
            k = rnd.randint(1, 6)
            if rnd.random() < 0.5:
                cig = "%dM%dI%dM" % (k, 3, rl - k - 3)
            else:
                cig = "%dM%dD%dM" % (k, 2, rl - k)
        recs.append((p0, bt.make_record(0, p0, cig, "".join(seq), qual=[rnd.choice([5, 20, 30, 40]) for _ in range(rl)], flag=rnd.choice([0, 16]), name="s%d" % i)))
    for i in range(extra):
        pos = rnd.randint(60, 160)
        rl = rnd.randint(10, 50)
        recs.append((pos, bt.make_record(0, pos, "%dM" % rl, "".join(ref[pos + j] if rnd.random() > 0.2 else rnd.choice("ACGT") for j in range(rl)),
                                         qual=[rnd.choice([12, 13, 35]) for _ in range(rl)], flag=rnd.choice([0, 16]), name="e%d" % i)))
    recs.sort(key=lambda x: x[0])
    return ref, bt.records(*[r for _, r in recs])


def test_token_limit_edit_reproduces_the_truncated_counts():
    """call_vC.cpp:481-483 cuts every token at 10000 characters, so the bases behind the cut of a sample's base string are
    never counted.  The host stage marks exactly those bases (quality 0): the oracle -- which builds the text and cuts it
    like the reference -- gives the same called_SNPs / indiv_called for the edited records, whose strings no longer reach
    the limit, as for the original ones; and without the edit (token_limit 0) the records stay untouched."""
    for seed in range(6):
        ref, s = _stack(seed)
        L = len(ref)
        small = bt.records(bt.make_record(0, 90, "30M", ref[90:120], name="o"))
        ds = core.Dataset(None, ["c1"], [L], [ref])
        edited = ds.pileup_qualities(s)
        ds.close()
        n_cut = int((edited != s).sum())
        # independently: the base characters at offset >= 10000 of the text mpileup would print for the original records
        expect = 0
        for line in orc.mpileup_text(["c1"], [L], [ref], [s]).split("\n"):
            if not line:
                continue
            bases, i = line.split("\t")[4], 0
            while i < len(bases):
                c = bases[i]
                if c == "^":
                    i += 2
                elif c in "+-":
                    j = i + 1
                    while bases[j].isdigit():
                        j += 1
                    i = j + int(bases[i + 1:j])
                else:
                    expect += 1 if (c not in "$*<>" and i >= 10000) else 0
                    i += 1
        assert n_cut == expect > 500
        kw = dict(mp=dict(min_baseq=13), sc=dict(min_coverage=1, calling_threshold=1, calling_min_fraction=0.0))
        want = orc.call(["c1"], [L], [ref], [small, s], **kw)
        got = orc.call(["c1"], [L], [ref], [small, edited], **kw)
        assert got[:2] == want[:2], seed
        text = orc.mpileup_text(["c1"], [L], [ref], [edited])
        assert max(len(l.split("\t")[4]) for l in text.split("\n") if l) <= 10000 + 60          # + one element and deletion marks
        full = orc.mpileup_text(["c1"], [L], [ref], [s])
        assert max(len(l.split("\t")[4]) for l in full.split("\n") if l) > 10000
        ds0 = core.Dataset(None, ["c1"], [L], [ref], core.default_params(token_limit=0))
        assert (ds0.pileup_qualities(s) == s).all()
        ds0.close()


def test_token_limit_marks_do_not_depend_on_the_quality_cutoff():
    """The bases behind the token limit are the same whatever -Q is (with -Q 0 no base is too low to be printed, so MORE characters
    sit in front of the cut, never fewer marks than the count of characters says), and the seam reports them as quality 0 in
    either case; stored qualities of 128 and more behave like 127 everywhere (they pass every cutoff) and come back clamped from a
    sample whose strings reach the limit."""
    ref, s = _stack(3)
    L = len(ref)
    cut = {}
    for q in (0, 13):
        ds = core.Dataset(None, ["c1"], [L], [ref], core.default_params(min_baseq=q))
        edited = ds.pileup_qualities(s)
        ds.close()
        changed = edited != s
        assert changed.sum() > 500 and (edited[changed] == 0).all()
        cut[q] = int(changed.sum())
    assert cut[0] >= cut[13]                                    # (at -Q 13 the low-quality bases in front of the cut are not printed)
