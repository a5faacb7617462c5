"""Oracle mpileup restatement against hand-derived known answers (samtools text semantics,
SURVEY.md Appendix C).  samtools itself is not available here: parity unpinned."""
import numpy as np

import bamtools as bt
import orc

REF = "ACGTACGTACGTACGTACGT"


def _sample():
    return bt.records(
        bt.make_record(0, 2, "5M", "GTACG"),
        bt.make_record(0, 4, "3M1I2M", "ACGTTA", flag=16),
        bt.make_record(0, 5, "2M2D2M", "TGCG", qual=[40, 10, 40, 40]),
    )


def test_text_of_three_reads():
    txt = orc.mpileup_text(["c1"], [20], [REF], [_sample()])
    assert txt.split("\n") == [
        "c1\t3\tG\t1\t^].\tI",
        "c1\t4\tT\t1\t.\tI",
        "c1\t5\tA\t2\t.^],\tII",
        "c1\t6\tC\t3\t.,^]T\tIII",
        "c1\t7\tG\t2\t.$,+1t\tII",          # third read's base has BQ 10 < 13: dropped with its -2TA suffix
        "c1\t8\tT\t2\t,*\tII",
        "c1\t9\tA\t2\t,$*\tII",
        "c1\t10\tC\t1\t.\tI",
        "c1\t11\tG\t1\t.$\tI",
        "",
    ]


def test_read_filters_and_empty_sample_columns():
    s1 = bt.records(
        bt.make_record(0, 0, "4M", "ACGT", flag=0x400),            # duplicate
        bt.make_record(0, 0, "4M", "ACGT", flag=0x100),            # secondary
        bt.make_record(0, 0, "4M", "ACGT", flag=0x200),            # QC fail
        bt.make_record(0, 0, "4M", "ACGT", flag=0x1),              # paired, not proper: orphan
        bt.make_record(0, 0, "4M", "ACGT", flag=0x4),              # unmapped
        bt.make_record(0, 1, "2M", "CG", flag=0x800, mapq=0),      # supplementary + mapq 0 are kept
        bt.make_record(0, 1, "2M", "CC", flag=0x3),                # proper pair kept
    )
    s2 = np.zeros(0, np.uint8)
    txt = orc.mpileup_text(["c1"], [20], [REF], [s1, s2])
    assert txt == ("c1\t2\tC\t2\t^!.^].\tII\t0\t*\t*\n"
                   "c1\t3\tG\t2\t.$C$\tII\t0\t*\t*\n")


def test_bed_excludes_first_position_of_each_contig():
    # metaSNV writes `name\t1\tLEN` (metaSNV.py:92); as BED that is [1, LEN): position 1 is never printed
    s = bt.records(bt.make_record(0, 0, "3M", "ACG"))
    full = orc.mpileup_text(["c1"], [20], [REF], [s])
    bed = orc.mpileup_text(["c1"], [20], [REF], [s], bed=[(0, 1, 20)])
    assert full.count("\n") == 3 and bed.count("\n") == 2
    assert bed.startswith("c1\t2\tC")


def test_lowercase_and_N_reference_and_read_N():
    ref = "acgNNcgt"
    s = bt.records(bt.make_record(0, 0, "8M", "ACNNACGT"))
    txt = orc.mpileup_text(["c1"], [8], [ref], [s])
    cols = [l.split("\t") for l in txt.strip().split("\n")]
    assert [c[2] for c in cols] == list(ref)                      # reference column keeps the FASTA's case
    bases = [c[4] for c in cols]
    # read N vs ref g -> 'N' (ignored by snpCall); read N vs ref N -> match; read A vs ref N -> 'A'
    assert bases[0] == "^]." and bases[2] == "N" and bases[3] == "." and bases[4] == "A" and bases[7] == ".$"


# ---------------------------------------------------------------------------------- overlapping mates
# mpileup without -x (metaSNV.py:160-165): htslib edits the qualities of proper-pair mates where both have an aligned base
# (sam.c tweak_overlap_quality).  Hand-derived: agreeing bases -> min(200, qa + qb) on the mate pushed first, 0 on the
# other; disagreeing -> 0.8 q (truncated) on the higher-quality one (tie: the first), 0 on the other; -Q 13 then drops
# the zeroed element together with its ^ / $ markers.
def _pair(a_seq, a_q, b_seq, b_q, a_pos=2, b_pos=4, a_cig="6M", b_cig="6M", tlen=8, a_flag=99, b_flag=147, b_name="p"):
    return bt.records(bt.make_record(0, a_pos, a_cig, a_seq, qual=a_q, flag=a_flag, name="p", mtid=0, mpos=b_pos, tlen=tlen),
                      bt.make_record(0, b_pos, b_cig, b_seq, qual=b_q, flag=b_flag, name=b_name, mtid=0, mpos=a_pos, tlen=-tlen))


def _lines(sample, **mp):
    return orc.mpileup_text(["c1"], [20], [REF], [sample], mp=mp or None).split("\n")[:-1]


def test_overlapping_mates_agreeing_bases_count_once_with_summed_quality():
    s = _pair("GTACGT", [30] * 6, "ACGTAC", [20] * 6)
    assert _lines(s) == ["c1\t3\tG\t1\t^].\t?", "c1\t4\tT\t1\t.\t?",
                         "c1\t5\tA\t1\t.\tS", "c1\t6\tC\t1\t.\tS", "c1\t7\tG\t1\t.\tS", "c1\t8\tT\t1\t.$\tS",      # 30 + 20 = 50 -> 'S'; the mate's ^] went with it
                         "c1\t9\tA\t1\t,\t5", "c1\t10\tC\t1\t,$\t5"]
    # -x switches the tweak off: both mates are printed
    assert _lines(s, ignore_overlaps=1)[2:6] == ["c1\t5\tA\t2\t.^],\t?5", "c1\t6\tC\t2\t.,\t?5", "c1\t7\tG\t2\t.,\t?5", "c1\t8\tT\t2\t.$,\t?5"]


def test_overlapping_mates_disagreeing_bases_keep_the_better_one_at_80_percent():
    # reference position 6 (C): first mate reads A (q30), second mate C (q35) -> second keeps int(0.8 * 35) = 28 -> '='
    assert _lines(_pair("GTAAGT", [30] * 6, "ACGTAC", [35] * 6))[3] == "c1\t6\tC\t1\t,\t="
    # tie: the mate pushed first keeps int(0.8 * 30) = 24 -> '9'
    assert _lines(_pair("GTAAGT", [30] * 6, "ACGTAC", [30] * 6))[3] == "c1\t6\tC\t1\tA\t9"
    # 0.8 * 15 = 12 < 13: a disagreeing pair of two q15 bases disappears altogether
    assert _lines(_pair("GTAAGT", [15] * 6, "ACGTAC", [15] * 6))[3] == "c1\t6\tC\t0\t*\t*"


def test_overlapping_mates_quality_cap_and_unpaired_reads():
    assert _lines(_pair("GTACGT", [120] * 6, "ACGTAC", [110] * 6))[2] == "c1\t5\tA\t1\t.\t~"          # min(200, 230) prints as '~'
    both = ["c1\t5\tA\t2\t.^],\t?5"]
    assert _lines(_pair("GTACGT", [30] * 6, "ACGTAC", [20] * 6, b_name="other"))[2:3] == both        # different templates
    assert _lines(_pair("GTACGT", [30] * 6, "ACGTAC", [20] * 6, a_flag=0, b_flag=16))[2:3] == both   # not flagged as a proper pair
    assert _lines(_pair("GTACGT", [30] * 6, "ACGTAC", [20] * 6, a_flag=99 | 8))[2:3] == both          # first mate says "mate unmapped": it never waits


def test_overlapping_mates_first_base_behind_a_gap_of_the_second_mate_is_left_alone():
    # first mate 10M at 1..10; second mate 2M2D4M from 3: htslib's shared cursor skips the second mate's first base behind
    # its deletion (position 7), so both mates are counted there; a gap in the FIRST mate has no such effect
    a = _pair("ACGTACGTAC", [30] * 10, "GTGTAC", [20] * 6, a_pos=0, b_pos=2, a_cig="10M", b_cig="2M2D4M", tlen=10)
    got = _lines(a)
    assert got[2:4] == ["c1\t3\tG\t1\t.\tS", "c1\t4\tT\t1\t.\tS"]
    assert got[6] == "c1\t7\tG\t2\t.,\t?5" and got[7:] == ["c1\t8\tT\t1\t.\tS", "c1\t9\tA\t1\t.\tS", "c1\t10\tC\t1\t.$\tS"]
    b = _pair("ACGTGTAC", [30] * 8, "GTACGTAC", [20] * 8, a_pos=0, b_pos=2, a_cig="4M2D4M", b_cig="8M", tlen=10)
    assert _lines(b)[6] == "c1\t7\tG\t1\t.\tS"


def test_cigar_in_the_cg_field_reads_like_the_plain_record():
    """htslib puts a CIGAR that travels in CG:B,I behind the placeholder `<l_seq>S<ref_len>N` back when it reads the record (sam.c
    bam_tag2cigar [EXT]): mpileup text and qaCompute's files of the two forms are the same; auxiliary fields around it change nothing."""
    import bamtools as bt
    import orc
    ref = ("ACGTTGCAAGGCTTAACCGGTTAACGTAGCTAGCTAGGATCCGATTACAGATTACAGGCATTACG" * 8)[:480]
    reads = [(10, "20M2D30M", ref[10:30] + ref[32:62]), (12, "5S40M3I10M", "NNNNN" + ref[12:52] + "GGG" + ref[52:62]), (40, "30M1X29M", ref[40:100])]
    forms = []
    for cg in (False, True):
        recs = [bt.make_record(0, pos, cig, seq, name="r%d" % k, aux=bt.aux_fields(nm=1, md="50", score=47), cg_form=cg) for k, (pos, cig, seq) in enumerate(reads)]
        s = bt.records(*recs)
        forms.append((orc.mpileup_text(["c"], [len(ref)], [ref], [s]), orc.qacompute(["c"], [len(ref)], s)))
    assert forms[0] == forms[1]
    assert forms[0][0].count("\n") > 80
    # a CG field that holds FEWER operations than the placeholder is not a real CIGAR: the record stays what it says (all clipped, ref-skip)
    odd = bt.make_record(0, 5, "30S30N", ref[5:35], name="odd", aux=b"CGBI" + __import__("struct").pack("<II", 1, 30 << 4))
    assert orc.qacompute(["c"], [len(ref)], bt.records(odd))[0] == orc.qacompute(["c"], [len(ref)], bt.records(bt.make_record(0, 5, "30S30N", ref[5:35], name="odd")))[0]


def test_token_cap_of_the_oracle_is_the_references_unless_a_test_shortens_it():
    """oracle/orc.h: orc_snpcall_opts.token_cap = 0 means snpCall's own 10000-character token (call_vC.cpp:482); the parity sweep shortens it so
    that shallow random pileups reach the cut (tests/fuzz_parity.py).  Same answers for 0 and 10000; a token shorter than the base strings
    drops the bases behind it, so some calls lose coverage or vanish."""
    from metasnv_amd import core
    sp = core.synth_params(n_species=1, contig_len=1500, n_samples=3, mean_cov=40.0, snv_density=0.02, seed=4711)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    kw = dict(min_coverage=2, calling_threshold=2)
    a = orc.call(syn.names, syn.lengths, syn.seqs, samples, sc=kw)
    b = orc.call(syn.names, syn.lengths, syn.seqs, samples, sc=dict(kw, token_cap=10000))
    c = orc.call(syn.names, syn.lengths, syn.seqs, samples, sc=dict(kw, token_cap=25))
    assert a[:2] == b[:2] and a[0].count("\n") > 10
    assert c[:2] != a[:2]
    cov = lambda text: sum(int(x) for line in text.splitlines() for x in line.split("\t")[4].split("|"))
    assert cov(c[0]) < cov(a[0])
