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
