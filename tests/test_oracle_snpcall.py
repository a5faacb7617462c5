"""Oracle (CPU restatement of call_vC.cpp) against the known-answer vectors of SURVEY.md Appendix E.

Provenance of tests/golden/snpcall_E/*: recorded during the survey from the reference source
compiled against a test-only stand-in for boost::icl (boost is absent from the image).  Under
this round's rules such a build does not count as a reference run, so the snpCall stage is
declared "parity unpinned" (oracle/orc.h); the vectors are kept as known answers.
"""
import os

import pytest

import orc


def _read(p):
    with open(p) as f:
        return f.read()


@pytest.mark.parametrize("case,fasta,genes", [
    ("E1", None, None),
    ("E2", "E2.ref.fa", "E2.annotation.tsv"),
    ("E3", None, None),
])
def test_survey_vectors(golden_dir, case, fasta, genes):
    d = os.path.join(golden_dir, "snpcall_E")
    rc, pop, ind, err = orc.snpcall_text(_read(os.path.join(d, case + ".mpileup")),
                                         fasta=os.path.join(d, fasta) if fasta else None,
                                         genes=os.path.join(d, genes) if genes else None)
    assert rc == 0, err
    assert pop == _read(os.path.join(d, case + ".called_SNPs"))
    assert ind == _read(os.path.join(d, case + ".indiv_called"))


@pytest.mark.parametrize("case", ["E4_refskip", "E5_iupac"])
def test_reference_crash_inputs_are_domain_errors(golden_dir, case):
    # '>' (CIGAR N) and IUPAC letters make the reference write through an empty vector (SIGSEGV)
    rc, pop, ind, err = orc.snpcall_text(_read(os.path.join(golden_dir, "snpcall_E", case + ".mpileup")))
    assert rc == orc.ERR_DOMAIN
    assert pop == ""


def test_first_line_is_dropped_and_counts_samples():
    mp = ("c\t1\tA\t4\tTTTT\tIIII\t4\tTTTT\tIIII\n"      # dropped although it would be called
          "c\t2\tA\t4\tTTTT\tIIII\t4\t....\tIIII\n")
    rc, pop, ind, _ = orc.snpcall_text(mp)
    assert rc == 0
    assert pop == "c\t-\t2\tA\t4|4\t4|T|.|4|0\n"
    assert ind == ""


def test_population_fraction_rule_and_individual_file():
    # cov 500: T=4 < 5.0 -> not population; sample 1 has 4 -> individual.  G=6 >= 5.0 -> population
    s1 = "." * 240 + "TTTT" + "GGG"
    s2 = "," * 250 + "ggg"
    mp = "c\t1\tA\t1\t.\tI\t1\t.\tI\n" + "c\t9\tA\t%d\t%s\t%s\t%d\t%s\t%s\n" % (len(s1), s1, "I" * len(s1), len(s2), s2, "I" * len(s2))
    rc, pop, ind, _ = orc.snpcall_text(mp)
    assert rc == 0
    assert pop == "c\t-\t9\tA\t247|253\t6|G|.|3|3\n"
    assert ind == "c\t-\t9\tA\t247|253\t4|T|.|4|0\n"


def test_lowercase_reference_suppresses_its_own_allele():
    mp = "c\t1\ta\t1\t.\tI\n" "c\t2\tt\t8\tTTTTCCCC\tIIIIIIII\n" "c\t3\tT\t8\tTTTTCCCC\tIIIIIIII\n"
    rc, pop, ind, _ = orc.snpcall_text(mp)
    assert pop == "c\t-\t2\tt\t8\t4|C|.|4\n" "c\t-\t3\tT\t8\t4|C|.|4,4|T|.|4\n"


def test_indel_and_read_start_markers_are_skipped():
    # ^<mapq char> may be any printable incl. '+', '-', 'A'; +n / -n skip exactly n characters
    mp = "c\t1\tA\t1\t.\tI\n" "c\t5\tC\t6\t^+T^AT+3ACGT-2acT$^-.\tIIIIII\n"
    rc, pop, ind, _ = orc.snpcall_text(mp)
    assert rc == 0
    assert pop == "c\t-\t5\tC\t5\t4|T|.|4\n"
