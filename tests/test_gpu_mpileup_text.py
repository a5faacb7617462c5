"""snpCall on its own input -- mpileup TEXT -- on the device (msnv_call_from_mpileup, csrc/textcall.hip).

The checker is the oracle's restatement of call_vC.cpp (oracle/orc_snpcall.c, run as a process on the same text), the known-answer
vectors of SURVEY.md Appendix E (tests/golden/snpcall_E: outputs of the reference source itself, see tests/test_oracle_snpcall.py
for their provenance), and the product's BAM path on the records the text was rendered from."""
import os
import random
import subprocess

import pytest

import orc
from metasnv_amd import core, _lib
from parity import run_product, synth_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "snpcall_E")


def _read(p):
    with open(p) as f:
        return f.read()


def product(text, tmp_path, fasta=None, ann=None, indiv=True, **kw):
    """(called_SNPs, indiv_called, stats) of the device path on `text`; kw: c, t, p like snpCall's options."""
    ctx = core.Context(0)
    pp, ip = str(tmp_path / "called"), str(tmp_path / "indiv")
    for f in (pp, ip):
        if os.path.exists(f):
            os.remove(f)
    p = core.default_params(min_coverage=kw.get("c", 4), calling_threshold=kw.get("t", 4), min_fraction=kw.get("p", 0.01))
    try:
        st = core.call_from_mpileup(ctx, pp, ip if indiv else None, text=text, fasta=fasta, ann=ann, params=p)
    finally:
        ctx.close()
    return _read(pp), (_read(ip) if indiv else ""), st


def same_as_oracle(text, tmp_path, fasta=None, ann=None, **kw):
    rc, pop, ind, err = orc.snpcall_text(text, fasta=fasta, genes=ann, **kw)
    assert rc == 0, err
    got = product(text, tmp_path, fasta=fasta, ann=ann, **kw)
    assert got[0] == pop
    assert got[1] == ind
    return got


@pytest.mark.parametrize("case,fasta,genes", [("E1", None, None), ("E2", "E2.ref.fa", "E2.annotation.tsv"), ("E3", None, None)])
def test_survey_vectors_through_the_device(case, fasta, genes, tmp_path):
    text = _read(os.path.join(GOLD, case + ".mpileup"))
    pop, ind, st = product(text, tmp_path, fasta=os.path.join(GOLD, fasta) if fasta else None, ann=os.path.join(GOLD, genes) if genes else None)
    assert pop == _read(os.path.join(GOLD, case + ".called_SNPs"))
    assert ind == _read(os.path.join(GOLD, case + ".indiv_called"))
    assert st["lines"] == text.count("\n") and st["called_lines"] == pop.count("\n")


@pytest.mark.parametrize("case", ["E4_refskip", "E5_iupac"])
def test_inputs_the_reference_crashes_on_are_domain_errors(case, tmp_path):
    text = _read(os.path.join(GOLD, case + ".mpileup"))
    with pytest.raises(_lib.MsnvError) as e:
        product(text, tmp_path)
    assert e.value.code == _lib.EDOMAIN and "pileup symbol" in str(e.value)
    assert not os.path.exists(tmp_path / "called")


def test_first_line_population_fraction_and_individual_file(tmp_path):
    mp = ("c\t1\tA\t4\tTTTT\tIIII\t4\tTTTT\tIIII\n"      # dropped although it would be called (call_vC.cpp:423)
          "c\t2\tA\t4\tTTTT\tIIII\t4\t....\tIIII\n")
    pop, ind, st = product(mp, tmp_path)
    assert pop == "c\t-\t2\tA\t4|4\t4|T|.|4|0\n" and ind == "" and st["samples"] == 2
    # cov 500: T=4 < 5.0 -> not population; sample 1 has 4 -> individual.  G=6 >= 5.0 -> population
    s1 = "." * 240 + "TTTT" + "GGG"
    s2 = "," * 250 + "ggg"
    mp = "c\t1\tA\t1\t.\tI\t1\t.\tI\n" + "c\t9\tA\t%d\t%s\t%s\t%d\t%s\t%s\n" % (len(s1), s1, "I" * len(s1), len(s2), s2, "I" * len(s2))
    pop, ind, _ = product(mp, tmp_path)
    assert pop == "c\t-\t9\tA\t247|253\t6|G|.|3|3\n"
    assert ind == "c\t-\t9\tA\t247|253\t4|T|.|4|0\n"
    same_as_oracle(mp, tmp_path)
    # without -i the individual calls are dropped (call_vC.cpp:653-660)
    assert product(mp, tmp_path, indiv=False)[:2] == (pop, "")


def test_tokeniser_and_base_string_quirks(tmp_path):
    first = "c\t1\tA\t1\t.\tI\t1\t.\tI\t1\t.\tI\n"               # three samples
    lines = [
        "c\t5\ta\t6\taaaaAA\tIIIIII\t5\tTTTTT\tIIIII\t0\t*\t*\n",           # lower-case reference character: its own allele is skipped (:580)
        "c\t6\tA\t6\t^]t^]t^]T.$,$+2ACt-3acgT\tIIIIII\t4\t^t.^T,tt\tIIII\t3\tT+12AC\tIII\n",   # ^x, $, indels; an indel that runs past the token
        "c\t7\tN\t4\t  TTTT\tIIII\t4\t \tIIII\t4\tGGGG\tIIII\n",          # leading blanks are skipped by toksplit; a blank-only token
        "c\t8\tA\t4\tTTTT\t\t\t,,..\tx\ty\tgggg\tI\n",                     # empty fields shift the columns: fields 7 and 10 are the base strings
        "c\t9\tA\t4\tTTTT\tIIII\t4\tCCCC\n",                               # short line: the last token is never processed (:490)
        "c\t10\tA\t4\tTTTT\tIIII\t4\tCCCC\t\n",                            # ... nor a token whose tab ends the line
        "c\t11\tA\t4\tTTTT\tIIII\t4\tCCCC\tI\n",                           # ... this one is
        "  c2\t 12\t T\t4\tGGGG\tIIII\t4\tgggg\tIIII\t1\t+\tI\n",          # blanks in front of name, position and reference character; a lone '+'
        "c2\t13x\tAC\t8\tT-0T+T^\tI\t4\t-TTTT\tI\t4\t+9\tI\n",             # atol("13x") = 13; refchar = first character; -0, +<no digits>, '^' last
        "c2\t\t\t4\tTTTT\tI\t4\tTTTT\tI\t1\t.\tI\n",                       # empty position (atol -> 0 -> "0") and empty reference character (NUL)
        "c2\t-3\tG\t4\tTTTT\tI\t4\tTTTT\tI\t1\t.\tI\n",                     # a negative position is printed as atol read it
    ]
    text = first + "".join(lines)
    pop, ind, st = same_as_oracle(text, tmp_path)
    assert pop.count("\n") >= 8 and st["samples"] == 3
    same_as_oracle(text, tmp_path, c=1, t=1, p=0.0)
    same_as_oracle(text, tmp_path, c=9, t=2, p=0.3)
    # a last line without a newline loses its last character instead (:475)
    same_as_oracle(text[:-1], tmp_path)
    same_as_oracle(first + "c\t5\tA\t4\tTTTT\tI\t4\tTTTT\tI\t1\tT\tII", tmp_path)
    # the same text through chunks of a few hundred bytes
    os.environ["MSNV_TEXT_CHUNK"] = "150"
    try:
        assert product(text, tmp_path)[:2] == (pop, ind)
    finally:
        del os.environ["MSNV_TEXT_CHUNK"]


def test_tokens_are_cut_at_10000_characters(tmp_path):
    """toksplit keeps 10 000 characters of a token (call_vC.cpp:92-111,482): bases behind the cut do not count, and an indel
    announced in front of the cut swallows nothing behind it."""
    first = "c\t1\tA\t1\t.\tI\t1\t.\tI\n"
    deep = "." * 9990 + "TTTTTTTTTT" + "GGGGGGGG"                  # the G's are behind the cut
    deep2 = "," * 9996 + "+9ACGTACGTA" + "tttt"                     # the insertion starts before the cut and ends behind it
    text = first + "c\t2\tA\t%d\t%s\tI\t%d\t%s\tI\n" % (len(deep), deep, len(deep2), deep2)
    pop, ind, _ = same_as_oracle(text, tmp_path)
    assert pop == "" and ind == "c\t-\t2\tA\t10000|9996\t10|T|.|10|0\n"          # 10 of 19 996: below the population fraction


def test_more_samples_than_the_first_line_is_a_domain_error(tmp_path):
    text = "c\t1\tA\t1\t.\tI\nc\t2\tA\t4\tTTTT\tI\t4\tTTTT\tI\n"
    rc, _, _, _ = orc.snpcall_text(text)
    assert rc == orc.ERR_DOMAIN
    with pytest.raises(_lib.MsnvError) as e:
        product(text, tmp_path)
    assert e.value.code == _lib.EDOMAIN and "more samples" in str(e.value)
    # ... unless the extra base string is the unprocessed last token
    same_as_oracle("c\t1\tA\t1\t.\tI\nc\t2\tA\t4\tTTTT\tI\t4\tTTTT\n", tmp_path)


def test_empty_and_degenerate_inputs(tmp_path):
    for text in ("", "\n", "c\t1\tA\n", "c\t1\tA\t1\t.\tI\n", "x\ny\n", "c\t1\tA\t1\t.\tI\n\n\nc\n"):
        same_as_oracle(text, tmp_path)


@pytest.mark.parametrize("seed", [1, 2])
def test_rendered_pileups_of_synthetic_cohorts(seed, tmp_path):
    """mpileup text rendered by the oracle's mpileup restatement from synthetic BAM records (indels, clips, low qualities, N's):
    the device's text path, the oracle's snpCall and the device's BAM path agree byte for byte."""
    syn, samples = synth_case(n_species=2, contig_len=2600, n_samples=9, mean_cov=9.0, sigma_cov=0.8, snv_density=0.03, error_rate=0.01,
                              frac_absent=0.1, lowercase_ref=1, seed=4300 + seed)
    text = orc.mpileup_text(syn.names, syn.lengths, syn.seqs, samples)
    assert text.count("\n") > 3000
    for kw in (dict(), dict(c=2, t=2), dict(c=10, t=3, p=0.2)):
        pop, ind, st = same_as_oracle(text, tmp_path, **kw)
        p = core.default_params(min_coverage=kw.get("c", 4), calling_threshold=kw.get("t", 4), min_fraction=kw.get("p", 0.01))
        bam = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
        assert (bam[0], bam[1]) == (pop, ind)
    assert pop.count("\n") > 20 and st["base_chars"] > 100000


def test_annotated_rendered_pileup(tmp_path):
    syn, samples = synth_case(n_species=2, contig_len=3000, n_samples=6, mean_cov=10.0, snv_density=0.03, seed=77)
    rnd = random.Random(5)
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    ann = str(tmp_path / "genes.tsv")
    with open(ann, "w") as f:
        f.write("gene_id\texternal_id\tsequence_id\ttype\tinfo\tlength\tstart\tend\tstrand\tsc\tstop\tgc\n")
        k = 0
        for n, ln in zip(syn.names, syn.lengths):
            p = 10
            while p + 400 < ln:
                e = p + 3 * rnd.randint(30, 120)
                f.write("%d\tg%d\t%s\tCDS\tx\t%d\t%d\t%d\t%s\tATG\tTAG\t0.4\n" % (k, k, n, e - p + 1, p, e, rnd.choice("+-")))
                k += 1
                p = e - rnd.choice([-40, 5, 30])                   # gaps and overlaps (first gene in file order wins)
    text = orc.mpileup_text(syn.names, syn.lengths, syn.seqs, samples)
    pop, ind, _ = same_as_oracle(text, tmp_path, fasta=fa, ann=ann)
    assert "[" in pop and pop.count("\n") > 20


def test_process_drop_in_reads_stdin_like_snpcall(tmp_path):
    """`msnv_snpcall -i INDIV -c 4 -t 4 < mpileup > called_SNPs`: snpCall's own command line (call_vC.cpp:346-410)."""
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "tools", "msnv_snpcall")
    text = _read(os.path.join(GOLD, "E2.mpileup"))
    ip = str(tmp_path / "ind")
    r = subprocess.run([exe, "-f", os.path.join(GOLD, "E2.ref.fa"), "-g", os.path.join(GOLD, "E2.annotation.tsv"), "-i", ip, "-c", "4", "-t", "4"],
                       input=text.encode(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == _read(os.path.join(GOLD, "E2.called_SNPs"))
    assert _read(ip) == _read(os.path.join(GOLD, "E2.indiv_called"))
    # snpCall's -a and -d are flags without an argument (call_vC.cpp:346 "hdab:f:g:i:c:p:t:"): what follows them is the next option
    r2 = subprocess.run([exe, "-a", "-f", os.path.join(GOLD, "E2.ref.fa"), "-d", "-g", os.path.join(GOLD, "E2.annotation.tsv"), "-i", ip, "-c", "4", "-t", "4"],
                        input=text.encode(), capture_output=True, timeout=300)
    assert r2.returncode == 0, r2.stderr.decode()
    assert r2.stdout.decode() == r.stdout.decode()
