"""Pins the oracle to the REFERENCE's own binaries wherever they exist -- and skips where they do not.

  oracle/_ref/snpCall, oracle/_ref/qaCompute   built by `make -C oracle ref` from /root/reference's unchanged sources against
                                               real boost / htslib (oracle/Makefile); absent in this image (no boost, no htslib)
  samtools                                     on PATH, or named by MSNV_SAMTOOLS; absent in this image

Every test compares the restatement (liborc.so / orc_snpcall) with the real tool on random inputs, byte for byte.  On a box
that has the libraries, `make -C oracle ref && pytest tests/test_ref_builds.py` turns "parity unpinned" into pinned without
new code.  The last test (GPU) runs the real pipe `samtools mpileup | snpCall` against the product."""
import os
import shutil
import subprocess

import pytest

import orc
from metasnv_amd import core
from parity import synth_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SNPCALL = os.path.join(ROOT, "oracle", "_ref", "snpCall")
REF_QACOMPUTE = os.path.join(ROOT, "oracle", "_ref", "qaCompute")
import reftools
SAMTOOLS = reftools.find_samtools()

need_snpcall = pytest.mark.skipif(not os.path.exists(REF_SNPCALL), reason="oracle/_ref/snpCall not built (needs real boost: make -C oracle ref BOOST_ROOT=...)")
need_qacompute = pytest.mark.skipif(not os.path.exists(REF_QACOMPUTE), reason="oracle/_ref/qaCompute not built (needs htslib: make -C oracle ref)")
need_samtools = pytest.mark.skipif(not SAMTOOLS, reason="samtools not found (PATH or MSNV_SAMTOOLS)")


def _write_project(tmp_path, syn, samples):
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    paths = []
    for i, s in enumerate(samples):
        p = str(tmp_path / ("s%04d.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, s)
        paths.append(p)
    lst = str(tmp_path / "all_samples")
    open(lst, "w").write("\n".join(paths) + "\n")
    return fa, paths, lst


def _write_ann(path, syn):
    with open(path, "w") as f:
        f.write("gene_id\texternal_id\tsequence_id\ttype\tinfo\tlength\tstart\tend\tstrand\tsc\tstop\tgc\n")
        n = 0
        for name, L in zip(syn.names, syn.lengths):
            for start, end, strand in ((5, 1205, "+"), (1000, 2500, "-"), (2600, 2600, "+"), (3000, L - 10, "-")):
                f.write("%d\tg%d\t%s\tCDS\tx\t%d\t%d\t%d\t%s\tATG\tTAG\t0.4\n" % (n, n, name, end - start + 1, start, end, strand))
                n += 1


def _ref_snpcall(text, tmp_path, fasta=None, genes=None, c=4, t=4):
    ind = str(tmp_path / "ref_indiv")
    cmd = [REF_SNPCALL, "-i", ind, "-c", str(c), "-t", str(t)]
    if fasta:
        cmd += ["-f", fasta]
    if genes:
        cmd += ["-g", genes]
    r = subprocess.run(cmd, input=text.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode(), open(ind).read()


@need_snpcall
@pytest.mark.parametrize("seed", range(6))
def test_snpcall_restatement_equals_the_reference_binary(tmp_path, seed):
    """call_vC.cpp, unchanged, against real boost, on the oracle's own mpileup text: with and without -f/-g, lower-case
    reference, thresholds, deep strings beyond the 10000-character token."""
    syn, samples = synth_case(n_species=3, contig_len=4000, n_samples=5 + seed, mean_cov=8.0 + 6 * seed, snv_density=0.03, error_rate=0.01,
                              lowercase_ref=seed % 2, frac_paired=0.5 if seed % 3 == 0 else 0.0, seed=500 + seed)
    text = orc.mpileup_text(syn.names, syn.lengths, syn.seqs, samples)
    fa, ann = str(tmp_path / "ref.fa"), str(tmp_path / "ann.tsv")
    syn.write_fasta(fa)
    _write_ann(ann, syn)
    for kw in (dict(), dict(fasta=fa, genes=ann), dict(c=2, t=1), dict(c=10, t=6, fasta=fa, genes=ann)):
        want = _ref_snpcall(text, tmp_path, **kw)
        rc, pop, ind, err = orc.snpcall_text(text, fasta=kw.get("fasta"), genes=kw.get("genes"), c=kw.get("c", 4), t=kw.get("t", 4))
        assert rc == 0 and (pop, ind) == want, kw


@need_snpcall
def test_snpcall_token_truncation_equals_the_reference_binary(tmp_path):
    from test_overlap_host import _stack
    ref, s = _stack(5)
    text = orc.mpileup_text(["c1"], [len(ref)], [ref], [s, s[:0]])
    want = _ref_snpcall(text, tmp_path, c=1, t=1)
    rc, pop, ind, err = orc.snpcall_text(text, c=1, t=1)
    assert rc == 0 and (pop, ind) == want


@need_qacompute
@pytest.mark.parametrize("seed", range(4))
def test_qacompute_restatement_equals_the_reference_binary(tmp_path, seed):
    """qaCompute.cpp, unchanged, against htslib: `qaCompute -c 10 -d -i BAM OUT` (metaSNV.py:63-65) on BAM files."""
    syn, samples = synth_case(n_species=4, contig_len=5000, n_samples=3, mean_cov=7.0 + 5 * seed, frac_absent=0.3, frac_paired=0.5 * (seed % 2), seed=700 + seed)
    fa, paths, lst = _write_project(tmp_path, syn, samples)
    for p, s in zip(paths, samples):
        if s.size == 0:
            continue
        out = p + ".cov"
        r = subprocess.run([REF_QACOMPUTE, "-c", "10", "-d", "-i", p, out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        want = orc.qacompute(syn.names, syn.lengths, s)
        assert open(out).read() == want[0] and open(out + ".detail").read() == want[1]


@need_samtools
@pytest.mark.parametrize("paired", [0.0, 0.8])
def test_mpileup_restatement_equals_samtools(tmp_path, paired):
    """`samtools mpileup -f REF [-l SPLIT] -B -b LIST` exactly as metaSNV.py:160-165 runs it, on BAM files: single-end reads and
    proper pairs with overlapping mates (the overlap handling is on, no -x).  Quality characters of deletion elements are not
    compared when pairs overlap (they depend on the engine's read-ahead; snpCall ignores '*')."""
    from test_overlap_host import _without_deletion_quals
    syn, samples = synth_case(n_species=2, contig_len=5000, n_samples=4, mean_cov=12.0, snv_density=0.02, frac_paired=paired, lowercase_ref=1, seed=900)
    fa, paths, lst = _write_project(tmp_path, syn, samples)
    bed = str(tmp_path / "best_split_0")
    open(bed, "w").write("%s\t1\t%d\n" % (syn.names[1], syn.lengths[1]))
    for extra, obed in (([], None), (["-l", bed], [(1, 1, syn.lengths[1])])):
        r = subprocess.run([SAMTOOLS, "mpileup", "-f", fa] + extra + ["-B", "-b", lst], capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        want, got = r.stdout.decode(), orc.mpileup_text(syn.names, syn.lengths, syn.seqs, samples, bed=obed)
        if paired:
            want, got = _without_deletion_quals(want), _without_deletion_quals(got)
        assert got == want


@pytest.mark.gpu
@need_samtools
@need_snpcall
def test_product_equals_the_real_pipe(tmp_path):
    """The reference pipeline itself -- samtools mpileup | snpCall -- against msnv_call on the same BAM files."""
    syn, samples = synth_case(n_species=3, contig_len=6000, n_samples=8, mean_cov=12.0, snv_density=0.03, frac_paired=0.5, seed=1000)
    fa, paths, lst = _write_project(tmp_path, syn, samples)
    mp = subprocess.run([SAMTOOLS, "mpileup", "-f", fa, "-B", "-b", lst], capture_output=True, timeout=900)
    assert mp.returncode == 0
    want = _ref_snpcall(mp.stdout.decode(), tmp_path, fasta=fa)
    from metasnv_amd import cli
    proj = str(tmp_path / "proj")
    cli.main([proj, lst, fa])
    assert open(os.path.join(proj, "snpCaller", "called_SNPs")).read() == want[0]
    assert open(os.path.join(proj, "snpCaller", "indiv_called")).read() == want[1]


@pytest.mark.gpu
@need_samtools
def test_real_mpileup_text_through_the_text_entry(tmp_path):
    """Real `samtools mpileup` text into msnv_call_from_mpileup (snpCall's own boundary): the same bytes as the product's BAM path
    on the same files, and -- when the reference snpCall could be built -- as the reference's on the same text."""
    from metasnv_amd import core, cli
    syn, samples = synth_case(n_species=3, contig_len=6000, n_samples=8, mean_cov=12.0, snv_density=0.03, frac_paired=0.5, seed=1001)
    fa, paths, lst = _write_project(tmp_path, syn, samples)
    mp = subprocess.run([SAMTOOLS, "mpileup", "-f", fa, "-B", "-b", lst], capture_output=True, timeout=900)
    assert mp.returncode == 0
    ctx = core.Context(0)
    pp, ip = str(tmp_path / "t.called"), str(tmp_path / "t.indiv")
    core.call_from_mpileup(ctx, pp, ip, text=mp.stdout)
    ctx.close()
    proj = str(tmp_path / "proj")
    cli.main([proj, lst, fa])
    assert open(pp).read() == open(os.path.join(proj, "snpCaller", "called_SNPs")).read()
    assert open(ip).read() == open(os.path.join(proj, "snpCaller", "indiv_called")).read()
    if os.path.exists(REF_SNPCALL):
        want = _ref_snpcall(mp.stdout.decode(), tmp_path, fasta=fa)
        assert (open(pp).read(), open(ip).read()) == (want[0], want[1])

