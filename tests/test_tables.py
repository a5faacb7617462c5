"""Python callers of the hot path (SURVEY.md section 8 row a17 + the split planner) against golden files
produced by running the reference's own scripts (tests/golden/make_python_goldens.py)."""
import filecmp
import os
import shutil

import pytest

from metasnv_amd import tables


@pytest.fixture()
def proj(tmp_path, golden_dir):
    src = os.path.join(golden_dir, "python_callers", "proj")
    dst = str(tmp_path / "proj")
    os.makedirs(os.path.join(dst, "cov"))
    for f in os.listdir(os.path.join(src, "cov")):
        if f.endswith(".cov") or f.endswith(".detail"):
            shutil.copy(os.path.join(src, "cov", f), os.path.join(dst, "cov", f))
    shutil.copy(os.path.join(src, "bed_header"), dst)
    return src, dst


def _same(a, b):
    assert open(a).read() == open(b).read(), (a, b)


def test_summary_and_tab_files_are_byte_identical(proj):
    src, dst = proj
    for bam in ("s1.bam", "s2.bam", "a0.bam"):
        c = os.path.join(dst, "cov", bam + ".cov")
        tables.species_summary(c, c + ".detail", c + ".summary")
        _same(c + ".summary", os.path.join(src, "cov", bam + ".cov.summary"))
    tables.collapse_tables(dst)
    _same(os.path.join(dst, "proj.all_cov.tab"), os.path.join(src, "proj.all_cov.tab"))
    _same(os.path.join(dst, "proj.all_perc.tab"), os.path.join(src, "proj.all_perc.tab"))


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_split_planner_matches_createOptimumSplit(proj, n, capsys):
    src, dst = proj
    for f in ("proj.all_cov.tab", "proj.all_perc.tab"):
        shutil.copy(os.path.join(src, f), dst)
    os.makedirs(os.path.join(dst, "bestsplits"))
    tables.plan_splits(os.path.join(dst, "proj.all_cov.tab"), os.path.join(dst, "proj.all_perc.tab"),
                       os.path.join(dst, "bed_header"), n, os.path.join(dst, "bestsplits", "best_split"))
    want = os.path.join(src, "bestsplits_%d" % n)
    assert sorted(os.listdir(os.path.join(dst, "bestsplits"))) == sorted(os.listdir(want))
    for f in os.listdir(want):
        _same(os.path.join(dst, "bestsplits", f), os.path.join(want, f))


def test_contig_sharding_follows_the_same_rule():
    from metasnv_amd import parallel
    names = ["spA.p.c1", "spA.p.c2", "spB.q.c1", "spC", "spB.q.c2"]
    lengths = [1000, 500, 2000, 77, 333]
    owner = parallel.shard_contigs(names, lengths, 2)
    assert owner[0] == owner[1] and owner[2] == owner[4]          # whole species stay together
    assert owner[2] != owner[0]                                    # heaviest (spB) alone, then spA, then spC joins the lighter
    assert parallel.shard_contigs(names, lengths, 1) == [0] * 5
