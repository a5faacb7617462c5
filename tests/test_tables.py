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


def test_float_repr_matches_python():
    """The frequencies of *.filtered.freq are printed with str(float) (metaSNV_Filtering.py:231)."""
    import ctypes as C
    import random
    from metasnv_amd._lib import lib
    rnd = random.Random(3)
    vals = [0.0, 1.0, 0.5, 1 / 3, 2 / 3, 0.1, 1e-4, 9.999e-5, 1 / 30000, 5e-324, 1.7976931348623157e308, 1e16, 9999999999999998.0,
            1e15, 123456789.125, 1e22, 1e-7, 0.30000000000000004, 100.0, 7.0, 2.5e-5, 1234.5e10, -0.0, -1.5, 4 / 200003]
    vals += [rnd.randint(0, c) / c for c in [rnd.randint(1, 300000) for _ in range(3000)]]
    vals += [rnd.random() * 10 ** rnd.randint(-12, 20) for _ in range(2000)]
    buf = C.create_string_buffer(64)
    for v in vals:
        n = lib.msnv_format_float(v, buf, 64)
        assert n > 0 and buf.value.decode() == repr(v), (v, buf.value, repr(v))
    assert lib.msnv_format_float(0.1, buf, 2) == -1


def test_relevant_taxa_matches_reference_answers(golden_dir):
    import json
    from metasnv_amd import filtering
    g = os.path.join(golden_dir, "python_callers", "filtering2")
    for a in json.load(open(os.path.join(g, "relevant_taxa.json"))):
        got = filtering.relevant_taxa(os.path.join(g, "proj", "proj.all_cov.tab"), os.path.join(g, "proj", "proj.all_perc.tab"), a["b"], a["d"], a["m"])
        assert got["SoI"] == a["SoI"], a


def test_float_parse_matches_pandas_default_converter():
    """computeDist reads the frequencies with pd.read_table (metaSNV_DistDiv.py:116), whose default converter is not
    correctly rounded; the library reproduces it (it decides the last digit of the printed distances)."""
    import ctypes as C
    import io
    import random
    pd = pytest.importorskip("pandas")
    from metasnv_amd._lib import lib
    rnd = random.Random(8)
    toks = ["0.0", "1.0", "0.5", "0.42857142857142855", "0.04748407749508153", "4.999925001124983e-06", "1e-05", "0.1", "0.30000000000000004",
            "123456789.12345678", "9.999850002249966e-06", "0.9999999999999999", "1e-10", "2.5e-05", "7", "12345678901234567890", "0.000123"]
    toks += [repr(rnd.randint(0, c) / c) for c in [rnd.randint(1, 300000) for _ in range(4000)]]
    toks += [repr(rnd.random() * 10 ** rnd.randint(-9, 6)) for _ in range(2000)]
    want = pd.read_csv(io.StringIO("x\n" + "\n".join(toks) + "\n"), dtype=float)["x"].values
    v = C.c_double()
    off = 0
    for tok, w in zip(toks, want):
        assert lib.msnv_parse_float(tok.encode(), C.byref(v)) == 0, tok
        assert v.value == w, (tok, repr(v.value), repr(w))
        off += v.value != float(tok)
    assert off > 100                                # the converter really is inexact: this is not strtod
    assert lib.msnv_parse_float(b"abc", C.byref(v)) != 0 and lib.msnv_parse_float(b"1.5x", C.byref(v)) != 0
