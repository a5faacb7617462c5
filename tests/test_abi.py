"""The C-ABI library loads, exports every symbol include/msnv.h declares, and refuses to
compute without a GPU (no CPU fallback anywhere in the product path)."""
import ctypes as C
import os
import re

import pytest

from metasnv_amd import _lib, core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "msnv.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(msnv_[a-z0-9_]+)\s*\(", hdr))
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(_lib.lib, name)
    assert _lib.lib.msnv_abi_version() == 3


def test_struct_layouts_match_the_header():
    assert C.sizeof(_lib.Site) == 32 and C.sizeof(_lib.SiteSample) == 10
    assert C.sizeof(_lib.Params) == 56
    p = core.default_params()
    assert (p.min_coverage, p.calling_threshold, p.min_fraction, p.min_baseq, p.flag_filter) == (4, 4, 0.01, 13, 0x704)
    assert (p.max_depth, p.drop_first_line, p.cov_max, p.cov_min_mapq) == (8000, 1, 10, 1)


@pytest.mark.skipif(core.device_count() > 0, reason="GPU present: the no-device failure path is not reachable")
def test_no_device_means_loud_failure_not_fallback():
    with pytest.raises(_lib.MsnvError) as e:
        core.Context(0)
    assert e.value.code == _lib.ENODEV
    assert "no CPU fallback" in str(e.value)


@pytest.mark.skipif(core.device_count() > 0, reason="GPU present: the no-device failure path is not reachable")
def test_drivers_refuse_to_run_without_a_gpu(tmp_path, golden_dir):
    """metaSNV_Filtering / metaSNV_DistDiv mirrors: FILTER I and argument handling are host logic, everything that
    computes needs the device -- they stop with an error instead of computing on the CPU."""
    import shutil
    from metasnv_amd import filtering, distdiv
    proj = str(tmp_path / "proj")
    shutil.copytree(os.path.join(golden_dir, "python_callers", "filtering2", "proj"), proj)
    with pytest.raises(SystemExit) as e:
        filtering.main([proj, "-m", "2", "-d", "1", "-b", "10"])
    assert "no CPU fallback" in str(e.value)
    dd = str(tmp_path / "dd" / "proj")                        # the table names derive from the directory name
    shutil.copytree(os.path.join(golden_dir, "python_callers", "distdiv", "proj"), dd)
    with pytest.raises(SystemExit) as e:
        distdiv.main(["--filt", os.path.join(dd, "filtered", "pop"), "--dist"])
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(SystemExit) as e:                       # options that are not built are refused, not skipped
        distdiv.main(["--filt", os.path.join(dd, "filtered", "pop"), "--div"])
    assert "not built" in str(e.value)


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under metasnv_amd/ may reference it
    for dp, _, files in os.walk(os.path.join(ROOT, "metasnv_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liborc" not in txt and "oracle/" not in txt and "import orc" not in txt, os.path.join(dp, f)


def test_drop_in_executables_exist_and_print_usage():
    import subprocess
    from metasnv_amd import _lib
    tools = os.path.join(os.path.dirname(_lib.LIB_PATH), "tools")
    for exe in ("msnv_qacompute", "msnv_snpcall"):
        r = subprocess.run([os.path.join(tools, exe)], capture_output=True, text=True)
        assert r.returncode == 1 and "Usage" in r.stderr       # qaCompute.cpp:356-359: wrong argument count -> usage, exit 1


def test_parameters_are_validated_when_a_dataset_is_created():
    """Per-sample counts are 16-bit on the device: a depth cap outside [1, 65535] is refused, and so are histogram cutoffs the
    device does not hold and negative thresholds (host-only dataset: no GPU needed to be told so)."""
    for bad in (dict(max_depth=0), dict(max_depth=70000), dict(cov_max=16), dict(cov_max=0), dict(min_coverage=-1), dict(min_fraction=-0.5)):
        with pytest.raises(_lib.MsnvError) as e:
            core.Dataset(None, ["c"], [10], None, core.default_params(**bad))
        assert e.value.code == _lib.EINVAL, bad
    core.Dataset(None, ["c"], [10], None, core.default_params(max_depth=65535, cov_max=15)).close()
    with pytest.raises(_lib.MsnvError) as e:                       # snpCall -t 0 prints every allele of every covered position: not supported
        core.Dataset(None, ["c"], [10], None, core.default_params(calling_threshold=0))
    assert e.value.code == _lib.EDOMAIN


def test_integration_md_stub_structs_match_the_bindings():
    """The ctypes stub INTEGRATION.md shows a maintainer declares the same struct layouts as the library's own bindings."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    ns = {"C": C}
    classes = re.findall(r"^class (_\w+)\(C\.Structure\):[^\n]*\n((?:    .*\n)+)", text, re.M)
    assert {c for c, _ in classes} >= {"_Params", "_CovArgs", "_CallArgs", "_MpileupArgs"}
    for name, body in classes:
        exec("class %s(C.Structure):\n%s" % (name, body), ns)
    from metasnv_amd import _lib
    for stub, real in (("_Params", _lib.Params), ("_CovArgs", _lib.CovArgs), ("_CallArgs", _lib.CallArgs), ("_MpileupArgs", _lib.MpileupArgs)):
        assert C.sizeof(ns[stub]) == C.sizeof(real), stub
        assert [f[0] for f in ns[stub]._fields_] == [f[0] for f in real._fields_], stub

