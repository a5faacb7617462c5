"""pytest configuration: `gpu` marker, oracle build, shared fixtures."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    # building the checker is not using it; the GPU box receives the prebuilt liborc.so too
    so = os.path.join(ROOT, "oracle", "liborc.so")
    src = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liborc.so"], stdout=subprocess.DEVNULL)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One block in every pytest log (the -q one the driver keeps as pytest.log too): which of the reference's own tools and
    libraries this box has -- what pins the oracle (tests/test_ref_builds.py) or leaves it "parity unpinned"."""
    try:
        import reftools
        r = reftools.report()
    except Exception as e:                                   # never cost a test run
        r = {"error": repr(e)}
    terminalreporter.write_line("")
    terminalreporter.write_line("reference tools on this box (tests/reftools.py): " + ", ".join("%s=%s" % (k, v) for k, v in r.items()))
    pinned = [k for k in ("samtools", "oracle_ref_snpCall", "oracle_ref_qaCompute") if r.get(k)]
    terminalreporter.write_line("oracle pinned against: " + (", ".join(pinned) if pinned else "nothing here -- the tests of tests/test_ref_builds.py skip (parity unpinned)"))
