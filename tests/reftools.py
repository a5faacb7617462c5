"""Where are the reference's own tools and their libraries on THIS box?  The oracle's restatements of samtools mpileup, snpCall
and qaCompute are pinned by tests/test_ref_builds.py the moment any of them exists; this module is the search both that test file
and the pytest summary use, so a box that has one of them flips the skipped tests without a code change.

  samtools   MSNV_SAMTOOLS, PATH, then the usual prefixes (conda, /usr/local, /opt)
  boost      BOOST_ROOT, then the usual include directories (boost/icl/split_interval_map.hpp: what call_vC.cpp includes)
  htslib     HTSLIB_CFLAGS / pkg-config htslib, then the usual include directories (htslib/sam.h: what qaCompute.cpp includes)
  pysam      importable (it bundles htslib and samtools' pileup engine)
  oracle/_ref/{snpCall,qaCompute}   the reference sources built by `make -C oracle ref` where /root/reference and the libraries are
"""
import glob
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIXES = ["/usr", "/usr/local", "/opt/conda", "/opt/miniconda3", "/opt/homebrew", os.path.expanduser("~/miniconda3"), os.path.expanduser("~/.local"), "/opt/rocm"]


def find_samtools():
    cand = [os.environ.get("MSNV_SAMTOOLS"), shutil.which("samtools")] + [os.path.join(p, "bin", "samtools") for p in PREFIXES]
    cand += glob.glob("/opt/*/bin/samtools") + glob.glob("/opt/conda/envs/*/bin/samtools")
    for c in cand:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


def find_boost():
    roots = [os.environ.get("BOOST_ROOT")] + PREFIXES
    for r in roots:
        if not r:
            continue
        for inc in (r, os.path.join(r, "include")):
            if os.path.isfile(os.path.join(inc, "boost", "icl", "split_interval_map.hpp")):
                return inc
    return None


def find_htslib():
    try:
        r = subprocess.run(["pkg-config", "htslib", "--cflags"], capture_output=True, text=True, timeout=20)
        if r.returncode == 0:
            return "pkg-config: " + (r.stdout.strip() or "(default include path)")
    except Exception:
        pass
    for r in PREFIXES:
        if os.path.isfile(os.path.join(r, "include", "htslib", "sam.h")):
            return os.path.join(r, "include")
    return None


def find_pysam():
    try:
        import pysam
        return getattr(pysam, "__version__", "yes")
    except Exception:
        return None


def report():
    ref = {n: os.path.exists(os.path.join(ROOT, "oracle", "_ref", n)) for n in ("snpCall", "qaCompute")}
    st = find_samtools()
    ver = None
    if st:
        try:
            ver = subprocess.run([st, "--version"], capture_output=True, text=True, timeout=20).stdout.splitlines()[0]
        except Exception:
            ver = "?"
    return {"samtools": st, "samtools_version": ver, "boost_include": find_boost(), "htslib": find_htslib(), "pysam": find_pysam(),
            "reference_sources": os.path.isdir("/root/reference/src"), "oracle_ref_snpCall": ref["snpCall"], "oracle_ref_qaCompute": ref["qaCompute"]}
