"""Out-of-bounds accesses of the kernels: the parity cohorts of every kernel family with guarded device buffers (MSNV_GUARD_ALLOC=1:
kernels.hip dev_alloc maps every buffer so that it ends at the end of its mapping, unmapped addresses behind it).  In a process of
its own: a GPU memory fault kills the process that caused it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deep", ["split", "wide"])
@pytest.mark.parametrize("case", ["narrow", "short_reads_dense_layout", "deep_wide", "sparse_whole_tile", "merged_and_split", "noisy_planes", "many_sites", "paired_aux_records", "bam_files_device_inflate"])
def test_no_access_past_the_end_of_a_device_buffer(case, deep):
    if deep == "wide" and case not in ("deep_wide", "merged_and_split"):
        pytest.skip("no deep runs in this cohort")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSNV_GUARD_ALLOC="1", MSNV_GUARD_FILL="255", MSNV_DEEP=deep)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_guard_worker.py"), case], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ("ok " + case) in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
