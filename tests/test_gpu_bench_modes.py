"""bench.py's strong-scaling mode: ONE fixed cohort through the product's N-rank path (parallel.resident_project_run --
decode sharding, owners by length x coverage, all-to-all of records, one dataset per rank, gather to rank 0), rehearsed with
two ranks sharing this GPU (gloo).  Reference counterpart: the split pool, metaSNV.py:196-215 + createOptimumSplit.py:46-62."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


SHAPE = ["--workload", "config3", "--species", "8", "--contig-len", "40000", "--samples", "24", "--steps", "3", "--warmup", "1"]


def test_strong_mode_shards_one_cohort_and_keeps_the_calls():
    one = _bench(["--gpus", "1"] + SHAPE)
    two = _bench(["--gpus", "2", "--dist-backend", "gloo"] + SHAPE)
    for line, n in ((one, 1), (two, 2)):
        assert line["scaling"] == "strong" and line["n_gpus"] == n and line["value"] > 0
        assert len(line["config"]["pileup_bases_per_rank"]) == n and sum(line["config"]["pileup_bases_per_rank"]) == line["config"]["pileup_bases_total"]
        assert line["roofline"]["frac"] > 0 and len(line["roofline"]["achieved_per_rank"]) == n
        assert line["gather"]["bytes_received_by_rank0"] > 0
    # the same cohort whatever the number of ranks: same bases, same called lines, same sites and cells
    assert two["config"]["pileup_bases_total"] == one["config"]["pileup_bases_total"]
    assert sum(two["config"]["called_SNPs_lines_per_rank"]) == one["config"]["called_SNPs_lines_per_rank"][0] > 10
    assert two["gather"]["sites_total"] == one["gather"]["sites_total"] and two["gather"]["cells_total"] == one["gather"]["cells_total"]
    assert sum(two["config"]["positions_per_rank"]) == one["config"]["positions_per_rank"][0]
    assert all(b > 0 for b in two["config"]["pileup_bases_per_rank"]) and two["imbalance_max_over_mean"] < 1.5
    # every stream was "decoded" by one rank
    assert sum(two["exchange"]["record_bytes_decoded_per_rank"]) == one["exchange"]["record_bytes_decoded_per_rank"][0]
    assert max(two["exchange"]["record_bytes_decoded_per_rank"]) < 0.75 * sum(two["exchange"]["record_bytes_decoded_per_rank"])
    # (a cohort in which every sample covers every species: the cell form is no smaller than sites x samples here; the sparse case is
    # tests/test_parallel.py::test_cell_form_of_a_sparse_cohort...)
    assert two["gather"]["bytes_received_by_rank0"] > 0 and two["gather"]["dense_form_would_be_bytes"] == two["gather"]["sites_total"] * 24 * 10


def test_one_rank_strong_rate_is_the_weak_rate():
    """N = 1: the strong mode is the weak mode's kernels on the same columns (same bases, same calls; rates within box noise)."""
    args = ["--gpus", "1", "--workload", "testdata", "--samples", "48", "--contig-len", "100000", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-annotation", "--no-overlap-extra"]
    weak = _bench(args + ["--mode", "weak"])
    strong = _bench(args + ["--mode", "strong"])
    assert strong["config"]["pileup_bases_total"] == weak["config"]["pileup_bases_per_gpu"]
    assert strong["config"]["called_SNPs_lines_per_rank"] == weak["config"]["called_SNPs_lines_per_rank"]
    assert abs(strong["roofline"]["kernel_ms_avg"] / weak["roofline"]["kernel_ms_avg"] - 1.0) < 0.15
    assert abs(strong["value"] / weak["value"] - 1.0) < 0.25


def test_two_ranks_with_the_plain_argv_print_the_strong_line_and_carry_the_weak_one():
    """N > 1 without --mode: the headline is the product's N-rank path on one cohort (scaling: strong), the weak replica line rides along;
    --mode weak keeps the weak line as the headline with the strong block inside."""
    common = ["--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1", "--samples", "24", "--contig-len", "60000", "--no-cpu-baseline", "--no-annotation",
              "--no-overlap-extra", "--strong-extra-shape", "8,40000"]
    line = _bench(common)
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["value"] > 0 and line["metric"].startswith("pileup Gbases/s")
    assert len(line["config"]["pileup_bases_per_rank"]) == 2
    w = line["weak_replicas"]
    assert w["scaling"] == "weak" and w["value"] > 0 and w["roofline"]["frac"] > 0
    line = _bench(common + ["--mode", "weak"])
    assert line["scaling"] == "weak" and line["n_gpus"] == 2
    st = line["strong_scaling"]
    assert "error" not in st, st
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and len(st["config"]["pileup_bases_per_rank"]) == 2 and st["value"] > 0
