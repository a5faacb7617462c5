"""Multi-GPU path on CPU: world_size 2, gloo backend.  The contig shards' site records are gathered with
torch.distributed and formatted on rank 0; the result must be the oracle's text for the whole dataset."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import orc
import recparse
from metasnv_amd import core
from parity import synth_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_case():
    syn, samples = synth_case(n_species=4, contig_len=3000, n_samples=5, mean_cov=12.0, snv_density=0.03, frac_absent=0.0, seed=23)
    names = ["%s.c%d" % (n, i) for i, n in enumerate(syn.names)]       # 4 species, one contig each
    pop, ind, nl, nb = orc.call(names, syn.lengths, syn.seqs, samples, mp=dict(min_baseq=13), sc=dict(min_coverage=2, calling_threshold=2))
    return names, syn.lengths, pop, ind


def test_formatter_roundtrip_single_process(tmp_path):
    names, lengths, pop, ind = _oracle_case()
    sites, samples, S = recparse.parse_calls(pop, ind, names)
    assert len(sites) > 100
    core.write_calls_records(names, S, sites, samples, str(tmp_path / "p"), str(tmp_path / "i"))
    assert open(tmp_path / "p").read() == pop
    assert open(tmp_path / "i").read() == ind


def test_cell_form_of_a_sparse_cohort_is_an_order_of_magnitude_smaller_than_the_dense_one(tmp_path):
    """BASELINE configs[3] in small: 500 samples, every called position carried by three of them.  What a rank ships to rank 0
    (gather_sites_root) is 32 + 4 B per site + 14 B per cell; the dense [sites][samples] records of round 2 were 5 KB per site.
    Same text from either form."""
    from metasnv_amd import parallel
    S, n = 500, 400
    rng = np.random.default_rng(3)
    sites = np.zeros(n, dtype=core.SITE_DTYPE)
    sites["tid"] = np.sort(rng.integers(0, 3, n)); sites["pos"] = np.arange(n) * 7; sites["cov"] = 30; sites["n"][:, 3] = 12
    sites["pop_mask"] = 8; sites["refchar"] = ord("A"); sites["dropped"][0] = 1
    dense = np.zeros((n, S), dtype=core.SAMPLE_DTYPE)
    for i in range(n):
        for k in rng.choice(S, 3, replace=False):
            dense[i, k]["cov"] = 10; dense[i, k]["n"][3] = 4
    row_off, cs, cells = core.dense_to_cells(dense)
    assert len(cells) == 3 * n and (core.cells_to_dense(S, row_off, cs, cells) == dense).all()
    st = {}
    m = parallel.gather_sites_root(sites, row_off, cs, cells, (0, 0), stats=st)
    assert st["bytes_received"] * 10 < dense.nbytes + sites.nbytes and (m[0]["pos"] == sites["pos"]).all()
    names = ["a.1", "b.1", "c.1"]
    core.write_calls_cells(names, S, m[0], m[1], m[2], m[3], str(tmp_path / "p1"), str(tmp_path / "i1"))
    core.write_calls_records(names, S, sites, dense, str(tmp_path / "p2"), str(tmp_path / "i2"))
    assert open(tmp_path / "p1").read() == open(tmp_path / "p2").read() and open(tmp_path / "p1").read().count("\n") == n - 1   # (0, 0) is the dropped first line
    # a split is cut out of the cell form without expanding it
    res = {"names": names, "sites": m[0], "row_off": m[1], "cell_sample": m[2], "cells": m[3], "ann": None, "first_from1": np.array([7, 0, 0])}
    s2, r2, c2, e2, _ = parallel.split_view(res, ["b.1"])
    keep = (sites["tid"] == 1) & (sites["pos"] >= 1)
    assert len(s2) == keep.sum() and len(e2) == 3 * keep.sum() and (core.cells_to_dense(S, r2, c2, e2) == dense[keep]).all()


def test_two_rank_gather_reproduces_the_full_output(tmp_path):
    names, lengths, pop, ind = _oracle_case()
    work = str(tmp_path)
    open(os.path.join(work, "names"), "w").write(" ".join(names))
    open(os.path.join(work, "lengths"), "w").write(" ".join(str(x) for x in lengths))
    open(os.path.join(work, "pop"), "w").write(pop)
    open(os.path.join(work, "ind"), "w").write(ind)
    open(os.path.join(work, "first"), "w").write("0 0")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), work]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    assert open(os.path.join(work, "out_pop")).read() == pop
    assert open(os.path.join(work, "out_ind")).read() == ind
    g = open(os.path.join(work, "gfirst")).read().split()
    assert g[:2] == ["0", "0"] and g[2] == "0"          # rank-local first lines were un-dropped, (0,0) is not a called site


def _records(buf):
    """[(tid, bytes)] of a raw BAM record stream."""
    out, o = [], 0
    b = bytes(buf)
    while o < len(b):
        n = int.from_bytes(b[o:o + 4], "little")
        out.append((int.from_bytes(b[o + 4:o + 8], "little", signed=True), b[o:o + 4 + n]))
        o += 4 + n
    return out


@pytest.mark.parametrize("batch,slice_kb", [(1, None), (2, None), (2, "4"), (1, "serial")])
def test_two_rank_decode_sharding_deals_every_record_to_its_owner(tmp_path, batch, slice_kb):
    """parallel.feed_sharded over gloo: each sample is read by exactly one rank (per-rank decoded bytes ~ 1/N of the job),
    and after the all-to-all every rank holds, per sample and in order, exactly the records of the contigs it owns.  Round k + 1 is
    decoded while round k is exchanged (round 5; "serial": MSNV_FEED_OVERLAP=0, one after the other)."""
    serial = slice_kb == "serial"
    slice_kb = None if serial else slice_kb
    work = str(tmp_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_feed_worker.py"), work, str(batch)]
    # slice_kb: the byte exchange in slices of 4 KB (parallel.exchange_records walks a large round in slices: uneven parts, parts that end early)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **({"MSNV_A2A_SLICE_KB": slice_kb} if slice_kb else {}), **({"MSNV_FEED_OVERLAP": "0"} if serial else {}))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    for k in (0, 1):                                                             # (several rounds stream through: the decoder ran ahead unless told not to)
        assert int(open(os.path.join(work, "n%d" % k)).read().split()[2]) == (0 if serial else 1)
    sp = core.synth_params(n_species=5, contig_len=2500, n_samples=7, mean_cov=6.0, frac_absent=0.3, seed=77)
    syn = core.Synth(sp)
    owner = np.load(os.path.join(work, "owner.npy"))
    assert set(owner.tolist()) == {0, 1}
    dec = [np.load(os.path.join(work, "decoded%d.npy" % k)).tolist() for k in (0, 1)]
    assert sorted(dec[0] + dec[1]) == list(range(sp.n_samples))                  # every BAM decoded once in the whole job
    assert abs(len(dec[0]) - len(dec[1])) <= batch
    total = 0
    nbytes = [int(open(os.path.join(work, "n%d" % k)).read().split()[1]) for k in (0, 1)]
    for i in range(sp.n_samples):
        full = syn.sample_records(i)
        total += full.size
        recs = _records(full)
        for k in (0, 1):
            got = np.fromfile(os.path.join(work, "r%d_s%d.bin" % (k, i)), dtype=np.uint8)
            want = b"".join(b for t, b in recs if t >= 0 and owner[t] == k)
            assert bytes(got) == want, (i, k)
        # statistics counted by the decoder reach every rank
        parts, st = core.partition_records(full, owner, 2, 1)
        for k in (0, 1):
            assert (np.load(os.path.join(work, "stats%d.npy" % k))[i] == st).all()
        assert st[0] == len(recs)
    assert nbytes[0] + nbytes[1] == total and max(nbytes) <= 0.75 * total


def test_two_rank_decode_failure_reaches_every_rank(tmp_path):
    """A rank whose decode fails (bad BAM) must not leave the others waiting in the all-to-all: the status word that travels
    with the sizes makes every rank raise in the same round."""
    work = str(tmp_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_feed_worker.py"), work, "1", "3"]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode != 0
    e = [open(os.path.join(work, "error%d" % k)).read() for k in (0, 1)]          # sample 3 is decoded by rank 1 (batch 1, round 2)
    assert e[1].startswith("MsnvError") and "malformed BAM (test)" in e[1]
    assert e[0].startswith("RankError") and "rank 1 reported error 3" in e[0]


def test_two_rank_owners_follow_length_times_coverage(tmp_path):
    """createOptimumSplit.py:46-62: species are dealt heaviest-first by genome length x coverage.  One species five times as deep
    as five others of the same length: by length alone the ranks would hold 70 % and 30 % of the bases; with the coverage of the
    first decode round (parallel.feed_sharded, owner=None) the heavy species gets a rank of its own and the ranks end within 20 %."""
    work = str(tmp_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_lpt_worker.py"), work]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    o0, o1 = np.load(os.path.join(work, "owner0.npy")), np.load(os.path.join(work, "owner1.npy"))
    assert (o0 == o1).all() and sorted(np.bincount(o0).tolist()) == [1, 5] and np.bincount(o0)[o0[0]] == 1
    b = [int(np.load(os.path.join(work, "bases%d.npy" % k)).sum()) for k in (0, 1)]
    assert min(b) > 0 and max(b) / (sum(b) / 2.0) < 1.2, b
    # by length alone: three species each, 70 / 30
    from metasnv_amd import parallel
    by_len = parallel.shard_contigs(["sp%d.x.c1" % k for k in range(6)], [20000] * 6, 2)
    assert sorted(np.bincount(by_len).tolist()) == [3, 3]


def test_shard_contigs_weights():
    from metasnv_amd import parallel
    names, lengths = ["a.1", "a.2", "b.1", "c.1"], [100, 100, 150, 50]
    assert parallel.shard_contigs(names, lengths, 2) == [0, 0, 1, 1]                     # by length: a (200) | b (150) + c (50)
    # all_cov.tab sums (the reference's own weights): c is 10x deeper than the others -> c (500) | a (200) + b (150)
    assert parallel.shard_contigs(names, lengths, 2, species_weight={"a": 1.0, "b": 1.0, "c": 10.0}) == [1, 1, 1, 0]
    # aligned bases per contig; a species nobody has seen yet still weighs something
    own = parallel.shard_contigs(names, lengths, 2, contig_bases=[1000, 1000, 0, 5000])
    assert own[3] != own[0] and own[0] == own[1]


def test_partition_counts_qacompute_statistics_and_drops_unmapped():
    from bamtools import make_record
    recs = [make_record(0, 5, "10M", "A" * 10),
            make_record(0, 7, "10M", "A" * 10, flag=2, mapq=0),               # mapq 0: "sub-par"
            make_record(1, 3, "10M", "A" * 10, flag=0x400 | 2),              # duplicate, proper pair
            make_record(2, 9, "10M", "A" * 10),
            make_record(-1, -1, "*", "A" * 10, flag=4, mapq=0)]              # unmapped
    buf = np.frombuffer(b"".join(recs), dtype=np.uint8)
    parts, st = core.partition_records(buf, [1, 0, -1], 2, 1)
    assert bytes(parts[0]) == recs[2] and bytes(parts[1]) == recs[0] + recs[1]
    assert dict(zip(core.STATS_FIELDS, st.tolist())) == dict(total_reads=5, unmapped=1, zero_quality=1, proper_pairs=1, duplicates=1, any_mapped=1)


def test_bench_launcher_command_starts_n_ranks():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert os.path.samefile(cmd[cmd.index("29999") + 1], os.path.join(ROOT, "bench.py"))
