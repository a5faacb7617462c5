"""Multi-GPU path on CPU: world_size 2, gloo backend.  The contig shards' site records are gathered with
torch.distributed and formatted on rank 0; the result must be the oracle's text for the whole dataset."""
import os
import socket
import subprocess
import sys

import numpy as np

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
