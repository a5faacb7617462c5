"""The RCCL code paths on ONE GPU: a process group of one rank with backend "nccl" runs every collective of
metasnv_amd/parallel.py on the device (the multi-rank rehearsals elsewhere in the suite share one GPU and therefore travel over
gloo).  The reference's counterpart of what these collectives frame is its pool of split processes, metaSNV.py:196-215."""
import os
import socket
import subprocess
import sys

import pytest

import orc
from metasnv_amd import core
from parity import run_oracle, synth_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun_one(script_args, env=None, timeout=900):
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    e = {k: v for k, v in os.environ.items() if k not in ("MSNV_DIST_BACKEND",)}
    e.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(env or {}))
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("deal", ["device", "host"])
def test_collectives_and_product_path_over_rccl(tmp_path, deal):
    """deal: the records are dealt to their owners by kernels (msnv_records_deal_device, the default over RCCL) or by host threads
    (msnv_records_partition); the worker asserts which one ran."""
    r = _torchrun_one([os.path.join(ROOT, "tests", "_nccl_worker.py"), str(tmp_path)], env=dict(MSNV_DEAL=deal))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    sp = core.synth_params(n_species=5, contig_len=4000, n_samples=6, mean_cov=11.0, snv_density=0.03, frac_absent=0.2, seed=55)
    syn = core.Synth(sp)
    names = ["%s.c" % n for n in syn.names]
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    orac = run_oracle(names, syn.lengths, syn.seqs, samples)
    assert open(tmp_path / "called_SNPs").read() == orac[0] and orac[0].count("\n") > 50
    assert open(tmp_path / "indiv_called").read() == orac[1]
    for i, s in enumerate(samples):                        # coverage rows gathered over RCCL -> qaCompute's files
        want = orc.qacompute(names, syn.lengths, s)
        assert open(tmp_path / ("s%d.cov" % i)).read() == want[0] and open(tmp_path / ("s%d.cov.detail" % i)).read() == want[1]


@pytest.mark.parametrize("plan_mb", [None, "1"])
def test_launcher_under_one_rccl_rank_matches_the_oracle(tmp_path, plan_mb):
    """plan_mb = "1": the split planner holds one megabyte of decoded records only, so the later rounds STREAM -- their BAMs are inflated,
    checked and dealt on the device (msnv_dataset_deal_bams_device), nothing of their inflated bytes on the host.
    metaSNV.py (reference argv) with the process group FORCED for a single rank (MSNV_DIST_FORCE=1, backend nccl): the BAMs go
    through msnv_bam_records_many -> partition -> all_to_all over RCCL -> pack, the coverage rows and the cell-form site records
    through the RCCL gather to rank 0.  Same bytes as the oracle for called_SNPs / indiv_called per split and every cov/ file."""
    syn, samples = synth_case(n_species=6, contig_len=3500, n_samples=8, mean_cov=11.0, snv_density=0.03, frac_absent=0.15, seed=91)
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    paths = []
    for i, s in enumerate(samples):
        p = str(tmp_path / ("s%04d.insilico.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, s)
        paths.append(p)
    lst = str(tmp_path / "all_samples")
    open(lst, "w").write("\n".join(paths) + "\n")
    proj, met = str(tmp_path / "proj"), str(tmp_path / "metrics.jsonl")
    r = _torchrun_one([os.path.join(ROOT, "metaSNV.py"), proj, lst, fa, "--n_splits", "2"], env=dict(MSNV_DIST_FORCE="1", MSNV_METRICS=met, **({"MSNV_PLAN_MB": plan_mb} if plan_mb else {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    total = 0
    for spf in sorted(os.listdir(os.path.join(proj, "bestsplits"))):
        bed = [(syn.names.index(l.split()[0]), int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(proj, "bestsplits", spf))]
        o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
        assert open(os.path.join(proj, "snpCaller", "called_SNPs." + spf)).read() == o[0], spf
        assert open(os.path.join(proj, "snpCaller", "indiv_called." + spf)).read() == o[1], spf
        total += o[0].count("\n")
    assert total > 50
    for i, p in enumerate(paths):
        want = orc.qacompute(syn.names, syn.lengths, samples[i])
        base = os.path.join(proj, "cov", os.path.basename(p) + ".cov")
        assert open(base).read() == want[0] and open(base + ".detail").read() == want[1]
    import json
    m = [json.loads(l) for l in open(met)]
    assert len(m) == 1 and m[0]["world"] == 1 and m[0]["gather_bytes_received"] > 0
    assert m[0].get("records_dealt_on_device_bytes", 0) == sum(int(s.size) for s in samples)       # every record was dealt by kernels ...
    # ... and no BAM reached the host inflated: the rounds the planner held were inflated into HBM and dealt from there, the streamed ones dealt as they came
    assert m[0].get("bams_inflated_on_device_bytes", 0) == sum(int(s.size) for s in samples)
