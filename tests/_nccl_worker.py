"""Worker of tests/test_gpu_nccl.py: ONE rank under torchrun with backend "nccl" (= RCCL on ROCm).  Every collective
metasnv_amd/parallel.py issues runs through RCCL on the GPU here -- all_gather of byte views (gather_fixed, gather_bytes),
all_to_all_single with uneven uint8 splits, empty parts and int64 sizes (exchange_records, gather_to_root) -- and then the
product path itself: decode-sharded feeding with the contig owners fixed from the first round, kernels, the gather of the
cell-form records and coverage rows to rank 0, checked against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["MSNV_DIST_FORCE"] = "1"          # (metasnv_amd/_lib.py then binds to the HIP runtime torch has loaded)

from metasnv_amd import core, parallel  # noqa: E402


def main():
    work = sys.argv[1]
    rank, world, local = parallel.init_from_env(force=True)
    assert (rank, world) == (0, 1) and parallel.backend() == "nccl", (rank, world, parallel.backend())
    rng = np.random.default_rng(5)

    # ---- all_gather of byte views: u64 accumulators beyond 2^53, odd shapes, an empty table
    a = rng.integers(0, 2 ** 63, size=(3, 5, 17), dtype=np.uint64) | np.uint64(1 << 62)
    g = parallel.gather_fixed(a)
    assert len(g) == 1 and g[0].dtype == np.uint64 and (g[0] == a).all()
    assert parallel.gather_fixed(np.zeros((0, 4), np.int32))[0].shape == (0, 4)
    for n in (0, 1, 13, 65537):
        b = rng.integers(0, 256, size=n, dtype=np.uint8)
        got = parallel.gather_bytes(b)
        assert len(got) == 1 and got[0].size == n and (got[0] == b).all()

    # ---- all_to_all_single: int64 {size, status} pairs, then uint8 with uneven splits; empty, odd and large parts
    for n in (0, 1, 7, 4097, 300 * 1000 * 1000 + 3):
        b = rng.integers(0, 256, size=n, dtype=np.uint8)
        got = parallel.exchange_records([b])
        assert len(got) == 1 and got[0].size == n and (got[0] == b).all()
        st = {}
        got = parallel.gather_to_root(b, st)
        assert len(got) == 1 and (got[0] == b).all() and st["bytes_received"] == n
    try:
        parallel.exchange_records([np.zeros(3, np.uint8)], status=7)
        raise AssertionError("a rank's error status must raise on every rank")
    except parallel.RankError as e:
        assert "error 7" in str(e)

    # ---- the product path over RCCL: records decoded "by one rank", owners from the first round, exchange, kernels, gathers
    sp = core.synth_params(n_species=5, contig_len=4000, n_samples=6, mean_cov=11.0, snv_density=0.03, frac_absent=0.2, seed=55)
    syn = core.Synth(sp)
    names = ["%s.c" % n for n in syn.names]
    ctx = core.Context(local)
    params = core.default_params()
    res = parallel.resident_project_run(ctx, None, None, [str(i) for i in range(sp.n_samples)], params, batch=2,
                                        make_dataset=lambda: core.Dataset(ctx, names, syn.lengths, syn.seqs, params),
                                        read_records=lambda p: syn.sample_records(int(p)))
    assert res["owner"] == [0] * len(names)
    m = res["metrics"]
    assert m["gather_bytes_received"] > 0 and m["inflated_record_bytes_per_rank"] == [sum(syn.sample_records(i).size for i in range(sp.n_samples))]
    # the streams the all-to-all left in HBM were packed there (csrc/devpack.hip): none of them came back to the host
    assert m.get("records_packed_on_device_bytes", 0) == sum(syn.sample_records(i).size for i in range(sp.n_samples)), m.get("records_packed_on_device_bytes")
    core.write_calls_cells(names, sp.n_samples, res["sites"], res["row_off"], res["cell_sample"], res["cells"],
                           os.path.join(work, "called_SNPs"), os.path.join(work, "indiv_called"))
    for i in range(sp.n_samples):
        out = os.path.join(work, "s%d.cov" % i)
        core.write_coverage_records(names, syn.lengths, params.cov_max, res["stats"][i], res["acc"][i], out, out + ".detail")
    # ... and they were dealt to their owners there too (csrc/devpack.hip: msnv_records_deal_device) -- unless the test asks for the host threads
    want_dealt = 0 if os.environ.get("MSNV_DEAL", "device")[:1] == "h" else sum(syn.sample_records(i).size for i in range(sp.n_samples))
    assert m.get("records_dealt_on_device_bytes", 0) == want_dealt, m.get("records_dealt_on_device_bytes")
    open(os.path.join(work, "metrics"), "w").write(repr(m))
    parallel.barrier()
    parallel.finalize()


if __name__ == "__main__":
    main()
