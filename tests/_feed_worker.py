"""Worker of the world_size-2 gloo test of the decode-sharded input (parallel.feed_sharded): every sample is "decoded" by
exactly one rank, its records are dealt by contig owner and exchanged, and every rank must end up with all samples in
order holding exactly its own contigs' records.  No GPU: the dataset is a recorder."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from metasnv_amd import core, parallel  # noqa: E402


class Recorder:
    def __init__(self):
        self.samples = []

    def add_sample_records(self, rec):
        self.samples.append(np.array(rec, dtype=np.uint8, copy=True))


def main():
    work, batch = sys.argv[1], int(sys.argv[2])
    rank, world, local = parallel.init_from_env()
    sp = core.synth_params(n_species=5, contig_len=2500, n_samples=7, mean_cov=6.0, frac_absent=0.3, seed=77)
    syn = core.Synth(sp)
    names = ["%s.c" % n for n in syn.names]
    owner = parallel.shard_contigs(names, syn.lengths, world)
    decoded = []

    fail_at = int(sys.argv[3]) if len(sys.argv) > 3 else -1

    def read_records(path):
        if int(path) == fail_at:
            raise core._lib.MsnvError(3, "sample %s: malformed BAM (test)" % path)
        decoded.append(int(path))
        return syn.sample_records(int(path))

    rec = Recorder()
    metrics = {}
    try:
        stats = parallel.feed_sharded(rec, [str(i) for i in range(sp.n_samples)], owner, 1, batch, read_records=read_records, metrics=metrics)
    except (core._lib.MsnvError, parallel.RankError) as e:
        # the rank that failed raises its own error, the others learn of it in the exchange: nobody waits in a collective
        open(os.path.join(work, "error%d" % rank), "w").write("%s: %s" % (type(e).__name__, e))
        parallel.abort(3)
    np.save(os.path.join(work, "stats%d.npy" % rank), stats)
    np.save(os.path.join(work, "decoded%d.npy" % rank), np.array(decoded, dtype=np.int64))
    np.save(os.path.join(work, "owner.npy"), np.array(owner, dtype=np.int32))
    for i, s in enumerate(rec.samples):
        s.tofile(os.path.join(work, "r%d_s%d.bin" % (rank, i)))
    open(os.path.join(work, "n%d" % rank), "w").write("%d %d %d" % (len(rec.samples), metrics["inflated_record_bytes"], int(metrics["decode_overlapped"])))
    parallel.barrier()
    parallel.finalize()


if __name__ == "__main__":
    main()
