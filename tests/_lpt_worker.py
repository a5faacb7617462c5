"""Worker of the world_size-2 gloo test of the rank sharding by genome length x coverage (createOptimumSplit.py:46-62):
six species of equal length, one of them five times as deep as the others.  parallel.feed_sharded fixes the contig owners from
the aligned bases of the first decode round; the ranks must end up with about the same number of bases.  No GPU: the dataset is a
recorder."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from metasnv_amd import core, parallel  # noqa: E402
from bamtools import make_record  # noqa: E402

NAMES = ["sp%d.x.c1" % k for k in range(6)]
LENGTHS = [20000] * 6
COV = [50, 10, 10, 10, 10, 10]              # species 0 is the heavy one


def sample_records(i):
    out = []
    for t, c in enumerate(COV):
        step = max(1, 100 // c)
        for pos in range((7 * i) % step, LENGTHS[t] - 100, step):
            out.append(make_record(t, pos, "100M", "A" * 100))
    return np.frombuffer(b"".join(out), dtype=np.uint8)


class Recorder:
    def __init__(self):
        self.samples, self.mask = [], None

    def set_contig_mask(self, m):
        assert not self.samples, "the owners must be fixed before any sample is added"
        self.mask = list(m)

    def add_sample_records(self, rec):
        self.samples.append(np.array(rec, dtype=np.uint8, copy=True))


def main():
    work = sys.argv[1]
    rank, world, local = parallel.init_from_env()
    rec = Recorder()
    metrics = {}
    parallel.feed_sharded(rec, [str(i) for i in range(6)], None, 1, 1, read_records=lambda p: sample_records(int(p)), metrics=metrics, plan=(NAMES, LENGTHS))
    owner = metrics["owner"]
    assert rec.mask == [o == rank for o in owner]
    mine = np.zeros(len(NAMES), dtype=np.uint64)
    for s in rec.samples:
        core.contig_bases(s, len(NAMES), into=mine)
    assert all(mine[t] == 0 for t in range(len(NAMES)) if owner[t] != rank)
    np.save(os.path.join(work, "bases%d.npy" % rank), mine)
    np.save(os.path.join(work, "owner%d.npy" % rank), np.array(owner))
    parallel.barrier()
    parallel.finalize()


if __name__ == "__main__":
    main()
