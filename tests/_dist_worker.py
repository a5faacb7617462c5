"""Worker of the world_size-2 gloo test: each rank holds the called-site records of its contig shard,
the records are gathered (counts first, then a padded all_gather) and rank 0 formats them."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from metasnv_amd import core, parallel  # noqa: E402
import recparse  # noqa: E402


def main():
    work = sys.argv[1]
    rank, world, local = parallel.init_from_env()
    names = open(os.path.join(work, "names")).read().split()
    lengths = [int(x) for x in open(os.path.join(work, "lengths")).read().split()]
    sites, samples, S = recparse.parse_calls(open(os.path.join(work, "pop")).read(), open(os.path.join(work, "ind")).read(), names)
    owner = parallel.shard_contigs(names, lengths, world)
    keep = np.array([owner[t] == rank for t in sites["tid"]], dtype=bool)
    first = [int(x) for x in open(os.path.join(work, "first")).read().split()]       # global first pileup line
    # every rank reports the first line of its own shard; here: the global one on its owner, a fake local one elsewhere
    mine_first = (first[0], first[1]) if owner[first[0]] == rank else ((int(sites["tid"][keep][0]), int(sites["pos"][keep][0])) if keep.any() else (-1, 0))
    local_sites = sites[keep].copy()
    if keep.any() and owner[first[0]] != rank:
        local_sites["dropped"][0] = 1                                                  # rank-local first line: must be un-dropped by the merge
    merged_sites, merged_samples, gfirst = parallel.gather_sites(local_sites, samples[keep], mine_first)
    cov = parallel.gather_fixed(np.full((3, 4), rank, dtype=np.uint64))
    assert [int(c[0, 0]) for c in cov] == list(range(world))
    if rank == 0:
        core.write_calls_records(names, S, merged_sites, merged_samples, os.path.join(work, "out_pop"), os.path.join(work, "out_ind"))
        open(os.path.join(work, "gfirst"), "w").write("%d %d %d" % (gfirst[0], gfirst[1], int(merged_sites["dropped"].sum())))
    parallel.barrier()
    parallel.finalize()


if __name__ == "__main__":
    main()
