"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" when no GPU is visible, which is how the CPU tests run it).

The path shards by CONTIG: reference positions are independent, the reference
already bins whole species for its process pool (src/createOptimumSplit.py:46-62), and every rank
sees all samples but only its contigs' reads.  There is no data-path collective.  The one exchange
is the final GATHER of small result tables to rank 0:
  * coverage accumulators: fixed size per (sample, contig) -> all_gather of equal tensors;
  * called-site records: variable length -> all_gather of the counts, then a padded all_gather.
"""
import os
import sys

import numpy as np

_dist = None
_rank, _world, _local = 0, 1, 0


def init_from_env():
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun); a no-op for a single process."""
    global _dist, _rank, _world, _local
    _rank = int(os.environ.get("RANK", "0"))
    _world = int(os.environ.get("WORLD_SIZE", "1"))
    _local = int(os.environ.get("LOCAL_RANK", "0"))
    if _world > 1 and _dist is None:
        import torch
        import torch.distributed as dist
        if torch.cuda.is_available() and os.environ.get("MSNV_DIST_BACKEND", "nccl") == "nccl":
            torch.cuda.set_device(_local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", _local))
        else:
            # no GPU (CPU tests of the gather logic), or MSNV_DIST_BACKEND=gloo: rehearsal of the N-rank product path
            # on a box with fewer GPUs than ranks -- the ranks share the visible GPUs, the tables travel over gloo
            if torch.cuda.is_available():
                _local = _local % torch.cuda.device_count()
            dist.init_process_group(backend="gloo")
        _dist = dist
    return _rank, _world, _local


def rank():
    return _rank


def world():
    return _world


def barrier():
    if _dist is not None:
        _dist.barrier()


def abort(code):
    if _dist is not None:
        try:
            _dist.destroy_process_group()
        except Exception:
            pass
    sys.exit(code)


def finalize():
    global _dist
    if _dist is not None:
        _dist.destroy_process_group()
        _dist = None


def _device():
    import torch
    return torch.device("cuda", _local) if (_dist is not None and _dist.get_backend() == "nccl") else torch.device("cpu")


# ------------------------------------------------------------------------------------ sharding policy
def shard_contigs(names, lengths, n_ranks, species_weight=None):
    """contig -> rank by the reference's own rule: whole species (name up to the first '.') are
    assigned heaviest-first to the lightest rank (createOptimumSplit.py:46-62).  The weight of a
    species is genome length x summed coverage when known (species_weight), else its length.
    Returns a list of rank ids, one per contig."""
    from .tables import species_of, lpt_assign
    length = {}
    for n, l in zip(names, lengths):
        length[species_of(n)] = length.get(species_of(n), 0) + int(l)
    weighted = [((species_weight or {}).get(sp, 1.0) * l if species_weight else l, sp) for sp, l in length.items()]
    bins = lpt_assign(weighted, n_ranks)
    owner = {sp: r for r, sps in enumerate(bins) for sp in sps}
    return [owner[species_of(n)] for n in names]


# ------------------------------------------------------------------------------------ gathers
def gather_fixed(array):
    """all_gather of equally shaped numpy arrays; returns the list (every rank gets it)."""
    if _dist is None:
        return [np.asarray(array)]
    import torch
    a = np.ascontiguousarray(array)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(_device())      # bytes: every dtype travels (u64 accumulators too)
    out = [torch.empty_like(t) for _ in range(_world)]
    _dist.all_gather(out, t)
    return [o.cpu().numpy().view(a.dtype).reshape(a.shape) for o in out]


def gather_bytes(blob):
    """Variable-length gather: counts first, then one padded all_gather.  blob: 1-D uint8 array."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8).reshape(-1)
    if _dist is None:
        return [blob]
    import torch
    dev = _device()
    n = torch.tensor([blob.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(_world)]
    _dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if blob.size:
        buf[:blob.size] = torch.from_numpy(blob).to(dev)
    out = [torch.empty_like(buf) for _ in range(_world)]
    _dist.all_gather(out, buf)
    return [o[:c].cpu().numpy() for o, c in zip(out, counts)]


def gather_sites(sites, samples, first_line, ann=None):
    """Gathers the called-site records of every rank and merges them in (tid, pos) order.
    first_line: this rank's (tid, pos) of the first pileup line, tid = -1 if none.  Only the globally
    first line is the one the reference drops (call_vC.cpp:423), so the `dropped` mark of every other
    rank-local first line is cleared.  Every rank returns the merged arrays; with `ann` (the device
    annotation records of this rank's sites) the merged annotation records are appended to the tuple."""
    from .core import SITE_DTYPE, SAMPLE_DTYPE
    n_samples = samples.shape[1] if samples.ndim == 2 and samples.shape[0] else 0
    parts_s = gather_bytes(np.ascontiguousarray(sites).view(np.uint8))
    parts_m = gather_bytes(np.ascontiguousarray(samples).view(np.uint8))
    firsts = gather_fixed(np.array([first_line[0], first_line[1], n_samples], dtype=np.int64))
    ns = max(int(f[2]) for f in firsts)
    all_sites = np.concatenate([p.view(SITE_DTYPE) for p in parts_s]) if parts_s else np.zeros(0, SITE_DTYPE)
    all_samples = np.concatenate([p.view(SAMPLE_DTYPE).reshape(-1, ns) if p.size else np.zeros((0, ns), SAMPLE_DTYPE) for p in parts_m])
    cand = [(int(f[0]), int(f[1])) for f in firsts if f[0] >= 0]
    gfirst = min(cand) if cand else (-1, -1)
    all_sites = all_sites.copy()
    for i in np.nonzero(all_sites["dropped"])[0]:
        if (int(all_sites["tid"][i]), int(all_sites["pos"][i])) != gfirst:
            all_sites["dropped"][i] = 0
    order = np.lexsort((all_sites["pos"], all_sites["tid"]))
    if ann is None:
        return all_sites[order], all_samples[order], gfirst
    from .core import ANN_DTYPE
    parts_a = gather_bytes(np.ascontiguousarray(ann, dtype=ANN_DTYPE).view(np.uint8))
    all_ann = np.concatenate([p.view(ANN_DTYPE) for p in parts_a]) if parts_a else np.zeros(0, ANN_DTYPE)
    return all_sites[order], all_samples[order], gfirst, all_ann[order]


def sharded_call(ctx, names, lengths, seqs, add_samples, params=None, species_weight=None,
                 called_path=None, indiv_path=None, ann_path=None, fasta_path=None):
    """One SNV-calling pass over all contigs on all ranks: every rank packs only its shard's
    reads (contig mask), runs the kernels, and rank 0 receives the records and writes the files.
    add_samples(dataset) must append every sample in all_samples order."""
    from . import core
    owner = shard_contigs(names, lengths, _world, species_weight)
    ds = core.Dataset(ctx, names, lengths, seqs, params)
    ds.set_contig_mask([o == _rank for o in owner])
    add_samples(ds)
    info = ds.finalize()
    stats = ds.run()
    sites, samples = ds.results()
    merged_ann = None
    if ann_path and fasta_path:                      # codon annotation runs on every rank's device, records are gathered
        ann, _ = ds.annotate(ann_path, fasta_path)
        merged_sites, merged_samples, gfirst, merged_ann = gather_sites(sites, samples, ds.first_line(), ann)
    else:
        merged_sites, merged_samples, gfirst = gather_sites(sites, samples, ds.first_line())
    if _rank == 0 and called_path:
        core.write_calls_records(names, ds.n_samples, merged_sites, merged_samples, called_path, indiv_path, ann_path, fasta_path, merged_ann)
    ds.close()
    return merged_sites, merged_samples, info, stats
