"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" when no GPU is visible, which is how the CPU tests run it).

The path shards by CONTIG: reference positions are independent, the reference
already bins whole species for its process pool (src/createOptimumSplit.py:46-62), and every rank
sees all samples but only its contigs' reads.  The kernels need no collective.  Two exchanges frame them:
  * before: the BAMs are dealt to the ranks for DECODING (every file is inflated once in the whole job; the
    reference's split processes each inflate every BAM, metaSNV.py:196-215) and the decoded records travel to
    the rank that owns their contig -- one all_to_all of byte streams per batch of BAMs (exchange_records);
  * after: the GATHER of small result tables to rank 0:
      - coverage accumulators: fixed size per (sample, contig) -> all_gather of equal tensors, summed;
      - called-site records: variable length -> all_gather of the counts, then a padded all_gather.
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


def exchange_records(parts):
    """All-to-all of byte streams: parts[q] (1-D uint8 array) goes to rank q; returns the list of the arrays this rank
    received, indexed by sender.  Sizes travel first (one all_to_all of `world` int64), then the bytes."""
    if _dist is None:
        return [np.ascontiguousarray(parts[0], dtype=np.uint8)]
    import torch
    dev = _device()
    sizes = [int(np.asarray(p).size) for p in parts]
    ts = torch.tensor(sizes, dtype=torch.int64, device=dev)
    tr = torch.zeros(_world, dtype=torch.int64, device=dev)
    _dist.all_to_all_single(tr, ts)
    rsizes = [int(x) for x in tr.cpu().tolist()]
    send = np.concatenate([np.ascontiguousarray(p, dtype=np.uint8).reshape(-1) for p in parts]) if sum(sizes) else np.zeros(0, np.uint8)
    tsend = torch.from_numpy(send).to(dev)
    trecv = torch.empty(sum(rsizes), dtype=torch.uint8, device=dev)
    _dist.all_to_all_single(trecv, tsend, output_split_sizes=rsizes, input_split_sizes=sizes)
    got = trecv.cpu().numpy()
    out, o = [], 0
    for n in rsizes:
        out.append(got[o:o + n])
        o += n
    return out


def deal_samples(n_samples, batch):
    """Decode schedule: rounds of world x batch consecutive samples; in a round, rank r decodes samples
    base + r * batch .. base + (r + 1) * batch - 1.  Yields (base, list of (sample, decoder rank))."""
    step = _world * batch
    for base in range(0, n_samples, step):
        yield base, [(i, (i - base) // batch) for i in range(base, min(n_samples, base + step))]


def feed_sharded(ds, bam_paths, owner, cov_min_mapq=1, batch=1, read_records=None, metrics=None):
    """Decode-sharded input of one dataset per rank: every BAM is read and inflated by ONE rank, its records are dealt by
    contig owner (core.partition_records) and exchanged, and every rank appends all samples in all_samples order holding
    only its contigs' records.  Returns stats[n_samples][6] (qaCompute's per-BAM statistics, counted by the decoder and
    all-gathered).  read_records(path) -> uint8 array replaces the BAM reader in tests."""
    from . import core
    n = len(bam_paths)
    if _world == 1 and read_records is None:             # nothing to exchange: decode + pack inside the library's thread pool
        ds.add_sample_bams(bam_paths, batch)
        if metrics is not None:
            metrics["inflated_record_bytes"] = None
        return np.stack([ds.sample_stats(i) for i in range(n)]) if n else np.zeros((0, len(core.STATS_FIELDS)), np.uint32)
    read_many = None
    if read_records is None:                             # the library reads a round's files in one call (device inflate when they are large)
        read_many = lambda paths: core.read_bam_records(paths, ctx=getattr(ds, "ctx", None), threads=max(1, len(paths)))
    stats = np.zeros((n, len(core.STATS_FIELDS)), dtype=np.uint32)
    inflated = 0
    for base, plan in deal_samples(n, batch):
        mine = [i for i, r in plan if r == _rank]
        # decode my samples of this round and deal every one of them to the owners
        per_dest = [[] for _ in range(_world)]          # per destination rank: (sample, bytes) in sample order
        if read_many is not None:
            decoded = read_many([bam_paths[i] for i in mine])
        elif len(mine) > 1:                              # the library releases the GIL: one decode thread per BAM of the round
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(mine)) as ex:
                decoded = list(ex.map(read_records, [bam_paths[i] for i in mine]))
        else:
            decoded = [read_records(bam_paths[i]) for i in mine]
        for i, rec in zip(mine, decoded):
            inflated += int(rec.size)
            parts, st = core.partition_records(rec, owner, _world, cov_min_mapq)
            stats[i] = st
            for q in range(_world):
                per_dest[q].append(parts[q])
        # one exchange per round: [sizes of my samples' parts | bytes], per destination
        send = []
        for q in range(_world):
            hdr = np.array([p.size for p in per_dest[q]], dtype=np.int64).view(np.uint8)
            send.append(np.concatenate([hdr] + per_dest[q]) if per_dest[q] else np.zeros(0, np.uint8))
        got = exchange_records(send)
        # unpack in sample order: sender r holds samples base + r * batch ...
        for i, r in plan:
            k = i - base - r * batch                     # index among the sender's samples of this round
            blob = got[r]
            n_from = len([1 for j, rr in plan if rr == r])
            sizes = blob[:8 * n_from].view(np.int64)
            o = 8 * n_from + int(sizes[:k].sum())
            ds.add_sample_records(blob[o:o + int(sizes[k])])
    allstats = gather_fixed(stats)
    stats = np.maximum.reduce(allstats) if len(allstats) > 1 else stats      # every row is non-zero on exactly one rank
    if metrics is not None:
        metrics["inflated_record_bytes"] = inflated
    return stats


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


def resident_project_run(ctx, first_bam, fasta_path, bam_paths, params, batch=1, want_coverage=True, ann_path=None,
                         after_coverage=None, species_weight=None):
    """ONE resident dataset per rank for a whole metaSNV.py run, whatever the number of ranks and splits (the reference forks
    one qaCompute process per BAM, metaSNV.py:55-78, and one `mpileup | snpCall` process per split, :196-221, each of which
    inflates every BAM again): contigs are sharded over the ranks by species (LPT), the BAMs are dealt to the ranks for
    decoding and their records exchanged (feed_sharded), every rank runs coverage and then calling [+ annotation] over its
    contigs, and every rank receives the summed coverage accumulators and the merged site records (rank 0 writes the files).

    after_coverage(result) is called on every rank between the two passes -- the driver writes cov/, the tables and the
    split plan there -- and must contain its own barriers.  Returns a dict:
      names, lengths, n_samples, stats[n_samples][6], acc[n_samples][n_contigs][17] (None without coverage),
      sites / samples / ann (merged, (tid, pos) order, `dropped` marks the global first line), first_any / first_from1
      (per contig, see core.Dataset.first_lines), metrics."""
    from . import core
    ds = core.Dataset.from_files(ctx, first_bam, fasta_path, params)
    names, lengths = ds.names, ds.lengths
    owner = shard_contigs(names, lengths, _world, species_weight)
    if _world > 1:
        ds.set_contig_mask([o == _rank for o in owner])
    metrics = {"rank": _rank, "world": _world, "contigs": int(sum(1 for o in owner if o == _rank))}
    res = {"names": names, "lengths": lengths, "n_samples": len(bam_paths), "metrics": metrics, "acc": None}
    try:
        res["stats"] = feed_sharded(ds, bam_paths, owner, params.cov_min_mapq, batch, metrics=metrics)
        metrics["dataset"] = ds.finalize()
        if want_coverage:
            metrics["coverage"] = ds.coverage_run()
            # every rank holds zeros outside its contigs, so the sum over ranks is the whole table
            res["acc"] = sum(a.astype(np.uint64) for a in gather_fixed(ds.coverage_accumulators()))
        if after_coverage:
            after_coverage(res)
        metrics["pileup"] = ds.run()
        sites, samples = ds.results()
        if ann_path and fasta_path:
            ann, _ = ds.annotate(ann_path, fasta_path)
            res["sites"], res["samples"], res["gfirst"], res["ann"] = gather_sites(sites, samples, ds.first_line(), ann)
        else:
            res["sites"], res["samples"], res["gfirst"] = gather_sites(sites, samples, ds.first_line())
            res["ann"] = None
        fa, f1 = ds.first_lines()                      # -1 outside this rank's contigs: the owner's value is the maximum
        res["first_any"] = np.maximum.reduce(gather_fixed(fa))
        res["first_from1"] = np.maximum.reduce(gather_fixed(f1))
        # per-rank inflated bytes for the report (the host stage shards with the ranks)
        infl = gather_fixed(np.array([metrics.get("inflated_record_bytes") or 0], dtype=np.int64))
        metrics["inflated_record_bytes_per_rank"] = [int(x[0]) for x in infl]
    finally:
        ds.close()
    return res


def split_view(res, split_contigs):
    """The records of one best_split_K invocation (`samtools mpileup -l best_split_K ... | snpCall`, metaSNV.py:160-176) cut
    out of a whole-run result: sites on the split's contigs at 0-based positions >= 1 (the split file's lines are
    `name 1 LEN`, parsed as BED [1, LEN): metaSNV.py:92), with the split's OWN first pileup line marked as dropped
    (call_vC.cpp:423; lines come in BAM-header order whatever the order of the split file)."""
    tid_of = {n: i for i, n in enumerate(res["names"])}
    tids = sorted(tid_of[n] for n in split_contigs if n in tid_of)          # samtools ignores names that are not in the header
    sites, samples, ann = res["sites"], res["samples"], res["ann"]
    keep = np.isin(sites["tid"], np.array(tids, dtype=np.int64)) & (sites["pos"] >= 1)
    s = sites[keep].copy()
    first = next(((t, int(res["first_from1"][t])) for t in tids if res["first_from1"][t] >= 0), (-1, -1))
    s["dropped"] = ((s["tid"] == first[0]) & (s["pos"] == first[1])).astype(np.uint8)
    return s, samples[keep], (ann[keep] if ann is not None else None)


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
