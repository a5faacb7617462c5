"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" when no GPU is visible, which is how the CPU tests run it).

The path shards by CONTIG: reference positions are independent, the reference
already bins whole species for its process pool (src/createOptimumSplit.py:46-62), and every rank
sees all samples but only its contigs' reads.  The kernels need no collective.  Two exchanges frame them:
  * before: the BAMs are dealt to the ranks for DECODING (every file is inflated once in the whole job; the
    reference's split processes each inflate every BAM, metaSNV.py:196-215) and the decoded records travel to
    the rank that owns their contig -- one all_to_all of byte streams per batch of BAMs (exchange_records).
    The contig owners are fixed by the reference's rule, whole species heaviest-first by genome length x
    coverage (createOptimumSplit.py:46-62), with the coverage taken from the first round of decoded BAMs;
  * after: the GATHER of the result tables TO RANK 0, compressed (gather_to_root: one all_to_all in which only
    rank 0 receives, exact sizes, no padding):
      - called-site records in the cell form (a cell per sample that holds something at the site, not
        n_samples entries per site: core.Dataset.results_cells);
      - coverage accumulators: the non-zero (sample, contig) rows only.
"""
import os
import sys

import numpy as np

_dist = None
_rank, _world, _local = 0, 1, 0


def init_from_env(force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun); a no-op for a single process unless `force` (or MSNV_DIST_FORCE=1):
    a process group of ONE rank, which is how the RCCL code paths run on a box with a single GPU (tests/test_gpu_nccl.py)."""
    global _dist, _rank, _world, _local
    _rank = int(os.environ.get("RANK", "0"))
    _world = int(os.environ.get("WORLD_SIZE", "1"))
    _local = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("MSNV_DIST_FORCE") == "1"
    if (_world > 1 or force) and _dist is None:
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


def backend():
    return _dist.get_backend() if _dist is not None else None


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


class RankError(RuntimeError):
    """Another rank failed in a step every rank takes part in (its own exception names the cause on that rank)."""


# ------------------------------------------------------------------------------------ sharding policy
def shard_contigs(names, lengths, n_ranks, species_weight=None, contig_bases=None):
    """contig -> rank by the reference's own rule: whole species (name up to the first '.') are
    assigned heaviest-first to the lightest rank (createOptimumSplit.py:46-62).  The weight of a
    species is genome length x summed coverage (`read = genomeLen[k]*coverage[k]`, :46-50):
      species_weight {species: summed coverage}  -- the all_cov.tab column sums, as the reference has them; or
      contig_bases   [aligned bases per contig]  -- length x coverage IS the aligned bases; counted on a SUBSET of the samples
                     (the first decode round), so a species none of them carries still gets 5 % of the mean coverage;
    else its length.  Returns a list of rank ids, one per contig."""
    from .tables import species_of, lpt_assign
    length, bases = {}, {}
    for i, (n, l) in enumerate(zip(names, lengths)):
        sp = species_of(n)
        length[sp] = length.get(sp, 0) + int(l)
        if contig_bases is not None:
            bases[sp] = bases.get(sp, 0) + int(contig_bases[i])
    if contig_bases is not None:
        prior = 0.05 * sum(bases.values()) / max(1, sum(length.values()))
        weighted = [(bases[sp] + prior * l, sp) for sp, l in length.items()]
    elif species_weight:
        weighted = [(species_weight.get(sp, 0.0) * l, sp) for sp, l in length.items()]
    else:
        weighted = [(l, sp) for sp, l in length.items()]
    bins = lpt_assign(weighted, n_ranks)
    owner = {sp: r for r, sps in enumerate(bins) for sp in sps}
    return [owner[species_of(n)] for n in names]


# ------------------------------------------------------------------------------------ collectives
def gather_fixed(array):
    """all_gather of equally shaped numpy arrays; returns the list (every rank gets it).  For SMALL tables only."""
    if _dist is None:
        return [np.asarray(array)]
    import torch
    a = np.ascontiguousarray(array)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(_device())      # bytes: every dtype travels (u64 accumulators too)
    out = [torch.empty_like(t) for _ in range(_world)]
    _dist.all_gather(out, t)
    return [o.cpu().numpy().view(a.dtype).reshape(a.shape) for o in out]


def gather_bytes(blob):
    """Variable-length gather to EVERY rank: counts first, then one padded all_gather.  blob: 1-D uint8 array."""
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


def agree(failure, what):
    """Every rank calls this behind a step any of them may have failed in (an exception object, or None): the ranks exchange their status,
    the failing rank re-raises its own exception, the others raise RankError naming it -- nobody is left waiting in the next collective."""
    if _dist is None:
        if failure is not None:
            raise failure
        return
    codes = gather_fixed(np.array([0 if failure is None else (int(getattr(failure, "code", 0)) or 99)], dtype=np.int64))
    bad = [(r, int(c[0])) for r, c in enumerate(codes) if int(c[0])]
    if failure is not None:
        raise failure
    if bad:
        raise RankError("rank %d reported error %d while %s" % (bad[0][0], bad[0][1], what))


class DeviceParts:
    """What exchange_records(..., keep_on_device=True) returns over RCCL: the received bytes as ONE tensor in HBM and where every
    sender's part lies in it.  The record streams inside go to the library where they are (core.Dataset.add_samples_records_device:
    parsed, filtered and packed by kernels); only the few header bytes a caller asks for with head() come to the host."""

    def __init__(self, tensor, sizes):
        self.tensor, self.sizes = tensor, [int(x) for x in sizes]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)

    def head(self, sender, n_bytes):
        o = self.offsets[sender]
        return self.tensor[o:o + n_bytes].cpu().numpy()

    def address(self, sender, offset=0):
        return self.tensor.data_ptr() + self.offsets[sender] + int(offset)


class DeviceSend:
    """A send buffer of exchange_records that was put together in HBM: one uint8 tensor holding the parts destination-major, and their sizes."""

    def __init__(self, tensor, sizes):
        self.tensor, self.sizes = tensor, [int(x) for x in sizes]


_a2a_checked = False


def _a2a_selfcheck(dev, slice_bytes):
    """Once per process, before the first byte exchange over RCCL: an all-to-all of a known pattern with parts of the size the exchange
    slices its buffers to, checked on the device.  This stack has returned wrong bytes for large exchanges (round 4: an all_to_all_single of
    more than 2^30 uint8 elements, profiles/a2a_check.py); a collective that does not deliver what was sent must stop the run, loudly."""
    global _a2a_checked
    if _a2a_checked or os.environ.get("MSNV_A2A_SELFCHECK", "1") == "0":
        return
    _a2a_checked = True
    import torch
    n = int(min(slice_bytes, 1 << 26))                       # (64 MB a part at most: the check must not need gigabytes)
    n -= n % 8
    base = torch.arange(n // 8, dtype=torch.int64, device=dev)
    send = [((base * 2654435761 + (_rank * 131 + q) * 40503) & 0x7fffffffffffffff).view(torch.uint8) for q in range(_world)]
    recv = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(_world)]
    _dist.all_to_all(recv, send)
    bad = 0
    for r in range(_world):
        want = ((base * 2654435761 + (r * 131 + _rank) * 40503) & 0x7fffffffffffffff).view(torch.uint8)
        bad += int((recv[r] != want).sum().item())
    if bad:
        raise RuntimeError("RCCL all_to_all self-check failed on rank %d: %d of %d bytes differ from what the senders sent (parts of %d bytes) -- "
                           "the collective of this stack cannot be trusted with the record exchange" % (_rank, bad, n * _world, n))


def exchange_records(parts, status=0, keep_on_device=False):
    """All-to-all of byte streams: parts[q] (1-D uint8 array) goes to rank q; returns the list of the arrays this rank
    received, indexed by sender.  {size, status} pairs travel first (one all_to_all of world x 2 int64), then the bytes.
    `status` != 0 says "this rank failed in the step that made the parts": every rank then raises RankError before the
    byte exchange instead of waiting in it for a rank that is gone.
    keep_on_device (RCCL only; ignored over gloo): the received bytes stay in HBM -- returns a DeviceParts instead of host arrays."""
    if _dist is None:
        if status:
            raise RankError("rank 0 reported error %d" % status)
        return [np.ascontiguousarray(parts[0], dtype=np.uint8)]
    import torch
    dev = _device()
    on_dev_send = isinstance(parts, DeviceSend)           # the send buffer already lies in HBM, destination-major (core.deal_records_device)
    sizes = [int(x) for x in parts.sizes] if on_dev_send else [int(np.asarray(p).size) for p in parts]
    # {size, status, my largest part}: the third word tells every rank how many slices the byte exchange needs (below)
    ts = torch.tensor([[s, int(status), max(sizes) if sizes else 0] for s in sizes], dtype=torch.int64, device=dev).reshape(-1)
    tr = torch.zeros(3 * _world, dtype=torch.int64, device=dev)
    _dist.all_to_all_single(tr, ts)
    got = tr.cpu().tolist()
    rsizes, rstatus, rmax = got[0::3], got[1::3], got[2::3]
    bad = [(r, int(c)) for r, c in enumerate(rstatus) if c]
    if bad:
        raise RankError("rank %d reported error %d while the records were decoded" % bad[0])
    if on_dev_send:
        tsend = parts.tensor[:sum(sizes)]
    else:
        send = np.concatenate([np.ascontiguousarray(p, dtype=np.uint8).reshape(-1) for p in parts]) if sum(sizes) else np.zeros(0, np.uint8)
        tsend = torch.from_numpy(send).to(dev)
    trecv = torch.empty(sum(rsizes), dtype=torch.uint8, device=dev)
    # The bytes travel in slices of at most 2^29 / world per part: an all_to_all_single of MORE THAN 2^30 uint8 elements returns wrong bytes
    # behind the first ~half of the buffer on this stack (RCCL of ROCm 7.2 under torch 2.10, found in round 4 at 1.1 GB per round:
    # profiles/a2a_check.py) -- and a round of a large cohort is several GB.  Every rank walks the same number of slices (the largest part
    # anywhere, from the size exchange above).
    slice_bytes = int(os.environ.get("MSNV_A2A_SLICE_KB", "0")) << 10 or max(1 << 20, (1 << 29) // _world)      # (MSNV_A2A_SLICE_KB: tests)
    n_slices = max(1, -(-max(rmax + [0]) // slice_bytes))
    soff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    roff = np.concatenate([[0], np.cumsum(rsizes)]).astype(np.int64)
    lists = dev.type == "cuda" and os.environ.get("MSNV_A2A_FORM", "lists") == "lists"
    if lists:
        _a2a_selfcheck(dev, slice_bytes)
    for j in range(n_slices):
        lo = j * slice_bytes
        ins = [min(slice_bytes, max(0, n - lo)) for n in sizes]
        outs = [min(slice_bytes, max(0, n - lo)) for n in rsizes]
        if lists:
            # RCCL: every part's slice is sent from and received INTO its place -- views of the two buffers, no staging tensor, no copy back
            # (round 4 concatenated the slices of several senders and copied the received ones back: an extra pass over HBM per round)
            _dist.all_to_all([trecv[int(roff[r]) + lo:int(roff[r]) + lo + outs[r]] for r in range(_world)],
                             [tsend[int(soff[q]) + lo:int(soff[q]) + lo + ins[q]] for q in range(_world)])
            continue
        if n_slices == 1:
            _dist.all_to_all_single(trecv, tsend, output_split_sizes=outs, input_split_sizes=ins)
            break
        live_in = [q for q in range(_world) if ins[q]]
        tin = (tsend[int(soff[live_in[0]]) + lo:int(soff[live_in[0]]) + lo + ins[live_in[0]]] if len(live_in) == 1
               else torch.cat([tsend[int(soff[q]) + lo:int(soff[q]) + lo + ins[q]] for q in live_in]) if live_in else tsend[:0])
        live_out = [r for r in range(_world) if outs[r]]
        direct = len(live_out) == 1                          # one sender in this slice: straight into its place
        tout = trecv[int(roff[live_out[0]]) + lo:int(roff[live_out[0]]) + lo + outs[live_out[0]]] if direct else torch.empty(sum(outs), dtype=torch.uint8, device=dev)
        _dist.all_to_all_single(tout, tin, output_split_sizes=outs, input_split_sizes=ins)
        if not direct:
            o = 0
            for r in live_out:
                trecv[int(roff[r]) + lo:int(roff[r]) + lo + outs[r]] = tout[o:o + outs[r]]
                o += outs[r]
    if keep_on_device and dev.type == "cuda":
        torch.cuda.current_stream().synchronize()        # the library reads the tensor on a stream of its own
        return DeviceParts(trecv, rsizes)
    got = trecv.cpu().numpy()
    out, o = [], 0
    for n in rsizes:
        out.append(got[o:o + n])
        o += n
    return out


def gather_to_root(blob, stats=None):
    """Variable-length gather to RANK 0 ONLY: the all-to-all of exchange_records in which every rank sends its bytes to
    rank 0 and nothing to the others (exact sizes, no padding, nothing lands on a rank that does not write).  Returns the list
    of arrays indexed by sender on rank 0, None elsewhere.  stats (dict): bytes_received accumulates what rank 0 took in."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8).reshape(-1)
    if _dist is None:
        if stats is not None:
            stats["bytes_received"] = stats.get("bytes_received", 0) + int(blob.size)
        return [blob]
    empty = np.zeros(0, np.uint8)
    got = exchange_records([blob if q == 0 else empty for q in range(_world)])
    if stats is not None and _rank == 0:
        stats["bytes_received"] = stats.get("bytes_received", 0) + int(sum(g.size for g in got))
    return got if _rank == 0 else None


def deal_samples(n_samples, batch):
    """Decode schedule: rounds of world x batch consecutive samples; in a round, rank r decodes samples
    base + r * batch .. base + (r + 1) * batch - 1.  Yields (base, list of (sample, decoder rank))."""
    step = _world * batch
    for base in range(0, n_samples, step):
        yield base, [(i, (i - base) // batch) for i in range(base, min(n_samples, base + step))]


def _stageable(paths):
    """Do the inflated record streams of these files fit the host's memory with room to spare?  (~4x the file bytes; MSNV_STAGE_MAX_MB
    overrides the default of a quarter of the physical memory.)"""
    try:
        total = sum(os.path.getsize(p) for p in paths) * 4
        limit = int(os.environ.get("MSNV_STAGE_MAX_MB", "0")) << 20
        if limit <= 0:
            limit = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") // 4
        return total <= limit
    except (OSError, ValueError):
        return False


def feed_sharded(ds, bam_paths, owner, cov_min_mapq=1, batch=1, read_records=None, metrics=None, plan=None, pack_threads=0, ctx_when_ready=None, feed_overlap=None):
    """Decode-sharded input of one dataset per rank: every BAM is read and inflated by ONE rank, its records are dealt by
    contig owner (core.partition_records) and exchanged, and every rank appends all samples in all_samples order holding
    only its contigs' records.  Returns stats[n_samples][6] (qaCompute's per-BAM statistics, counted by the decoder and
    all-gathered).  read_records(path) -> uint8 array replaces the BAM reader in tests.

    owner = None: the contig owners are fixed HERE, after the first round has been decoded and before anything is dealt --
    plan = (names, lengths); the ranks add up the aligned bases per contig of their first-round samples (= genome length x
    coverage, the reference's split weight, createOptimumSplit.py:46-50), shard_contigs assigns the species, the dataset gets
    its contig mask and metrics["owner"] the assignment."""
    from . import core
    n = len(bam_paths)
    if _dist is None and read_records is None:           # nothing to exchange: decode + pack inside the library's thread pool
        if owner is None and metrics is not None:
            metrics["owner"] = [0] * len(plan[0])
        if metrics is not None:
            metrics["inflated_record_bytes"] = None
        if getattr(ds, "ctx", None) is None and callable(ctx_when_ready):
            # (cli.py's choice for hosts with few cores; MSNV_ONESHOT=device | host) wait for the context, then everything on the device (BGZF blocks inflated and checked there, records packed
            # there) instead of inflating on host threads meanwhile.  On a host that grants the job 16 cores the two are close (one-shot run of
            # the 160 BAMs: 0.82 vs 0.89 s wall, profiles/e2e_ab.sh); with 32 real cores the host threads finish under the runtime's start-up
            for p in bam_paths:                              # (the page cache may fill while the runtime comes up: a hint, not a read)
                try:
                    fd = os.open(p, os.O_RDONLY)
                    try:
                        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_WILLNEED)
                    finally:
                        os.close(fd)
                except (OSError, AttributeError):
                    pass
            ds.attach_context(ctx_when_ready())
            ds.add_sample_bams(bam_paths, batch)
            return np.stack([ds.sample_stats(i) for i in range(n)]) if n else np.zeros((0, len(core.STATS_FIELDS)), np.uint32)
        if getattr(ds, "ctx", None) is None and hasattr(ds, "stage_sample_bams") and os.environ.get("MSNV_PACK", "device")[:1] != "h" and _stageable(bam_paths):
            # the device is still coming up (cli.py brings the HIP runtime up on a thread of its own): the files are read and inflated now,
            # their records are packed by kernels once the context is attached (finalize); the statistics exist then
            ds.stage_sample_bams(bam_paths, batch)
            return None
        ds.add_sample_bams(bam_paths, batch)
        return np.stack([ds.sample_stats(i) for i in range(n)]) if n else np.zeros((0, len(core.STATS_FIELDS)), np.uint32)
    read_many = None
    if read_records is None:                             # the library reads a round's files in one call (device inflate when they are large)
        read_many = lambda paths: core.read_bam_records(paths, ctx=getattr(ds, "ctx", None), threads=max(1, len(paths)))
    stats = np.zeros((n, len(core.STATS_FIELDS)), dtype=np.uint32)
    inflated = 0
    batch = max(1, min(batch, -(-n // _world)))          # fewer samples than world x batch: every rank still decodes its share
    class DealtRound:
        """A round that left decode_round already dealt on the device: the send tensor with the parts destination-major behind their gaps."""
        def __init__(self, tensor, part_bytes, stats, record_bytes, gap):
            self.tensor, self.part_bytes, self.stats, self.record_bytes, self.gap = tensor, part_bytes, stats, record_bytes, gap

    class HeldOnDevice:
        """A round decoded BEFORE the owners are known, held in HBM: its record streams in one tensor (inflated there), their statistics and the
        aligned bases per contig the planner asked for; deliver() deals them from where they lie."""
        def __init__(self, tensor, offsets, sizes, stats, bases):
            self.tensor, self.offsets, self.sizes, self.stats, self.bases = tensor, offsets, sizes, stats, bases

    def device_route(need_owner=True):
        return ((owner is not None or not need_owner) and _dist is not None and getattr(ds, "ctx", None) is not None and hasattr(ds, "deal_bams_device")
                and os.environ.get("MSNV_PACK", "device")[:1] != "h" and os.environ.get("MSNV_DEAL", "device")[:1] != "h"
                and os.environ.get("MSNV_INFLATE", "device")[:1] == "d" and _device().type == "cuda")

    def deal_files_on_device(paths):
        import torch
        torch.cuda.set_device(_device())                         # (the round ahead runs on a thread of its own: torch's current device is per thread)
        gap = 8 * len(paths)
        comp = sum(os.path.getsize(p) for p in paths)
        cap = 8 * comp + (1 << 20) + _world * gap             # (BAM inflates 2.5-4 x; the call answers MSNV_ECAPACITY when this is not enough)
        tsend = torch.empty(cap, dtype=torch.uint8, device=_device())
        torch.cuda.current_stream().synchronize()
        try:
            pb, st, rb = ds.deal_bams_device(paths, owner, _world, tsend.data_ptr(), cap, gap=gap, cov_min_mapq=cov_min_mapq, host_threads=max(1, len(paths)))
        except core._lib.MsnvError as e:
            if e.code in (core._lib.EDOMAIN, core._lib.ECAPACITY):
                return None
            raise
        return DealtRound(tsend, pb, st, rb, gap)

    def hold_files_on_device(paths, n_contigs):
        import torch
        comp = sum(os.path.getsize(p) for p in paths)
        cap = 8 * comp + (1 << 20) + 32 * len(paths)
        t = torch.empty(cap, dtype=torch.uint8, device=_device())
        torch.cuda.current_stream().synchronize()
        cb = np.zeros(n_contigs, dtype=np.uint64)
        try:
            offs, sizes, st = ds.inflate_bams_device(paths, t.data_ptr(), cap, contig_bases=cb, host_threads=max(1, len(paths)))
        except core._lib.MsnvError as e:
            if e.code in (core._lib.EDOMAIN, core._lib.ECAPACITY):
                return None
            raise
        used = int(offs[-1] + sizes[-1]) + 32 if len(offs) else 32
        held_t = t[:used].clone()                            # (the generous buffer goes back to the allocator's cache: the next round takes it again)
        del t
        return HeldOnDevice(held_t, offs, sizes, st, cb)

    def decode_round(plan_round, hold_contigs=0):
        """My samples of this round, decoded; a failure here is carried to every rank by the exchange (status word).
        hold_contigs > 0: the split planner's call -- over RCCL the files are inflated on the device and HELD there (HeldOnDevice)."""
        mine = [i for i, r in plan_round if r == _rank]
        failure, decoded = None, []
        try:
            if hold_contigs and read_many is not None and mine and device_route(need_owner=False):
                held_dev = hold_files_on_device([bam_paths[i] for i in mine], hold_contigs)
                if held_dev is not None:
                    return mine, held_dev, None
            if read_many is not None and mine and device_route():
                # owners known, RCCL, device pack: the round's files are inflated, CRC-checked and dealt ON THE DEVICE (core.Dataset.deal_bams_device:
                # nothing of the inflated bytes on the host); a round that does not fit one batch of the device inflate, or an output that turns
                # out too small for its inflated bytes, takes the host route below
                dealt = deal_files_on_device([bam_paths[i] for i in mine])
                if dealt is not None:
                    return mine, dealt, None
            if read_many is not None:
                decoded = read_many([bam_paths[i] for i in mine])
            elif len(mine) > 1:                              # the library releases the GIL: one decode thread per BAM of the round
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=len(mine)) as ex:
                    decoded = list(ex.map(read_records, [bam_paths[i] for i in mine]))
            else:
                decoded = [read_records(bam_paths[i]) for i in mine]
        except Exception as e:                               # noqa: BLE001 -- re-raised by deliver(), after the other ranks have been told
            failure, decoded = e, []
        return mine, decoded, failure

    def size_tables(tsend, pb, gap):
        """The size table of my samples' parts (int64 each) into the gap the dealer left in front of every part; returns the bytes per destination."""
        import torch
        sizes, o = [], 0
        for q in range(_world):
            tsend[o:o + gap] = torch.from_numpy(np.ascontiguousarray(pb[:, q], dtype=np.int64).view(np.uint8).copy()).to(_device())
            sizes.append(gap + int(pb[:, q].sum()))
            o += sizes[-1]
        return sizes

    def deliver(base, plan_round, mine, decoded, failure):
        """Deals the round's records to the contig owners, exchanges them, appends the round's samples to this rank's dataset."""
        nonlocal inflated
        if failure is None and pending[0] is not None:      # the previous round's append failed here: this round's status word says so
            failure, pending[0] = pending[0], None
        per_dest = [[] for _ in range(_world)]              # per destination rank: (sample, bytes) in sample order
        # over RCCL with the device pack, the dealing itself runs on the device (core.deal_records_device: one upload of the round's streams,
        # kernels put every record where the all-to-all sends it from): no host walk over the records, no host copy of the parts
        # (MSNV_DEAL=host keeps the host threads)
        dev_send = None
        if failure is None and isinstance(decoded, HeldOnDevice):
            try:
                import torch
                gap = 8 * len(mine)
                cap = int(decoded.sizes.sum()) + _world * gap
                tsend = torch.empty(max(cap, 16), dtype=torch.uint8, device=_device())
                torch.cuda.current_stream().synchronize()
                base_ptr = decoded.tensor.data_ptr()
                pb, _st = core.deal_records_device(ds.ctx, [base_ptr + int(o) for o in decoded.offsets], owner, _world, tsend.data_ptr(), cap, gap=gap, cov_min_mapq=cov_min_mapq,
                                                   on_device=True, sizes=[int(x) for x in decoded.sizes])
                sizes = size_tables(tsend, pb, gap)
                for i, row, nb in zip(mine, decoded.stats, decoded.sizes):
                    inflated += int(nb)
                    stats[i] = row
                dev_send = DeviceSend(tsend, sizes)
                if metrics is not None:
                    metrics["records_dealt_on_device_bytes"] = metrics.get("records_dealt_on_device_bytes", 0) + int(decoded.sizes.sum())
                    metrics["bams_inflated_on_device_bytes"] = metrics.get("bams_inflated_on_device_bytes", 0) + int(decoded.sizes.sum())
            except Exception as e:                           # noqa: BLE001
                failure, dev_send = e, None
            decoded = []
        if failure is None and isinstance(decoded, DealtRound):
            try:
                import torch
                pb, gap, tsend = decoded.part_bytes, decoded.gap, decoded.tensor
                sizes = size_tables(tsend, pb, gap)
                for i, row, nb in zip(mine, decoded.stats, decoded.record_bytes):
                    inflated += int(nb)
                    stats[i] = row
                dev_send = DeviceSend(tsend, sizes)
                if metrics is not None:
                    metrics["records_dealt_on_device_bytes"] = metrics.get("records_dealt_on_device_bytes", 0) + int(decoded.record_bytes.sum())
                    metrics["bams_inflated_on_device_bytes"] = metrics.get("bams_inflated_on_device_bytes", 0) + int(decoded.record_bytes.sum())
            except Exception as e:                           # noqa: BLE001
                failure, dev_send = e, None
            decoded = []
        deal_on_device = (dev_send is None and failure is None and len(decoded) > 0 and _dist is not None and getattr(ds, "ctx", None) is not None and hasattr(ds, "add_samples_records_device")
                          and os.environ.get("MSNV_PACK", "device")[:1] != "h" and os.environ.get("MSNV_DEAL", "device")[:1] != "h" and _device().type == "cuda")
        if deal_on_device:
            try:
                import torch
                gap = 8 * len(mine)
                cap = int(sum(int(r.size) for r in decoded)) + _world * gap
                tsend = torch.empty(max(cap, 16), dtype=torch.uint8, device=_device())
                torch.cuda.current_stream().synchronize()
                pb, st = core.deal_records_device(ds.ctx, decoded, owner, _world, tsend.data_ptr(), cap, gap=gap, cov_min_mapq=cov_min_mapq)
                sizes = size_tables(tsend, pb, gap)
                for i, rec, row in zip(mine, decoded, st):
                    inflated += int(rec.size)
                    stats[i] = row
                dev_send = DeviceSend(tsend, sizes)
                if metrics is not None:
                    metrics["records_dealt_on_device_bytes"] = metrics.get("records_dealt_on_device_bytes", 0) + int(sum(int(r.size) for r in decoded))
            except Exception as e:                           # noqa: BLE001
                failure, dev_send = e, None
        if failure is None and dev_send is None:
            try:
                # one walk over every sample's records (msnv_records_partition: owner of every record's contig, qaCompute's statistics); the
                # library releases the GIL, so the round's samples are dealt side by side (one after the other this was 0.3 s per GB and rank)
                deal = lambda rec: core.partition_records(rec, owner, _world, cov_min_mapq)
                if len(decoded) > 1:
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(max_workers=min(len(decoded), max(1, batch))) as ex:
                        dealt = list(ex.map(deal, decoded))
                else:
                    dealt = [deal(rec) for rec in decoded]
                for i, rec, (parts, st) in zip(mine, decoded, dealt):
                    inflated += int(rec.size)
                    stats[i] = st
                    for q in range(_world):
                        per_dest[q].append(parts[q])
            except Exception as e:                           # noqa: BLE001
                failure, per_dest = e, [[] for _ in range(_world)]
        # one exchange per round: [sizes of my samples' parts | bytes], per destination
        send = []
        for q in range(_world):
            hdr = np.array([p.size for p in per_dest[q]], dtype=np.int64).view(np.uint8)
            send.append(np.concatenate([hdr] + per_dest[q]) if per_dest[q] else np.zeros(0, np.uint8))
        # over RCCL the received streams stay in HBM and are parsed / filtered / packed there (csrc/devpack.hip): no copy to the host, no host
        # pack, no second upload (MSNV_PACK=host keeps the round trip)
        on_device = getattr(ds, "ctx", None) is not None and hasattr(ds, "add_samples_records_device") and os.environ.get("MSNV_PACK", "device")[:1] != "h"
        try:
            got = exchange_records(dev_send if dev_send is not None else send, status=0 if failure is None else int(getattr(failure, "code", 0)) or 99, keep_on_device=on_device)
        except RankError:
            if failure is not None:
                raise failure
            raise
        # unpack in sample order: sender r holds samples base + r * batch ...
        n_from = [0] * _world
        for _, r in plan_round:
            n_from[r] += 1
        if isinstance(got, DeviceParts):
            sizes_of = [got.head(r, 8 * n_from[r]).view(np.int64) for r in range(_world)]
            starts = [8 * n_from[r] + np.concatenate([[0], np.cumsum(sizes_of[r])]).astype(np.int64) for r in range(_world)]
            ptrs, sizes = [], []
            for i, r in plan_round:
                k = i - base - r * batch
                ptrs.append(got.address(r, starts[r][k]) if int(sizes_of[r][k]) else 0)
                sizes.append(int(sizes_of[r][k]))
            try:
                ds.add_samples_records_device(ptrs, sizes)
            except Exception as e:                           # noqa: BLE001 -- told to the other ranks by the next round's status word / agree()
                pending[0] = e
            if metrics is not None:
                metrics["records_packed_on_device_bytes"] = metrics.get("records_packed_on_device_bytes", 0) + int(sum(sizes))
            return
        sizes_of = [got[r][:8 * n_from[r]].view(np.int64) for r in range(_world)]
        starts = [8 * n_from[r] + np.concatenate([[0], np.cumsum(sizes_of[r])]).astype(np.int64) for r in range(_world)]
        streams = []
        for i, r in plan_round:
            k = i - base - r * batch                     # index among the sender's samples of this round
            o = int(starts[r][k])
            streams.append(got[r][o:o + int(sizes_of[r][k])])
        try:
            if hasattr(ds, "add_samples_records"):       # the round's samples are packed side by side by the library
                ds.add_samples_records(streams, pack_threads)
            else:
                for s in streams:
                    ds.add_sample_records(s)
        except Exception as e:                               # noqa: BLE001
            pending[0] = e

    # Files a rank decodes per round: the caller's (= --threads, what host decode threads want) -- but the DEVICE route inflates a round's files
    # in one launch of one wavefront per BGZF block, and seven files are ~3 000 blocks for 5 000 wavefront slots: as many files as fit three
    # quarters of a batch of the device inflate (the feed of the benchmark's 160 BAMs: 0.65 s at 7 files a round, 0.43 s at 32; round 6).
    # Every rank computes the same number from the same list.  MSNV_FEED_BATCH overrides.
    if os.environ.get("MSNV_FEED_BATCH"):
        batch = max(1, int(os.environ["MSNV_FEED_BATCH"]))
    elif read_many is not None and device_route(need_owner=False):
        try:
            largest = max(os.path.getsize(p) for p in bam_paths) if bam_paths else 0
            room = (int(os.environ.get("MSNV_INFLATE_BATCH_MB", "1024")) << 20) * 3 // 4
            batch = max(batch, min(64, max(1, room // max(1, largest + 32))))
        except OSError:
            pass
    batch = max(1, min(batch, -(-n // _world)))
    pending = [None]                                         # a failure of THIS rank while it appended a round (pack: ENOMEM, EDOMAIN ...)
    rounds = list(deal_samples(n, batch))
    k = 0
    # where the wall seconds of the feed go (metrics): decoding (read + inflate, or a test's / benchmark's generator) apart from dealing,
    # exchanging and packing -- a benchmark whose "decoder" is a synthetic generator reports the two separately
    import time
    split = {"decode_s": 0.0, "deliver_s": 0.0}
    _decode_round, _deliver = decode_round, deliver

    def decode_round(plan_round, hold_contigs=0):            # noqa: F811
        t0 = time.perf_counter()
        try:
            return _decode_round(plan_round, hold_contigs)
        finally:
            split["decode_s"] += time.perf_counter() - t0

    def deliver(*a):                                         # noqa: F811
        t0 = time.perf_counter()
        try:
            return _deliver(*a)
        finally:
            split["deliver_s"] += time.perf_counter() - t0
    if owner is None:
        # The contig owners are fixed by the reference's rule -- whole species, heaviest first, weight = genome length x coverage = aligned
        # bases (createOptimumSplit.py:46-62, which runs AFTER qaCompute has seen every BAM) -- from as many rounds as fit the planning
        # budget: decoded rounds are HELD (MSNV_PLAN_MB of record bytes per rank, default a quarter of the host memory divided by the
        # ranks) and dealt once the owners are known; whatever follows streams through.  A small cohort is planned on all of its reads.
        names, lengths = plan
        local = np.zeros(len(names), dtype=np.uint64)
        budget = int(os.environ.get("MSNV_PLAN_MB", "0")) << 20
        if budget <= 0:
            try:
                budget = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") // 4 // max(1, _world)
            except (OSError, ValueError):
                budget = 1 << 30
        if read_many is not None and device_route(need_owner=False):      # rounds held in HBM (hold_files_on_device): a third of what is free there at most
            import torch
            budget = min(budget, int(torch.cuda.mem_get_info()[0]) // 3)
        held, held_bytes = [], 0
        while k < len(rounds):
            base, plan_round = rounds[k]
            mine, decoded, failure = decode_round(plan_round, hold_contigs=len(names))
            if failure is None and isinstance(decoded, HeldOnDevice):
                local += decoded.bases                          # (counted by the kernel that measured the streams)
            elif failure is None:
                try:
                    if len(decoded) > 1:                     # (one walk over every record's CIGAR: the round's samples side by side, the library releases the GIL)
                        from concurrent.futures import ThreadPoolExecutor
                        with ThreadPoolExecutor(max_workers=min(len(decoded), max(1, batch))) as ex:
                            for part in ex.map(lambda rec: core.contig_bases(rec, len(names)), decoded):
                                local += part
                    else:
                        for rec in decoded:
                            core.contig_bases(rec, len(names), into=local)
                except Exception as e:                       # noqa: BLE001
                    failure, decoded = e, []
            held.append((base, plan_round, mine, decoded, failure))
            held_bytes += int(decoded.sizes.sum()) if isinstance(decoded, HeldOnDevice) else sum(int(r.size) for r in decoded)
            k += 1
            stop = gather_fixed(np.array([1 if (held_bytes > budget or failure is not None) else 0], dtype=np.int64))
            if any(int(x[0]) for x in stop):
                break
        total = sum(a.astype(np.uint64) for a in gather_fixed(local))
        owner = shard_contigs(names, lengths, _world, contig_bases=total)
        if _world > 1 and hasattr(ds, "set_contig_mask"):
            ds.set_contig_mask([o == _rank for o in owner])
        if metrics is not None:
            metrics["owner"] = list(owner)
            metrics["first_round_bases"] = int(total.sum())
            metrics["plan_rounds"] = len(held)
            metrics["plan_samples"] = int(sum(len(h[1]) for h in held))
        while held:
            deliver(*held.pop(0))
    # The rounds that stream through (round 5): round k + 1 is DECODED WHILE round k is dealt, exchanged and packed -- on a thread of its own
    # when the decoder stays on the host (a test's / benchmark's read_records, MSNV_INFLATE=host: the library releases the GIL and touches
    # nothing of the context), and as a read-ahead hint for the files of the next round when it does not: the device route inflates and
    # deals with the context's own staging buffers and stream, which the pack of round k is using (one context, one thread at a time).
    def read_ahead(plan_round):
        if read_records is not None:
            return
        for i, r in plan_round:
            if r != _rank:
                continue
            try:
                fd = os.open(bam_paths[i], os.O_RDONLY)
                try:
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_WILLNEED)
                finally:
                    os.close(fd)
            except (OSError, AttributeError):
                pass
    host_decoder = read_records is not None or os.environ.get("MSNV_INFLATE", "device")[:1] == "h"
    want_overlap = feed_overlap if feed_overlap is not None else os.environ.get("MSNV_FEED_OVERLAP", "1") != "0"      # (an argument of the run; the environment is the default)
    overlap = host_decoder and want_overlap and len(rounds) - k > 1
    # ... and on the DEVICE route (round 6): the upload, inflate, CRC check and dealing of round k + 1 run on a second context of the device
    # (core.Dataset.set_feed_context: its own stream, staging buffers and pinned words) from a thread of its own, under the exchange and the
    # pack of round k on the dataset's context -- the inflate kernel is bound by latency per symbol and leaves most of the chip to the pack
    feed_ctx = None
    if not overlap and want_overlap and len(rounds) - k > 1 and read_many is not None and device_route() and hasattr(ds, "set_feed_context"):
        try:
            feed_ctx = core.Context(ds.ctx.device)
            ds.set_feed_context(feed_ctx)
            overlap = True
        except Exception:                                        # noqa: BLE001 -- no second context: one round after the other, as before
            feed_ctx = None
    if overlap:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as ahead:
            nxt = ahead.submit(decode_round, rounds[k][1])
            while k < len(rounds):
                base, plan_round = rounds[k]
                got = nxt.result()
                nxt = ahead.submit(decode_round, rounds[k + 1][1]) if k + 1 < len(rounds) else None
                deliver(base, plan_round, *got)
                k += 1
    while k < len(rounds):
        base, plan_round = rounds[k]
        if k + 1 < len(rounds):
            read_ahead(rounds[k + 1][1])
        deliver(base, plan_round, *decode_round(plan_round))
        k += 1
    if feed_ctx is not None:
        ds.set_feed_context(None)
        feed_ctx.close()
    if metrics is not None:
        metrics.update(split)
        metrics["decode_overlapped"] = bool(overlap)
        metrics["decode_on_second_context"] = feed_ctx is not None
    agree(pending[0], "the last round of records was appended")
    allstats = gather_fixed(stats)
    stats = np.maximum.reduce(allstats) if len(allstats) > 1 else stats      # every row is non-zero on exactly one rank
    if metrics is not None:
        metrics["inflated_record_bytes"] = inflated
        if "owner" not in metrics and owner is not None:
            metrics["owner"] = list(owner)
    return stats


# ------------------------------------------------------------------------------------ gather of the result tables
def _merge_cells(parts, want_ann):
    """parts: per rank (sites, counts, cell_sample, cells, ann or None), each in (tid, pos) order -> merged in (tid, pos) order."""
    from .core import SITE_DTYPE, SAMPLE_DTYPE, ANN_DTYPE
    sites = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, SITE_DTYPE)
    counts = np.concatenate([p[1] for p in parts]).astype(np.int64) if parts else np.zeros(0, np.int64)
    cell_sample = np.concatenate([p[2] for p in parts]) if parts else np.zeros(0, np.uint32)
    cells = np.concatenate([p[3] for p in parts]) if parts else np.zeros(0, SAMPLE_DTYPE)
    order = np.lexsort((sites["pos"], sites["tid"]))
    old_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    new_counts = counts[order]
    new_off = np.concatenate([[0], np.cumsum(new_counts)]).astype(np.int64)
    # cell j of new row i comes from old cell old_off[order[i]] + j
    idx = np.repeat(old_off[order] - new_off[:-1], new_counts) + np.arange(int(new_off[-1]), dtype=np.int64)
    ann = None
    if want_ann:
        ann = np.concatenate([p[4] for p in parts])[order] if parts else np.zeros(0, ANN_DTYPE)
    return sites[order].copy(), new_off.astype(np.uint64), cell_sample[idx], cells[idx], ann


def gather_sites_root(sites, row_off, cell_sample, cells, first_line, ann=None, stats=None):
    """The called-site records of every rank, in the cell form, gathered TO RANK 0 and merged in (tid, pos) order.
    first_line: this rank's (tid, pos) of the first pileup line, tid = -1 if none.  Only the globally first line is the one
    the reference drops (call_vC.cpp:423), so the `dropped` mark of every other rank-local first line is cleared.
    Rank 0 returns (sites, row_off, cell_sample, cells, ann or None, gfirst); the other ranks (None, ..., gfirst).
    What travels per site: 32 B + 4 B (cell count) [+ 36 B annotation] + 14 B per non-empty (site, sample) cell."""
    from .core import SITE_DTYPE, SAMPLE_DTYPE, ANN_DTYPE
    sites = np.ascontiguousarray(sites, dtype=SITE_DTYPE)
    counts = np.diff(np.asarray(row_off).astype(np.int64)).astype(np.uint32)
    head = np.array([len(sites), int(np.asarray(row_off)[-1]) if len(row_off) else 0, 1 if ann is not None else 0], dtype=np.int64)
    pieces = [head.view(np.uint8), sites.view(np.uint8).reshape(-1), counts.view(np.uint8),
              np.ascontiguousarray(cell_sample, dtype=np.uint32).view(np.uint8), np.ascontiguousarray(cells, dtype=SAMPLE_DTYPE).view(np.uint8).reshape(-1)]
    if ann is not None:
        pieces.append(np.ascontiguousarray(ann, dtype=ANN_DTYPE).view(np.uint8).reshape(-1))
    firsts = gather_fixed(np.array([first_line[0], first_line[1]], dtype=np.int64))
    cand = [(int(f[0]), int(f[1])) for f in firsts if f[0] >= 0]
    gfirst = min(cand) if cand else (-1, -1)
    got = gather_to_root(np.concatenate(pieces), stats)
    if got is None:
        return None, None, None, None, None, gfirst
    parts = []
    for b in got:
        ns, nc, has_ann = (int(x) for x in b[:24].view(np.int64))
        o = 24
        s = b[o:o + ns * SITE_DTYPE.itemsize].view(SITE_DTYPE); o += ns * SITE_DTYPE.itemsize
        c = b[o:o + 4 * ns].view(np.uint32); o += 4 * ns
        cs = b[o:o + 4 * nc].view(np.uint32); o += 4 * nc
        ce = b[o:o + nc * SAMPLE_DTYPE.itemsize].view(SAMPLE_DTYPE); o += nc * SAMPLE_DTYPE.itemsize
        a = b[o:o + ns * ANN_DTYPE.itemsize].view(ANN_DTYPE) if has_ann else None
        parts.append((s, c, cs, ce, a))
    m_sites, m_off, m_cs, m_cells, m_ann = _merge_cells(parts, ann is not None)
    for i in np.nonzero(m_sites["dropped"])[0]:
        if (int(m_sites["tid"][i]), int(m_sites["pos"][i])) != gfirst:
            m_sites["dropped"][i] = 0
    return m_sites, m_off, m_cs, m_cells, m_ann, gfirst


class SparseAcc:
    """Coverage accumulators [n_samples][n_contigs][COV_WORDS] held as their non-zero (sample, contig) rows; acc[i] is sample
    i's dense [n_contigs][COV_WORDS] table (what core.write_coverage_records takes)."""

    def __init__(self, n_samples, n_contigs, sample, contig, rows):
        from .core import COV_WORDS
        order = np.lexsort((contig, sample))
        self.n_samples, self.n_contigs, self.words = n_samples, n_contigs, COV_WORDS
        self.sample, self.contig, self.rows = sample[order], contig[order], rows[order]
        self.start = np.searchsorted(self.sample, np.arange(n_samples + 1))

    def __len__(self):
        return self.n_samples

    def __getitem__(self, i):
        out = np.zeros((self.n_contigs, self.words), dtype=np.uint64)
        lo, hi = int(self.start[i]), int(self.start[i + 1])
        np.add.at(out, self.contig[lo:hi], self.rows[lo:hi])          # (a contig is owned by one rank: one row per (sample, contig))
        return out

    def dense(self):
        return np.stack([self[i] for i in range(self.n_samples)]) if self.n_samples else np.zeros((0, self.n_contigs, self.words), np.uint64)


def gather_coverage_root(acc, stats=None, rows=None, shape=None):
    """Coverage accumulators of every rank -> SparseAcc of the whole job on rank 0 (None elsewhere): only the (sample, contig) rows
    that hold something travel -- 8 B of index + 136 B each.  rows = (sample[n], contig[n], acc[n][COV_WORDS]) with shape =
    (n_samples, n_contigs) (core.Dataset.coverage_rows: what the device keeps), or acc = the dense [n_samples][n_contigs][COV_WORDS]
    table of this rank (zeros outside its contigs)."""
    from .core import COV_WORDS
    if rows is not None:
        nz_s, nz_c, rr = (np.asarray(rows[0], dtype=np.uint32), np.asarray(rows[1], dtype=np.uint32), np.ascontiguousarray(rows[2], dtype=np.uint64))
        S, NC = shape
        W = COV_WORDS
    else:
        acc = np.ascontiguousarray(acc, dtype=np.uint64)
        S, NC, W = acc.shape
        nz_s, nz_c = np.nonzero(acc.any(axis=2))
        rr = acc[nz_s, nz_c]
    blob = np.concatenate([np.array([len(nz_s)], dtype=np.int64).view(np.uint8), nz_s.astype(np.uint32).view(np.uint8),
                           nz_c.astype(np.uint32).view(np.uint8), np.ascontiguousarray(rr).view(np.uint8).reshape(-1)])
    got = gather_to_root(blob, stats)
    if got is None:
        return None
    ss, cc, rws = [], [], []
    for b in got:
        n = int(b[:8].view(np.int64)[0])
        o = 8
        ss.append(b[o:o + 4 * n].view(np.uint32)); o += 4 * n
        cc.append(b[o:o + 4 * n].view(np.uint32)); o += 4 * n
        rws.append(b[o:o + 8 * W * n].view(np.uint64).reshape(n, W))
    return SparseAcc(S, NC, np.concatenate(ss).astype(np.int64), np.concatenate(cc).astype(np.int64), np.concatenate(rws))


def resident_project_run(ctx, first_bam, fasta_path, bam_paths, params, batch=1, want_coverage=True, ann_path=None,
                         after_coverage=None, species_weight=None, make_dataset=None, read_records=None, run_passes=None, inflate_after_context=False, feed_overlap=None):
    """ONE resident dataset per rank for a whole metaSNV.py run, whatever the number of ranks and splits (the reference forks
    one qaCompute process per BAM, metaSNV.py:55-78, and one `mpileup | snpCall` process per split, :196-221, each of which
    inflates every BAM again): contigs are sharded over the ranks by species (LPT on length x coverage), the BAMs are dealt to
    the ranks for decoding and their records exchanged (feed_sharded), every rank runs coverage and then calling [+ annotation]
    over its contigs, and RANK 0 receives the coverage rows and the merged site records (it writes the files).

    species_weight {species: summed coverage} fixes the owners up front (a previous run's all_cov.tab, --use_prev_cov); without
    it they are fixed after the first decode round (feed_sharded).  make_dataset() / read_records(path) replace the BAM-file
    dataset and reader (bench.py's strong-scaling mode feeds synthetic record streams through this very function);
    run_passes(ds) -> stats replaces the single ds.run() (bench.py times K passes there).

    after_coverage(result) is called on every rank between the two passes -- the driver writes cov/, the tables and the
    split plan there -- and must contain its own barriers.  Returns a dict:
      names, lengths, n_samples, stats[n_samples][6], acc (SparseAcc on rank 0, None elsewhere / without coverage),
      sites / row_off / cell_sample / cells / ann (merged, (tid, pos) order, `dropped` marks the global first line; rank 0 only),
      first_any / first_from1 (per contig, see core.Dataset.first_lines), metrics."""
    import time
    from . import core
    # ctx may be a zero-argument callable that returns the context when it is first needed (cli.py creates it on a thread of its own:
    # the HIP runtime takes ~0.5 s to come up, which a one-process run spends reading and packing BAMs with host threads)
    lazy_ctx = callable(ctx)
    t_open = time.perf_counter()
    ds = make_dataset() if make_dataset else core.Dataset.from_files(None if lazy_ctx else ctx, first_bam, fasta_path, params)
    t_open = time.perf_counter() - t_open
    names, lengths = ds.names, ds.lengths
    owner = None
    if species_weight is not None or _dist is None:
        owner = shard_contigs(names, lengths, _world, species_weight)
        if _world > 1:
            ds.set_contig_mask([o == _rank for o in owner])
    metrics = {"rank": _rank, "world": _world, "open_dataset_s": t_open}
    res = {"names": names, "lengths": lengths, "n_samples": len(bam_paths), "metrics": metrics, "acc": None}
    try:
        t0 = time.perf_counter()
        res["stats"] = feed_sharded(ds, bam_paths, owner, params.cov_min_mapq, batch, read_records=read_records, metrics=metrics, plan=(names, lengths),
                                    ctx_when_ready=ctx if (lazy_ctx and inflate_after_context) else None, feed_overlap=feed_overlap)
        owner = metrics.pop("owner", owner)
        res["owner"] = owner
        metrics["contigs"] = int(sum(1 for o in owner if o == _rank))
        metrics["feed_s"] = time.perf_counter() - t0
        if lazy_ctx and getattr(ds, "ctx", None) is None:
            t0 = time.perf_counter()
            ds.attach_context(ctx())
            metrics["wait_for_context_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        err = None
        try:
            metrics["dataset"] = ds.finalize()
        except Exception as e:                               # noqa: BLE001 -- every rank learns of it before the next collective
            err = e
        agree(err, "the datasets were finalized")
        metrics["finalize_s"] = time.perf_counter() - t0
        if res["stats"] is None:                           # staged streams: packed by finalize
            res["stats"] = np.stack([ds.sample_stats(i) for i in range(len(bam_paths))]) if bam_paths else np.zeros((0, len(core.STATS_FIELDS)), np.uint32)
        if hasattr(ds, "pack_stats"):
            metrics["pack_on_device"] = ds.pack_stats()
        gstats = {}
        if want_coverage:
            t0 = time.perf_counter()
            metrics["coverage"] = ds.coverage_run()
            metrics["coverage_run_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            res["acc"] = gather_coverage_root(None, gstats, rows=ds.coverage_rows(), shape=(len(bam_paths), len(names)))
            metrics["gather_coverage_s"] = time.perf_counter() - t0
        if after_coverage:
            after_coverage(res)
        err = None
        t0 = time.perf_counter()
        try:
            metrics["pileup"] = run_passes(ds) if run_passes else ds.run()
        except Exception as e:                               # noqa: BLE001
            err = e
        agree(err, "the calling pass ran")
        metrics["calling_pass_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        sites, row_off, cell_sample, cells = ds.results_cells()
        ann = None
        if ann_path and fasta_path:
            ann, _ = ds.annotate(ann_path, fasta_path)
        metrics["sites_local"] = int(len(sites)); metrics["cells_local"] = int(len(cells))
        res["sites"], res["row_off"], res["cell_sample"], res["cells"], res["ann"], res["gfirst"] = \
            gather_sites_root(sites, row_off, cell_sample, cells, ds.first_line(), ann, gstats)
        metrics["gather_sites_s"] = time.perf_counter() - t0
        metrics["gather_bytes_received"] = gstats.get("bytes_received", 0)
        fa, f1 = ds.first_lines()                      # -1 outside this rank's contigs: the owner's value is the maximum
        res["first_any"] = np.maximum.reduce(gather_fixed(fa))
        res["first_from1"] = np.maximum.reduce(gather_fixed(f1))
        # per-rank inflated bytes and pileup bases for the report (the host stage shards with the ranks)
        infl = gather_fixed(np.array([metrics.get("inflated_record_bytes") or 0, metrics["dataset"]["n_pileup_bases"]], dtype=np.int64))
        metrics["inflated_record_bytes_per_rank"] = [int(x[0]) for x in infl]
        metrics["pileup_bases_per_rank"] = [int(x[1]) for x in infl]
    finally:
        t0 = time.perf_counter()
        ds.close()
        metrics["close_dataset_s"] = time.perf_counter() - t0
    return res


def split_view(res, split_contigs):
    """The records of one best_split_K invocation (`samtools mpileup -l best_split_K ... | snpCall`, metaSNV.py:160-176) cut
    out of a whole-run result: sites on the split's contigs at 0-based positions >= 1 (the split file's lines are
    `name 1 LEN`, parsed as BED [1, LEN): metaSNV.py:92), with the split's OWN first pileup line marked as dropped
    (call_vC.cpp:423; lines come in BAM-header order whatever the order of the split file).
    Returns (sites, row_off, cell_sample, cells, ann)."""
    tid_of = {n: i for i, n in enumerate(res["names"])}
    tids = sorted(tid_of[n] for n in split_contigs if n in tid_of)          # samtools ignores names that are not in the header
    sites, ann = res["sites"], res["ann"]
    keep = np.isin(sites["tid"], np.array(tids, dtype=np.int64)) & (sites["pos"] >= 1)
    s = sites[keep].copy()
    first = next(((t, int(res["first_from1"][t])) for t in tids if res["first_from1"][t] >= 0), (-1, -1))
    s["dropped"] = ((s["tid"] == first[0]) & (s["pos"] == first[1])).astype(np.uint8)
    counts = np.diff(res["row_off"].astype(np.int64))
    cell_keep = np.repeat(keep, counts)
    row_off = np.concatenate([[0], np.cumsum(counts[keep])]).astype(np.uint64)
    return s, row_off, res["cell_sample"][cell_keep], res["cells"][cell_keep], (ann[keep] if ann is not None else None)


def sharded_call(ctx, names, lengths, seqs, add_samples, params=None, species_weight=None,
                 called_path=None, indiv_path=None, ann_path=None, fasta_path=None):
    """One SNV-calling pass over all contigs on all ranks: every rank packs only its shard's
    reads (contig mask), runs the kernels, and rank 0 receives the records and writes the files.
    add_samples(dataset) must append every sample in all_samples order.  Returns (sites, dense per-sample records, info, stats)
    on rank 0 and (None, None, info, stats) elsewhere."""
    from . import core
    owner = shard_contigs(names, lengths, _world, species_weight)
    ds = core.Dataset(ctx, names, lengths, seqs, params)
    ds.set_contig_mask([o == _rank for o in owner])
    add_samples(ds)
    info = ds.finalize()
    stats = ds.run()
    sites, row_off, cell_sample, cells = ds.results_cells()
    ann = None
    if ann_path and fasta_path:                      # codon annotation runs on every rank's device, records are gathered
        ann, _ = ds.annotate(ann_path, fasta_path)
    m_sites, m_off, m_cs, m_cells, m_ann, gfirst = gather_sites_root(sites, row_off, cell_sample, cells, ds.first_line(), ann)
    dense = None
    if _rank == 0:
        if called_path:
            core.write_calls_cells(names, ds.n_samples, m_sites, m_off, m_cs, m_cells, called_path, indiv_path, ann_path, fasta_path, m_ann)
        dense = core.cells_to_dense(ds.n_samples, m_off, m_cs, m_cells)
    ds.close()
    return m_sites, dense, info, stats
