#!/usr/bin/env python
"""bench.py -- pileup Gbases/s of the MI355X-native SNV-calling hot path (BASELINE.json metric).

One "step" = one pass of the hot path (pileup histogram -> gates -> per-sample gather -> calling
rule) over the synthetic "testdata"-shaped batch (BASELINE.json configs[1]: 160 BAMs x 3
refGenomes), with the packed read columns already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: contigs shard across ranks with no data-path collective.  With the plain argv and N > 1 the line that is
printed is the STRONG one -- one cohort (the generator of BASELINE configs[3]'s per-GPU shard, 500 samples) through the product's N-rank
path, value = cohort bases / the slowest rank's pass, "scaling": "strong" -- and the weak replica line rides along as "weak_replicas"; at
N = 1 the line is the metric's own configuration (BASELINE configs[1]).
  --mode weak   every rank holds its own testdata-shaped shard -- three species cannot be dealt to eight ranks -- and only a small table
                is gathered after the timed region (the line itself at N = 1; forces the weak line as the headline at N > 1).
  --mode strong (default for config3 / config4shard): ONE fixed multi-species cohort through the product's N-rank path
                (metasnv_amd/parallel.py: resident_project_run) -- record streams "decoded" by one rank each, contig owners by
                species LPT on length x coverage from the first round, all-to-all of the records over RCCL, one dataset per rank,
                K timed passes, then the gather of coverage rows and cell-form site records to rank 0 -- with per-rank bases,
                imbalance, exchange / gather seconds and bytes beside the kernel line.  With N > 1 the weak line also carries a
                small strong-scaling block (--no-strong-extra skips it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


# Synthetic workloads.  testdata: 160 samples x 3 refGenomes x 300 kb, ~10x, 10 % of the (sample, species) pairs
# absent.  config3 (BASELINE configs[2]): 100 species x (1-50 contigs, 3 Mbp), 160 samples, each carries 10 random species at
# LogNormal(ln 10, 0.7)x.  config4shard (one GPU's share of BASELINE configs[3]): 1500 species x ~2.07 Mbp (3.1 G positions), 500 samples,
# each carries 20 of the 12 000 species = 2.5 of this shard's, at 5x.
WORKLOADS = {
    "testdata": dict(n_species=3, contig_len=300000, n_samples=160, mean_cov=10.0),
    "config3": dict(n_species=100, contig_len=3000000, n_samples=160, mean_cov=10.0, sigma_cov=0.7, contigs_per_species_max=50, species_per_sample=10, frac_absent=0.0),
    "config4shard": dict(n_species=1500, contig_len=2070000, n_samples=500, mean_cov=5.0, sigma_cov=0.3, contigs_per_species_max=20, species_per_sample=2.5, frac_absent=0.0),
}


def workload_params(a, rank=0):
    kw = dict(WORKLOADS[a.workload])
    label = {"testdata": "testdata shape (BASELINE configs[1])", "config3": "ProGenomes2-subset shape (BASELINE configs[2])",
             "config4shard": "one GPU's contig shard of BASELINE configs[3]"}[a.workload]
    if a.workload != "testdata":
        scale = a.scale if a.scale is not None else 0.25
        kw["n_species"] = max(2, int(round(kw["n_species"] * scale)))
        want = kw["species_per_sample"] * scale                 # the same share of the species per sample, as an expectation:
        kw["species_per_sample"] = max(1, int(-(-want // 1)))    # draw ceil(want) species and keep each with probability want / ceil(want)
        kw["frac_absent"] = 1.0 - want / kw["species_per_sample"]
        label += " at %g of its species" % scale
    for arg, key in (("samples", "n_samples"), ("contig_len", "contig_len"), ("species", "n_species"), ("mean_cov", "mean_cov")):
        if getattr(a, arg) is not None:
            kw[key] = getattr(a, arg)
    kw["seed"] = 1 + rank
    if a.read_len != 100:
        kw["read_len"] = a.read_len
    if a.error_rate is not None:
        kw["error_rate"] = a.error_rate
    return kw, label


TRAFFIC_PROFILES = ["r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03fin_pmc.json", "r03_pmc.json", "r02d_pmc.json"]       # newest first; written by profiles/collect.sh + summarize.py


def measured_traffic(samples, species, contig_len, mean_cov):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 read correction applied) and the file they come from.
    Only valid for the workload it was collected on; anything else reports null.  The counters cannot be collected inside
    this process (rocprofv3 wraps the command), so the figure is the one of the committed profile of THIS kernel build."""
    if (samples, species, contig_len, mean_cov) != (160, 3, 300000, 10.0):
        return None, None
    for name in TRAFFIC_PROFILES:
        try:
            p = json.load(open(os.path.join(ROOT, "profiles", name)))
            return p["hbm_traffic"]["total_bytes_per_launch"], "profiles/" + name
        except Exception:
            continue
    return None, None


def pileup_counters(samples, species, contig_len, mean_cov, kernel_ms):
    """What limits the dominant kernel, COMPUTED from the counter passes committed under profiles/ (the newest one that holds SQ_INSTS_VALU of
    msnv_pileup_tiles_narrow32): vector-instruction issue share = wavefront instructions x 4 cycles / (1024 SIMDs x this run's kernel time at
    2.4 GHz).  Only for the workload the counters were collected on; None otherwise."""
    if (samples, species, contig_len, mean_cov) != (160, 3, 300000, 10.0) or not kernel_ms:
        return None
    for name in TRAFFIC_PROFILES:
        try:
            c = json.load(open(os.path.join(ROOT, "profiles", name)))["counters"]
            k = [v for kk, v in c.items() if "pileup_tiles_narrow32" in kk][0]
            valu = k["SQ_INSTS_VALU"]["avg_per_launch"]
        except Exception:
            continue
        share = valu * 4.0 / (1024.0 * kernel_ms * 1e-3 * 2.4e9)
        out = {"source": "profiles/" + name, "valu_wave_instructions_per_launch": valu, "valu_issue_share_of_this_run": share,
               "limited_by": "vector-instruction issue" if share > 0.5 else "HBM / latency"}
        if "SQ_INSTS_LDS" in k:
            out["lds_wave_instructions_per_launch"] = k["SQ_INSTS_LDS"]["avg_per_launch"]
        return out
    return None


class ResidentRecords:
    """The workload's raw alignment-record streams in ONE device buffer (16-byte aligned streams, 256 bytes of room behind the last): where
    the "records resident in HBM -> calls" region starts.  Allocated through torch when this process already runs one (N ranks), else through
    the HIP runtime the library itself is linked to."""

    def __init__(self, syn, n_samples, device):
        import ctypes as C
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 8)) as ex:      # (the generator runs in the library, outside the GIL)
            recs = list(ex.map(syn.sample_records, range(n_samples)))
        self.offsets, self.sizes, o = [], [], 0
        for r in recs:
            self.offsets.append(o); self.sizes.append(int(r.size)); o += (int(r.size) + 15) & ~15
        self.capacity = o + 256
        self.bytes = int(sum(self.sizes))
        if "torch" in sys.modules:
            import torch
            host = np.zeros(self.capacity, dtype=np.uint8)
            for r, off in zip(recs, self.offsets):
                host[off:off + r.size] = r
            self._t = torch.from_numpy(host).to("cuda:%d" % device)
            torch.cuda.synchronize()
            self.ptr = self._t.data_ptr()
        else:
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipSetDevice(device) == 0
            buf = C.c_void_p()
            assert hip.hipMalloc(C.byref(buf), C.c_size_t(self.capacity)) == 0
            for r, off in zip(recs, self.offsets):
                if r.size:
                    assert hip.hipMemcpy(C.c_void_p(buf.value + off), C.c_void_p(r.ctypes.data), C.c_size_t(r.size), 1) == 0
            hip.hipDeviceSynchronize()
            self._hip, self._buf, self.ptr = hip, buf, buf.value
        del recs
        time.sleep(0.4)     # (freeing host memory the runtime has uploaded from stalls the next GPU operation for milliseconds: let that pass, outside every timed region)

    def build(self, core, ctx, syn):
        """A fresh dataset from the resident records: wall milliseconds of the per-read stage (kernels), of finalize, and the library's HIP-event split."""
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        t0 = time.perf_counter()
        ds.add_samples_records_resident(self.ptr, self.capacity, self.offsets, self.sizes)
        t1 = time.perf_counter()
        info = ds.finalize()
        t2 = time.perf_counter()
        return ds, info, 1e3 * (t1 - t0), 1e3 * (t2 - t1)

    def close(self):
        if hasattr(self, "_hip"):
            self._hip.hipFree(self._buf)
        self._t = None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="testdata", choices=sorted(WORKLOADS),
                    help="testdata = BASELINE configs[1] (the metric's configuration, default); config3 / config4shard = the SURVEY.md 8d shapes of "
                         "BASELINE configs[2] / the per-GPU shard of configs[3], scaled by --scale")
    ap.add_argument("--mode", default=None, choices=["weak", "strong"], help="weak: one shard per rank (default for testdata); strong: one fixed cohort sharded over the ranks "
                                                                             "through the product's N-rank path (default for config3 / config4shard)")
    ap.add_argument("--no-strong-extra", action="store_true", help="N > 1, weak mode: skip the small strong-scaling block behind the timed region")
    ap.add_argument("--strong-extra-shape", default=None, help="species,contig_len of the N-rank cohort (a config3-shaped one; tests shrink it); default: the generator of BASELINE configs[3]'s per-GPU shard "
                                                                "(config4shard) at --scale (0.25)")
    ap.add_argument("--scale", type=float, default=None, help="fraction of the named workload's species (config3 / config4shard; default 0.25: the full shapes need ~100 GB of host staging)")
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--contig-len", type=int, default=None)
    ap.add_argument("--species", type=int, default=None)
    ap.add_argument("--mean-cov", type=float, default=None)
    ap.add_argument("--read-len", type=int, default=100, help="synthetic read length (BASELINE: 100)")
    ap.add_argument("--error-rate", type=float, default=None, help="synthetic sequencing error rate (BASELINE: 0.001)")
    ap.add_argument("--cpu-samples", type=int, default=64, help="samples of the workload the CPU oracle is timed on")
    ap.add_argument("--sync-each-step", action="store_true", help="one msnv_pileup_run call (with its host sync) per step instead of one batched call")
    ap.add_argument("--no-overlap-extra", action="store_true", help="skip the extra timed batch with overlapped passes (profiling runs)")
    ap.add_argument("--build-from-host", action="store_true", help="build the dataset the round-4 way (streams made and uploaded group by group inside the build) instead of from records resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-annotation", action="store_true", help="skip the --db_ann codon-annotation kernel (BASELINE configs[4]) after the timed region")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def cpu_baseline(sp_kwargs, n_cpu_samples):
    """The oracle (CPU restatement, kind "port") timed on a bounded sample of the same workload."""
    import orc
    from metasnv_amd import core
    sp = core.synth_params(**sp_kwargs)
    syn = core.Synth(sp)
    samples, nbytes = [], 0
    for i in range(n_cpu_samples):                     # bounded sample: ~10-30 s of single-threaded CPU work
        samples.append(syn.sample_records(i))
        nbytes += samples[-1].size
        if nbytes > 1.3e9:
            break
    n_cpu_samples = len(samples)
    t0 = time.perf_counter()
    pop, ind, n_lines, n_bases = orc.call(syn.names, syn.lengths, syn.seqs, samples)
    dt = time.perf_counter() - t0
    out = {"value": n_bases / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
           "what": "port (oracle): the repository's plain-C restatement of samtools mpileup + snpCall (oracle/), NOT the reference binaries -- "
                   "boost / htslib / samtools are absent from the image, so call_vC.cpp and samtools cannot be built or run here",
           "sample": "first %d of %d samples, all %d contigs: %d pileup bases in %.1f s (mpileup+snpCall restatement, 1 thread)"
                     % (n_cpu_samples, sp.n_samples, len(syn.names), n_bases, dt),
           "called_lines": pop.count("\n")}
    try:
        out["all_cores"] = cpu_baseline_all_cores(orc, syn, samples, n_bases)
    except Exception as e:                             # the extra must never cost the bench line
        out["all_cores"] = {"error": repr(e)}
    try:
        out["snpcall_alone"] = snpcall_alone(syn, samples[:32])
    except Exception as e:                             # the extra must never cost the bench line
        out["snpcall_alone"] = {"error": repr(e)}
    return out


def cpu_baseline_all_cores(orc, syn, samples, n_bases_whole):
    """the same restatement on ALL host cores, the way the reference itself uses several -- one `samtools mpileup -l
    SPLIT | snpCall` per split, every split reading every BAM (/root/reference/metaSNV.py:196-215): the contigs are cut into one position range
    per core, each range is one thread's orc.call with that range as its BED (the C restatement holds no global state; ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from metasnv_amd import core
    # (the cores' worth of CPU time the process may use -- the container's cgroup quota, 16 on the pool's boxes whatever os.cpu_count() says: 256
    # threads on 16 cores, each split walking every BAM, ran at HALF the one-core rate, profiles/r06_bench.json of the first closing run)
    cores = max(1, min(os.cpu_count() or 1, core.host_cores()))
    total = sum(syn.lengths)
    per = max(1, -(-total // cores))
    beds = []
    for tid, L in enumerate(syn.lengths):
        for b in range(0, L, per):
            beds.append((tid, b, min(L, b + per)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(lambda bed: orc.call(syn.names, syn.lengths, syn.seqs, samples, bed=[bed])[3], beds))
    dt = time.perf_counter() - t0
    return {"value": sum(res) / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "port", "splits": len(beds), "seconds": dt,
            "sample": "the same %d samples, %d position ranges (BED splits like metaSNV.py --n_splits), one thread each: %d pileup bases (whole run: %d)" % (len(samples), len(beds), sum(res), n_bases_whole)}


def snpcall_alone(syn, samples, n_pos=100000):
    """snpCall ALONE on mpileup text -- the stage BASELINE.md section 2's figure is about: the oracle's restatement of call_vC.cpp on
    text rendered from a bounded slice (<= 32 samples, the first n_pos positions of contig 0), one thread, as a process reading
    stdin like the reference; beside it the product's text entry (msnv_call_from_mpileup: the text parsed and called on the
    device) on the same bytes, and whether the two outputs are identical."""
    import tempfile
    import orc
    from metasnv_amd import core
    bed = [(0, 0, min(syn.lengths[0], n_pos))]
    text = orc.mpileup_text(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
    n_bases = orc.call(syn.names, syn.lengths, syn.seqs, samples, bed=bed)[3]
    t0 = time.perf_counter()
    rc, pop, ind, err = orc.snpcall_text(text)
    dt = time.perf_counter() - t0
    ctx = core.Context(0)
    try:
        with tempfile.TemporaryDirectory() as td:
            pp, ip = os.path.join(td, "c"), os.path.join(td, "i")
            core.call_from_mpileup(ctx, pp, ip, text=text)                          # warm-up (allocations, clocks)
            t0 = time.perf_counter()
            st = core.call_from_mpileup(ctx, pp, ip, text=text)
            wall = time.perf_counter() - t0
            same = rc == 0 and open(pp).read() == pop and open(ip).read() == ind
    finally:
        ctx.close()
    return {"sample": "%d samples x %d positions of contig 0: %.1f MB of mpileup text, %d pileup bases" % (len(samples), bed[0][2], len(text) / 1e6, n_bases),
            "cpu": {"value": n_bases / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port", "text_MB_per_s": len(text) / dt / 1e6, "seconds": dt},
            "device": {"kernel_ms": st["kernel_ms"], "text_GB_per_s_kernel": st["text_bytes"] / max(st["kernel_ms"], 1e-9) / 1e6,
                       "Gbases_per_s_kernel": n_bases / max(st["kernel_ms"], 1e-9) / 1e6, "wall_s_with_transfer_and_output": wall,
                       "Gbases_per_s_wall": n_bases / wall / 1e9},
            "identical_output": bool(same)}


def end_to_end(sp, syn, n_bases, host_threads, cpu_rate_gbases=None):
    """BAM files -> project directory through the LAUNCHER with the reference's argv (metaSNV.py DIR all_samples REF --threads T; the
    driver this replaces: /root/reference/metaSNV.py:224-292): every sample of the workload written as a BAM file, one process run,
    wall seconds and what they are made of (the library's host-stage timers + the driver's own)."""
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from metasnv_amd import core
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "ref.fa")
        syn.write_fasta(fa)
        paths = [os.path.join(td, "s%04d.insilicoRefs.unique.sorted.bam" % i) for i in range(sp.n_samples)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
            list(ex.map(lambda i: core.write_bam(paths[i], syn.names, syn.lengths, syn.sample_records(i)), range(sp.n_samples)))
        t_write = time.perf_counter() - t0
        lst, met, proj = os.path.join(td, "all_samples"), os.path.join(td, "metrics.jsonl"), os.path.join(td, "proj")
        open(lst, "w").write("\n".join(paths) + "\n")
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["MSNV_METRICS"] = met
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "metaSNV.py"), proj, lst, fa, "--threads", str(host_threads)],
                           env=env, capture_output=True, text=True, timeout=1800)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stdout[-500:] + r.stderr[-1500:])}
        m = json.loads(open(met).read().strip().splitlines()[-1])
        ht, cw = m.get("host_timers", {}), m.get("cli_wall", {})
        import glob
        called = sum(sum(1 for _ in open(f)) for f in glob.glob(os.path.join(proj, "snpCaller", "called_SNPs*")))   # (--threads T makes T splits: metaSNV.py:275-276)
        out = {"bams": sp.n_samples, "bam_bytes": sum(os.path.getsize(x) for x in paths), "write_bams_s_not_counted": t_write, "host_threads": host_threads,
               "argv": "metaSNV.py DIR all_samples REF --threads %d (= %d best_split outputs, metaSNV.py:275-276)" % (host_threads, host_threads),
               "wall_s": wall, "Gbases_per_s": n_bases / wall / 1e9, "called_SNPs_lines": called,
               "split_wall_s": {"process_start_hip_runtime_and_context": max(0.0, wall - cw.get("total_s", 0.0)),
                                "decode_and_pack": ht.get("add_bams_wall_s"), "pack_on_device_incl_upload_of_records": ht.get("pack_device_wall_s"), "finalize_index_and_upload": ht.get("finalize_upload_wall_s"),
                                "kernels_coverage_ms": m["coverage"]["ms_coverage"] if "coverage" in m else None, "kernels_pileup_pass_ms": m["pileup"]["ms_total"],
                                "coverage_files": cw.get("coverage_files_s"), "tables_and_splits": cw.get("tables_and_splits_s"), "calls_text": cw.get("calls_text_s")},
               "thread_seconds": {"file_read": ht.get("read_s"), "inflate_host_and_crc": ht.get("inflate_host_s"), "parse_and_pack": ht.get("pack_s")},
               "device_inflate_wall_s": ht.get("inflate_device_wall_s"), "pack_on_device": m.get("pack_on_device")}
        if cpu_rate_gbases:
            out["cpu_baseline_extrapolated_s"] = {"seconds": n_bases / (cpu_rate_gbases * 1e9), "what": "the oracle's mpileup + snpCall restatement, 1 thread, at its measured rate on all %d samples (qaCompute not included)" % sp.n_samples}
        return out


def synth_annotation(syn, path, seed=7):
    """SURVEY.md section 8d annotation shape: CDS of 300-3000 bp (90 % a multiple of 3), 50 % on the '-' strand,
    ~85 % coding density, 5 % of the genes overlapping their predecessor."""
    import random
    rnd = random.Random(seed)
    n = 0
    with open(path, "w") as f:
        f.write("gene_id\texternal_id\tsequence_id\ttype\tinfo\tlength\tstart\tend\tstrand\tsc\tstop\tgc\n")
        for name, length in zip(syn.names, syn.lengths):
            p = 1
            while True:
                glen = rnd.randrange(100, 1000) * 3 if rnd.random() < 0.9 else rnd.randrange(300, 3000)
                start = max(1, p - rnd.randrange(10, 200)) if rnd.random() < 0.05 else p + rnd.randrange(0, int(glen * 0.35))
                end = start + glen - 1
                if end > length - 3:
                    break
                f.write("%d\tg%06d\t%s\tCDS\tx\t%d\t%d\t%d\t%s\tATG\tTAG\t0.4\n" % (n, n, name, glen, start, end, rnd.choice("+-")))
                n += 1
                p = end + 1
    return n


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(n, argv, port=None):
    """`python bench.py --gpus N` outside torchrun: the N rank processes are started as a CHILD (torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1) before this process has made any HIP / torch.cuda call -- a process that has
    touched the GPU must never exec another program on this pool -- and the parent only forwards the child's exit code."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def strong_run(a, rank, world, local, dist, brief=False):
    """ONE fixed cohort through the product's N-rank path (parallel.resident_project_run, the function metaSNV.py runs under
    torchrun; reference: the split pool, metaSNV.py:196-215 + createOptimumSplit.py:46-62): every rank generates ("decodes") 1/N of
    the samples' record streams, the contig owners come from the first round's length x coverage, the records travel in an
    all-to-all per round, every rank packs and uploads its contigs' share and runs K timed passes (barrier + sync on both sides, max
    over ranks), then the coverage rows and the cell-form site records are gathered to rank 0.  Returns the JSON line (rank 0)."""
    import numpy as np
    from metasnv_amd import core, parallel
    sp_kwargs, wl_label = workload_params(a, 0)            # the same cohort on every rank
    sp = core.synth_params(**sp_kwargs)
    syn = core.Synth(sp)
    ctx = core.Context(local)
    params = core.default_params()
    timing = {}

    def barrier():
        parallel.barrier()

    def run_passes(ds):
        st = ds.run()                                       # sizes the sparse buffers
        for _ in range(a.warmup):
            st = ds.run()
        if not a.sync_each_step:
            ds.reserve_passes(a.steps)
        barrier()
        t0 = time.perf_counter()
        sts = ds.run_many(a.steps) if not a.sync_each_step else [ds.run() for _ in range(a.steps)]
        barrier()
        timing["dt"] = time.perf_counter() - t0
        timing["k_ms"] = sum(x["ms_pileup"] for x in sts) / len(sts)
        timing["alg"] = sts[-1]["algorithmic_bytes"]
        timing["called"] = sts[-1]["n_called_pop"]
        timing["total_ms"] = sum(x["ms_total"] for x in sts) / len(sts)
        return sts[-1]

    threads = a.host_threads or max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    # (the "decoder" of this benchmark is the synthetic generator: run ahead of the pack -- what feed_sharded does with a host decoder -- its
    # threads would sit in the pack stage's event brackets, which this line reports as pack_on_device; MSNV_FEED_OVERLAP=1 to see it overlapped)
    feed_overlap = os.environ.get("MSNV_FEED_OVERLAP", "0") != "0"      # (the environment is read, never written: the choice is an argument of the run and is printed)
    t_all = time.perf_counter()
    res = parallel.resident_project_run(ctx, None, None, [str(i) for i in range(sp.n_samples)], params, batch=threads, want_coverage=True, feed_overlap=feed_overlap,
                                        make_dataset=lambda: core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, params),
                                        read_records=lambda p: syn.sample_records(int(p)), run_passes=run_passes)
    t_all = time.perf_counter() - t_all
    m = res["metrics"]
    info = m["dataset"]
    mine = np.array([timing["dt"], float(info["n_pileup_bases"]), timing["k_ms"], float(timing["alg"]), float(timing["called"]), float(m["sites_local"]),
                     float(m["cells_local"]), m["feed_s"], m["finalize_s"], m.get("gather_coverage_s", 0.0), m["gather_sites_s"], float(info["n_positions"]),
                     float(m.get("inflated_record_bytes") or 0), float(info["device_bytes"]), timing["total_ms"], m["coverage"]["ms_coverage"]], dtype=np.float64)
    allr = parallel.gather_fixed(mine)
    ctx.close()
    if rank != 0:
        return None
    dt_max = max(float(r[0]) for r in allr)
    bases = [float(r[1]) for r in allr]
    gbs = [float(r[3]) / (float(r[2]) * 1e-3) / 1e9 if r[2] > 0 else 0.0 for r in allr]
    # a rank's share of the timed region is its kernel time: the slowest rank is the one with the most bases (imbalance = max / mean)
    slow = max(range(world), key=lambda r: float(allr[r][2]))
    n_sites = int(len(res["sites"]))
    dense_bytes = n_sites * sp.n_samples * 10
    line = {
        "metric": "pileup Gbases/s across all samples",
        "value": sum(bases) * a.steps / dt_max / 1e9, "unit": "Gbases/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_max / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: ONE cohort of %d synthetic BAM-record streams x %d species (%d contigs, %d bp per species), ~%gx, single-end %d bp, sharded over %d rank(s)"
                               % (wl_label, sp.n_samples, sp.n_species, len(syn.names), sp.contig_len, sp.mean_cov, sp.read_len, world),
                   "samples": sp.n_samples, "parallelism": "species LPT (length x first-round coverage) over %d rank(s); all-to-all of records before, gather to rank 0 after; no data-path collective" % world,
                   "pileup_bases_total": int(sum(bases)), "pileup_bases_per_rank": [int(b) for b in bases],
                   "positions_per_rank": [int(r[11]) for r in allr], "called_SNPs_lines_per_rank": [int(r[4]) for r in allr]},
        "imbalance_max_over_mean": max(bases) / (sum(bases) / world) if sum(bases) else None,
        # `value` is the pass alone (columns resident, like the N = 1 line); the same cohort with everything in front of the pass counted ONCE --
        # the slowest rank's feed (generate / decode, deal, all-to-all, per-read stage) and finalize -- plus one pass:
        "feed_inclusive": {"Gbases_per_s": sum(bases) / (max(float(r[7]) for r in allr) + max(float(r[8]) for r in allr) + dt_max / a.steps) / 1e9,
                           "seconds": {"feed_max": max(float(r[7]) for r in allr), "finalize_max": max(float(r[8]) for r in allr), "one_pass": dt_max / a.steps},
                           "what": "cohort bases / (slowest feed + slowest finalize + one pass); the feed of this benchmark is the host's synthetic generator, not a BAM decoder"},
        "roofline": {"bound": "hbm", "kernel": "msnv_pileup_tiles_* (narrow32 + merged [+ wide] between one pair of HIP events)", "achieved": gbs[slow], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": gbs[slow] / HBM_PEAK_GBS, "traffic": None, "traffic_source": None, "rank": slow, "achieved_per_rank": gbs,
                     "kernel_ms_avg": float(allr[slow][2]), "algorithmic_bytes_per_launch": int(allr[slow][3]),
                     "algorithmic_definition": "SURVEY.md 8d: per pileup read 16 B header + 4 B per CIGAR op + 0.5 B/base + 1 B/base quality"},
        "kernel_ms": {"pileup_per_rank": [float(r[2]) for r in allr], "pipeline_total_per_rank": [float(r[14]) for r in allr], "coverage_per_rank": [float(r[15]) for r in allr]},
        "exchange": {"backend": parallel.backend() or "none (one process)", "feed_s_per_rank": [float(r[7]) for r in allr], "feed_overlap": feed_overlap,
                     "what": "generate 1/N of the samples, count first-round bases, partition by owner, all_to_all per round of %d samples x %d ranks, pack" % (threads, world),
                     "record_bytes_decoded_per_rank": [int(r[12]) for r in allr], "finalize_upload_s_per_rank": [float(r[8]) for r in allr]},
        "gather": {"to": "rank 0", "sites_total": n_sites, "cells_total": int(len(res["cells"])), "bytes_received_by_rank0": int(m["gather_bytes_received"]),
                   "dense_form_would_be_bytes": dense_bytes, "seconds_sites_per_rank": [float(r[10]) for r in allr], "seconds_coverage_per_rank": [float(r[9]) for r in allr],
                   "sites_per_rank": [int(r[5]) for r in allr], "cells_per_rank": [int(r[6]) for r in allr]},
        "host": {"wall_s_whole_run": t_all, "device_bytes_per_rank": [int(r[13]) for r in allr], "host_threads_per_rank": threads,
                 # rank 0's feed, split: the synthetic generator stands where the BAM decoder stands in a real run; deal + exchange + pack is the product's stage
                 "feed_decode_s_rank0": m.get("decode_s"), "feed_deal_exchange_pack_s_rank0": m.get("deliver_s"), "pack_on_device_rank0": m.get("pack_on_device")},
    }
    if brief:
        for k in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data"):
            line.pop(k)
    return line


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(launcher_command(a.gpus, sys.argv[1:]), env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is what runs\n" % (a.gpus, world))
    # the process group is the product's own (metasnv_amd/parallel.py: nccl = RCCL; gloo = rehearsal, ranks may share a GPU)
    os.environ["MSNV_DIST_BACKEND"] = a.dist_backend
    from metasnv_amd import parallel
    rank, world, local = parallel.init_from_env()
    dist = parallel._dist
    if world > 1:
        import torch
        torch.cuda.set_device(local)

    from metasnv_amd import core
    if core.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the pileup path has no CPU fallback")
    mode = a.mode or ("weak" if a.workload == "testdata" else "strong")
    if mode == "strong":
        line = strong_run(a, rank, world, local, dist)
        if rank == 0:
            print(json.dumps(line))
        parallel.finalize()
        return

    # ---- build this rank's shard: same shape on every rank, different seed (weak scaling)
    sp_kwargs, wl_label = workload_params(a, rank)
    sp = core.synth_params(**sp_kwargs)
    syn = core.Synth(sp)
    ctx = core.Context(local)
    core.host_timers(reset=True)
    from_records = None
    if os.environ.get("MSNV_PACK", "d")[0] != "h" and not a.build_from_host and a.workload == "testdata":      # (the larger shapes make their streams group by group: they do not fit host memory at once)
        # The dataset is built from RAW RECORDS RESIDENT IN HBM (msnv_dataset_add_sample_records_resident): the streams go up first, outside
        # every timed region; then records -> packed columns -> tile index is timed -- seven builds (the first pays the allocations, the next two still miss the allocator's cache now and then; the median is quoted), the last one is kept for the passes.
        t0 = time.perf_counter()
        rr = ResidentRecords(syn, sp.n_samples, local)
        t_synth = time.perf_counter() - t0
        builds = []
        ds = None
        for rep in range(7):
            if ds is not None:
                ds.close()
            ds, info, pack_ms, fin_ms = rr.build(core, ctx, syn)
            builds.append({"pack_wall_ms": pack_ms, "finalize_wall_ms": fin_ms, "pack_kernel_ms": {k: v for k, v in ds.pack_stats().items() if k.endswith("_ms")}})
        rr.close()
        from_records = {"record_bytes": rr.bytes, "builds": builds}
        t_pack, t_up = builds[-1]["pack_wall_ms"] * 1e-3, builds[-1]["finalize_wall_ms"] * 1e-3
        ht_build = dict(core.host_timers(), synth_wall_s=t_synth)
    else:
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        t0 = time.perf_counter()
        ds.add_synth_samples(sp, 0, sp.n_samples, a.host_threads)
        t_pack = time.perf_counter() - t0
        ht_build = core.host_timers()
        t0 = time.perf_counter()
        info = ds.finalize()
        t_up = time.perf_counter() - t0
    pack = ds.pack_stats()          # the per-read stage as kernels (csrc/devpack.hip): all zero under MSNV_PACK=host

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # (the builds above gave gigabytes of device and pinned memory back; the driver unmaps them in the background and GPU work that runs meanwhile
    # is stalled for milliseconds at a time -- 0.43 vs 0.8-1.3 ms per step, every second run on a fresh box.  Let that pass, outside every timed region)
    time.sleep(0.5)
    ds.run()                               # (the first pass sizes the sparse buffers)
    # (... and the clocks, which fell while the device idled, come back up: untimed batches of 20 passes in front of the W warmup steps, until two in a
    # row take what the fastest one took -- three batches, 25 ms, as a rule; one run in ten needed more)
    best, close = None, 0
    for _ in range(25):
        tb = time.perf_counter(); ds.run_many(20); tb = time.perf_counter() - tb
        best = tb if best is None else min(best, tb)
        close = close + 1 if tb <= 1.02 * best else 0
        if close >= 2 and _ >= 2:
            break
    for _ in range(a.warmup):
        ds.run()
    if not a.sync_each_step:
        ds.reserve_passes(a.steps)         # events + pinned counters of the batch: a one-time allocation, not a step
    barrier()
    t0 = time.perf_counter()
    ms_pileup, ms_total = [], []
    if a.sync_each_step:
        for _ in range(a.steps):
            st = ds.run()                  # blocks until the pass has finished (stream sync inside)
            ms_pileup.append(st["ms_pileup"]); ms_total.append(st["ms_total"])
    else:
        # the K passes are enqueued back to back (each with its own HIP-event pair around the pileup kernel) and the
        # host waits once, like a queue of shards would be driven; --sync-each-step gives the one-call-per-pass form
        for st in ds.run_many(a.steps):
            ms_pileup.append(st["ms_pileup"]); ms_total.append(st["ms_total"])
    barrier()
    dt = time.perf_counter() - t0

    # the same K passes with the tails overlapped on a second stream (reported beside the main line, never as `value`:
    # the roofline above is measured on kernels that have the chip to themselves)
    overlapped = None
    if not a.sync_each_step and not a.no_overlap_extra:
        ds.run_many(2, overlap=True)       # untimed: allocates the second set of per-pass intermediates and the second stream
        barrier()
        t0o = time.perf_counter()
        so = ds.run_many(a.steps, overlap=True)
        barrier()
        dto = time.perf_counter() - t0o
        overlapped = {"ms_per_step": dto / a.steps * 1e3, "pileup_kernel_ms_sharing_the_chip": sum(x["ms_pileup"] for x in so) / len(so),
                      "per_gpu_value": info["n_pileup_bases"] * a.steps / dto / 1e9, "unit": "Gbases/s"}

    def coverage_counters(kernel_ms):
        """What bounds msnv_coverage_tiles, from the counter passes committed under profiles/ (profiles/cov_prof.sh: rocprofv3 --pmc in
        separate runs; they cannot be collected inside this process).  Issue-slot share = VALU wave-instructions x 4 cycles / (1024
        SIMDs x the kernel's cycles at 2.4 GHz)."""
        for name in TRAFFIC_PROFILES[:2] + ["r03cov_pmc.json", "r02cov5_pmc.json"]:      # (the bench command's own counter passes hold msnv_coverage_tiles since round 5)
            try:
                c = json.load(open(os.path.join(ROOT, "profiles", name)))["counters"]
                k = [v for kk, v in c.items() if "coverage_tiles" in kk][0]
                valu = k["SQ_INSTS_VALU"]["avg_per_launch"]
                out = {"source": "profiles/" + name, "valu_wave_instructions_per_launch": valu,
                       "valu_issue_share_of_this_run": valu * 4.0 / (1024.0 * kernel_ms * 1e-3 * 2.4e9)}
                if "SQ_INSTS_LDS" in k:
                    out["lds_wave_instructions_per_launch"] = k["SQ_INSTS_LDS"]["avg_per_launch"]
                if "FETCH_SIZE" in k:
                    out["fetch_bytes_x2_per_launch"] = 2 * 1024.0 * k["FETCH_SIZE"]["avg_per_launch"]
                out["limited_by"] = "vector-instruction issue" if out["valu_issue_share_of_this_run"] > 0.5 else "latency / occupancy"
                return out
            except Exception:
                continue
        return None

    cov_extra = None
    if rank == 0:
        # the qaCompute half of the path on the same resident columns (msnv_coverage_tiles), outside the timed region
        ms = []
        for _ in range(5):
            ms.append(ds.coverage_run()["ms_coverage"])
        cov_ms = sum(ms[1:]) / len(ms[1:])
        cov_extra = {"kernel_ms": cov_ms, "bytes_per_M_interval": 8, "intervals": info["n_reads_pileup"],
                     "roofline": {"bound": "hbm", "achieved": 8.0 * info["n_reads_pileup"] / (cov_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": 8.0 * info["n_reads_pileup"] / (cov_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "counters": coverage_counters(cov_ms)}}

    ann_extra = None
    if not a.no_annotation and rank == 0:
        # configs[4]: gene / codon annotation of the called sites on the device (outside the timed region)
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            fa, an = os.path.join(td, "ref.fa"), os.path.join(td, "ann.tsv")
            syn.write_fasta(fa)
            n_genes = synth_annotation(syn, an)
            t0a = time.perf_counter()
            recs, _ = ds.annotate(an, fa)                 # first call parses + uploads the tables
            t_first = time.perf_counter() - t0a
            ms = [ds.annotate(an, fa)[1] for _ in range(5)]
            ann_extra = {"genes": n_genes, "sites": int(len(recs)), "sites_in_gene": int((recs["gene"] >= 0).sum()),
                         "kernel_ms": sum(ms) / len(ms), "first_call_s_incl_parse_upload": t_first}

    dist_extra = None
    if not a.no_annotation and rank == 0 and a.workload == "testdata":
        # metaSNV_DistDiv.py --dist on the device, on a table of the size this workload's calls give
        # (samples x called positions; random frequencies, 10 % of the cells uninformative), outside the timed region
        import random
        import tempfile
        import ctypes as C
        from metasnv_amd import _lib
        rnd = random.Random(11)
        n_pos = 6000
        with tempfile.TemporaryDirectory() as td:
            fp = os.path.join(td, "sp.filtered.freq")
            with open(fp, "w") as f:
                f.write("\t" + "\t".join("s%d.bam" % i for i in range(sp.n_samples)) + "\n")
                for k in range(n_pos):
                    f.write("c:-:%d:A>T:.\t%s\n" % (k + 1, "\t".join("-1" if rnd.random() < 0.1 else repr(rnd.randint(0, 40) / 40) for _ in range(sp.n_samples))))
            ns, npos, ms = C.c_int32(), C.c_uint64(), C.c_double()
            _lib.check(_lib.lib.msnv_dist_file(ctx._h, fp.encode(), (fp + ".mann").encode(), (fp + ".allele").encode(), 0.6, C.byref(ns), C.byref(npos), C.byref(ms)))
            dist_extra = {"samples": ns.value, "positions": int(npos.value), "pairs": ns.value * (ns.value + 1) // 2, "kernel_ms": ms.value,
                          "table_bytes_read_per_pair": 16 * int(npos.value)}

    decode_extra = None
    if not a.no_annotation and rank == 0 and world == 1 and a.workload == "testdata":
        # the host stage in front of the kernels -- 16 of the workload's samples written as BAM files and
        # read back through msnv_dataset_add_sample_bams (read + BGZF inflate + parse + pack) with 8 host threads, the blocks inflated
        # by the host decoder and on the device (csrc/inflate_k.hip); outside the timed region
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as td:
                fa = os.path.join(td, "ref.fa")
                syn.write_fasta(fa)
                paths = []
                for i in range(min(16, sp.n_samples)):
                    pth = os.path.join(td, "s%03d.bam" % i)
                    core.write_bam(pth, syn.names, syn.lengths, syn.sample_records(i))
                    paths.append(pth)
                decode_extra = {"files": len(paths), "bam_bytes": sum(os.path.getsize(x) for x in paths), "host_threads": 8}
                keep = os.environ.get("MSNV_INFLATE")
                for inflate_mode in ("device", "host", "device"):      # (the first device round pins the staging buffers)
                    os.environ["MSNV_INFLATE"] = inflate_mode
                    d2 = core.Dataset.from_files(ctx, paths[0], fa)
                    t0h = time.perf_counter()
                    d2.add_sample_bams(paths, 8)
                    dth = time.perf_counter() - t0h
                    nb = d2.finalize()["n_pileup_bases"]
                    d2.close()
                    decode_extra["%s_inflate" % inflate_mode] = {"seconds": dth, "Gbases_per_s": nb / dth / 1e9}
                if keep is None:
                    os.environ.pop("MSNV_INFLATE", None)
                else:
                    os.environ["MSNV_INFLATE"] = keep
                _, cnt = core.bgzf_inflate(paths[0], ctx)
                decode_extra["inflate_kernel"] = {"blocks": cnt["blocks"], "ms": cnt["kernel_ms"], "inflated_bytes": cnt["bytes"]}
        except Exception as e:                             # the extra must never cost the bench line
            decode_extra = {"error": repr(e)}

    bases = info["n_pileup_bases"]
    k_ms = sum(ms_pileup) / len(ms_pileup)
    alg = st["algorithmic_bytes"]
    slowest = 0
    if dist is not None:
        import torch
        t = torch.tensor([dt, float(bases), float(st["n_called_pop"]), float(st["n_called_indiv"]), k_ms, float(alg)], dtype=torch.float64,
                         device="cuda" if a.dist_backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)       # the only collective: tiny result table over RCCL/xGMI
        dt_max = max(float(g[0]) for g in gathered)
        total_bases = sum(float(g[1]) for g in gathered)
        called = [int(g[2]) for g in gathered]
        # the roofline of the job is the one of its slowest rank (lowest achieved bandwidth of the dominant kernel)
        per_rank_gbs = [float(g[5]) / (float(g[4]) * 1e-3) / 1e9 for g in gathered]
        slowest = min(range(world), key=lambda r: per_rank_gbs[r])
        k_ms, alg = float(gathered[slowest][4]), int(gathered[slowest][5])
    else:
        dt_max, total_bases, called = dt, float(bases), [int(st["n_called_pop"])]
        per_rank_gbs = [alg / (k_ms * 1e-3) / 1e9]

    if rank == 0:
        achieved = alg / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(sp.n_samples, sp.n_species, sp.contig_len, sp.mean_cov) if (world == 1 and a.workload == "testdata") else (None, None)
        resident = info["bytes_headers"] + info["bytes_seq"] + (info["bytes_qual"] + 7) // 8
        line = {
            "metric": "pileup Gbases/s across all samples",
            "value": total_bases * a.steps / dt_max / 1e9,
            "unit": "Gbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d synthetic BAM-record streams x %d species (%d contigs, %d bp per species), ~%gx, single-end %d bp"
                                   % (wl_label, sp.n_samples, sp.n_species, len(syn.names), sp.contig_len, sp.mean_cov, sp.read_len),
                       "samples": sp.n_samples, "pairs_per_gpu": info["n_pairs"], "work_items_per_gpu": info["n_work"], "positions_per_gpu": info["n_positions"], "pileup_bases_per_gpu": bases,
                       "reads_per_gpu": info["n_reads"], "parallelism": "contig shards x%d, no data-path collective" % world,
                       "called_SNPs_lines_per_rank": called},
            "roofline": {"bound": "hbm", "kernel": "msnv_pileup_tiles_narrow32" if a.workload == "testdata" else "msnv_pileup_tiles_* (narrow32 + merged [+ wide] between one pair of HIP events)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # the three figures that say what `frac` does not (filled below / here): the same SURVEY bytes over the WHOLE region from raw records
                         # resident in HBM to the calls, that region's milliseconds, and the counters' HBM bytes of the dominant kernel over its time
                         "frac_from_records": None, "total_ms_from_records": None,
                         "frac_hbm_traffic": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and k_ms) else None,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": k_ms,
                         "rank": slowest, "achieved_per_rank": per_rank_gbs,
                         "bytes_per_pileup_base": alg / max(1, bases),
                         "algorithmic_definition": "SURVEY.md 8d: per pileup read 16 B header + 4 B per CIGAR op + 0.5 B/base + 1 B/base quality",
                         # what the kernel has to read of the RESIDENT format: piece headers, 4-bit bases and ONE BIT per base of quality ("below the
                         # -Q cutoff", packed on the host at upload; the reference's input carries a byte, which is what `achieved` counts)
                         "resident_bytes_per_launch": resident, "frac_resident": resident / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else None,
                         # `achieved` / `frac` price the SURVEY 8d bytes (the reference's input: a byte of quality per base) against the kernel's time, as the
                         # bench contract defines them; what the kernel MOVES is less (one bit of quality per base): frac_hbm_traffic = measured HBM bytes
                         # of the committed counter passes / this run's kernel time / peak, null away from the profiled workload
                         "counters": pileup_counters(sp.n_samples, sp.n_species, sp.contig_len, sp.mean_cov, k_ms) if (world == 1 and a.workload == "testdata") else None},
            "positions_per_s": info["n_positions"] * world * a.steps / dt_max,
            "kernel_ms": {"pileup": k_ms, "pipeline_total": sum(ms_total) / len(ms_total)},
            "host": {"pack_s": t_pack, "synth_generator_wall_s": ht_build.get("synth_wall_s"), "pack_on_device_wall_s": ht_build.get("pack_device_wall_s"),
                     "finalize_upload_s": t_up, "device_bytes": info["device_bytes"],
                     "what": "pack_s = wall seconds of building the samples: %s" % ("record streams resident in HBM parsed / filtered / cut into pieces by kernels (csrc/devpack.hip); "
                             "synth_generator_wall_s = making the streams on the host and uploading them, before" if from_records else
                             "synthetic record streams made by the host threads and packed " + ("by kernels (csrc/devpack.hip)" if pack["records"] else "by the host threads too (MSNV_PACK=host, csrc/pack.cpp)"))},
        }
        # what limits the dominant kernel comes from the counters, not from the roofline it is priced against: the path is integer counting (no
        # MFMA work), `achieved` / `peak` stay HBM GB/s as the contract defines them, `bound` says which unit the counters show saturated
        if achieved > 6300.0:
            line["roofline"]["note"] = ("SURVEY bytes per second, not HBM traffic: `achieved` prices the reference's input (1.70 B per pileup base, a byte of quality per base) over the "
                                        "kernel's time and exceeds what HBM delivers (~6.3 TB/s); the bytes the kernel moves are `traffic` (frac_hbm_traffic)")
        cnt = line["roofline"]["counters"]
        if cnt and cnt.get("limited_by", "").startswith("vector"):
            line["roofline"]["bound"] = "valu"
            line["roofline"]["bound_note"] = ("the counters (%s) put the kernel at %.0f %% of its vector-instruction issue slots: it is VALU-issue bound, not HBM bound; achieved / peak / frac "
                                              "are still the SURVEY 8d bytes over the kernel's time against the HBM peak" % (cnt["source"], 100.0 * cnt["valu_issue_share_of_this_run"]))
        if from_records:
            # Second roofline block: the timed region starts at RAW alignment records resident in HBM -- the SURVEY 8d bytes and then some
            # (36-byte fixed part, read name, CIGAR, 4-bit bases, one byte of quality per base) -- and ends at the calls: the per-read stage
            # (kernels of csrc/devpack.hip; WALL milliseconds of the call, host side included), finalize (the tile index; wall milliseconds,
            # INSIDE the region since round 5) and one launch of the pileup kernel.  Median of seven builds.
            bl = sorted(from_records["builds"], key=lambda b: b["pack_wall_ms"] + b["finalize_wall_ms"])[len(from_records["builds"]) // 2]
            total_ms = bl["pack_wall_ms"] + bl["finalize_wall_ms"] + k_ms
            pk = sum(bl["pack_kernel_ms"].values())
            line["roofline_from_records"] = {
                "bound": "hbm", "kernel": "msnv_scan_sub2 (record boundaries + per-record measure in one walk) + msnv_scan_write2 + msnv_depth2 + msnv_emit_block (+ small scans), finalize (msnv_fin_*), then msnv_pileup_tiles_narrow32",
                "timed_region": "raw BAM records resident in HBM -> per-read stage (wall) -> finalize: tile index (wall) -> one pileup kernel launch",
                "achieved": alg / (total_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "total_ms": total_ms, "pack_wall_ms": bl["pack_wall_ms"], "finalize_ms": bl["finalize_wall_ms"], "pileup_kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": alg, "record_bytes_resident": from_records["record_bytes"],
                "pack_kernels_ms": pk, "pack_stage_ms": bl["pack_kernel_ms"], "whole_pass_ms": sum(ms_total) / len(ms_total),
                "pack_alone": {"achieved_on_record_bytes": from_records["record_bytes"] / (bl["pack_wall_ms"] * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": from_records["record_bytes"] / (bl["pack_wall_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "Gbases_per_s": bases / (bl["pack_wall_ms"] * 1e-3) / 1e9},
                "all_builds": from_records["builds"],
                "records": int(pack["records"]), "pieces": int(pack["pieces"]), "samples_through_the_host_prepass": int(pack["prepass_samples"]),
                "record_scans_through_the_careful_kernel": int(pack["scan_segments_redone"])}
            line["roofline"]["frac_from_records"] = line["roofline_from_records"]["frac"]
            line["roofline"]["total_ms_from_records"] = total_ms
        if overlapped:
            line["overlapped_passes"] = overlapped
        if cov_extra:
            line["coverage_pass"] = cov_extra
        if ann_extra:
            line["annotation"] = ann_extra
        if dist_extra:
            line["distances"] = dist_extra
        if decode_extra:
            line["host_decode"] = decode_extra
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(sp_kwargs, min(a.cpu_samples, sp.n_samples))
        if not a.no_annotation and world == 1 and a.workload == "testdata":
            try:
                ds.close(); ctx.close()                       # the launcher below is a process of its own with its own context
                line["end_to_end"] = end_to_end(sp, syn, bases, a.host_threads or min(32, os.cpu_count() or 8), line.get("cpu_baseline", {}).get("value"))
            except Exception as e:                             # the extra must never cost the bench line
                line["end_to_end"] = {"error": repr(e)}
    strong = None
    if (world > 1 or (a.mode is None and a.workload == "testdata")) and not a.no_strong_extra and mode == "weak":
        # (at N = 1 with the plain argv too: the one-rank point of the cohort the N > 1 lines are about -- the base of the scaling curve)
        # the product's N-rank path on ONE fixed cohort (32 species x 300 kb, 160 samples carrying six each): what the weak line above
        # cannot show -- LPT imbalance, the all-to-all of the records, the gather to rank 0
        ds.close()
        import copy
        b = copy.copy(a)
        if a.strong_extra_shape:
            xs, xl = (int(x) for x in a.strong_extra_shape.split(","))
            b.workload, b.scale, b.species, b.contig_len, b.mean_cov = "config3", 0.32, xs, xl, None
        else:
            # the cohort every rank count shares: the generator of one GPU's contig shard of BASELINE configs[3] (500 samples, sparse coverage,
            # 375 species x ~2 Mbp at the default scale), dealt to the N ranks by the product's own path
            b.workload, b.scale, b.species, b.contig_len, b.mean_cov, b.samples = "config4shard", (a.scale if a.scale is not None else 0.25), None, None, None, None
        b.steps, b.warmup = max(3, min(a.steps, 10)), 1
        # the extra must never cost the bench line: not by raising, and not by one rank waiting in a collective for a rank that is gone --
        # after `limit` seconds every rank leaves, rank 0 with the line it already has
        import threading
        limit = float(os.environ.get("MSNV_STRONG_EXTRA_LIMIT_S", "300"))
        printed = threading.Lock()                                # the line is printed exactly once: by the watchdog or by the normal path
        def give_up():
            if not printed.acquire(blocking=False):
                return
            if rank == 0:
                line["strong_scaling"] = {"error": "no result within %.0f s" % limit}
                print(json.dumps(line), flush=True)
            os._exit(3)                                           # a rank that hung in a collective is not a success (rank 0 has printed the line it had)
        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            strong = strong_run(b, rank, world, local, dist, brief=False)
        except Exception as e:
            strong = {"error": repr(e)}
        watchdog.cancel()
        if not printed.acquire(blocking=False):                   # the watchdog fired between the run's return and the cancel: it prints and exits
            time.sleep(60)
        if rank == 0:
            if world > 1 and a.mode is None and a.workload == "testdata" and strong and "error" not in strong:
                # N > 1 with the driver's plain argv: the HEADLINE is the product's N-rank path on ONE cohort (strong scaling: three species
                # cannot shard eight ways, and N independent replicas of them measure nothing); the replica line rides along
                head = strong
                head["weak_replicas"] = {k: line[k] for k in ("value", "unit", "ms_per_step", "scaling", "steps", "warmup") if k in line}
                head["weak_replicas"]["workload"] = line["config"]["workload"]
                head["weak_replicas"]["roofline"] = {k: line["roofline"][k] for k in ("bound", "kernel", "achieved", "frac", "kernel_ms_avg", "achieved_per_rank") if k in line["roofline"]}
                print(json.dumps(head))
            else:
                for k in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data"):
                    if strong and "error" not in strong:
                        strong.pop(k, None)
                line["strong_scaling"] = strong
                print(json.dumps(line))
        parallel.finalize()
        return
    if rank == 0:
        print(json.dumps(line))
    parallel.finalize()


if __name__ == "__main__":
    main()
