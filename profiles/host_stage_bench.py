"""profiles/host_stage_bench.py [samples] -- msnv_dataset_add_sample_bams (read + BGZF inflate + record parse + pack) on BAM files of
the benchmark shape, with the blocks inflated by the host decoder and on the device (csrc/inflate_k.hip), at 1 / 8 / all threads."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print("host cores", os.cpu_count())
sp = core.synth_params(n_species=3, contig_len=300000, n_samples=n, mean_cov=10.0, seed=1)
syn = core.Synth(sp); ctx = core.Context(0)
os.makedirs("/tmp/hb", exist_ok=True)
fa = "/tmp/hb/ref.fa"; syn.write_fasta(fa)
paths = []
for i in range(n):
    p = "/tmp/hb/s%04d.bam" % i
    core.write_bam(p, syn.names, syn.lengths, syn.sample_records(i)); paths.append(p)
print("%d BAM files, %.0f MB" % (n, sum(os.path.getsize(p) for p in paths) / 1e6))
for rep in range(2):
    for mode in ("host", "device"):
        os.environ["MSNV_INFLATE"] = mode
        for th in (1, 8, 32, 0):
            ds = core.Dataset.from_files(ctx, paths[0], fa)
            t0 = time.perf_counter(); ds.add_sample_bams(paths, th); t1 = time.perf_counter()
            info = ds.finalize(); st = ds.run()
            print("round %d  %-6s threads %3s: %.2f s = %.2f Gbases/s host stage (sites %d)" % (rep, mode, th or "all", t1 - t0, info["n_pileup_bases"] / (t1 - t0) / 1e9, st["n_sites"]))
            ds.close()
