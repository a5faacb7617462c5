import os, sys, time
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
print("cores", os.cpu_count())
sp = core.synth_params(n_species=3, contig_len=300000, n_samples=32, mean_cov=10.0, seed=1)
syn = core.Synth(sp); ctx = core.Context(0)
os.makedirs("/tmp/hb", exist_ok=True)
fa = "/tmp/hb/ref.fa"; syn.write_fasta(fa)
paths = []
t0 = time.perf_counter()
recs = [syn.sample_records(i) for i in range(sp.n_samples)]
print("synth %.2fs for %d samples, %.1f MB records" % (time.perf_counter() - t0, len(recs), sum(r.size for r in recs) / 1e6))
t0 = time.perf_counter()
for i, r in enumerate(recs):
    p = "/tmp/hb/s%04d.bam" % i
    core.write_bam(p, syn.names, syn.lengths, r); paths.append(p)
print("write_bam %.2fs, %.1f MB on disk" % (time.perf_counter() - t0, sum(os.path.getsize(p) for p in paths) / 1e6))
for th in (1, 4, 16, 0):
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    t0 = time.perf_counter(); ds.add_sample_bams(paths, th); t1 = time.perf_counter()
    info = ds.finalize(); t2 = time.perf_counter()
    print("threads %2d: add_sample_bams %.2fs  finalize %.2fs  bases %.3g -> %.3f Gbases/s host" % (th, t1 - t0, t2 - t1, info["n_pileup_bases"], info["n_pileup_bases"] / (t1 - t0) / 1e9))
    ds.close()
for th in (1, 0):
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    t0 = time.perf_counter()
    for r in recs: ds.add_sample_records(r)
    t1 = time.perf_counter()
    print("add_sample_records (pack only, serial calls): %.2fs" % (t1 - t0)); ds.close(); break
