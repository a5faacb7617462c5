// metasnv_amd/csrc/api.cpp -- C ABI of libmsnv.so (include/msnv.h): contexts, datasets, the
// pipeline driver, result mapping and the reference-format text writers.
#include <algorithm>
#include <sys/stat.h>

#include <atomic>
#include <future>
#include <chrono>
#include <climits>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "device.h"
#include "devpack.h"
#include "filter.h"

namespace msnv {
int pack_sample(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, SampleCols &sc);
int pileup_qualities(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, uint8_t *out);
int finalize_dataset(msnv_dataset &ds);
int write_calls_text(msnv_dataset &ds, const char *called_path, const char *indiv_path,
                     const msnv_site_ann *ann, const std::vector<std::string> *gene_names);
int coverage_run(msnv_dataset &ds, msnv_run_stats *stats);
int coverage_write(msnv_dataset &ds, int sample, const char *cov_path, const char *detail_path);
std::vector<std::string> synth_contigs(const msnv_synth_params &p);
void synth_sample_records(const msnv_synth_params &p, int sample, const std::vector<std::string> &contigs, std::vector<uint8_t> &out);
}  // namespace msnv

using namespace msnv;

extern "C" int msnv_abi_version(void) { return 3; }

extern "C" void msnv_params_default(msnv_params *p) {
    if (!p) return;
    p->min_coverage = 4; p->calling_threshold = 4; p->min_fraction = 0.01;      // call_vC.cpp:26-36
    p->min_baseq = 13; p->flag_filter = 0x704; p->count_orphans = 0; p->max_depth = 8000; p->min_mapq = 0;
    p->drop_first_line = 1;
    p->cov_max = 10; p->cov_min_mapq = 1;                                         // metaSNV.py:63-65, qaCompute.cpp:302
    p->ignore_overlaps = 0;                                                       // no -x in metaSNV.py:160-165
    p->token_limit = 10000;                                                       // call_vC.cpp:481-483
}

namespace msnv { void warm_devpack(void *), warm_kernels(void *), warm_textcall(void *), warm_annotate(void *); }
extern "C" int msnv_ctx_create(int device_id, msnv_ctx **out) {
    clear_error();
    if (!out) return fail(MSNV_EINVAL, "msnv_ctx_create: NULL out");
    *out = nullptr;
    if (int rc = dev_set_device(device_id)) return rc;
    msnv_ctx *c = new msnv_ctx();
    c->device = device_id;
    if (int rc = dev_stream_create(&c->stream)) { delete c; return rc; }
    warm_devpack(c->stream); warm_kernels(c->stream); warm_textcall(c->stream); warm_annotate(c->stream);      // (code objects loaded now, not inside the first stage that needs them)
    *out = c;
    return MSNV_OK;
}

namespace msnv { void dev_inflate_release(msnv_ctx *ctx); }
extern "C" void msnv_ctx_destroy(msnv_ctx *ctx) {
    if (!ctx) return;
    dev_inflate_release(ctx);
    devpack_ctx_release(ctx);
    dev_cache_trim();
    dev_stream_destroy(ctx->stream);
    if (ctx->stream2) dev_stream_destroy(ctx->stream2);
    delete ctx;
}

// ------------------------------------------------------------------------------ dataset
extern "C" int msnv_dataset_create(msnv_ctx *ctx, const msnv_ref_desc *ref, const msnv_params *params, msnv_dataset **out) {
    clear_error();
    if (!ref || !out) return fail(MSNV_EINVAL, "msnv_dataset_create: NULL argument");      // ctx may be NULL: host-stage-only dataset
    msnv_params P;
    if (params) P = *params; else msnv_params_default(&P);
    // per-sample counts are 16-bit on the device (msnv_site_sample): the depth cap keeps them there
    if (P.max_depth < 1 || P.max_depth > 65535) return fail(MSNV_EINVAL, "max_depth (mpileup -d) must be in [1, 65535], got %d", P.max_depth);
    if (P.cov_max < 1 || P.cov_max >= COV_BINS) return fail(MSNV_EINVAL, "cov_max (qaCompute -c) must be in [1, %d], got %d", COV_BINS - 1, P.cov_max);
    if (P.min_coverage < 0 || P.calling_threshold < 0 || P.min_baseq < 0 || P.min_mapq < 0 || P.cov_min_mapq < 0 || P.token_limit < 0 || !(P.min_fraction >= 0.0))
        return fail(MSNV_EINVAL, "negative cutoff in msnv_params");
    // snpCall -t 0: every allele of every position that passes the gates becomes an individual call (getSum >= 0 holds for every sample,
    // call_vC.cpp:593-600) -- gigabytes of zero entries the device path has no representation for (its per-sample evidence is sparse)
    if (P.calling_threshold < 1) return fail(MSNV_EDOMAIN, "calling_threshold (snpCall -t) must be at least 1 (with 0 the reference prints every allele of every covered position)");
    msnv_dataset *ds = new msnv_dataset();
    ds->ctx = ctx;
    ds->params = P;
    for (int i = 0; i < ref->n_contigs; ++i) {
        ds->names.emplace_back(ref->names[i]);
        ds->lengths.push_back(ref->lengths[i]);
        if (ref->seqs && ref->seqs[i]) { ds->seqs.emplace_back(ref->seqs[i], (size_t)ref->seq_lens[i]); ds->has_seq.push_back(1); }
        else { ds->seqs.emplace_back(); ds->has_seq.push_back(0); }
    }
    ds->sel.assign(ds->names.size(), 1);
    ds->bed_beg.assign(ds->names.size(), 0);
    ds->bed_end.assign(ds->names.size(), INT64_MAX);
    *out = ds;
    return MSNV_OK;
}

extern "C" int msnv_dataset_create_from_files(msnv_ctx *ctx, const char *bam_path, const char *fasta_path,
                                              const msnv_params *params, msnv_dataset **out) {
    clear_error();
    if (!bam_path || !out) return fail(MSNV_EINVAL, "msnv_dataset_create_from_files: NULL argument");
    BamHeader h;
    if (int rc = bam_read_header(bam_path, h)) return rc;
    std::vector<FastaSeq> fa;
    if (fasta_path) if (int rc = fasta_read(fasta_path, fa)) return rc;
    std::vector<const char *> names, seqs;
    std::vector<int64_t> lens, slens;
    std::unordered_map<std::string, const FastaSeq *> by_name;          // (a database of a million contigs: no scan per header line; the FIRST record of a name wins, like the scan did)
    by_name.reserve(fa.size() * 2);
    for (const FastaSeq &f : fa) by_name.emplace(f.name, &f);
    for (size_t i = 0; i < h.names.size(); ++i) {
        names.push_back(h.names[i].c_str());
        lens.push_back(h.lengths[i]);
        const auto it = by_name.find(h.names[i]);
        const FastaSeq *hit = it == by_name.end() ? nullptr : it->second;
        seqs.push_back(hit ? hit->seq.data() : nullptr);
        slens.push_back(hit ? (int64_t)hit->seq.size() : 0);
    }
    msnv_ref_desc rd{(int32_t)names.size(), names.data(), lens.data(), seqs.data(), slens.data()};
    return msnv_dataset_create(ctx, &rd, params, out);
}

extern "C" int msnv_dataset_attach_ctx(msnv_dataset *ds, msnv_ctx *ctx) {
    clear_error();
    if (!ds || !ctx) return fail(MSNV_EINVAL, "msnv_dataset_attach_ctx: NULL argument");
    if (ds->finalized) return fail(MSNV_EINVAL, "dataset is already finalized");
    ds->ctx = ctx;
    return MSNV_OK;
}

extern "C" int msnv_dataset_set_feed_ctx(msnv_dataset *ds, msnv_ctx *ctx) {
    clear_error();
    if (!ds) return fail(MSNV_EINVAL, "msnv_dataset_set_feed_ctx: NULL dataset");
    if (ctx && ds->ctx && ctx->device != ds->ctx->device) return fail(MSNV_EINVAL, "msnv_dataset_set_feed_ctx: the feed context must be one of the dataset's device (%d), not of device %d", ds->ctx->device, ctx->device);
    ds->feed_ctx = ctx;
    return MSNV_OK;
}

extern "C" void msnv_dataset_destroy(msnv_dataset *ds) {
    if (!ds) return;
    if (ds->ctx && (ds->dp.ready || !ds->dp.round_bufs.empty())) { (void)dev_set_device(ds->ctx->device); devpack_release(*ds); }
    if (ds->dev) { dev_free_all(*ds->dev); delete ds->dev; }
    delete ds;
}

extern "C" int msnv_dataset_set_bed(msnv_dataset *ds, int32_t n, const int32_t *tid, const int64_t *beg, const int64_t *end) {
    clear_error();
    if (!ds || (n && (!tid || !beg || !end))) return fail(MSNV_EINVAL, "msnv_dataset_set_bed: NULL argument");
    if (!ds->samples.empty()) return fail(MSNV_EINVAL, "msnv_dataset_set_bed must precede the first sample");
    const size_t NC = ds->names.size();
    std::vector<uint8_t> seen(NC, 0);
    for (int i = 0; i < n; ++i) {
        if (tid[i] < 0 || (size_t)tid[i] >= NC) return fail(MSNV_EINVAL, "BED region %d names contig %d which does not exist", i, tid[i]);
        if (seen[(size_t)tid[i]]) return fail(MSNV_EDOMAIN, "more than one BED region on contig %s (metaSNV writes one: metaSNV.py:92)", ds->names[(size_t)tid[i]].c_str());
        seen[(size_t)tid[i]] = 1;
        ds->bed_beg[(size_t)tid[i]] = beg[i]; ds->bed_end[(size_t)tid[i]] = end[i];
    }
    // contigs absent from the BED produce no pileup lines at all
    for (size_t c = 0; c < NC; ++c) if (!seen[c]) { ds->bed_beg[c] = 0; ds->bed_end[c] = 0; ds->sel[c] = 0; }
    ds->has_bed = true;
    return MSNV_OK;
}

extern "C" int msnv_dataset_set_bed_file(msnv_dataset *ds, const char *bed_path) {
    clear_error();
    if (!ds || !bed_path) return fail(MSNV_EINVAL, "msnv_dataset_set_bed_file: NULL argument");
    std::vector<BedRegion> regs;
    if (int rc = bed_read(bed_path, regs)) return rc;
    std::vector<int32_t> tid; std::vector<int64_t> b, e;
    for (const BedRegion &r : regs) {
        int t = -1;
        for (size_t c = 0; c < ds->names.size(); ++c) if (ds->names[c] == r.name) { t = (int)c; break; }
        if (t < 0) continue;     // samtools ignores BED names that are not in the header
        tid.push_back(t); b.push_back(r.beg); e.push_back(r.end);
    }
    return msnv_dataset_set_bed(ds, (int32_t)tid.size(), tid.data(), b.data(), e.data());
}

extern "C" int msnv_dataset_set_contig_mask(msnv_dataset *ds, const uint8_t *mask, int32_t n) {
    clear_error();
    if (!ds || !mask) return fail(MSNV_EINVAL, "msnv_dataset_set_contig_mask: NULL argument");
    if (!ds->samples.empty()) return fail(MSNV_EINVAL, "msnv_dataset_set_contig_mask must precede the first sample");
    if ((size_t)n != ds->names.size()) return fail(MSNV_EINVAL, "contig mask has %d entries, header has %zu contigs", n, ds->names.size());
    for (int i = 0; i < n; ++i) if (!mask[i]) ds->sel[(size_t)i] = 0;
    return MSNV_OK;
}

// Where the per-read stage runs (record parse, read filters, CIGAR walk, -Q test, piece cutting): on the device (devpack.hip) whenever the
// dataset has a device context -- MSNV_PACK=host keeps it on the host threads (pack.cpp), the same bytes either way.  A dataset created
// without a context (host-stage entry points only) packs on the host.
// What every entry point that appends samples (and finalize) asks first.  staged_ok: msnv_dataset_stage_sample_bams itself and finalize -- behind a
// staging call no other add_sample_* call may follow (msnv.h): the staged streams are packed last, so the sample order, and with it every
// per-sample output column, would come out wrong with no error.
static int check_open(const msnv_dataset *ds, bool staged_ok = false) {
    if (ds->finalized) return fail(MSNV_EINVAL, "dataset is already finalized");
    if (ds->poisoned) return fail(MSNV_EINVAL, "an earlier call failed after part of its samples had been packed on the device: the dataset cannot be used further (destroy it)");
    if (!staged_ok && !ds->staged.empty()) return fail(MSNV_EINVAL, "the dataset holds staged streams (msnv_dataset_stage_sample_bams): they are packed last, by msnv_dataset_finalize");
    return MSNV_OK;
}
static bool pack_on_device(const msnv_dataset *ds) {
    if (!ds->ctx) return false;
    const char *e = getenv("MSNV_PACK");                 // (per call: tests switch it)
    return !(e && e[0] == 'h');
}
// Appends n streams as n samples through the device pack, in rounds of at most MSNV_PACK_ROUND_MB (default 6144) of record bytes.
static int add_streams_device(msnv_dataset *ds, const uint8_t *const *records, const uint64_t *n_bytes, int n, bool streams_on_device, const uint8_t *in_place_base = nullptr, uint64_t in_place_capacity = 0) {
    HostTimerScope ts(HT_PACK_DEVICE_WALL);
    fin_trace_reset();
    struct Mark { ~Mark() { fin_trace("pack: whole call"); } } mark;
    const uint64_t round_bytes = [] { const char *e = getenv("MSNV_PACK_ROUND_MB"); const long long v = e ? atoll(e) : 6144; return (uint64_t)std::max<long long>(1, v) << 20; }();
    const size_t first = ds->samples.size();
    const size_t rounds_at_entry = ds->dp.rounds.size();
    ds->samples.resize(first + (size_t)n);
    int rc = MSNV_OK;
    try {
        for (int i0 = 0; i0 < n && !rc;) {
            int i1 = i0; uint64_t b = 0;
            while (i1 < n && i1 - i0 < 2048 && (i1 == i0 || b + n_bytes[i1] <= round_bytes)) { b += n_bytes[i1]; ++i1; }
            rc = devpack_add_round(*ds, first + (size_t)i0, records + i0, n_bytes + i0, i1 - i0, streams_on_device, in_place_base, in_place_capacity);
            i0 = i1;
        }
    } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "packing on the device failed: %s", e.what()); }
    if (rc) {
        ds->samples.resize(first);
        // rounds of this call that went through left their tables behind (dp.rounds, first_sample): finalize would index with them
        if (ds->dp.rounds.size() != rounds_at_entry) ds->poisoned = true;
    }
    return rc;
}

// A call that appends its samples in SEVERAL add_streams_device calls (groups of files, batches of the device inflate, groups of synthetic
// samples) and fails in a later one: the samples of the calls that went through are dropped with the rest (msnv.h: a failed add_* call adds
// nothing), and since their rounds' tables stay behind in dp.rounds the dataset is poisoned like in add_streams_device itself.
static int fail_multi_add(msnv_dataset *ds, size_t first, size_t rounds_at_entry, int rc) {
    ds->samples.resize(first);
    if (ds->dp.rounds.size() != rounds_at_entry) ds->poisoned = true;
    return rc;
}

extern "C" int msnv_dataset_add_sample_records_device(msnv_dataset *ds, const void *const *dev_records, const uint64_t *n_bytes, int32_t n) {
    clear_error();
    if (!ds || n < 0 || (n && (!dev_records || !n_bytes))) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_device: bad argument");
    if (int rc = check_open(ds)) return rc;
    if (!ds->ctx) return fail(MSNV_ENODEV, "msnv_dataset_add_sample_records_device: the dataset has no device context");
    for (int i = 0; i < n; ++i) if (n_bytes[i] && !dev_records[i]) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_device: stream %d is NULL", i);
    return add_streams_device(ds, reinterpret_cast<const uint8_t *const *>(dev_records), n_bytes, n, true);
}

extern "C" int msnv_dataset_add_sample_records_resident(msnv_dataset *ds, void *dev_buffer, uint64_t capacity, const uint64_t *offsets, const uint64_t *n_bytes, int32_t n) {
    clear_error();
    if (!ds || n < 0 || (n && (!dev_buffer || !offsets || !n_bytes))) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_resident: bad argument");
    if (int rc = check_open(ds)) return rc;
    if (!ds->ctx) return fail(MSNV_ENODEV, "msnv_dataset_add_sample_records_resident: the dataset has no device context");
    if (reinterpret_cast<uintptr_t>(dev_buffer) & 15u) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_resident: the buffer must start on 16 bytes");
    uint64_t prev_end = 0;
    for (int i = 0; i < n; ++i) {
        if (offsets[i] < prev_end || offsets[i] + n_bytes[i] < offsets[i] || offsets[i] + n_bytes[i] + 256 > capacity)
            return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_resident: stream %d must follow the one before it and leave 256 bytes of the buffer behind it", i);
        prev_end = offsets[i] + n_bytes[i];
    }
    std::vector<const uint8_t *> ptrs((size_t)n);
    for (int i = 0; i < n; ++i) ptrs[(size_t)i] = static_cast<const uint8_t *>(dev_buffer) + offsets[i];
    return add_streams_device(ds, ptrs.data(), n_bytes, n, true, static_cast<const uint8_t *>(dev_buffer), capacity);
}

extern "C" int msnv_dataset_add_sample_records(msnv_dataset *ds, const uint8_t *records, uint64_t n_bytes) {
    clear_error();
    if (!ds || (n_bytes && !records)) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records: NULL argument");
    if (int rc = check_open(ds)) return rc;
    if (pack_on_device(ds)) return add_streams_device(ds, &records, &n_bytes, 1, false);
    ds->samples.emplace_back();
    int rc;
    try { rc = pack_sample(*ds, records, n_bytes, ds->samples.back()); }
    catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "packing a sample failed: %s", e.what()); }      // nothing is thrown across the C ABI
    if (rc) ds->samples.pop_back();
    return rc;
}

extern "C" int msnv_dataset_add_sample_records_many(msnv_dataset *ds, const uint8_t *const *records, const uint64_t *n_bytes, int32_t n, int32_t host_threads) {
    clear_error();
    if (!ds || n < 0 || (n && (!records || !n_bytes))) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_many: bad argument");
    if (int rc = check_open(ds)) return rc;
    for (int i = 0; i < n; ++i) if (n_bytes[i] && !records[i]) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_records_many: stream %d is NULL", i);
    if (pack_on_device(ds)) return add_streams_device(ds, records, n_bytes, n, false);
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)n));
    const size_t first = ds->samples.size();
    ds->samples.resize(first + (size_t)n);
    std::atomic<int> next{0}, err{0};
    std::vector<std::string> msgs((size_t)n);
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || err.load()) break;
            int rc;
            try { rc = pack_sample(*ds, records[i], n_bytes[i], ds->samples[first + (size_t)i]); }
            catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "packing sample %d failed: %s", i, e.what()); }
            if (rc) { msgs[(size_t)i] = msnv_last_error(); err.store(rc); }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    if (err.load()) {
        ds->samples.resize(first);
        for (const std::string &m : msgs) if (!m.empty()) return fail(err.load(), "%s", m.c_str());
        return fail(err.load(), "packing failed");
    }
    return MSNV_OK;
}

extern "C" int msnv_dataset_pileup_qualities(const msnv_dataset *ds, const uint8_t *records, uint64_t n_bytes, uint8_t *out) {
    clear_error();
    if (!ds || (n_bytes && (!records || !out))) return fail(MSNV_EINVAL, "msnv_dataset_pileup_qualities: NULL argument");
    return pileup_qualities(*ds, records, n_bytes, out);
}

static int check_header(const msnv_dataset &ds, const BamHeader &h, const char *path) {
    // metaSNV assumes every BAM shares the header of the first (metaSNV.py:82-83)
    if (h.names.size() != ds.names.size()) return fail(MSNV_EFORMAT, "%s: header has %zu contigs, expected %zu", path, h.names.size(), ds.names.size());
    for (size_t i = 0; i < h.names.size(); ++i)
        if (h.names[i] != ds.names[i] || h.lengths[i] != ds.lengths[i]) return fail(MSNV_EFORMAT, "%s: contig %zu differs from the first BAM's header", path, i);
    return MSNV_OK;
}

extern "C" int msnv_dataset_add_sample_bam(msnv_dataset *ds, const char *bam_path) {
    clear_error();
    if (!ds || !bam_path) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_bam: NULL argument");
    BamHeader h; ByteBuf buf; uint64_t rec_off = 0;
    if (int rc = bam_read(bam_path, h, buf, rec_off, 4)) return rc;
    if (int rc = check_header(*ds, h, bam_path)) return rc;
    return msnv_dataset_add_sample_records(ds, buf.data() + rec_off, buf.size() - rec_off);      // (device pack when the dataset has a context)
}

namespace msnv {
struct InfBlock { unsigned long long in_off, out_off; uint32_t in_size, out_size; };
int dev_inflate_staging(msnv_ctx *ctx, uint64_t in_bytes, uint64_t out_bytes, uint8_t **in, uint8_t **out);
void dev_inflate_release(msnv_ctx *ctx);
void dev_inflate_release_device(msnv_ctx *ctx);
int dev_inflate(msnv_ctx *ctx, uint64_t comp_bytes, const std::vector<InfBlock> &blocks, uint64_t out_bytes, std::vector<uint32_t> &status, double *ms_kernel);
int dev_inflate_device_buffers(msnv_ctx *ctx, uint64_t in_bytes, uint64_t out_bytes);
int dev_inflate_resident(msnv_ctx *ctx, const uint8_t *host_in, uint64_t comp_bytes, const std::vector<InfBlock> &blocks, const std::vector<uint32_t> &blk_in_file,
                         uint32_t check_every, std::vector<uint32_t> &status, double *ms_kernel);
int dev_inflate_patch(msnv_ctx *ctx, uint64_t out_off, const uint8_t *data, uint32_t n);
}

// BGZF files inflated on the device (inflate_k.hip).  The files of a batch are read by `threads` host threads straight into the
// context's pinned staging buffer (no copy of the compressed bytes), their blocks are indexed there, the device inflates all blocks of
// the batch into the pinned output buffer, and `consume(f0, f1, out, ext)` parses files [f0, f1) in place (no copy of the inflated
// bytes either; the buffer is reused by the next batch).  Blocks the device refuses are inflated by the host decoder, which words
// the error of a malformed file.  ext[k]: offset and size of file f0 + k in `out`.
// counters (optional): [0] blocks, [1] blocks inflated on the host after all, [2] kernel microseconds, [3] inflated bytes.
struct InflatedExt { uint64_t off, size; };
// RESIDENT form (res != nullptr; the device pack's: add_bams_device_pack): the inflated bytes never leave HBM.  The files are read into
// pageable memory (no pinning: a context's first gigabyte of pinned staging costs 0.25 s), the batch goes up as it is, every block's CRC-32
// is checked by a kernel (inflate_k.hip: msnv_crc_blocks), only the status words come back; the BAM headers are read from the leading
// blocks of every file by the host decoder (res->hdr / res->rec_off, per file of the batch); a block the device refused or that did not
// check is inflated by the host decoder and patched into the device buffer.  consume() then gets out = nullptr and dev_valid = true.
struct ResidentBatch { std::vector<BamHeader> hdr; std::vector<uint64_t> rec_off; };
// MSNV_FEED_TRACE=1: wall milliseconds between the steps of the device feed, on stderr
static void feed_mark(const char *what) {
    static const bool on = [] { const char *e = getenv("MSNV_FEED_TRACE"); return e && e[0] == '1'; }();
    if (!on) return;
    static double last = 0;
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[feed] %-44s %8.3f ms\n", what, last ? (now - last) * 1e3 : 0.0);
    last = now;
}
template <typename Consume>
static int bgzf_read_files_device(msnv_ctx *ctx, const char *const *paths, int n, int threads, Consume consume, uint64_t counters[4], ResidentBatch *res = nullptr) {
    feed_mark("enter");
    if (int rc = dev_set_device(ctx->device)) return rc;
    std::vector<uint64_t> fsize((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        FILE *f = fopen(paths[i], "rb");
        if (!f) return fail(MSNV_EIO, "cannot open %s", paths[i]);
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fclose(f);
        if (sz < 0) return fail(MSNV_EIO, "cannot stat %s", paths[i]);
        fsize[(size_t)i] = (uint64_t)sz;
    }
    double ms = 0;
    uint64_t n_blocks = 0, n_host = 0, n_bytes = 0;
    // the batch buffers in HBM go back on EVERY way out (a caller that falls back to the host path after an error must not find up to ~4.6 GB
    // of staging still attached to the context); the pinned half stays for the next call
    struct ReleaseDevice { msnv_ctx *c; ~ReleaseDevice() { dev_inflate_release_device(c); } } release_device{ctx};
    // compressed bytes per batch (tests shrink it).  1 GB (~3.6 GB inflated) where the batch goes through pinned staging; 2 GB for the resident form,
    // which pins nothing: the benchmark's 160 BAMs (1.35 GB) are then ONE batch -- as two, the second one's files were read (page faults of fresh
    // buffers) while the first one's 1 GB went up from pageable memory (the runtime pinning it page by page), and the launcher waited 90 ms
    // for that read behind the first batch (MSNV_FEED_TRACE=1: round 5)
    const uint64_t batch_in = [&] { const char *e = getenv("MSNV_INFLATE_BATCH_MB"); const long long v = e ? atoll(e) : (res ? 2048 : 1024); return (uint64_t)std::max<long long>(1, v) << 20; }();
    // RESIDENT form: a batch's host work -- files read into pageable memory, blocks indexed, BAM headers read from the leading blocks -- is done by
    // load_batch, and the NEXT batch is loaded (std::async) while the device inflates, checks and packs the current one
    struct Loaded {
        int f0 = 0, f1 = 0; uint64_t ib = 0;
        std::vector<uint64_t> in_off; ByteBuf host_in;
        std::vector<std::vector<BgzfBlock>> blocks; std::vector<uint64_t> total;
        std::vector<BamHeader> hdr; std::vector<uint64_t> rec_off;
        int rc = MSNV_OK; std::string msg;
    };
    auto batch_extent = [&](int f0, std::vector<uint64_t> &in_off, uint64_t &ib) -> int {
        int f1 = f0; ib = 0; in_off.clear();
        while (f1 < n && (f1 == f0 || ib + fsize[(size_t)f1] <= batch_in)) { in_off.push_back(ib); ib += (fsize[(size_t)f1] + 31) & ~15ull; ++f1; }   // 16 bytes of slack behind every file
        return f1;
    };
    auto load_batch = [&](int f0) -> std::unique_ptr<Loaded> {
        std::unique_ptr<Loaded> L(new Loaded());
        L->f0 = f0; L->f1 = batch_extent(f0, L->in_off, L->ib);
        const int nf = L->f1 - f0;
        auto failed = [&](int rc) { L->rc = rc; L->msg = msnv_last_error(); return std::move(L); };
        try {
            if (!L->host_in.alloc(L->ib + 64)) return failed(fail(MSNV_ENOMEM, "out of memory for %llu compressed bytes", (unsigned long long)L->ib));
            L->blocks.resize((size_t)nf); L->total.assign((size_t)nf, 0); L->hdr.assign((size_t)nf, BamHeader()); L->rec_off.assign((size_t)nf, 0);
            std::atomic<int> next{0}, err{0};
            std::vector<std::string> msgs((size_t)nf);
            auto w = [&]() {
                for (;;) {
                    const int k = next.fetch_add(1);
                    if (k >= nf || err.load()) break;
                    const char *path = paths[f0 + k];
                    int rc = MSNV_OK;
                    try {
                        uint8_t *dst = L->host_in.data() + L->in_off[(size_t)k];
                        const uint64_t sz = fsize[(size_t)(f0 + k)];
                        {
                            HostTimerScope ts(HT_READ);
                            FILE *f = fopen(path, "rb");
                            if (!f) rc = fail(MSNV_EIO, "cannot open %s", path);
                            else {
                                if (sz && fread(dst, 1, sz, f) != sz) rc = fail(MSNV_EIO, "short read on %s", path);
                                fclose(f);
                            }
                        }
                        if (!rc) { memset(dst + sz, 0, 16); rc = bgzf_index_bytes(dst, sz, path, L->blocks[(size_t)k], L->total[(size_t)k]); }
                        if (!rc) rc = bam_header_from_blocks(dst, L->blocks[(size_t)k], path, L->hdr[(size_t)k], L->rec_off[(size_t)k]);      // (leading blocks, host decoder)
                    } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", path, e.what()); }
                    if (rc) { msgs[(size_t)k] = msnv_last_error(); err.store(rc); }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < std::max(1, std::min(threads, nf)); ++t) th.emplace_back(w);
            for (auto &t : th) t.join();
            if (err.load()) { L->rc = err.load(); for (const std::string &m : msgs) if (!m.empty()) { L->msg = m; break; } if (L->msg.empty()) L->msg = "BGZF read failed"; }
        } catch (const std::exception &e) { L->rc = MSNV_ENOMEM; L->msg = e.what(); }
        return L;
    };
    std::future<std::unique_ptr<Loaded>> ahead;
    struct WaitAhead { std::future<std::unique_ptr<Loaded>> &f; ~WaitAhead() { if (f.valid()) f.wait(); } } wait_ahead{ahead};      // (the loader reads this frame's variables: never leave it running)
    for (int f0 = 0; f0 < n;) {
        int f1 = f0; uint64_t ib = 0;
        std::vector<uint64_t> in_off;
        std::unique_ptr<Loaded> loaded;
        if (res) {
            loaded = ahead.valid() ? ahead.get() : load_batch(f0);
            feed_mark("batch loaded (files read, blocks indexed)");
            if (loaded->rc) return fail(loaded->rc, "%s", loaded->msg.c_str());
            f1 = loaded->f1; ib = loaded->ib; in_off = loaded->in_off;
            if (f1 < n) ahead = std::async(std::launch::async, load_batch, f1);
        } else f1 = batch_extent(f0, in_off, ib);
        uint8_t *in_stage = nullptr, *out = nullptr;
        // A batch whose staging cannot be had (pinned host memory or HBM: MSNV_ENOMEM) or whose launch fails is inflated by the host
        // decoder instead -- the call must not fail where the host path would have worked (a multi-GB BAM sizes the staging to itself)
        bool host_batch = false;
        ByteBuf host_in, host_out;
        if (res) {
            host_in = std::move(loaded->host_in);
            in_stage = host_in.data();
        } else if (int rc = dev_inflate_staging(ctx, ib, 0, &in_stage, &out)) {
            if (rc != MSNV_ENOMEM) return rc;
            fprintf(stderr, "libmsnv: no staging for the device inflate (%s); this batch is inflated on the host\n", msnv_last_error());
            clear_error();
            host_batch = true;
            if (!host_in.alloc(ib + 64)) return fail(MSNV_ENOMEM, "out of memory for %llu compressed bytes", (unsigned long long)ib);
            in_stage = host_in.data();
        }
        const int nf = f1 - f0;
        std::vector<std::vector<BgzfBlock>> blocks((size_t)nf);
        std::vector<uint64_t> total((size_t)nf, 0);
        if (res) { blocks = std::move(loaded->blocks); total = std::move(loaded->total); }
        std::atomic<int> next{0}, err{0};
        std::vector<std::string> msgs((size_t)nf);
        auto loader = [&]() {
            for (;;) {
                const int k = next.fetch_add(1);
                if (k >= nf || err.load()) break;
                const char *path = paths[f0 + k];
                int rc = MSNV_OK;
                try {
                    HostTimerScope ts(HT_READ);
                    uint8_t *dst = in_stage + in_off[(size_t)k];
                    const uint64_t sz = fsize[(size_t)(f0 + k)];
                    FILE *f = fopen(path, "rb");
                    if (!f) rc = fail(MSNV_EIO, "cannot open %s", path);
                    else {
                        if (sz && fread(dst, 1, sz, f) != sz) rc = fail(MSNV_EIO, "short read on %s", path);
                        fclose(f);
                    }
                    if (!rc) { memset(dst + sz, 0, 16); rc = bgzf_index_bytes(dst, sz, path, blocks[(size_t)k], total[(size_t)k]); }
                } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", path, e.what()); }
                if (rc) { msgs[(size_t)k] = msnv_last_error(); err.store(rc); }
            }
        };
        if (!res) {
            std::vector<std::thread> th;
            for (int t = 0; t < std::max(1, std::min(threads, nf)); ++t) th.emplace_back(loader);
            for (auto &t : th) t.join();
        }
        if (err.load()) { for (const std::string &m : msgs) if (!m.empty()) return fail(err.load(), "%s", m.c_str()); return fail(err.load(), "BGZF read failed"); }
        uint64_t ob = 0;
        std::vector<InfBlock> list;
        std::vector<int> origin;                                     // file (of the batch) of every entry
        std::vector<uint32_t> blk_in_file;                           // ... and its index among the file's blocks (MSNV_INFLATE_CHECK counts per file)
        std::vector<InflatedExt> ext((size_t)nf);
        for (int k = 0; k < nf; ++k) {
            ext[(size_t)k] = InflatedExt{ob, total[(size_t)k]};
            uint32_t bi = 0;
            for (const BgzfBlock &bl : blocks[(size_t)k]) {
                const uint32_t this_block = bi++;
                if (bl.out_size == 0) continue;
                list.push_back(InfBlock{in_off[(size_t)k] + bl.in_off, ob + bl.out_off, bl.in_size, bl.out_size});
                origin.push_back(k); blk_in_file.push_back(this_block);
            }
            ob += (total[(size_t)k] + 15) & ~15ull;
            n_bytes += total[(size_t)k];
        }
        if (res) {
            res->hdr = std::move(loaded->hdr); res->rec_off = std::move(loaded->rec_off);      // (read by load_batch)
            feed_mark("block list");
            const int rc_buf = dev_inflate_device_buffers(ctx, ib, ob);
            feed_mark("device buffers");
            if (int rc = rc_buf) {
                if (rc != MSNV_ENOMEM) return rc;
                fprintf(stderr, "libmsnv: no staging for the device inflate (%s); this batch is inflated on the host\n", msnv_last_error());
                clear_error();
                host_batch = true;
            }
        } else if (!host_batch) {
            uint8_t *same_in = nullptr;
            if (int rc = dev_inflate_staging(ctx, ib, ob, &same_in, &out)) {      // (the input staging does not move: it only grows when ib does)
                if (rc != MSNV_ENOMEM) return rc;
                fprintf(stderr, "libmsnv: no staging for the device inflate (%s); this batch is inflated on the host\n", msnv_last_error());
                clear_error();
                host_batch = true;
            }
        }
        if (host_batch) {
            if (!host_out.alloc(ob + 64)) return fail(MSNV_ENOMEM, "out of memory for %llu inflated bytes", (unsigned long long)ob);
            out = host_out.data();
        }
        std::vector<uint32_t> status;
        bool dev_valid = false;
        const uint32_t check_every = inflate_check_every();       // (one reading for both decoders: msnv_internal.h)
        if (!host_batch) {
            HostTimerScope ts(HT_INFLATE_DEVICE_WALL);
            int rc = MSNV_OK;
            if (res && getenv("MSNV_TEST_RESIDENT_FAIL")) rc = fail_quiet(MSNV_ENOMEM, "resident inflate refused (MSNV_TEST_RESIDENT_FAIL)");      // (tests: the fallback below)
            else rc = res ? dev_inflate_resident(ctx, in_stage, ib, list, blk_in_file, check_every, status, &ms) : dev_inflate(ctx, ib, list, ob, status, &ms);
            feed_mark("upload + inflate + check");
            if (rc) {
                if (rc != MSNV_ENOMEM && rc != MSNV_EHIP) return rc;
                fprintf(stderr, "libmsnv: the device inflate failed (%s); this batch is inflated on the host\n", msnv_last_error());
                clear_error();
                host_batch = true;
                // (a resident batch has no host copy of its output yet: the host decoder needs one -- round 4 wrote through a NULL pointer here)
                if (res) {
                    if (host_out.size() < ob + 64) { if (!host_out.alloc(ob + 64)) return fail(MSNV_ENOMEM, "out of memory for %llu inflated bytes", (unsigned long long)ob); }
                    out = host_out.data();
                }
            }
        }
        if (host_batch) status.assign(list.size(), 1u);
        // Every block's output is checked against the CRC-32 of its BGZF trailer, as htslib does for the reference's tools (a block that
        // does not check is handed to the host decoder like one the device refused); the host threads share the blocks.
        // MSNV_INFLATE_CHECK=n: every n-th block only (0 = none: benchmarks).
        if (check_every && !(res && !host_batch)) {                // (resident batches were checked by msnv_crc_blocks)
            std::atomic<size_t> nxt{0};
            auto checker = [&]() {
                HostTimerScope ts(HT_INFLATE_HOST);
                for (;;) {
                    const size_t e0 = nxt.fetch_add(64);
                    if (e0 >= list.size()) break;
                    for (size_t e = e0; e < std::min(list.size(), e0 + 64); ++e) {
                        if (status[e] || blk_in_file[e] % check_every) continue;
                        const uint8_t *trailer = in_stage + list[e].in_off + list[e].in_size;
                        const uint32_t want = (uint32_t)trailer[0] | (uint32_t)trailer[1] << 8 | (uint32_t)trailer[2] << 16 | (uint32_t)trailer[3] << 24;
                        if (bgzf_crc32(out + list[e].out_off, list[e].out_size) != want) status[e] = 2u;
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < std::max(1, std::min<int>(threads, (int)(list.size() / 64) + 1)); ++t) th.emplace_back(checker);
            for (auto &t : th) t.join();
        }
        if (res && !host_batch) {
            // resident batch: the few blocks the device refused or that did not check are inflated by the host decoder and patched into HBM
            std::vector<uint8_t> tmp;
            uint64_t done = 0;
            for (size_t e = 0; e < list.size(); ++e) {
                if (!status[e]) continue;
                HostTimerScope ts(HT_INFLATE_HOST);
                ++done;
                tmp.resize((size_t)list[e].out_size + 64);
                bool ok = bgzf_inflate_block_host(in_stage + list[e].in_off, list[e].in_size, tmp.data(), list[e].out_size);
                if (ok && check_every) {
                    const uint8_t *trailer = in_stage + list[e].in_off + list[e].in_size;
                    const uint32_t want = (uint32_t)trailer[0] | (uint32_t)trailer[1] << 8 | (uint32_t)trailer[2] << 16 | (uint32_t)trailer[3] << 24;
                    ok = bgzf_crc32(tmp.data(), list[e].out_size) == want;
                }
                if (!ok) return fail(MSNV_EFORMAT, "%s: BGZF inflate failed (malformed DEFLATE stream or CRC-32 mismatch)", paths[f0 + origin[e]]);
                if (int rc = dev_inflate_patch(ctx, list[e].out_off, tmp.data(), list[e].out_size)) return rc;
            }
            n_host += done;
            dev_valid = true;
        } else {   // blocks the device refused (or all of them, for a host batch): the host decoder, shared by the host threads
            std::atomic<size_t> nxt{0};
            std::atomic<int> bad{-1};
            std::atomic<uint64_t> done{0};
            auto redo = [&]() {
                HostTimerScope ts(HT_INFLATE_HOST);
                for (;;) {
                    const size_t e0 = nxt.fetch_add(16);
                    if (e0 >= list.size() || bad.load() >= 0) break;
                    for (size_t e = e0; e < std::min(list.size(), e0 + 16); ++e) {
                        if (!status[e]) continue;
                        done.fetch_add(1);
                        bool ok = bgzf_inflate_block_host(in_stage + list[e].in_off, list[e].in_size, out + list[e].out_off, list[e].out_size);
                        if (ok && check_every) {                     // the host decoder's bytes answer to the same trailer
                            const uint8_t *trailer = in_stage + list[e].in_off + list[e].in_size;
                            const uint32_t want = (uint32_t)trailer[0] | (uint32_t)trailer[1] << 8 | (uint32_t)trailer[2] << 16 | (uint32_t)trailer[3] << 24;
                            ok = bgzf_crc32(out + list[e].out_off, list[e].out_size) == want;
                        }
                        if (!ok) { int expect = -1; bad.compare_exchange_strong(expect, origin[e]); }
                    }
                }
            };
            bool any = host_batch;
            for (size_t e = 0; e < list.size() && !any; ++e) any = status[e] != 0u;
            if (any) {
                std::vector<std::thread> th;
                for (int t = 0; t < std::max(1, host_batch ? threads : std::min(threads, 4)); ++t) th.emplace_back(redo);
                for (auto &t : th) t.join();
            }
            n_host += done.load();
            if (bad.load() >= 0) return fail(MSNV_EFORMAT, "%s: BGZF inflate failed (malformed DEFLATE stream or CRC-32 mismatch)", paths[f0 + bad.load()]);
            dev_valid = !host_batch && done.load() == 0;               // every block of the batch as the device wrote it: ctx->dev_out holds the same bytes as `out`
        }
        n_blocks += list.size();
        feed_mark("blocks settled");
        if (int rc = consume(f0, f1, (const uint8_t *)out, ext, dev_valid)) return rc;
        feed_mark("batch consumed (statistics, pack)");
        f0 = f1;
    }
    if (counters) { counters[0] = n_blocks; counters[1] = n_host; counters[2] = (uint64_t)(ms * 1000.0); counters[3] = n_bytes; }
    return MSNV_OK;
}

// Test / measurement hook: the inflated bytes of one BGZF file, through the device (on_device != 0; needs ctx) or the host decoder.
extern "C" int msnv_bgzf_inflate(msnv_ctx *ctx, const char *path, int32_t on_device, uint8_t **out, uint64_t *n_out, uint64_t counters[4]) {
    clear_error();
    if (!path || !out || !n_out) return fail(MSNV_EINVAL, "msnv_bgzf_inflate: NULL argument");
    *out = nullptr; *n_out = 0;
    try {
        ByteBuf buf;
        if (on_device) {
            if (!ctx) return fail(MSNV_ENODEV, "msnv_bgzf_inflate: the device path needs a context");
            const char *p[1] = {path};
            auto take = [&](int, int, const uint8_t *data, const std::vector<InflatedExt> &ext, bool) -> int {
                if (!buf.alloc((size_t)ext[0].size)) return fail(MSNV_ENOMEM, "%s: out of memory for %llu inflated bytes", path, (unsigned long long)ext[0].size);
                memcpy(buf.data(), data + ext[0].off, (size_t)ext[0].size);
                return MSNV_OK;
            };
            if (int rc = bgzf_read_files_device(ctx, p, 1, 1, take, counters)) return rc;
        } else {
            if (int rc = bgzf_read_all(path, buf, 1)) return rc;
            if (counters) for (int i = 0; i < 4; ++i) counters[i] = 0;
        }
        *out = buf.p; *n_out = buf.n;
        buf.p = nullptr; buf.n = 0;                              // released by msnv_free
        return MSNV_OK;
    } catch (const std::exception &e) { return fail(MSNV_ENOMEM, "msnv_bgzf_inflate: %s", e.what()); }
}

// Where the BGZF blocks of a call's files are inflated: on the device when that is the faster way for THIS call.  The device path
// needs pinned staging for a batch (up to 1 GB compressed + its inflated bytes), and pinning costs ~0.25 s per GB the first time a
// context does it -- more than 32 host threads need for the whole job of the benchmark shape (160 BAMs, 1.35 GB: device path cold
// 1.5 s, host decoder 0.4 s; profiles/r03d end-to-end).  So: device when the estimated host time (compressed bytes / threads x
// ~90 MB/s per thread) exceeds the estimated device time (staging still to pin + both transfers at ~25 GB/s + a launch).
// MSNV_INFLATE=host | zlib keeps everything on the host, MSNV_INFLATE=device forces the device whatever the size.
static bool want_device_inflate(msnv_ctx *ctx, const char *const *paths, int n, int threads, bool resident = false) {
    if (!ctx) return false;
    const char *e = getenv("MSNV_INFLATE");
    if (e) return e[0] == 'd';
    uint64_t bytes = 0, largest = 0;
    for (int i = 0; i < n; ++i) {
        FILE *f = fopen(paths[i], "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END);
        const long z = ftell(f);
        fclose(f);
        if (z > 0) { bytes += (uint64_t)z; largest = std::max<uint64_t>(largest, (uint64_t)z); }
    }
    if (bytes < (64ull << 20)) return false;                       // the host decoder is done before the staging is set up
    const double batch_in = (double)std::max<uint64_t>(std::min<uint64_t>(bytes, 1024ull << 20), largest), batch_out = 3.6 * batch_in;
    const double to_pin = std::max(0.0, batch_in - (double)ctx->pin_in_cap) + std::max(0.0, batch_out - (double)ctx->pin_out_cap);
    // (resident: add_bams_device_pack -- nothing is pinned, the compressed bytes go up from pageable memory at ~40 GB/s, the inflated bytes stay
    // in HBM and are checked there; the kernel writes ~21 GB/s of output on a 160-BAM job: profiles/r04e_inflate_*)
    const double est_dev = resident ? (double)bytes / 40e9 + 3.6 * (double)bytes / 21e9 + 0.03 : to_pin * 0.25e-9 + (double)bytes * (1.0 + 3.6) / 25e9 + 0.02;
    const double est_host = (double)bytes / ((double)std::max(1, threads) * 90e6);
    return est_dev < est_host;
}

// The record streams of several BAM files (the N-rank driver deals them to the ranks that own their contigs): through the device
// inflate when a context is given and the files are large enough (the rule of msnv_dataset_add_sample_bams), else one host thread
// per file.  records[i] (released with msnv_free) holds n_bytes[i] bytes: the alignment records behind the header of bam_paths[i].
extern "C" int msnv_bam_records_many(msnv_ctx *ctx, const char *const *bam_paths, int32_t n, int32_t host_threads, uint8_t **records, uint64_t *n_bytes) {
    clear_error();
    if (n < 0 || (n && (!bam_paths || !records || !n_bytes))) return fail(MSNV_EINVAL, "msnv_bam_records_many: bad argument");
    for (int i = 0; i < n; ++i) { records[i] = nullptr; n_bytes[i] = 0; }
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)n));
    const bool on_device = want_device_inflate(ctx, bam_paths, n, nthreads);
    std::atomic<int> err{0};
    std::vector<std::string> msgs((size_t)std::max(n, 0));
    auto keep = [&](int i, const uint8_t *data, uint64_t size) -> int {      // header parsed, records copied out
        BamHeader h; uint64_t rec_off = 0;
        if (int rc = bam_parse_header_bytes(data, size, bam_paths[i], h, rec_off)) return rc;
        const uint64_t nb = size - rec_off;
        uint8_t *p = (uint8_t *)malloc(nb ? nb : 1);
        if (!p) return fail(MSNV_ENOMEM, "%s: out of memory for %llu record bytes", bam_paths[i], (unsigned long long)nb);
        memcpy(p, data + rec_off, nb);
        records[i] = p; n_bytes[i] = nb;
        return MSNV_OK;
    };
    auto run_threads = [&](int lo, int hi, auto body) {
        std::atomic<int> nxt{lo};
        auto w = [&]() {
            for (;;) {
                const int i = nxt.fetch_add(1);
                if (i >= hi || err.load()) break;
                int rc;
                try { rc = body(i); } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", bam_paths[i], e.what()); }
                if (rc) { msgs[(size_t)i] = msnv_last_error(); err.store(rc); }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < std::max(1, std::min(nthreads, hi - lo)); ++t) th.emplace_back(w);
        for (auto &t : th) t.join();
    };
    int rc = MSNV_OK;
    try {
        if (on_device) {
            auto consume = [&](int f0, int f1, const uint8_t *out, const std::vector<InflatedExt> &ext, bool) -> int {
                run_threads(f0, f1, [&](int i) { return keep(i, out + ext[(size_t)(i - f0)].off, ext[(size_t)(i - f0)].size); });
                return err.load();
            };
            uint64_t cnt[4];
            rc = bgzf_read_files_device(ctx, bam_paths, n, nthreads, consume, cnt);
        } else {
            run_threads(0, n, [&](int i) { ByteBuf buf; if (int r = bgzf_read_all(bam_paths[i], buf, 1)) return r; return keep(i, buf.data(), buf.size()); });
            rc = err.load();
        }
    } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "msnv_bam_records_many: %s", e.what()); }
    if (rc) {
        for (int i = 0; i < n; ++i) { free(records[i]); records[i] = nullptr; n_bytes[i] = 0; }
        for (const std::string &m : msgs) if (!m.empty()) return fail(rc, "%s", m.c_str());
        return rc;
    }
    return MSNV_OK;
}

// BAM files -> samples with the per-read stage on the device: the files are read and inflated group by group (host threads, or the device
// inflate when the rule of want_device_inflate picks it), the record streams of a group go to HBM and are packed there (devpack.hip).
static int add_bams_device_pack(msnv_dataset *ds, const char *const *bam_paths, int n, int nthreads) {
    const bool inflate_on_device = want_device_inflate(ds->ctx, bam_paths, n, nthreads, true);
    if (inflate_on_device) {
        // the inflated bytes of a batch are in the context's device buffer (and, for the CRC check and the header parse, in its pinned
        // twin): the record streams are handed over where they lie in HBM
        // (resident form of bgzf_read_files_device: the batch's bytes exist in HBM only, its headers were read from the files' leading blocks)
        ResidentBatch rb;
        const size_t first = ds->samples.size(), rounds_at_entry = ds->dp.rounds.size();
        auto consume = [&](int f0, int f1, const uint8_t *out, const std::vector<InflatedExt> &ext, bool dev_valid) -> int {
            std::vector<const uint8_t *> ptrs; std::vector<uint64_t> sizes;
            // (a batch the host decoder had to take -- no room for it in HBM -- is in host memory: it goes up from there)
            const uint8_t *base = dev_valid ? static_cast<const uint8_t *>(ds->ctx->dev_out) : out;
            for (int i = f0; i < f1; ++i) {
                const uint64_t size = ext[(size_t)(i - f0)].size, rec_off = rb.rec_off[(size_t)(i - f0)];
                if (int rc = check_header(*ds, rb.hdr[(size_t)(i - f0)], bam_paths[i])) return rc;
                if (rec_off > size) return fail(MSNV_EFORMAT, "%s: truncated BAM header", bam_paths[i]);
                ptrs.push_back(base + ext[(size_t)(i - f0)].off + rec_off);
                sizes.push_back(size - rec_off);
            }
            // in HBM: the records are read where the inflate kernel wrote them (no copy into a round buffer) when the batch's buffer leaves
            // the kernels' read-ahead room behind its last stream
            bool in_place = dev_valid && !(reinterpret_cast<uintptr_t>(base) & 15u) && !getenv("MSNV_PACK_COPY");
            for (size_t k = 0; k < ptrs.size() && in_place; ++k) {
                if (k > 0 && ptrs[k] < ptrs[k - 1] + sizes[k - 1]) in_place = false;
                if ((uint64_t)(ptrs[k] - base) + sizes[k] + 256 > ds->ctx->dev_out_cap) in_place = false;
            }
            return add_streams_device(ds, ptrs.data(), sizes.data(), f1 - f0, dev_valid, in_place ? base : nullptr, in_place ? ds->ctx->dev_out_cap : 0);
        };
        int rc;
        try { uint64_t cnt[4]; rc = bgzf_read_files_device(ds->ctx, bam_paths, n, nthreads, consume, cnt, &rb); }
        catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "device inflate: %s", e.what()); }
        return rc ? fail_multi_add(ds, first, rounds_at_entry, rc) : MSNV_OK;      // (a later batch failed: the earlier batches' samples go too)
    }
    const size_t first = ds->samples.size(), rounds_at_entry = ds->dp.rounds.size();
    const int group = std::max(nthreads, 16);
    for (int g0 = 0; g0 < n; g0 += group) {
        const int g1 = std::min(n, g0 + group);
        std::vector<ByteBuf> bufs((size_t)(g1 - g0));
        std::vector<uint64_t> rec_off((size_t)(g1 - g0), 0);
        std::atomic<int> next{g0}, err{0};
        std::vector<std::string> msgs((size_t)(g1 - g0));
        auto worker = [&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= g1 || err.load()) break;
                int rc;
                try {
                    BamHeader h;
                    rc = bam_read(bam_paths[i], h, bufs[(size_t)(i - g0)], rec_off[(size_t)(i - g0)], 1);
                    if (!rc) rc = check_header(*ds, h, bam_paths[i]);
                } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", bam_paths[i], e.what()); }
                if (rc) { msgs[(size_t)(i - g0)] = msnv_last_error(); err.store(rc); }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < std::min(nthreads, g1 - g0); ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
        int rc = err.load();
        if (rc) { for (const std::string &m : msgs) if (!m.empty()) { fail(rc, "%s", m.c_str()); break; } }
        if (!rc) {
            std::vector<const uint8_t *> ptrs; std::vector<uint64_t> sizes;
            for (int i = g0; i < g1; ++i) { ptrs.push_back(bufs[(size_t)(i - g0)].data() + rec_off[(size_t)(i - g0)]); sizes.push_back(bufs[(size_t)(i - g0)].size() - rec_off[(size_t)(i - g0)]); }
            rc = add_streams_device(ds, ptrs.data(), sizes.data(), g1 - g0, false);
        }
        if (rc) return fail_multi_add(ds, first, rounds_at_entry, rc);
    }
    return MSNV_OK;
}

// The N-rank feed's decode + deal step without a host copy of the inflated bytes: the files' BGZF blocks are inflated and checked on the device
// (resident form of bgzf_read_files_device), the record streams are dealt from where the inflate kernel wrote them (records_deal_device).
extern "C" int msnv_dataset_deal_bams_device(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads, const int32_t *contig_owner, int32_t n_parts,
                                             int32_t cov_min_mapq, uint8_t *out, uint64_t capacity, uint64_t gap, uint64_t *part_bytes, msnv_sample_stats *stats, uint64_t *record_bytes) {
    clear_error();
    if (!ds || n < 0 || (n && (!bam_paths || !part_bytes || !stats || !record_bytes)) || !contig_owner) return fail(MSNV_EINVAL, "msnv_dataset_deal_bams_device: bad argument");
    if (!ds->ctx) return fail(MSNV_ENODEV, "msnv_dataset_deal_bams_device needs a dataset with a device context");
    if (n == 0) return MSNV_OK;
    msnv_ctx *const fc = ds->feed_ctx ? ds->feed_ctx : ds->ctx;      // (msnv_dataset_set_feed_ctx: a round ahead of the dataset's own context)
    if (int rc = dev_set_device(fc->device)) return rc;
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)n));
    {   // one batch of the device inflate only: the parts of a call lie destination-major in `out`
        const uint64_t batch_in = [] { const char *e = getenv("MSNV_INFLATE_BATCH_MB"); const long long v = e ? atoll(e) : 1024; return (uint64_t)std::max<long long>(1, v) << 20; }();
        uint64_t ib = 0;
        for (int i = 0; i < n; ++i) {
            FILE *f = fopen(bam_paths[i], "rb");
            if (!f) return fail(MSNV_EIO, "cannot open %s", bam_paths[i]);
            fseek(f, 0, SEEK_END);
            const long z = ftell(f);
            fclose(f);
            if (z < 0) return fail(MSNV_EIO, "cannot stat %s", bam_paths[i]);
            ib += ((uint64_t)z + 31) & ~15ull;
        }
        if (n > 1 && ib > batch_in) return fail_quiet(MSNV_EDOMAIN, "msnv_dataset_deal_bams_device: the files of the call do not fit one batch of the device inflate (%llu bytes)", (unsigned long long)ib);
    }
    const int NC = (int)ds->names.size();
    ResidentBatch rb;
    int calls = 0;
    auto consume = [&](int f0, int f1, const uint8_t *host_out, const std::vector<InflatedExt> &ext, bool dev_valid) -> int {
        if (calls++ || f0 != 0 || f1 != n) return fail(MSNV_EINVAL, "internal: msnv_dataset_deal_bams_device expects one batch");
        std::vector<const uint8_t *> ptrs; std::vector<uint64_t> sizes;
        const uint8_t *base = dev_valid ? static_cast<const uint8_t *>(fc->dev_out) : host_out;      // (a batch the host decoder had to take lies in host memory)
        for (int i = 0; i < n; ++i) {
            const uint64_t size = ext[(size_t)i].size, rec_off = rb.rec_off[(size_t)i];
            if (int rc = check_header(*ds, rb.hdr[(size_t)i], bam_paths[i])) return rc;
            if (rec_off > size) return fail(MSNV_EFORMAT, "%s: truncated BAM header", bam_paths[i]);
            ptrs.push_back(base + ext[(size_t)i].off + rec_off);
            sizes.push_back(size - rec_off);
            record_bytes[i] = size - rec_off;
        }
        return records_deal_device(fc, ptrs.data(), sizes.data(), n, dev_valid, contig_owner, NC, n_parts, cov_min_mapq, out, capacity, gap, part_bytes, stats, nullptr);
    };
    try { uint64_t cnt[4]; return bgzf_read_files_device(fc, bam_paths, n, nthreads, consume, cnt, &rb); }
    catch (const std::exception &e) { return fail(MSNV_ENOMEM, "msnv_dataset_deal_bams_device: %s", e.what()); }
}

// ... and the step BEFORE the owners are known (the split planner holds decoded rounds): the files' record streams, inflated and checked on the
// device, are left in the caller's device buffer -- stream i at rec_off[i], rec_bytes[i] long -- with their statistics and the aligned bases per
// contig the planner weighs contigs by; msnv_records_deal_device deals them from there later.
extern "C" int msnv_dataset_inflate_bams_device(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads, uint8_t *out, uint64_t capacity,
                                                uint64_t *rec_off, uint64_t *rec_bytes, msnv_sample_stats *stats, uint64_t *contig_bases) {
    clear_error();
    if (!ds || n < 0 || (n && (!bam_paths || !rec_off || !rec_bytes || !stats || !out))) return fail(MSNV_EINVAL, "msnv_dataset_inflate_bams_device: bad argument");
    if (!ds->ctx) return fail(MSNV_ENODEV, "msnv_dataset_inflate_bams_device needs a dataset with a device context");
    if (n == 0) return MSNV_OK;
    msnv_ctx *const fc = ds->feed_ctx ? ds->feed_ctx : ds->ctx;
    if (int rc = dev_set_device(fc->device)) return rc;
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)n));
    {
        const uint64_t batch_in = [] { const char *e = getenv("MSNV_INFLATE_BATCH_MB"); const long long v = e ? atoll(e) : 1024; return (uint64_t)std::max<long long>(1, v) << 20; }();
        uint64_t ib = 0;
        for (int i = 0; i < n; ++i) {
            FILE *f = fopen(bam_paths[i], "rb");
            if (!f) return fail(MSNV_EIO, "cannot open %s", bam_paths[i]);
            fseek(f, 0, SEEK_END);
            const long z = ftell(f);
            fclose(f);
            if (z < 0) return fail(MSNV_EIO, "cannot stat %s", bam_paths[i]);
            ib += ((uint64_t)z + 31) & ~15ull;
        }
        if (n > 1 && ib > batch_in) return fail_quiet(MSNV_EDOMAIN, "msnv_dataset_inflate_bams_device: the files of the call do not fit one batch of the device inflate (%llu bytes)", (unsigned long long)ib);
    }
    const int NC = (int)ds->names.size();
    ResidentBatch rb;
    int calls = 0;
    auto consume = [&](int f0, int f1, const uint8_t *host_out, const std::vector<InflatedExt> &ext, bool dev_valid) -> int {
        if (calls++ || f0 != 0 || f1 != n) return fail(MSNV_EINVAL, "internal: msnv_dataset_inflate_bams_device expects one batch");
        std::vector<const uint8_t *> ptrs; std::vector<uint64_t> sizes;
        const uint8_t *base = dev_valid ? static_cast<const uint8_t *>(fc->dev_out) : host_out;
        uint64_t o = 0;
        for (int i = 0; i < n; ++i) {
            const uint64_t size = ext[(size_t)i].size, ro = rb.rec_off[(size_t)i];
            if (int rc = check_header(*ds, rb.hdr[(size_t)i], bam_paths[i])) return rc;
            if (ro > size) return fail(MSNV_EFORMAT, "%s: truncated BAM header", bam_paths[i]);
            rec_off[i] = o; rec_bytes[i] = size - ro;
            if (o + (size - ro) + 32 > capacity) return fail_quiet(MSNV_ECAPACITY, "msnv_dataset_inflate_bams_device: the output holds %llu bytes, more are needed", (unsigned long long)capacity);
            if (size - ro) if (int rc = dev_copy_bytes(out + o, base + ext[(size_t)i].off + ro, size - ro, dev_valid, fc->stream)) return rc;
            ptrs.push_back(out + o); sizes.push_back(size - ro);
            o += (size - ro + 31) & ~15ull;                       // (16 readable bytes behind every stream)
        }
        std::vector<int32_t> nobody((size_t)std::max(1, NC), -1);
        std::vector<uint64_t> pb((size_t)n, 0);
        return records_deal_device(fc, ptrs.data(), sizes.data(), n, true, nobody.data(), NC, 1, ds->params.cov_min_mapq, nullptr, 0, 0, pb.data(), stats, contig_bases);
    };
    try { uint64_t cnt[4]; return bgzf_read_files_device(fc, bam_paths, n, nthreads, consume, cnt, &rb); }
    catch (const std::exception &e) { return fail(MSNV_ENOMEM, "msnv_dataset_inflate_bams_device: %s", e.what()); }
}

extern "C" int msnv_dataset_add_sample_bams(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads) {
    clear_error();
    HostTimerScope ts_all(HT_ADD_WALL);
    if (!ds || (n && !bam_paths)) return fail(MSNV_EINVAL, "msnv_dataset_add_sample_bams: NULL argument");
    if (int rc = check_open(ds)) return rc;
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)n));
    if (pack_on_device(ds)) return add_bams_device_pack(ds, bam_paths, n, nthreads);
    const size_t first = ds->samples.size();
    ds->samples.resize(first + (size_t)n);
    std::atomic<int> next{0}, err{0};
    std::vector<std::string> msgs((size_t)n);
    // MSNV_INFLATE=device: the BGZF blocks of the files are inflated on the device (inflate_k.hip: a wavefront per block, thousands of
    // blocks at a time), batch by batch; the host threads read the files in front of it and parse / pack the batch's bytes in place
    bool on_device = want_device_inflate(ds->ctx, bam_paths, n, nthreads);
    auto pack_one = [&](int i, const uint8_t *data, uint64_t size, BamHeader &h, uint64_t rec_off) -> int {
        if (int rc = check_header(*ds, h, bam_paths[i])) return rc;
        return pack_sample(*ds, data + rec_off, size - rec_off, ds->samples[first + (size_t)i]);
    };
    if (on_device) {
        auto consume = [&](int f0, int f1, const uint8_t *out, const std::vector<InflatedExt> &ext, bool) -> int {
            std::atomic<int> nxt{f0};
            auto w = [&]() {
                for (;;) {
                    const int i = nxt.fetch_add(1);
                    if (i >= f1 || err.load()) break;
                    int rc;
                    try {
                        BamHeader h; uint64_t rec_off = 0;
                        const uint8_t *data = out + ext[(size_t)(i - f0)].off; const uint64_t size = ext[(size_t)(i - f0)].size;
                        rc = bam_parse_header_bytes(data, size, bam_paths[i], h, rec_off);
                        if (!rc) rc = pack_one(i, data, size, h, rec_off);
                    } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", bam_paths[i], e.what()); }
                    if (rc) { msgs[(size_t)i] = msnv_last_error(); err.store(rc); }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < std::max(1, std::min(nthreads, f1 - f0)); ++t) th.emplace_back(w);
            for (auto &t : th) t.join();
            return err.load();
        };
        int rc;
        try { uint64_t cnt[4]; rc = bgzf_read_files_device(ds->ctx, bam_paths, n, nthreads, consume, cnt); }
        catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "device inflate: %s", e.what()); }
        if (rc) {
            ds->samples.resize(first);
            for (const std::string &m : msgs) if (!m.empty()) return fail(rc, "%s", m.c_str());
            return rc;
        }
        return MSNV_OK;
    }
    auto worker = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n || err.load()) break;
            int rc;
            try {                                              // an exception in a worker thread would be std::terminate
                BamHeader h; ByteBuf buf; uint64_t rec_off = 0;
                rc = bam_read(bam_paths[i], h, buf, rec_off, 1);
                if (!rc) rc = pack_one(i, buf.data(), buf.size(), h, rec_off);
            } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", bam_paths[i], e.what()); }
            if (rc) { msgs[(size_t)i] = msnv_last_error(); err.store(rc); }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    if (err.load()) {
        ds->samples.resize(first);
        for (const std::string &m : msgs) if (!m.empty()) return fail(err.load(), "%s", m.c_str());
        return fail(err.load(), "BAM decode failed");
    }
    return MSNV_OK;
}

// BAM files read, inflated and header-checked by the host threads, their record streams KEPT as they are: msnv_dataset_finalize packs them,
// on the device when the dataset has been given its context by then (msnv_dataset_attach_ctx).  For a one-shot driver that brings the HIP
// runtime up on a thread of its own while the files are read (metasnv_amd/cli.py): the per-read stage then still runs as kernels.
extern "C" int msnv_dataset_stage_sample_bams(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads) {
    clear_error();
    HostTimerScope ts_all(HT_ADD_WALL);
    if (!ds || n < 0 || (n && !bam_paths)) return fail(MSNV_EINVAL, "msnv_dataset_stage_sample_bams: bad argument");
    if (int rc = check_open(ds, true)) return rc;
    if (!ds->samples.empty()) return fail(MSNV_EINVAL, "msnv_dataset_stage_sample_bams: the dataset already holds packed samples (staged streams are packed last)");
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)n));
    const size_t first = ds->staged.size();
    ds->staged.resize(first + (size_t)n); ds->staged_off.resize(first + (size_t)n, 0);
    std::atomic<int> next{0}, err{0};
    std::vector<std::string> msgs((size_t)n);
    // largest files first: the threads take files from one queue, so the last ones to be started are the small ones and no thread is left
    // alone with a large file at the end (160 files on 32 threads: the tail was a file's ~60 ms)
    std::vector<int> order((size_t)n);
    {
        std::vector<long> fsz((size_t)n, 0);
        for (int i = 0; i < n; ++i) { order[(size_t)i] = i; struct stat sb; if (stat(bam_paths[i], &sb) == 0) fsz[(size_t)i] = (long)sb.st_size; }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fsz[(size_t)a] > fsz[(size_t)b]; });
    }
    auto worker = [&]() {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n || err.load()) break;
            const int i = order[(size_t)k];
            int rc;
            try {
                BamHeader h;
                rc = bam_read(bam_paths[i], h, ds->staged[first + (size_t)i], ds->staged_off[first + (size_t)i], 1);
                if (!rc) rc = check_header(*ds, h, bam_paths[i]);
            } catch (const std::exception &e) { rc = fail(MSNV_ENOMEM, "%s: %s", bam_paths[i], e.what()); }
            if (rc) { msgs[(size_t)i] = msnv_last_error(); err.store(rc); }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    if (err.load()) {
        ds->staged.resize(first); ds->staged_off.resize(first);
        for (const std::string &m : msgs) if (!m.empty()) return fail(err.load(), "%s", m.c_str());
        return fail(err.load(), "BAM decode failed");
    }
    return MSNV_OK;
}

extern "C" int msnv_dataset_add_synth_samples(msnv_dataset *ds, const msnv_synth_params *p, int32_t first, int32_t count, int32_t host_threads) {
    clear_error();
    if (!ds || !p || count < 0) return fail(MSNV_EINVAL, "msnv_dataset_add_synth_samples: bad argument");
    if (int rc = check_open(ds)) return rc;
    if ((size_t)msnv_synth_contig_count(p) != ds->names.size()) return fail(MSNV_EINVAL, "synthetic parameters describe %d contigs, dataset has %zu", msnv_synth_contig_count(p), ds->names.size());
    int nthreads = host_threads > 0 ? host_threads : (int)msnv_default_threads();
    nthreads = std::min(nthreads, std::max(1, (int)count));
    const std::vector<std::string> contigs = synth_contigs(*p);
    if (pack_on_device(ds)) {
        // record streams are made by the host threads, group by group, and packed in HBM
        const size_t base0 = ds->samples.size(), rounds_at_entry = ds->dp.rounds.size();
        const int group = std::max(nthreads, 1);
        for (int g0 = 0; g0 < count; g0 += group) {
            const int g1 = std::min<int>(count, g0 + group);
            std::vector<std::vector<uint8_t>> recs((size_t)(g1 - g0));
            std::atomic<int> nxt{g0}, bad{0};
            const auto t_synth = std::chrono::steady_clock::now();     // (wall seconds of MAKING the group's streams: the generator, not the product)
            auto w = [&]() {
                for (;;) {
                    const int i = nxt.fetch_add(1);
                    if (i >= g1) break;
                    try { synth_sample_records(*p, first + i, contigs, recs[(size_t)(i - g0)]); } catch (const std::exception &) { bad.store(1); }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < std::min(nthreads, g1 - g0); ++t) th.emplace_back(w);
            for (auto &t : th) t.join();
            if (bad.load()) return fail_multi_add(ds, base0, rounds_at_entry, fail(MSNV_ENOMEM, "making a synthetic sample failed"));
            host_timer_add(HT_SYNTH_WALL, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_synth).count());
            std::vector<const uint8_t *> ptrs; std::vector<uint64_t> sizes;
            for (auto &r : recs) { ptrs.push_back(r.data()); sizes.push_back(r.size()); }
            if (int rc = add_streams_device(ds, ptrs.data(), sizes.data(), g1 - g0, false)) return fail_multi_add(ds, base0, rounds_at_entry, rc);
        }
        return MSNV_OK;
    }
    const size_t base = ds->samples.size();
    ds->samples.resize(base + (size_t)count);
    std::atomic<int> next{0}, err{0};
    std::string msg;
    auto worker = [&]() {
        std::vector<uint8_t> rec;
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= count || err.load()) break;
            int rc;
            try {
                synth_sample_records(*p, first + i, contigs, rec);
                rc = pack_sample(*ds, rec.data(), rec.size(), ds->samples[base + (size_t)i]);
            } catch (const std::exception &) { rc = MSNV_ENOMEM; }
            if (rc) { err.store(rc); }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    if (err.load()) { ds->samples.resize(base); return fail(err.load(), "packing a synthetic sample failed"); }
    return MSNV_OK;
}

extern "C" int msnv_dataset_finalize(msnv_dataset *ds) {
    clear_error();
    if (!ds) return fail(MSNV_EINVAL, "msnv_dataset_finalize: NULL dataset");
    if (int rc = check_open(ds, true)) return rc;
    if (!ds->ctx) return fail(MSNV_ENODEV, "this dataset was created without a device context: only the host-stage entry points work on it (no CPU fallback)");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    // The staged streams go back to the system on threads of their own, and only once the device dataset stands.  Giving gigabytes back
    // is not free where the kernel clears pages on release (init_on_free: the GPU box needs 0.14 s for the 3 GB of the benchmark shape,
    // huge pages and all -- profiles/thp_probe.sh), it is serial per munmap, and it holds the lock of the address space that every
    // hipMalloc and page fault of the caller needs: so several threads share the buffers, and they start when finalize_dataset is done
    // (beside it: finalize 0.03 -> 0.15 s).  MSNV_STAGE_FREE=s: on the caller's thread.
    struct FreeLater {
        std::vector<ByteBuf> bufs;
        ~FreeLater() {
            if (bufs.empty()) return;
            if (const char *e = getenv("MSNV_STAGE_FREE")) if (e[0] == 's') { bufs.clear(); return; }
            if (const char *e = getenv("MSNV_STAGE_FREE")) if (e[0] == 'n') { static std::vector<ByteBuf> keep; for (ByteBuf &b : bufs) keep.push_back(std::move(b)); return; }
            const size_t nt = std::min<size_t>(bufs.size(), std::min<size_t>(16, msnv_default_threads()));
            std::vector<std::vector<ByteBuf>> share(nt);
            for (size_t i = 0; i < bufs.size(); ++i) share[i % nt].push_back(std::move(bufs[i]));
            // (pages first, under the SHARED lock of the address space -- msnv_drop_pages -- so that the threads really work side by side
            // and the caller's allocations get in between; the unmapping that follows finds nothing left to give back)
            for (size_t t = 0; t < nt; ++t) std::thread([b = std::move(share[t])]() mutable { for (ByteBuf &x : b) { msnv_drop_pages(x.data(), x.size()); } b.clear(); }).detach();
        }
    } free_later;
    if (!ds->staged.empty()) {                                   // streams staged while the device was still coming up: packed now
        std::vector<const uint8_t *> ptrs; std::vector<uint64_t> sizes;
        for (size_t i = 0; i < ds->staged.size(); ++i) { ptrs.push_back(ds->staged[i].data() + ds->staged_off[i]); sizes.push_back(ds->staged[i].size() - ds->staged_off[i]); }
        int rc;
        if (pack_on_device(ds)) { rc = add_streams_device(ds, ptrs.data(), sizes.data(), (int)ptrs.size(), false); free_later.bufs.swap(ds->staged); }
        else {
            free_later.bufs.swap(ds->staged);                    // (msnv_dataset_add_sample_records_many refuses a dataset with staged streams)
            rc = msnv_dataset_add_sample_records_many(ds, ptrs.data(), sizes.data(), (int32_t)ptrs.size(), 0);
        }
        ds->staged.clear(); ds->staged_off.clear();
        if (rc) return rc;
    }
    HostTimerScope ts(HT_UPLOAD_WALL);
    try { return finalize_dataset(*ds); }
    catch (const std::exception &e) { return fail(MSNV_ENOMEM, "building the device dataset failed: %s", e.what()); }
}

// What the device pack of this dataset cost so far (cumulative over its rounds; devpack.hip): kernel milliseconds by stage (HIP events on the
// context's stream), wall seconds of the transfers and of the host pre-pass, and counts.
extern "C" int msnv_dataset_pack_stats(const msnv_dataset *ds, double *out, int32_t n) {
    clear_error();
    if (!ds || !out || n < 0) return fail(MSNV_EINVAL, "msnv_dataset_pack_stats: bad argument");
    if (devpack_sync_pending(*const_cast<msnv_dataset *>(ds))) clear_error();      // (the last round's kernels: their time belongs to the figures)
    const DevPackTables &t = ds->dp;
    const double v[MSNV_PACK_STATS] = {t.ms_scan, t.ms_measure, t.ms_depth, t.ms_emit, t.ms_sort, t.wall_upload_s, t.wall_download_s, t.wall_prepass_s,
                                       (double)t.raw_bytes, (double)t.n_records, (double)t.n_pieces, (double)t.n_prepass_samples, (double)t.n_scan_redone, (double)t.n_deep_runs_split, (double)t.n_dense_samples, (double)t.n_quick_redone, (double)t.n_device_edit_samples};
    for (int i = 0; i < n; ++i) out[i] = i < MSNV_PACK_STATS ? v[i] : 0.0;
    return MSNV_OK;
}

// Inspection hook (tests/test_gpu_devpack.py: the device pack and the host pack build the same dataset): the bytes of one device
// column or index table of a finalized dataset.
extern "C" int msnv_dataset_fetch_column(msnv_dataset *ds, const char *name, uint8_t *out, uint64_t capacity, uint64_t *n_bytes) {
    clear_error();
    if (!ds || !name || !n_bytes) return fail(MSNV_EINVAL, "msnv_dataset_fetch_column: NULL argument");
    if (!ds->finalized || !ds->dev) return fail(MSNV_EINVAL, "msnv_dataset_fetch_column: dataset is not finalized");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    const DeviceCols &d = *ds->dev;
    const std::string k(name);
    const void *p = nullptr; uint64_t n = 0;
    const uint64_t S = d.n_samples, nt = d.n_tiles;
    if (k == "hdr") { p = d.hdr; n = d.n_reads * sizeof(ReadHdr); }
    else if (k == "hdr4") { p = d.hdr4; n = d.hdr4 ? d.n_reads * 4 : 0; }
    else if (k == "hdr8m") { p = d.hdr8m; n = d.n_hdr8m * sizeof(PieceHdr); }
    else if (k == "blk") { p = d.blk; n = d.blk ? d.n_blk * 4 : 0; }
    else if (k == "seq") { p = d.seq; n = d.n_seq_bytes; }
    else if (k == "qual") { p = d.qual; n = d.n_seq_bytes / 4; }
    else if (k == "s_read_base") { p = d.s_read_base; n = (S + 1) * 8; }
    else if (k == "s_seq_base") { p = d.s_seq_base; n = (S + 1) * 8; }
    else if (k == "ref4") { p = d.ref4; n = (nt * TILE / 8 + 1) * 4; }
    else if (k == "pairs") { p = d.pairs; n = (uint64_t)d.n_pairs * sizeof(TilePair); }
    else if (k == "work") { p = d.work; n = (uint64_t)d.n_work * sizeof(WorkItem); }
    else if (k == "chunks") { p = d.chunks; n = d.n_chunks * sizeof(ChunkDesc); }
    else if (k == "cov_iv") { p = d.cov_iv; n = d.n_cov_iv * sizeof(Pair32); }
    else if (k == "cov_pairs") { p = d.cov_pairs; n = (uint64_t)d.n_cov_pairs * sizeof(TilePair); }
    else if (k == "cov_work") { p = d.cov_work; n = (uint64_t)d.n_cov_work * sizeof(WorkItem); }
    else return fail(MSNV_EINVAL, "msnv_dataset_fetch_column: no column named %s", name);
    *n_bytes = n;
    if (!out) return MSNV_OK;                                      // size query
    if (capacity < n) return fail(MSNV_ECAPACITY, "msnv_dataset_fetch_column: %s holds %llu bytes, capacity %llu", name, (unsigned long long)n, (unsigned long long)capacity);
    if (n && !p) return fail(MSNV_EINVAL, "msnv_dataset_fetch_column: %s is not allocated in this layout", name);
    return dev_download(out, p, n);
}

extern "C" int msnv_dataset_info_get(const msnv_dataset *ds, msnv_dataset_info *out) {
    clear_error();
    if (!ds || !out) return fail(MSNV_EINVAL, "msnv_dataset_info_get: NULL argument");
    *out = ds->info;
    if (ds->dev) { out->n_whole_tile_items = ds->dev->n_fused_tiles; out->n_listed_tiles = ds->dev->last_ovf_tiles; }
    return MSNV_OK;
}

// ------------------------------------------------------------------------------ pipeline
namespace msnv { int dev_reserve_passes(DeviceCols &d, int n); }
extern "C" int msnv_pileup_reserve(msnv_dataset *ds, int32_t n) {
    clear_error();
    if (!ds || !ds->finalized || n <= 0) return fail(MSNV_EINVAL, "msnv_pileup_reserve: dataset is not finalized or n <= 0");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    return dev_reserve_passes(*ds->dev, n);
}

extern "C" int msnv_pileup_run_many(msnv_dataset *ds, int32_t n, int32_t overlap, msnv_run_stats *stats) {
    clear_error();
    if (!ds || !ds->finalized || n <= 0) return fail(MSNV_EINVAL, "msnv_pileup_run_many: dataset is not finalized or n <= 0");
    if (!ds->have_results) return fail(MSNV_EINVAL, "msnv_pileup_run_many: run msnv_pileup_run once first (it sizes the sparse buffers)");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    RunCounts c{};
    std::vector<msnv_run_stats> tmp((size_t)n);
    int rc = dev_run_pipeline_many(*ds->dev, ds->params, ds->ctx->stream, n, overlap != 0, tmp.data(), &c);
    if (rc == MSNV_ECAPACITY) return fail(rc, "%s", msnv_last_error());
    if (rc) return rc;
    ds->have_results = true; ds->results_fetched = false; ds->ann_valid = false;
    ds->last_counts_sites = c.n_sites;
    ds->last_stats = tmp.back();
    if (stats) memcpy(stats, tmp.data(), tmp.size() * sizeof(msnv_run_stats));
    return MSNV_OK;
}

static void gpos_to_contig(const msnv_dataset &ds, uint32_t gpos, int32_t &tid, int32_t &pos) {
    const uint32_t tile = gpos / TILE;
    tid = (int32_t)ds.tile_contig[tile];
    pos = (int32_t)(gpos - ds.tile_base[(size_t)tid] * TILE);
}

extern "C" int msnv_pileup_run(msnv_dataset *ds, msnv_run_stats *stats) {
    clear_error();
    if (!ds || !ds->finalized) return fail(MSNV_EINVAL, "msnv_pileup_run: dataset is not finalized");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    DeviceCols &d = *ds->dev;
    msnv_run_stats st{};
    RunCounts c{};
    int rc = dev_run_pipeline(d, ds->params, ds->ctx->stream, &st, &c);
    for (int attempt = 0; rc == MSNV_ECAPACITY && attempt < 3; ++attempt) {      // (sparse buffers; whole-tile items sent back to the unfused path)
        // grow the sparse buffers to what the failed pass asked for and run again
        auto grow = [&](void **p, uint32_t &cap, uint32_t need, size_t elem) -> int {
            if (need <= cap) return MSNV_OK;
            dev_free(*p); *p = nullptr;
            cap = (uint32_t)std::min<uint64_t>(0x7fffffffull, (uint64_t)need + need / 4 + 1024);
            return dev_alloc(p, (uint64_t)cap * elem, &d.device_bytes);
        };
        if (int r2 = grow((void **)&d.events, d.cap_events, c.n_events, sizeof(Pair32))) return r2;
        if (int r2 = grow((void **)&d.overflow, d.cap_overflow, c.n_overflow, sizeof(Pair32))) return r2;
        const uint32_t old_cap_sites = d.cap_sites;
        if (int r2 = grow((void **)&d.sites, d.cap_sites, c.n_sites, sizeof(SiteRec))) return r2;
        if (d.cap_sites != old_cap_sites) {                 // the list of sites msnv_decide_sites looks at is sized like the sites
            dev_free(d.unc_sites); d.unc_sites = nullptr;
            if (int r2 = dev_alloc((void **)&d.unc_sites, (uint64_t)d.cap_sites * sizeof(uint32_t), &d.device_bytes)) return r2;
        }
        clear_error();
        rc = dev_run_pipeline(d, ds->params, ds->ctx->stream, &st, &c);
    }
    if (rc == MSNV_ECAPACITY) return fail(rc, "%s", msnv_last_error());
    if (rc) return rc;

    ds->have_results = true;
    ds->results_fetched = false;
    ds->ann_valid = false;
    ds->last_counts_sites = c.n_sites;
    ds->last_stats = st;
    if (stats) *stats = st;
    return MSNV_OK;
}

// Downloads the records of the last pass and maps them back to (contig, position) in output order.
static int fetch_results(msnv_dataset *ds) {
    if (ds->results_fetched) return MSNV_OK;
    DeviceCols &d = *ds->dev;
    const uint32_t n = ds->last_counts_sites;
    std::vector<SiteRec> sites(n);
    std::vector<uint8_t> flags(n);
    std::vector<uint32_t> tbase(ds->n_tiles + 1), tcnt(ds->n_tiles + 1);
    if (int r2 = dev_download(sites.data(), d.sites, (uint64_t)n * sizeof(SiteRec))) return r2;
    if (int r2 = dev_download(flags.data(), d.site_flags, n)) return r2;
    if (int r2 = dev_download(tbase.data(), d.tile_site_base, (uint64_t)ds->n_tiles * 4)) return r2;
    if (int r2 = dev_download(tcnt.data(), d.tile_site_cnt, (uint64_t)ds->n_tiles * 4)) return r2;
    // per-sample cells are stored per (site, slot of the tile) on the device (kernels.hip: CellMap); expanded to all samples here
    std::vector<unsigned long long> tcell(ds->n_tiles + 1);
    if (int r2 = dev_download(tcell.data(), d.tile_cell_base, (uint64_t)ds->n_tiles * sizeof(unsigned long long))) return r2;
    const uint64_t n_cells = d.last_cells;
    std::vector<msnv_site_sample> raw((size_t)n_cells);
    {   // five u16 columns on the device (coverage and the four allele counts, device.h): zipped into records here
        std::vector<uint16_t> col((size_t)n_cells);
        if (int r2 = dev_download(col.data(), d.cov_col, col.size() * sizeof(uint16_t))) return r2;
        for (size_t i = 0; i < raw.size(); ++i) raw[i].cov = col[i];
        for (int x = 0; x < 4; ++x) {
            if (int r2 = dev_download(col.data(), d.ncol + (uint64_t)x * d.cap_cells, col.size() * sizeof(uint16_t))) return r2;
            for (size_t i = 0; i < raw.size(); ++i) raw[i].n[x] = col[i];
        }
    }

    ds->sites.clear(); ds->site_samples.clear(); ds->site_dev_index.clear();
    ds->site_row.clear(); ds->site_cell_sample.clear(); ds->site_cells.clear();
    ds->sites.reserve(n);
    ds->site_row.push_back(0);
    for (uint32_t t = 0; t < ds->n_tiles; ++t) {
        const uint64_t slot0 = ds->tile_slot_base[t], n_slots = ds->tile_slot_base[t + 1] - slot0;
        for (uint32_t j = 0; j < tcnt[t]; ++j) {
            const uint32_t i = tbase[t] + j;
            if (!flags[i]) continue;               // passed the gates but neither rule fired
            msnv_site s{};
            gpos_to_contig(*ds, sites[i].gpos, s.tid, s.pos);
            s.cov = sites[i].cov;
            for (int x = 0; x < 4; ++x) s.n[x] = sites[i].n[x];
            s.pop_mask = flags[i] & 15; s.ind_mask = flags[i] >> 4;
            const std::string &seq = ds->seqs[(size_t)s.tid];
            s.refchar = (uint8_t)((ds->has_seq[(size_t)s.tid] && (size_t)s.pos < seq.size()) ? seq[(size_t)s.pos] : 'N');
            s.dropped = (ds->params.drop_first_line && s.tid == ds->first_tid && s.pos == ds->first_pos) ? 1 : 0;
            ds->sites.push_back(s);
            ds->site_dev_index.push_back(i);
            const uint64_t cell0 = tcell[t] + (uint64_t)j * ds->tile_slot_stride[t];
            if (cell0 + n_slots > n_cells) return fail(MSNV_EHIP, "internal: the device reported %llu cells but tile %u needs cell %llu", (unsigned long long)n_cells, t, (unsigned long long)(cell0 + n_slots));
            // the site's row of cells: the tile's slots in SAMPLE order (pack.cpp sorts a tile's pairs by kind, so the slots are not),
            // without the samples that hold nothing at this position
            const size_t c_lo = ds->site_cells.size();
            for (uint64_t c = 0; c < n_slots; ++c) {
                const msnv_site_sample &v = raw[(size_t)(cell0 + c)];
                if (!(v.cov | v.n[0] | v.n[1] | v.n[2] | v.n[3])) continue;
                ds->site_cell_sample.push_back(ds->slot_sample[(size_t)(slot0 + c)]);
                ds->site_cells.push_back(v);
            }
            const size_t n_c = ds->site_cells.size() - c_lo;
            bool sorted = true;
            for (size_t k = 1; k < n_c && sorted; ++k) sorted = ds->site_cell_sample[c_lo + k - 1] < ds->site_cell_sample[c_lo + k];
            if (!sorted) {
                std::vector<std::pair<uint32_t, msnv_site_sample>> tmp(n_c);
                for (size_t k = 0; k < n_c; ++k) tmp[k] = {ds->site_cell_sample[c_lo + k], ds->site_cells[c_lo + k]};
                std::sort(tmp.begin(), tmp.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
                for (size_t k = 0; k < n_c; ++k) { ds->site_cell_sample[c_lo + k] = tmp[k].first; ds->site_cells[c_lo + k] = tmp[k].second; }
            }
            ds->site_row.push_back((uint64_t)ds->site_cells.size());
        }
    }
    ds->results_fetched = true;
    return MSNV_OK;
}

extern "C" int msnv_results_count(const msnv_dataset *ds, uint64_t *n_sites) {
    clear_error();
    if (!ds || !n_sites) return fail(MSNV_EINVAL, "msnv_results_count: NULL argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    if (int rc = fetch_results(const_cast<msnv_dataset *>(ds))) return rc;
    *n_sites = ds->sites.size();
    return MSNV_OK;
}

extern "C" int msnv_results_fetch(msnv_dataset *ds, msnv_site *sites, msnv_site_sample *samples, uint64_t capacity) {
    clear_error();
    if (!ds || !sites || !samples) return fail(MSNV_EINVAL, "msnv_results_fetch: NULL argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    if (int rc = fetch_results(ds)) return rc;
    if (capacity < ds->sites.size()) return fail(MSNV_ECAPACITY, "capacity %llu < %zu sites", (unsigned long long)capacity, ds->sites.size());
    memcpy(sites, ds->sites.data(), ds->sites.size() * sizeof(msnv_site));
    const size_t S = ds->samples.size();
    memset(samples, 0, ds->sites.size() * S * sizeof(msnv_site_sample));
    for (size_t i = 0; i < ds->sites.size(); ++i)
        for (uint64_t c = ds->site_row[i]; c < ds->site_row[i + 1]; ++c) samples[i * S + ds->site_cell_sample[(size_t)c]] = ds->site_cells[(size_t)c];
    return MSNV_OK;
}

extern "C" int msnv_results_cells_count(const msnv_dataset *ds, uint64_t *n_sites, uint64_t *n_cells) {
    clear_error();
    if (!ds || !n_sites || !n_cells) return fail(MSNV_EINVAL, "msnv_results_cells_count: NULL argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    if (int rc = fetch_results(const_cast<msnv_dataset *>(ds))) return rc;
    *n_sites = ds->sites.size(); *n_cells = ds->site_cells.size();
    return MSNV_OK;
}

extern "C" int msnv_results_fetch_cells(msnv_dataset *ds, msnv_site *sites, uint64_t *row_off, uint32_t *cell_sample, msnv_site_sample *cells,
                                        uint64_t cap_sites, uint64_t cap_cells) {
    clear_error();
    if (!ds || !row_off) return fail(MSNV_EINVAL, "msnv_results_fetch_cells: NULL argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    if (int rc = fetch_results(ds)) return rc;
    if (cap_sites < ds->sites.size() || cap_cells < ds->site_cells.size())
        return fail(MSNV_ECAPACITY, "capacity %llu sites / %llu cells < %zu / %zu", (unsigned long long)cap_sites, (unsigned long long)cap_cells, ds->sites.size(), ds->site_cells.size());
    if ((ds->sites.size() && !sites) || (ds->site_cells.size() && (!cell_sample || !cells))) return fail(MSNV_EINVAL, "msnv_results_fetch_cells: NULL argument");
    if (!ds->sites.empty()) memcpy(sites, ds->sites.data(), ds->sites.size() * sizeof(msnv_site));
    memcpy(row_off, ds->site_row.data(), ds->site_row.size() * sizeof(uint64_t));
    if (!ds->site_cells.empty()) {
        memcpy(cell_sample, ds->site_cell_sample.data(), ds->site_cell_sample.size() * sizeof(uint32_t));
        memcpy(cells, ds->site_cells.data(), ds->site_cells.size() * sizeof(msnv_site_sample));
    }
    return MSNV_OK;
}

// ---- gene / codon annotation (snpCall -g): tables are built once per (annotation, FASTA) pair and stay on the device
static int annotate_sites(msnv_dataset *ds, const char *ann_path, const char *fasta_path, double *ms_kernel) {
    if (!ann_path || !fasta_path) return fail(MSNV_EINVAL, "annotation needs both the gene table and the FASTA (call_vC.cpp:448)");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    DeviceCols &d = *ds->dev;
    const std::string key = std::string(ann_path) + "\n" + fasta_path;
    if (!d.ann.ready || d.ann.key != key) {
        Annotation an;
        if (int rc = load_annotation(ann_path, fasta_path, an)) return rc;
        AnnHost h;
        if (int rc = ann_build(*ds, an, h)) return rc;
        if (int rc = dev_ann_upload(d, h)) return rc;
        ann_gene_names(an, ds->names, d.ann.gene_names);
        d.ann.key = key;
    }
    uint32_t drop_gpos = UINT32_MAX;
    if (ds->params.drop_first_line && ds->first_tid >= 0 && ds->tile_base[(size_t)ds->first_tid] != UINT32_MAX)
        drop_gpos = ds->tile_base[(size_t)ds->first_tid] * TILE + (uint32_t)ds->first_pos;
    uint32_t err[2] = {UINT32_MAX, UINT32_MAX};
    if (int rc = dev_annotate(d, ds->last_counts_sites, drop_gpos, ds->ctx->stream, ms_kernel, err)) return rc;
    for (int k = 0; k < 2; ++k) {
        if (err[k] == UINT32_MAX) continue;
        int32_t tid, pos;
        gpos_to_contig(*ds, err[k], tid, pos);
        return fail(MSNV_EDOMAIN, k == 0 ? "contig %s has genes but no FASTA record (position %d; reference: undefined behaviour)"
                                         : "codon at %s:%d runs past the contig end (reference: undefined behaviour)",
                    ds->names[(size_t)tid].c_str(), pos + 1);
    }
    ds->ann_valid = true;
    return MSNV_OK;
}

static int fetch_ann(msnv_dataset *ds, std::vector<msnv_site_ann> &out) {
    if (int rc = fetch_results(ds)) return rc;
    std::vector<msnv_site_ann> raw(ds->last_counts_sites);
    if (int rc = dev_download(raw.data(), ds->dev->ann.out, raw.size() * sizeof(msnv_site_ann))) return rc;
    out.resize(ds->sites.size());
    for (size_t i = 0; i < out.size(); ++i) out[i] = raw[ds->site_dev_index[i]];
    return MSNV_OK;
}

extern "C" int msnv_annotate_run(msnv_dataset *ds, const char *ann_path, const char *fasta_path, double *ms_kernel) {
    clear_error();
    if (!ds) return fail(MSNV_EINVAL, "msnv_annotate_run: NULL argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    return annotate_sites(ds, ann_path, fasta_path, ms_kernel);
}

extern "C" int msnv_results_fetch_ann(msnv_dataset *ds, msnv_site_ann *ann, uint64_t capacity) {
    clear_error();
    if (!ds || !ann) return fail(MSNV_EINVAL, "msnv_results_fetch_ann: NULL argument");
    if (!ds->have_results || !ds->ann_valid) return fail(MSNV_EINVAL, "no annotation: call msnv_annotate_run after msnv_pileup_run");
    std::vector<msnv_site_ann> v;
    if (int rc = fetch_ann(ds, v)) return rc;
    if (capacity < v.size()) return fail(MSNV_ECAPACITY, "capacity %llu < %zu sites", (unsigned long long)capacity, v.size());
    if (!v.empty()) memcpy(ann, v.data(), v.size() * sizeof(msnv_site_ann));
    return MSNV_OK;
}

extern "C" int msnv_write_calls(msnv_dataset *ds, const char *called_path, const char *indiv_path,
                                const char *ann_path, const char *fasta_path) {
    clear_error();
    if (!ds || !called_path) return fail(MSNV_EINVAL, "msnv_write_calls: NULL argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    if (int rc = fetch_results(ds)) return rc;
    if (!(ann_path && fasta_path)) return write_calls_text(*ds, called_path, indiv_path, nullptr, nullptr);   // call_vC.cpp:448
    if (int rc = annotate_sites(ds, ann_path, fasta_path, nullptr)) return rc;
    std::vector<msnv_site_ann> ann;
    if (int rc = fetch_ann(ds, ann)) return rc;
    return write_calls_text(*ds, called_path, indiv_path, ann.data(), &ds->dev->ann.gene_names);
}

extern "C" int msnv_dataset_first_line(const msnv_dataset *ds, int32_t *tid, int32_t *pos) {
    clear_error();
    if (!ds || !tid || !pos) return fail(MSNV_EINVAL, "msnv_dataset_first_line: NULL argument");
    if (!ds->finalized) return fail(MSNV_EINVAL, "msnv_dataset_first_line: dataset is not finalized");
    *tid = ds->first_tid; *pos = (int32_t)ds->first_pos;
    return MSNV_OK;
}

extern "C" int msnv_dataset_first_lines(const msnv_dataset *ds, int32_t *first_any, int32_t *first_from1, int32_t n) {
    clear_error();
    if (!ds || !first_any || !first_from1) return fail(MSNV_EINVAL, "msnv_dataset_first_lines: NULL argument");
    if ((size_t)n != ds->names.size()) return fail(MSNV_EINVAL, "msnv_dataset_first_lines: %d entries, header has %zu contigs", n, ds->names.size());
    if (ds->has_bed) return fail(MSNV_EINVAL, "msnv_dataset_first_lines: the dataset was restricted with a BED file");
    for (int c = 0; c < n; ++c) { first_any[c] = -1; first_from1[c] = -1; }
    for (const SampleCols &sc : ds->samples) {
        if (sc.first_any.empty()) continue;
        for (int c = 0; c < n; ++c) {
            const int32_t a = sc.first_any[(size_t)c], b = sc.first_from1[(size_t)c];
            if (a >= 0 && (first_any[c] < 0 || a < first_any[c])) first_any[c] = a;
            if (b >= 0 && (first_from1[c] < 0 || b < first_from1[c])) first_from1[c] = b;
        }
    }
    return MSNV_OK;
}

// formatter over records that did not come from a local run: `tmp` holds names, sample count and the records in either form
static int write_gathered(msnv_dataset &tmp, const msnv_ref_desc *ref, uint64_t n_sites, const char *called_path, const char *indiv_path,
                          const char *ann_path, const char *fasta_path, const msnv_site_ann *ann) {
    for (uint64_t i = 0; i < n_sites; ++i)
        if (tmp.sites[i].tid < 0 || tmp.sites[i].tid >= ref->n_contigs) return fail(MSNV_EINVAL, "record %llu names contig %d", (unsigned long long)i, tmp.sites[i].tid);
    const bool annotated = ann_path && fasta_path;
    if (!annotated) return write_calls_text(tmp, called_path, indiv_path, nullptr, nullptr);
    Annotation an;                          // gene names only; the codon work was done on the ranks' devices
    if (int rc = load_annotation(ann_path, nullptr, an)) return rc;
    std::vector<std::string> gene_names;
    ann_gene_names(an, tmp.names, gene_names);
    for (uint64_t i = 0; i < n_sites; ++i)
        if (ann[i].gene >= (int32_t)gene_names.size()) return fail(MSNV_EINVAL, "annotation record %llu names gene %d of %zu", (unsigned long long)i, ann[i].gene, gene_names.size());
    return write_calls_text(tmp, called_path, indiv_path, ann, &gene_names);
}

extern "C" int msnv_write_calls_records(const msnv_ref_desc *ref, int32_t n_samples, const msnv_site *sites,
                                        const msnv_site_sample *samples, uint64_t n_sites,
                                        const char *called_path, const char *indiv_path,
                                        const char *ann_path, const char *fasta_path, const msnv_site_ann *ann) {
    clear_error();
    if (!ref || !called_path || n_samples < 0 || (n_sites && (!sites || !samples))) return fail(MSNV_EINVAL, "msnv_write_calls_records: bad argument");
    if (ann_path && fasta_path && n_sites && !ann) return fail(MSNV_EINVAL, "msnv_write_calls_records: annotation records are required with ann_path (the annotation is computed on the device)");
    msnv_dataset tmp;                       // formatter state only: names, sample count, records
    for (int i = 0; i < ref->n_contigs; ++i) tmp.names.emplace_back(ref->names[i]);
    tmp.samples.resize((size_t)n_samples);
    tmp.sites.assign(sites, sites + n_sites);
    tmp.site_samples.assign(samples, samples + n_sites * (uint64_t)n_samples);
    return write_gathered(tmp, ref, n_sites, called_path, indiv_path, ann_path, fasta_path, ann);
}

extern "C" int msnv_write_calls_cells(const msnv_ref_desc *ref, int32_t n_samples, const msnv_site *sites, const uint64_t *row_off,
                                      const uint32_t *cell_sample, const msnv_site_sample *cells, uint64_t n_sites,
                                      const char *called_path, const char *indiv_path,
                                      const char *ann_path, const char *fasta_path, const msnv_site_ann *ann) {
    clear_error();
    if (!ref || !called_path || n_samples < 0 || !row_off || (n_sites && !sites)) return fail(MSNV_EINVAL, "msnv_write_calls_cells: bad argument");
    if (ann_path && fasta_path && n_sites && !ann) return fail(MSNV_EINVAL, "msnv_write_calls_cells: annotation records are required with ann_path (the annotation is computed on the device)");
    const uint64_t n_cells = row_off[n_sites];
    if (row_off[0] != 0 || (n_cells && (!cell_sample || !cells))) return fail(MSNV_EINVAL, "msnv_write_calls_cells: bad row offsets");
    for (uint64_t i = 0; i < n_sites; ++i) if (row_off[i + 1] < row_off[i]) return fail(MSNV_EINVAL, "msnv_write_calls_cells: row offsets of record %llu decrease", (unsigned long long)i);
    for (uint64_t c = 0; c < n_cells; ++c) if (cell_sample[c] >= (uint32_t)n_samples) return fail(MSNV_EINVAL, "msnv_write_calls_cells: cell %llu names sample %u of %d", (unsigned long long)c, cell_sample[c], n_samples);
    msnv_dataset tmp;
    for (int i = 0; i < ref->n_contigs; ++i) tmp.names.emplace_back(ref->names[i]);
    tmp.samples.resize((size_t)n_samples);
    tmp.sites.assign(sites, sites + n_sites);
    tmp.site_row.assign(row_off, row_off + n_sites + 1);
    tmp.site_cell_sample.assign(cell_sample, cell_sample + n_cells);
    tmp.site_cells.assign(cells, cells + n_cells);
    return write_gathered(tmp, ref, n_sites, called_path, indiv_path, ann_path, fasta_path, ann);
}

namespace msnv {
int text_call(msnv_ctx *ctx, const char *text, uint64_t n_text, const msnv_params &p, const char *ref_fasta, const char *ann_path,
              const char *called_path, const char *indiv_path, uint64_t stats[8]);
}

extern "C" int msnv_call_from_mpileup(msnv_ctx *ctx, const msnv_mpileup_args *a, uint64_t stats[8]) {
    clear_error();
    if (!ctx || !a || !a->out_called_path) return fail(MSNV_EINVAL, "msnv_call_from_mpileup: NULL argument");
    if (a->params.min_coverage < 0 || a->params.calling_threshold < 0 || !(a->params.min_fraction >= 0.0)) return fail(MSNV_EINVAL, "negative cutoff in msnv_params");
    if (a->ann_path && !a->ref_fasta) return fail(MSNV_EINVAL, "annotation needs both the gene table and the FASTA (call_vC.cpp:448)");
    try {
        if (a->text) return text_call(ctx, a->text, a->text_bytes, a->params, a->ref_fasta, a->ann_path, a->out_called_path, a->out_indiv_path, stats);
        // the whole text in memory (a pileup of the benchmark shape is ~3.6 GB; the reference streams it line by line)
        const bool from_stdin = !a->mpileup_path || !strcmp(a->mpileup_path, "-");
        FILE *f = from_stdin ? stdin : fopen(a->mpileup_path, "rb");
        if (!f) return fail(MSNV_EIO, "cannot open %s", a->mpileup_path);
        // read straight into one buffer that grows with realloc (large blocks are remapped, not copied: no second copy of a
        // multi-GB pileup at the peak, which a growing std::string makes); a regular file is sized up front
        size_t cap = 64u << 20, len = 0;
        if (!from_stdin && fseek(f, 0, SEEK_END) == 0) { const long z = ftell(f); if (z > 0) cap = (size_t)z + 1; fseek(f, 0, SEEK_SET); }
        char *buf = (char *)malloc(cap);
        if (!buf) { if (!from_stdin) fclose(f); return fail(MSNV_ENOMEM, "out of memory for the pileup text"); }
        size_t n;
        while ((n = fread(buf + len, 1, cap - len, f)) > 0) {
            len += n;
            if (len == cap) {
                char *nb = (char *)realloc(buf, cap + cap / 2);
                if (!nb) { free(buf); if (!from_stdin) fclose(f); return fail(MSNV_ENOMEM, "out of memory for the pileup text"); }
                buf = nb; cap += cap / 2;
            }
        }
        const bool bad = ferror(f) != 0;
        if (!from_stdin) fclose(f);
        if (bad) { free(buf); return fail(MSNV_EIO, "read error on %s", from_stdin ? "stdin" : a->mpileup_path); }
        buf[len] = 0;                                                  // (len < cap here: the loop grows the buffer when it fills)
        int rc;
        try { rc = text_call(ctx, buf, len, a->params, a->ref_fasta, a->ann_path, a->out_called_path, a->out_indiv_path, stats); }
        catch (...) { free(buf); throw; }
        free(buf);
        return rc;
    } catch (const std::exception &e) { return fail(MSNV_ENOMEM, "msnv_call_from_mpileup: %s", e.what()); }
}

extern "C" int msnv_coverage_run(msnv_dataset *ds, msnv_run_stats *stats) {
    clear_error();
    if (!ds || !ds->finalized) return fail(MSNV_EINVAL, "msnv_coverage_run: dataset is not finalized");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    return coverage_run(*ds, stats);
}

// BASELINE configs[2]: qaCompute + snpCall from ONE resident dataset (the BAMs are decoded, packed and uploaded once;
// the two passes have different read filters and index spaces, so they stay two kernels over shared columns).
extern "C" int msnv_fused_run(msnv_dataset *ds, msnv_run_stats *pileup_stats, msnv_run_stats *coverage_stats) {
    if (int rc = msnv_coverage_run(ds, coverage_stats)) return rc;
    return msnv_pileup_run(ds, pileup_stats);
}

extern "C" int msnv_write_coverage(msnv_dataset *ds, int32_t sample_idx, const char *cov_path, const char *detail_path) {
    clear_error();
    if (!ds || !cov_path || !detail_path) return fail(MSNV_EINVAL, "msnv_write_coverage: NULL argument");
    return coverage_write(*ds, sample_idx, cov_path, detail_path);
}

// ------------------------------------------------------------------------------ multi-GPU: decode sharding + gathered coverage
namespace msnv {
int records_partition(const uint8_t *rec, uint64_t n_bytes, const int32_t *owner, int n_contigs, int n_parts, int cov_min_mapq,
                      uint8_t *out, uint64_t *part_bytes, msnv_sample_stats &st);
int coverage_write_rows(const std::vector<std::string> &names, const std::vector<int64_t> &lengths, int max_cov, const msnv_sample_stats &sc,
                        const unsigned long long *acc, const char *cov_path, const char *detail_path, int sample);
}
static_assert(MSNV_COV_WORDS == 1 + COV_BINS, "msnv.h and device.h disagree on the accumulator width");

extern "C" int msnv_records_partition(const uint8_t *records, uint64_t n_bytes, const int32_t *contig_owner, int32_t n_contigs,
                                      int32_t n_parts, int32_t cov_min_mapq, uint8_t *out, uint64_t *part_bytes, msnv_sample_stats *stats) {
    clear_error();
    if ((n_bytes && (!records || !out)) || !contig_owner || n_contigs < 0 || n_parts <= 0 || !part_bytes)
        return fail(MSNV_EINVAL, "msnv_records_partition: bad argument");
    msnv_sample_stats st{};
    const int rc = records_partition(records, n_bytes, contig_owner, n_contigs, n_parts, cov_min_mapq, out, part_bytes, st);
    if (!rc && stats) *stats = st;
    return rc;
}

extern "C" int msnv_records_deal_device(msnv_ctx *ctx, const uint8_t *const *streams, const uint64_t *n_bytes, int32_t n, int32_t on_device, const int32_t *owner, int32_t n_contigs,
                                        int32_t n_parts, int32_t cov_min_mapq, uint8_t *out, uint64_t capacity, uint64_t gap, uint64_t *part_bytes, msnv_sample_stats *stats,
                                        uint64_t *contig_bases) {
    clear_error();
    if (!ctx || n < 0 || n_contigs < 0 || (n && (!streams || !n_bytes || !part_bytes || !stats)) || (n_contigs && !owner) || (!out && capacity))      // (out = NULL, capacity = 0: measure only)
        return fail(MSNV_EINVAL, "msnv_records_deal_device: bad argument");
    for (int i = 0; i < n; ++i) if (n_bytes[i] && !streams[i]) return fail(MSNV_EINVAL, "msnv_records_deal_device: stream %d is NULL", i);
    try { return records_deal_device(ctx, streams, n_bytes, n, on_device != 0, owner, n_contigs, n_parts, cov_min_mapq, out, capacity, gap, part_bytes, stats, contig_bases); }
    catch (const std::exception &e) { return fail(MSNV_ENOMEM, "msnv_records_deal_device: %s", e.what()); }
}

extern "C" int msnv_records_contig_bases(const uint8_t *records, uint64_t n_bytes, int32_t n_contigs, uint64_t *bases) {
    clear_error();
    if ((n_bytes && !records) || n_contigs < 0 || (n_contigs && !bases)) return fail(MSNV_EINVAL, "msnv_records_contig_bases: bad argument");
    uint64_t off = 0;
    while (off < n_bytes) {
        RecView r;
        if (!rec_parse(records + off, n_bytes - off, r)) return fail(MSNV_EFORMAT, "malformed BAM record at byte %llu", (unsigned long long)off);
        off += r.size;
        if ((r.flag & BAM_FUNMAP) || r.tid < 0) continue;
        if (r.tid >= n_contigs) return fail(MSNV_EFORMAT, "record refers to contig %d but the header has %d", r.tid, n_contigs);
        uint64_t m = 0;
        for (int k = 0; k < r.n_cigar; ++k) { const uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u; if (t == C_M || t == C_EQ || t == C_X) m += c >> 4; }
        bases[r.tid] += m;
    }
    return MSNV_OK;
}

extern "C" int msnv_dataset_sample_stats(const msnv_dataset *ds, int32_t sample_idx, msnv_sample_stats *out) {
    clear_error();
    if (!ds || !out) return fail(MSNV_EINVAL, "msnv_dataset_sample_stats: NULL argument");
    if (sample_idx < 0 || (size_t)sample_idx >= ds->samples.size()) return fail(MSNV_EINVAL, "sample index %d out of range", sample_idx);
    *out = ds->samples[(size_t)sample_idx].st;
    return MSNV_OK;
}

extern "C" int msnv_coverage_fetch(msnv_dataset *ds, uint64_t *acc, uint64_t capacity_words) {
    clear_error();
    if (!ds || !acc) return fail(MSNV_EINVAL, "msnv_coverage_fetch: NULL argument");
    if (!ds->have_coverage) return fail(MSNV_EINVAL, "no coverage results: call msnv_coverage_run first");
    const uint64_t words = (uint64_t)ds->samples.size() * ds->names.size() * (1 + COV_BINS);
    if (capacity_words < words) return fail(MSNV_ECAPACITY, "capacity %llu < %llu words", (unsigned long long)capacity_words, (unsigned long long)words);
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "accumulators are 64-bit");
    memset(acc, 0, words * sizeof(uint64_t));
    for (size_t r = 0; r < ds->cov_row_sample.size(); ++r)
        memcpy(acc + ((uint64_t)ds->cov_row_sample[r] * ds->names.size() + ds->cov_row_contig[r]) * (1 + COV_BINS), &ds->cov_acc[r * (1 + COV_BINS)], (1 + COV_BINS) * sizeof(uint64_t));
    return MSNV_OK;
}

extern "C" int msnv_coverage_rows_count(const msnv_dataset *ds, uint64_t *n_rows) {
    clear_error();
    if (!ds || !n_rows) return fail(MSNV_EINVAL, "msnv_coverage_rows_count: NULL argument");
    if (!ds->have_coverage) return fail(MSNV_EINVAL, "no coverage results: call msnv_coverage_run first");
    *n_rows = ds->cov_row_sample.size();
    return MSNV_OK;
}

extern "C" int msnv_coverage_fetch_rows(msnv_dataset *ds, uint32_t *sample, uint32_t *contig, uint64_t *acc, uint64_t capacity_rows) {
    clear_error();
    if (!ds) return fail(MSNV_EINVAL, "msnv_coverage_fetch_rows: NULL argument");
    if (!ds->have_coverage) return fail(MSNV_EINVAL, "no coverage results: call msnv_coverage_run first");
    const size_t n = ds->cov_row_sample.size();
    if (capacity_rows < n) return fail(MSNV_ECAPACITY, "capacity %llu < %zu rows", (unsigned long long)capacity_rows, n);
    if (n && (!sample || !contig || !acc)) return fail(MSNV_EINVAL, "msnv_coverage_fetch_rows: NULL argument");
    if (n) {
        memcpy(sample, ds->cov_row_sample.data(), n * sizeof(uint32_t));
        memcpy(contig, ds->cov_row_contig.data(), n * sizeof(uint32_t));
        memcpy(acc, ds->cov_acc.data(), n * (1 + COV_BINS) * sizeof(uint64_t));
    }
    return MSNV_OK;
}

extern "C" int msnv_write_coverage_records(const msnv_ref_desc *ref, int32_t max_cov, const msnv_sample_stats *stats, const uint64_t *acc,
                                           const char *cov_path, const char *detail_path) {
    clear_error();
    if (!ref || !stats || !acc || !cov_path || !detail_path || ref->n_contigs < 0 || (ref->n_contigs && (!ref->names || !ref->lengths)))
        return fail(MSNV_EINVAL, "msnv_write_coverage_records: bad argument");
    std::vector<std::string> names; std::vector<int64_t> lens;
    for (int i = 0; i < ref->n_contigs; ++i) { names.emplace_back(ref->names[i]); lens.push_back(ref->lengths[i]); }
    return coverage_write_rows(names, lens, max_cov, *stats, reinterpret_cast<const unsigned long long *>(acc), cov_path, detail_path, 0);
}

// ------------------------------------------------------------------------------ filter_two (section 8 f1)
namespace msnv {
int filter_files(msnv_ctx *ctx, const char *const *paths, int n_paths, uint32_t n_samples, const FilterSpecies &sp,
                 double min_cov, double min_prop, const char *out_dir, uint64_t *n_lines_kept, double *ms_kernel);
}
namespace msnv { void py_repr(double x, std::string &out); }
static int make_filter_species(const msnv_filter_species *species, int32_t n_species, int32_t n_samples, FilterSpecies &sp, const char *who);
extern "C" int msnv_format_float(double x, char *buf, int32_t cap) {
    std::string s;
    py_repr(x, s);
    if (!buf || (int32_t)s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

extern "C" int msnv_filter_files(msnv_ctx *ctx, const char *const *snp_paths, int32_t n_paths, int32_t n_samples,
                                 const msnv_filter_species *species, int32_t n_species, double min_cov_c, double min_prop_p,
                                 const char *out_dir, uint64_t *n_positions_kept, double *ms_kernel) {
    clear_error();
    if (!ctx || n_paths < 0 || (n_paths && !snp_paths) || n_samples <= 0 || n_species < 0 || (n_species && !species) || !out_dir)
        return fail(MSNV_EINVAL, "msnv_filter_files: bad argument");
    if (int rc = dev_set_device(ctx->device)) return rc;
    FilterSpecies sp;
    if (int rc = make_filter_species(species, n_species, n_samples, sp, "msnv_filter_files")) return rc;
    if (ms_kernel) *ms_kernel = 0;
    if (n_positions_kept) *n_positions_kept = 0;
    if (!n_species) return MSNV_OK;
    return filter_files(ctx, snp_paths, n_paths, (uint32_t)n_samples, sp, min_cov_c, min_prop_p, out_dir, n_positions_kept, ms_kernel);
}

namespace msnv {
int filter_resident(msnv_dataset &ds, int which, const FilterSpecies &sp, double min_cov, double min_prop, const char *out_dir,
                    const msnv_site_ann *ann, const std::vector<std::string> *gene_names, uint64_t *n_lines_kept, double *ms_kernel);
}
static int make_filter_species(const msnv_filter_species *species, int32_t n_species, int32_t n_samples, FilterSpecies &sp, const char *who) {
    sp.soi_off.push_back(0);
    for (int i = 0; i < n_species; ++i) {
        const msnv_filter_species &s = species[i];
        if (!s.species || s.n_soi <= 0 || !s.soi || !s.soi_names)
            return fail(MSNV_EINVAL, "%s: species %d has no samples of interest (the reference divides by their number)", who, i);
        sp.name.emplace_back(s.species);
        sp.soi_names.emplace_back();
        for (int k = 0; k < s.n_soi; ++k) {
            if (s.soi[k] < 0 || s.soi[k] >= n_samples) return fail(MSNV_EINVAL, "%s: sample index %d out of range", who, s.soi[k]);
            sp.soi_idx.push_back((uint32_t)s.soi[k]);
            sp.soi_names.back().emplace_back(s.soi_names[k]);
        }
        sp.soi_off.push_back((uint32_t)sp.soi_idx.size());
    }
    return MSNV_OK;
}

extern "C" int msnv_filter_resident(msnv_dataset *ds, int32_t which, const msnv_filter_species *species, int32_t n_species,
                                    double min_cov_c, double min_prop_p, const char *out_dir, const char *ann_path, const char *fasta_path,
                                    uint64_t *n_positions_kept, double *ms_kernel) {
    clear_error();
    if (!ds || n_species < 0 || (n_species && !species) || !out_dir || (which != 0 && which != 1)) return fail(MSNV_EINVAL, "msnv_filter_resident: bad argument");
    if (!ds->have_results) return fail(MSNV_EINVAL, "no results: call msnv_pileup_run first");
    if (int rc = dev_set_device(ds->ctx->device)) return rc;
    if (ms_kernel) *ms_kernel = 0;
    if (n_positions_kept) *n_positions_kept = 0;
    if (!n_species) return MSNV_OK;
    FilterSpecies sp;
    if (int rc = make_filter_species(species, n_species, (int32_t)ds->samples.size(), sp, "msnv_filter_resident")) return rc;
    if (int rc = fetch_results(ds)) return rc;
    std::vector<msnv_site_ann> ann;
    const bool annotated = ann_path && fasta_path;
    if (annotated) {
        if (int rc = annotate_sites(ds, ann_path, fasta_path, nullptr)) return rc;
        if (int rc = fetch_ann(ds, ann)) return rc;
    }
    return filter_resident(*ds, which, sp, min_cov_c, min_prop_p, out_dir, annotated ? ann.data() : nullptr,
                           annotated ? &ds->dev->ann.gene_names : nullptr, n_positions_kept, ms_kernel);
}

// ------------------------------------------------------------------------------ --dist (section 8 f3)
namespace msnv {
int dist_file(msnv_ctx *ctx, const char *freq_path, const char *mann_path, const char *allele_path, double threshold,
              int32_t *n_samples_out, uint64_t *n_pos_out, double *ms_kernel);
}
namespace msnv { bool pandas_strtod(const char *s, const char *end, double &out); }
extern "C" int msnv_parse_float(const char *text, double *value) {
    clear_error();
    if (!text || !value) return fail(MSNV_EINVAL, "msnv_parse_float: NULL argument");
    if (!pandas_strtod(text, text + strlen(text), *value)) return fail_quiet(MSNV_EFORMAT, "'%s' is not a plain number", text);
    return MSNV_OK;
}

extern "C" int msnv_dist_file(msnv_ctx *ctx, const char *freq_path, const char *mann_path, const char *allele_path, double threshold,
                              int32_t *n_samples, uint64_t *n_positions, double *ms_kernel) {
    clear_error();
    if (!ctx || !freq_path || !mann_path || !allele_path) return fail(MSNV_EINVAL, "msnv_dist_file: NULL argument");
    if (int rc = dev_set_device(ctx->device)) return rc;
    if (ms_kernel) *ms_kernel = 0;
    return dist_file(ctx, freq_path, mann_path, allele_path, threshold, n_samples, n_positions, ms_kernel);
}

// ------------------------------------------------------------------------------ one-call forms
extern "C" int msnv_call(msnv_ctx *ctx, const msnv_call_args *a) {
    clear_error();
    if (!ctx || !a || !a->bam_paths || a->n_bams <= 0 || !a->out_called_path)
        return fail(MSNV_EINVAL, "msnv_call: bam_paths / out_called_path are required");
    msnv_dataset *ds = nullptr;
    int rc = msnv_dataset_create_from_files(ctx, a->bam_paths[0], a->ref_fasta, &a->params, &ds);
    if (rc) return rc;
    if (!rc && a->bed_split_path) rc = msnv_dataset_set_bed_file(ds, a->bed_split_path);
    if (!rc && a->contig_rank_mask) rc = msnv_dataset_set_contig_mask(ds, a->contig_rank_mask, a->n_contig_rank_mask);
    if (!rc) rc = msnv_dataset_add_sample_bams(ds, a->bam_paths, a->n_bams, a->host_threads);
    if (!rc) rc = msnv_dataset_finalize(ds);
    if (!rc) rc = msnv_pileup_run(ds, nullptr);
    if (!rc) rc = msnv_write_calls(ds, a->out_called_path, a->out_indiv_path, a->ann_path, a->ref_fasta);
    msnv_dataset_destroy(ds);
    return rc;
}

extern "C" int msnv_coverage(msnv_ctx *ctx, const msnv_cov_args *a) {
    clear_error();
    if (!ctx || !a || !a->bam_path || !a->out_cov_path || !a->out_detail_path) return fail(MSNV_EINVAL, "msnv_coverage: NULL argument");
    msnv_params p;
    msnv_params_default(&p);
    p.cov_max = a->max_cov > 0 ? a->max_cov : 10;
    p.cov_min_mapq = a->min_mapq;
    msnv_dataset *ds = nullptr;
    int rc = msnv_dataset_create_from_files(ctx, a->bam_path, nullptr, &p, &ds);
    if (rc) return rc;
    rc = msnv_dataset_add_sample_bam(ds, a->bam_path);
    if (!rc) rc = msnv_dataset_finalize(ds);
    if (!rc) rc = msnv_coverage_run(ds, nullptr);
    if (!rc) rc = msnv_write_coverage(ds, 0, a->out_cov_path, a->out_detail_path);
    msnv_dataset_destroy(ds);
    return rc;
}
