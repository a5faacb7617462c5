// metasnv_amd/csrc/coverage.cpp -- host side of the genome-coverage path (qaCompute -c N -d -i):
// runs the device pass and writes OUT / OUT.detail exactly as qaCompute.cpp:192-217,226-263,439,
// 623-657 formats them.  The numbers come from the device accumulators; only printf happens here.
#include <cstring>

#include "device.h"

namespace msnv {

int coverage_run(msnv_dataset &ds, msnv_run_stats *stats) {
    DeviceCols &d = *ds.dev;
    msnv_run_stats st{};
    if (int rc = dev_run_coverage(d, ds.params.cov_max, ds.ctx->stream, &st)) return rc;
    const size_t n = (size_t)d.n_cov_rows * (1 + COV_BINS);
    std::vector<unsigned long long> copies((size_t)d.cov_copies * n);
    if (int rc = dev_download(copies.data(), d.cov_acc, copies.size() * sizeof(unsigned long long))) return rc;
    ds.cov_acc.assign(n, 0);                                 // tile t added to copy t % cov_copies (device.h)
    for (uint32_t k = 0; k < d.cov_copies; ++k)
        for (size_t i = 0; i < n; ++i) ds.cov_acc[i] += copies[(size_t)k * n + i];
    ds.have_coverage = true;
    if (stats) stats->ms_coverage = st.ms_coverage;
    return MSNV_OK;
}

// OUT / OUT.detail of one sample from its accumulator rows acc[contig][1 + COV_BINS] and its read statistics.
int coverage_write_rows(const std::vector<std::string> &names, const std::vector<int64_t> &lengths, int max_cov, const msnv_sample_stats &sc,
                        const unsigned long long *acc, const char *cov_path, const char *detail_path, int sample) {
    if (max_cov < 1 || max_cov >= COV_BINS) return fail(MSNV_EINVAL, "coverage histogram cutoff must be in [1, %d]", COV_BINS - 1);
    // a BAM without mapped reads makes qaCompute read target_name[-1] (qaCompute.cpp:596)
    if (!sc.any_mapped) return fail(MSNV_EDOMAIN, "sample %d has no mapped reads (qaCompute: undefined behaviour, README.md:59)", sample);
    FILE *out = fopen(cov_path, "wt");
    if (!out) return fail(MSNV_EIO, "qaCompute: Filed to create output file %s", cov_path);
    FILE *det = fopen(detail_path, "wt");
    if (!det) { fclose(out); return fail(MSNV_EIO, "qaCompute: Unable to create detailed output file %s", detail_path); }
    const size_t NC = names.size();
    std::vector<unsigned long long> global_hist((size_t)max_cov + 1, 0);
    unsigned long long total_len = 0;
    fprintf(out, "Chromosome\tSeq_lem\tAvg_Cov\n");                                  // qaCompute.cpp:439
    for (size_t c = 0; c < NC; ++c) {
        total_len += (unsigned long long)lengths[c];                                    // :425-427
        // contigs without reads print zeros through printSkipped (:226-263); contigs with reads through
        // compute_print_cov (:192-217): the same bytes when the sums are zero.
        const unsigned long long *a = acc + c * (1 + COV_BINS);
        const int L = (int)lengths[c];
        fprintf(det, "%s\t%d\t", names[c].c_str(), L);
        for (int k = 1; k <= max_cov; ++k) {
            unsigned long long cum = 0;
            for (int x = k; x <= max_cov; ++x) cum += a[1 + x];
            fprintf(det, "%d\t", (int)cum);
        }
        fprintf(det, "\n");
        fprintf(out, "%s\t%d\t%3.5f\n", names[c].c_str(), L, L ? (double)a[0] / L : 0.0);
        for (int x = 1; x <= max_cov; ++x) global_hist[(size_t)x] += a[1 + x];
    }
    fprintf(out, "\nCov*X\tPercentage\tNr. of bases\n");                                // :623-640
    for (int i = 1; i <= max_cov; ++i) {
        unsigned long long cum = 0;
        for (int x = i; x <= max_cov; ++x) cum += global_hist[(size_t)x];
        fprintf(out, "%d\t%3.5f\t%lu\n", i, (double)cum / total_len * 100, (unsigned long)cum);
    }
    fprintf(out, "\nOther\n");                                                          // :642-654
    const double pu = 100 * ((double)sc.unmapped / sc.total_reads);
    const double pz = 100 * ((double)sc.zero_quality / sc.total_reads);
    const int32_t pairs = (int32_t)(sc.total_reads / 2);
    const double pp = (double)(100 * (double)sc.proper_pairs / 2) / pairs;
    fprintf(out, "Total number of reads: %u\n", sc.total_reads);
    fprintf(out, "Total number of duplicates found and ignored: %u\n", sc.duplicates);
    fprintf(out, "Percentage of unmapped reads: %3.5f\n", pu);
    fprintf(out, "Percentage of sub-par quality mappings: %3.5f\n", pz);
    fprintf(out, "Number of proper paired reads: %u\n", sc.proper_pairs);
    fprintf(out, "Percentage of proper pairs: %3.5f\n", pp);
    fclose(out); fclose(det);
    return MSNV_OK;
}

int coverage_write(msnv_dataset &ds, int sample, const char *cov_path, const char *detail_path) {
    if (!ds.have_coverage) return fail(MSNV_EINVAL, "no coverage results: call msnv_coverage_run first");
    if (sample < 0 || (size_t)sample >= ds.samples.size()) return fail(MSNV_EINVAL, "sample index %d out of range", sample);
    // the sample's rows expanded to one per contig; contigs without reads and contigs outside this shard carry zeros (multi-GPU: rank 0
    // receives every rank's rows before it writes, parallel.py)
    std::vector<unsigned long long> dense(ds.names.size() * (1 + COV_BINS), 0ull);
    for (uint64_t r = ds.cov_row_start[(size_t)sample]; r < ds.cov_row_start[(size_t)sample + 1]; ++r)
        memcpy(&dense[(size_t)ds.cov_row_contig[(size_t)r] * (1 + COV_BINS)], &ds.cov_acc[(size_t)r * (1 + COV_BINS)], (1 + COV_BINS) * sizeof(unsigned long long));
    return coverage_write_rows(ds.names, ds.lengths, ds.params.cov_max, ds.samples[(size_t)sample].st, dense.data(), cov_path, detail_path, sample);
}

}  // namespace msnv
