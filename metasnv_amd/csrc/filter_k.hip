// metasnv_amd/csrc/filter_k.hip -- metaSNV_Filtering.py filter_two on the device.
//
//   position filter   metaSNV_Filtering.py:183-195   nr_good = #{SoI samples with coverage >= c and != 0};
//                                                     keep iff float(nr_good) / len(SoI) >= p
//   allele frequency  metaSNV_Filtering.py:222-231   count / coverage (true division of a float by an int) or -1
//
// One wavefront per output row (= one alternative allele of one called position); lanes stride over the species'
// samples of interest.  IEEE fp64 division is correctly rounded on the device as in CPython, so the printed
// repr() of every frequency is identical.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "filter.h"
#include "msnv_internal.h"

namespace msnv {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(MSNV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

__global__ __launch_bounds__(256) void msnv_filter_freq(const uint32_t *__restrict__ cov, const uint32_t *__restrict__ cnt, uint32_t n_samples,
                                                        const uint32_t *__restrict__ row_line, const uint32_t *__restrict__ line_species,
                                                        const uint32_t *__restrict__ soi_off, const uint32_t *__restrict__ soi_idx,
                                                        const unsigned long long *__restrict__ row_out, uint32_t n_rows,
                                                        double min_cov, double min_prop, double *__restrict__ freq, uint8_t *__restrict__ line_pass) {
    const uint32_t row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const uint32_t line = row_line[row];
    const uint32_t sp = line_species[line];
    const uint32_t s0 = soi_off[sp], n_soi = soi_off[sp + 1] - s0;
    const uint32_t *c = cov + (uint64_t)line * n_samples;
    const uint32_t *a = cnt + (uint64_t)row * n_samples;
    double *out = freq + row_out[row];
    uint32_t good = 0;
    for (uint32_t i = lane; i < n_soi; i += 64) {
        const uint32_t idx = soi_idx[s0 + i];
        const uint32_t cv = c[idx];
        const bool ok = !((double)cv < min_cov || cv == 0u);                 // :187 / :224
        good += ok ? 1u : 0u;
        out[i] = ok ? (double)a[idx] / (double)cv : -1.0;                     // :225-227 (-1 is printed as an int)
    }
    for (int off = 32; off; off >>= 1) good += __shfl_xor(good, off);
    if (lane == 0) line_pass[line] = ((double)good / (double)n_soi < min_prop) ? 0 : 1;   // :193
}

int dev_filter_batch(const FilterBatch &b, const FilterSpecies &sp, double min_cov, double min_prop, void *stream_,
                     std::vector<double> &freq, std::vector<uint8_t> &line_pass, double *ms_kernel) {
    hipStream_t st = (hipStream_t)stream_;
    const uint32_t n_rows = (uint32_t)b.row_line.size(), n_lines = (uint32_t)b.line_species.size();
    freq.assign(b.n_out, 0.0); line_pass.assign(n_lines, 0);
    if (!n_rows) return MSNV_OK;
    struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } };
    Buf d_cov, d_cnt, d_rl, d_ls, d_so, d_si, d_ro, d_fr, d_lp;
    auto up = [&](Buf &buf, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(&buf.p, bytes ? bytes : 16));
        if (bytes) HIP_TRY(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
        return MSNV_OK;
    };
    if (int rc = up(d_cov, b.cov.data(), b.cov.size() * 4)) return rc;
    if (int rc = up(d_cnt, b.cnt.data(), b.cnt.size() * 4)) return rc;
    if (int rc = up(d_rl, b.row_line.data(), b.row_line.size() * 4)) return rc;
    if (int rc = up(d_ls, b.line_species.data(), b.line_species.size() * 4)) return rc;
    if (int rc = up(d_so, sp.soi_off.data(), sp.soi_off.size() * 4)) return rc;
    if (int rc = up(d_si, sp.soi_idx.data(), sp.soi_idx.size() * 4)) return rc;
    if (int rc = up(d_ro, b.row_out.data(), b.row_out.size() * 8)) return rc;
    HIP_TRY(hipMalloc(&d_fr.p, std::max<size_t>(16, b.n_out * sizeof(double))));
    HIP_TRY(hipMalloc(&d_lp.p, std::max<size_t>(16, n_lines)));
    HIP_TRY(hipMemsetAsync(d_lp.p, 0, n_lines, st));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipError_t he = hipEventRecord(e0, st);
    hipLaunchKernelGGL(msnv_filter_freq, dim3((n_rows + 3) / 4), dim3(256), 0, st, (const uint32_t *)d_cov.p, (const uint32_t *)d_cnt.p, b.n_samples,
                       (const uint32_t *)d_rl.p, (const uint32_t *)d_ls.p, (const uint32_t *)d_so.p, (const uint32_t *)d_si.p,
                       (const unsigned long long *)d_ro.p, n_rows, min_cov, min_prop, (double *)d_fr.p, (uint8_t *)d_lp.p);
    if (he == hipSuccess) he = hipGetLastError();
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess && b.n_out) he = hipMemcpyAsync(freq.data(), d_fr.p, b.n_out * sizeof(double), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipMemcpyAsync(line_pass.data(), d_lp.p, n_lines, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    float t = 0;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(MSNV_EHIP, "filter kernel: %s", hipGetErrorString(he));
    if (ms_kernel) *ms_kernel += t;
    return MSNV_OK;
}

}  // namespace msnv
