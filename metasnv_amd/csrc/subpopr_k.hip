// metasnv_amd/csrc/subpopr_k.hip -- SURVEY.md section 8 row f4: allele frequencies of the genotyping positions.
//
//   msnv_allele_freq_pct   src/subpopr/inst/convertSNVtoAlleleFreq.py:19-22
//       freq = -1                               if int(cov) < minDepth
//       freq = float(count) / int(cov) * 100    otherwise (two IEEE operations, in this order)
// One thread per (allele row, sample).  The quotient and the product are rounded separately (__ddiv_rn, __dmul_rn: no
// contraction), so every double is the one CPython computes and repr() of it prints the same digits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "device.h"
#include "msnv_internal.h"

namespace msnv {

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MSNV_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

// out[row * S + s]: the frequency, or NaN for "below the depth cutoff" (the host prints -1 there)
__global__ void msnv_allele_freq_pct(const uint32_t *cov, const uint32_t *cnt, const uint32_t *row_line, uint32_t n_samples,
                                     uint64_t n_cells, long long min_depth, double *out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / n_samples), s = (uint32_t)(i % n_samples);
        const uint32_t c = cov[(uint64_t)row_line[row] * n_samples + s];
        double f = __longlong_as_double(0x7ff8000000000000ll);
        if ((long long)c >= min_depth) f = __dmul_rn(__ddiv_rn((double)cnt[i], (double)c), 100.0);
        out[i] = f;
    }
}

int dev_allele_freq(const std::vector<uint32_t> &cov, const std::vector<uint32_t> &cnt, const std::vector<uint32_t> &row_line,
                    uint32_t n_samples, long long min_depth, void *stream_, std::vector<double> &freq, double *ms_kernel) {
    hipStream_t st = (hipStream_t)stream_;
    const uint64_t n_cells = cnt.size();
    freq.assign(n_cells, 0.0);
    if (!n_cells) return MSNV_OK;
    struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } };
    Buf d_cov, d_cnt, d_rl, d_fr;
    auto up = [&](Buf &buf, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(&buf.p, bytes ? bytes : 16));
        if (bytes) HIP_TRY(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
        return MSNV_OK;
    };
    if (int rc = up(d_cov, cov.data(), cov.size() * 4)) return rc;
    if (int rc = up(d_cnt, cnt.data(), cnt.size() * 4)) return rc;
    if (int rc = up(d_rl, row_line.data(), row_line.size() * 4)) return rc;
    HIP_TRY(hipMalloc(&d_fr.p, n_cells * sizeof(double)));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipError_t he = hipEventRecord(e0, st);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_cells + 255) / 256, 65536);
    hipLaunchKernelGGL(msnv_allele_freq_pct, dim3(blocks), dim3(256), 0, st, (const uint32_t *)d_cov.p, (const uint32_t *)d_cnt.p,
                       (const uint32_t *)d_rl.p, n_samples, n_cells, min_depth, (double *)d_fr.p);
    if (he == hipSuccess) he = hipGetLastError();
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess) he = hipMemcpyAsync(freq.data(), d_fr.p, n_cells * sizeof(double), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    float t = 0;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(MSNV_EHIP, "allele frequency kernel: %s", hipGetErrorString(he));
    if (ms_kernel) *ms_kernel += t;
    return MSNV_OK;
}

}  // namespace msnv
