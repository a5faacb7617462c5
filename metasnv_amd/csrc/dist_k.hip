// metasnv_amd/csrc/dist_k.hip -- metaSNV_DistDiv.py --dist on the device.
//
//   computeDist   metaSNV_DistDiv.py:105-124: for every pair of samples of one species' *.filtered.freq table
//     mann   = np.abs(d1 - d2).mean()            pandas: NaN (= "-1" in the file) skipped, sum / count of the rest
//     allele = (np.abs(d1 - d2) > 0.6).mean()    NaN compares False and still counts in the denominator
//
// Bit-exact floating point: pandas' nanmean is `values.sum() / count` on the float64 array with NaN replaced by 0,
// and numpy's sum of a contiguous float64 array is its PAIRWISE summation (blocks of <= 128 elements accumulated in 8
// interleaved partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail; longer arrays split at
// n/2 rounded down to a multiple of 8, recursively).  One thread per sample pair replays exactly that order, so the
// printed repr() of every distance equals the reference's.  No multiplications: nothing for the compiler to contract.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "msnv_internal.h"

namespace msnv {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(MSNV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

__device__ __forceinline__ double absdiff0(const double *__restrict__ x, const double *__restrict__ y, long k) {
    const double a = x[k], b = y[k];
    return (a != a || b != b) ? 0.0 : fabs(a - b);            // NaN -> 0 (pandas nanops: fill_value 0)
}

__device__ double pairwise_sum(const double *__restrict__ x, const double *__restrict__ y, long lo, long n) {
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += absdiff0(x, y, lo + i);
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = absdiff0(x, y, lo + j);
        long i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += absdiff0(x, y, lo + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += absdiff0(x, y, lo + i);
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(x, y, lo, n2) + pairwise_sum(x, y, lo + n2, n - n2);
}

// xt: [n_samples][n_pos] (sample-major), NaN = not informative
__global__ void msnv_dist_pairs(const double *__restrict__ xt, int n_samples, long n_pos, double threshold,
                                double *__restrict__ mann, double *__restrict__ allele) {
    const long pair = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n_pairs = (long)n_samples * (n_samples + 1) / 2;
    if (pair >= n_pairs) return;
    // unrank (i <= j) from the row-major upper triangle
    long i = 0, rem = pair;
    while (rem >= n_samples - i) { rem -= n_samples - i; ++i; }
    const long j = i + rem;
    const double *x = xt + i * n_pos, *y = xt + j * n_pos;
    const double sum = 0.0 + pairwise_sum(x, y, 0, n_pos);
    long count = 0, above = 0;
    for (long k = 0; k < n_pos; ++k) {
        const double a = x[k], b = y[k];
        const bool ok = !(a != a || b != b);
        count += ok ? 1 : 0;
        above += (ok && fabs(a - b) > threshold) ? 1 : 0;
    }
    const double nanv = nan("");
    const double m = count > 0 ? sum / (double)count : nanv;
    const double al = n_pos > 0 ? (double)above / (double)n_pos : nanv;
    mann[i * n_samples + j] = m; mann[j * n_samples + i] = m;
    allele[i * n_samples + j] = al; allele[j * n_samples + i] = al;
}

int dev_dist(const double *xt_host, int n_samples, long n_pos, double threshold, void *stream_, double *mann, double *allele, double *ms_kernel) {
    hipStream_t st = (hipStream_t)stream_;
    struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } };
    Buf d_x, d_m, d_a;
    const size_t xb = (size_t)n_samples * (size_t)std::max<long>(n_pos, 1) * sizeof(double), mb = (size_t)n_samples * n_samples * sizeof(double);
    HIP_TRY(hipMalloc(&d_x.p, xb));
    HIP_TRY(hipMalloc(&d_m.p, std::max<size_t>(mb, 16)));
    HIP_TRY(hipMalloc(&d_a.p, std::max<size_t>(mb, 16)));
    if (n_pos > 0) HIP_TRY(hipMemcpyAsync(d_x.p, xt_host, (size_t)n_samples * n_pos * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipDeviceSetLimit(hipLimitStackSize, 8192));     // pairwise_sum recurses log2(n / 128) deep
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const long n_pairs = (long)n_samples * (n_samples + 1) / 2;
    hipError_t he = hipEventRecord(e0, st);
    if (he == hipSuccess && n_pairs) {
        hipLaunchKernelGGL(msnv_dist_pairs, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, st, (const double *)d_x.p, n_samples, n_pos, threshold,
                           (double *)d_m.p, (double *)d_a.p);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess && mb) he = hipMemcpyAsync(mann, d_m.p, mb, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess && mb) he = hipMemcpyAsync(allele, d_a.p, mb, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    float t = 0;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(MSNV_EHIP, "distance kernel: %s", hipGetErrorString(he));
    if (ms_kernel) *ms_kernel = t;
    return MSNV_OK;
}

}  // namespace msnv
