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
#include <vector>

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

// numpy's pairwise summation as data: the leaves (blocks of <= 128 elements, in array order) and the order in which their sums
// are combined (a postfix program: 0 = take the next leaf, 1 = add the two on top) -- the same for every pair of samples,
// built once on the host from n_pos.  ONE WAVEFRONT per pair of samples: eight lanes share a leaf, lane j accumulating the
// elements j, 8 + j, 16 + j ... exactly like numpy's r[j], so every load instruction of the wavefront reads whole 64-byte lines
// (the first form had one thread per pair walking two rows: lane stride n_pos x 8 bytes, nothing coalesced, device recursion).
struct DistLeaf { long lo; int n; int pad; };
constexpr int DIST_MAX_LEAVES_LDS = 2048;                     // leaf sums kept in LDS (n_pos <= 262144); beyond that in global scratch

__global__ __launch_bounds__(64) void msnv_dist_pairs(const double *__restrict__ xt, int n_samples, long n_pos, double threshold,
                                                      const DistLeaf *__restrict__ leaves, int n_leaves, const unsigned char *__restrict__ prog, int n_prog,
                                                      double *__restrict__ scratch, double *__restrict__ mann, double *__restrict__ allele) {
    __shared__ double s_leaf[DIST_MAX_LEAVES_LDS];
    __shared__ double s_stack[64];
    const long pair = blockIdx.x;
    // unrank (i <= j) from the row-major upper triangle
    long i = 0, rem = pair;
    while (rem >= n_samples - i) { rem -= n_samples - i; ++i; }
    const long j = i + rem;
    const double *x = xt + i * n_pos, *y = xt + j * n_pos;
    const int lane = threadIdx.x, sub = lane & 7, grp = lane >> 3;
    double *leaf_sum = n_leaves <= DIST_MAX_LEAVES_LDS ? s_leaf : scratch + (long)pair * n_leaves;
    long count = 0, above = 0;
    for (int l = grp; l < n_leaves; l += 8) {
        const DistLeaf lf = leaves[l];
        double res;
        if (lf.n < 8) {                                          // only a table shorter than 8 positions: plain left-to-right sum
            res = 0.0;
            for (int k = 0; k < lf.n; ++k) res += absdiff0(x, y, lf.lo + k);
        } else {
            const int n8 = lf.n - (lf.n % 8);
            double r = absdiff0(x, y, lf.lo + sub);              // r[sub]
            for (int k = 8; k < n8; k += 8) r += absdiff0(x, y, lf.lo + k + sub);
            // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)): the partner's value is added in the same order in every lane
            const double a1 = r + __shfl_xor(r, 1);              // lanes 0,1 hold r0 + r1 (lane 1 computes r1 + r0: the same double)
            const double a2 = a1 + __shfl_xor(a1, 2);
            res = a2 + __shfl_xor(a2, 4);
            for (int k = n8; k < lf.n; ++k) res += absdiff0(x, y, lf.lo + k);
        }
        if (sub == 0) leaf_sum[l] = res;
        // counts (order does not matter): every lane looks at its own elements
        for (int k = sub; k < lf.n; k += 8) {
            const double a = x[lf.lo + k], b = y[lf.lo + k];
            const bool ok = !(a != a || b != b);
            count += ok ? 1 : 0;
            above += (ok && fabs(a - b) > threshold) ? 1 : 0;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) { count += __shfl_xor(count, o); above += __shfl_xor(above, o); }
    __syncthreads();
    if (lane == 0) {
        int sp = 0, next = 0;
        for (int k = 0; k < n_prog; ++k) {
            if (prog[k] == 0) s_stack[sp++] = leaf_sum[next++];
            else { --sp; s_stack[sp - 1] = s_stack[sp - 1] + s_stack[sp]; }
        }
        const double sum = 0.0 + (n_leaves ? s_stack[0] : 0.0);
        const double nanv = nan("");
        const double m = count > 0 ? sum / (double)count : nanv;
        const double al = n_pos > 0 ? (double)above / (double)n_pos : nanv;
        mann[i * n_samples + j] = m; mann[j * n_samples + i] = m;
        allele[i * n_samples + j] = al; allele[j * n_samples + i] = al;
    }
}

// the leaves and the combination order of numpy's pairwise sum over n elements (numpy/core/src/umath/loops_utils.h pairwise_sum)
static void pairwise_plan(long lo, long n, std::vector<DistLeaf> &leaves, std::vector<unsigned char> &prog) {
    if (n <= 128) { leaves.push_back(DistLeaf{lo, (int)n, 0}); prog.push_back(0); return; }
    long n2 = n / 2;
    n2 -= n2 % 8;
    pairwise_plan(lo, n2, leaves, prog);
    pairwise_plan(lo + n2, n - n2, leaves, prog);
    prog.push_back(1);
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
    std::vector<DistLeaf> leaves; std::vector<unsigned char> prog;
    if (n_pos > 0) pairwise_plan(0, n_pos, leaves, prog);
    const long n_pairs = (long)n_samples * (n_samples + 1) / 2;
    Buf d_l, d_p, d_s;
    HIP_TRY(hipMalloc(&d_l.p, std::max<size_t>(leaves.size() * sizeof(DistLeaf), 16)));
    HIP_TRY(hipMalloc(&d_p.p, std::max<size_t>(prog.size(), 16)));
    if (!leaves.empty()) HIP_TRY(hipMemcpyAsync(d_l.p, leaves.data(), leaves.size() * sizeof(DistLeaf), hipMemcpyHostToDevice, st));
    if (!prog.empty()) HIP_TRY(hipMemcpyAsync(d_p.p, prog.data(), prog.size(), hipMemcpyHostToDevice, st));
    if ((int)leaves.size() > DIST_MAX_LEAVES_LDS) HIP_TRY(hipMalloc(&d_s.p, (size_t)n_pairs * leaves.size() * sizeof(double)));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipError_t he = hipEventRecord(e0, st);
    if (he == hipSuccess && n_pairs) {
        hipLaunchKernelGGL(msnv_dist_pairs, dim3((unsigned)n_pairs), dim3(64), 0, st, (const double *)d_x.p, n_samples, n_pos, threshold,
                           (const DistLeaf *)d_l.p, (int)leaves.size(), (const unsigned char *)d_p.p, (int)prog.size(), (double *)d_s.p,
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
