// metasnv_amd/csrc/dist.cpp -- host side of metaSNV_DistDiv.py --dist (metaSNV_DistDiv.py:105-124): read one
// `<species>.filtered.freq` table the way `pd.read_table(f, index_col=0, na_values=['-1']).T` does, run the pair
// kernel (dist_k.hip), write `<species>.mann.dist` / `<species>.allele.dist` the way `DataFrame.to_csv(sep='\t')` does
// (header = tab + sample names, floats as repr(), NaN as the empty string).
#include <cmath>
#include <cstring>

#include "dataset.h"

namespace msnv {

void py_repr(double x, std::string &out);
int dev_dist(const double *xt_host, int n_samples, long n_pos, double threshold, void *stream, double *mann, double *allele, double *ms_kernel);

// pandas' default float converter (read_table(float_precision=None) -> precise_xstrtod, pandas/_libs/src/parser/
// tokenizer.c; pandas is a dependency of the reference, version unpinned -- restated from pandas 2.x and pinned by
// tests/test_tables.py against the installed pandas): at most 17 digit characters are accumulated (a leading "0"
// before the decimal point counts), the rest is dropped, and the result is scaled by ONE multiplication or division
// with a power of ten.  It is not correctly rounded -- about every tenth 17-digit repr() lands one ulp off -- and the
// reference's distances carry those errors, so they are reproduced here instead of calling strtod.
static const double k_pow10[] = {
    1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23, 1e24,
    1e25, 1e26, 1e27, 1e28, 1e29, 1e30, 1e31, 1e32, 1e33, 1e34, 1e35, 1e36, 1e37, 1e38, 1e39, 1e40, 1e41, 1e42, 1e43, 1e44, 1e45, 1e46, 1e47, 1e48,
    1e49, 1e50, 1e51, 1e52, 1e53, 1e54, 1e55, 1e56, 1e57, 1e58, 1e59, 1e60, 1e61, 1e62, 1e63, 1e64};
bool pandas_strtod(const char *s, const char *end, double &out) {
    const char *p = s;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    double number = 0.0;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    const int max_digits = 17;
    while (p < end && *p >= '0' && *p <= '9') {
        if (num_digits < max_digits) { number = number * 10.0 + (double)(*p - '0'); ++num_digits; } else ++exponent;
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (num_digits < max_digits && p < end && *p >= '0' && *p <= '9') { number = number * 10.0 + (double)(*p - '0'); ++p; ++num_digits; ++num_decimals; }
        if (num_digits >= max_digits) while (p < end && *p >= '0' && *p <= '9') ++p;
        exponent -= num_decimals;
    }
    if (num_digits == 0) return false;
    if (neg) number = -number;
    if (p < end && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < end && (*p == '-' || *p == '+')) { eneg = *p == '-'; ++p; }
        if (!(p < end && *p >= '0' && *p <= '9')) return false;
        int n = 0;
        while (p < end && *p >= '0' && *p <= '9') { if (n < 100000) n = n * 10 + (*p - '0'); ++p; }
        exponent += eneg ? -n : n;
    }
    if (p != end) return false;
    if (exponent > 64 || exponent < -64) return false;       // far outside anything a frequency table holds
    if (exponent > 0) number *= k_pow10[exponent]; else number /= k_pow10[-exponent];
    out = number;
    return true;
}

static bool is_na_token(const char *s, size_t n) {            // '-1' (na_values) + pandas' default NA strings
    static const char *na[] = {"-1", "", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                               "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
    for (const char *t : na) if (strlen(t) == n && memcmp(t, s, n) == 0) return true;
    return false;
}

static int write_matrix(const char *path, const std::vector<std::string> &names, const std::vector<double> &m) {
    FILE *f = fopen(path, "w");
    if (!f) return fail(MSNV_EIO, "Cannot open %s", path);
    std::string line;
    for (const std::string &n : names) { line.push_back('\t'); line += n; }
    line.push_back('\n');
    fwrite(line.data(), 1, line.size(), f);
    const size_t S = names.size();
    for (size_t i = 0; i < S; ++i) {
        line.assign(names[i]);
        for (size_t j = 0; j < S; ++j) {
            line.push_back('\t');
            const double v = m[i * S + j];
            if (v == v) py_repr(v, line);                     // NaN -> '' (na_rep)
        }
        line.push_back('\n');
        fwrite(line.data(), 1, line.size(), f);
    }
    fclose(f);
    return MSNV_OK;
}

int dist_file(msnv_ctx *ctx, const char *freq_path, const char *mann_path, const char *allele_path, double threshold,
              int32_t *n_samples_out, uint64_t *n_pos_out, double *ms_kernel) {
    FILE *in = fopen(freq_path, "r");
    if (!in) return fail(MSNV_EIO, "Cannot open %s", freq_path);
    char *line = nullptr; size_t cap = 0; ssize_t len;
    std::vector<std::string> names;
    std::vector<double> rows;                                   // [pos][sample]
    uint64_t lineno = 0, n_pos = 0;
    while ((len = getline(&line, &cap, in)) >= 0) {
        ++lineno;
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (lineno == 1) {                                      // header: empty index label, then the sample names
            const char *s = line, *end = line + len;
            bool first = true;
            while (s <= end) {
                const char *t = (const char *)memchr(s, '\t', (size_t)(end - s));
                if (!t) t = end;
                if (!first) names.emplace_back(s, t);
                first = false;
                if (t == end) break;
                s = t + 1;
            }
            continue;
        }
        if (len == 0) continue;                                 // pandas skips blank lines
        const char *s = line, *end = line + len;
        const char *t = (const char *)memchr(s, '\t', (size_t)(end - s));
        size_t col = 0;
        if (t) {
            s = t + 1;
            while (true) {
                t = (const char *)memchr(s, '\t', (size_t)(end - s));
                const char *e = t ? t : end;
                double v;
                if (is_na_token(s, (size_t)(e - s))) v = std::nan("");
                else {
                    if (!pandas_strtod(s, e, v)) { free(line); fclose(in); return fail(MSNV_EFORMAT, "%s:%llu: '%s' is not a number", freq_path, (unsigned long long)lineno, std::string(s, e).c_str()); }
                }
                if (col < names.size()) rows.push_back(v);
                ++col;
                if (!t) break;
                s = t + 1;
            }
        }
        if (col != names.size()) { free(line); fclose(in); return fail(MSNV_EFORMAT, "%s:%llu: %zu values for %zu samples", freq_path, (unsigned long long)lineno, col, names.size()); }
        ++n_pos;
    }
    free(line);
    fclose(in);
    const size_t S = names.size();
    std::vector<double> xt(S * n_pos);                          // sample-major for the kernel
    for (uint64_t p = 0; p < n_pos; ++p) for (size_t s = 0; s < S; ++s) xt[s * n_pos + p] = rows[p * S + s];
    std::vector<double> mann(S * S, std::nan("")), allele(S * S, std::nan(""));
    if (S) if (int rc = dev_dist(xt.data(), (int)S, (long)n_pos, threshold, ctx->stream, mann.data(), allele.data(), ms_kernel)) return rc;
    if (n_samples_out) *n_samples_out = (int32_t)S;
    if (n_pos_out) *n_pos_out = n_pos;
    if (int rc = write_matrix(mann_path, names, mann)) return rc;
    return write_matrix(allele_path, names, allele);
}

}  // namespace msnv
