// metasnv_amd/csrc/subpopr.cpp -- SURVEY.md section 8 row f4: the raw-SNV consumers of subpopr.
//
//   msnv_genotyping_subset   src/subpopr/inst/getGenotypingSNVSubset.py:20-48
//       positions of every species' *_hap_positions.tab (field 2 = contig:gene:pos:base -> code contig:pos, :29-31), then
//       one scan of the called_SNPs* files that copies every line whose contig:pos (fields 1 and 3, :43-44) is wanted
//       into <species>.pos of every species that listed it (:46-47).  Text only: no device work.
//   msnv_snv_allele_freq     src/subpopr/inst/convertSNVtoAlleleFreq.py:7-24
//       per allele of every line of a .pos file: id contig:gene:pos:base (:9,15-16) and per sample -1 below the depth
//       cutoff or count / coverage * 100 (:19-22), printed with str() -- the arithmetic runs on the device (subpopr_k.hip).
// The callers (metasnv_amd/subpopr.py) do the globbing and the reference's messages; the order of the path lists is the
// order the files are read in, as in the reference (directory order of glob.glob).
#include <cerrno>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "dataset.h"
#include "device.h"

namespace msnv {

void py_repr(double x, std::string &out);
int dev_allele_freq(const std::vector<uint32_t> &cov, const std::vector<uint32_t> &cnt, const std::vector<uint32_t> &row_line,
                    uint32_t n_samples, long long min_depth, void *stream, std::vector<double> &freq, double *ms_kernel);

namespace {

int read_file(const char *path, std::string &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MSNV_EIO, "Cannot open %s: %s", path, strerror(errno));
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return MSNV_OK;
}

// str.rstrip(): Python's whitespace set for ASCII text
inline bool py_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f' || c == '\x1c' || c == '\x1d' || c == '\x1e' || c == '\x1f'; }

// str.split(sep) into string views (every field, empty ones included)
void split_char(const char *s, const char *e, char sep, std::vector<std::pair<const char *, const char *>> &out) {
    out.clear();
    const char *b = s;
    for (const char *p = s; p <= e; ++p)
        if (p == e || *p == sep) { out.emplace_back(b, p); b = p + 1; }
}

// int(text) for the digit strings these files hold (optional sign, surrounding blanks); false when Python would raise
bool py_int(const char *s, const char *e, long long &v) {
    while (s < e && py_space(*s)) ++s;
    while (e > s && py_space(e[-1])) --e;
    bool neg = false;
    if (s < e && (*s == '+' || *s == '-')) { neg = *s == '-'; ++s; }
    if (s == e) return false;
    unsigned long long x = 0;
    for (const char *p = s; p < e; ++p) {
        if (*p < '0' || *p > '9' || x > 400000000ull) return false;
        x = x * 10 + (unsigned)(*p - '0');
    }
    v = neg ? -(long long)x : (long long)x;
    return true;
}

}  // namespace
}  // namespace msnv

using namespace msnv;

extern "C" int msnv_genotyping_subset(const char *const *hap_paths, int32_t n_hap, const char *const *snp_paths, int32_t n_snp,
                                      const char *out_dir, uint64_t *n_positions, uint64_t *n_lines_written) {
    clear_error();
    if (!hap_paths || !snp_paths || !out_dir || n_hap <= 0 || n_snp <= 0) return fail(MSNV_EINVAL, "msnv_genotyping_subset: hap_paths / snp_paths / out_dir are required");
    static const char suffix[] = "_hap_positions.tab";
    std::vector<FILE *> files;                                        // one <species>.pos per species, opened at its first table
    std::unordered_map<std::string, int> file_of;                     // species + ".pos" -> index
    std::unordered_map<std::string, std::vector<int>> wanted;         // "contig:pos" -> the files that listed it, in listing order
    auto close_all = [&]() { for (FILE *f : files) if (f) fclose(f); };
    std::vector<std::pair<const char *, const char *>> f, c;
    for (int i = 0; i < n_hap; ++i) {
        std::string base = hap_paths[i];
        const size_t slash = base.rfind('/');
        if (slash != std::string::npos) base.erase(0, slash + 1);
        for (size_t p; (p = base.find(suffix)) != std::string::npos;) base.erase(p, sizeof suffix - 1);   // str.replace: every occurrence (:21)
        const std::string key = base + ".pos";
        int fi;
        auto it = file_of.find(key);
        if (it == file_of.end()) {
            const std::string path = std::string(out_dir) + "/" + key;
            FILE *o = fopen(path.c_str(), "w");
            if (!o) { close_all(); return fail(MSNV_EIO, "Cannot open %s: %s", path.c_str(), strerror(errno)); }
            fi = (int)files.size(); files.push_back(o); file_of.emplace(key, fi);
        } else fi = it->second;
        std::string text;
        if (int rc = read_file(hap_paths[i], text)) { close_all(); return rc; }
        const char *p = text.data(), *end = p + text.size();
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        p = nl ? nl + 1 : end;                                       // the header line (:26)
        while (p < end) {
            nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end, *next = nl ? nl + 1 : end;
            while (le > p && py_space(le[-1])) --le;                  // line.rstrip()
            split_char(p, le, '\t', f);
            if (f.size() < 2) { close_all(); return fail(MSNV_EFORMAT, "%s: a line has no second field (the reference stops with an IndexError)", hap_paths[i]); }
            split_char(f[1].first, f[1].second, ':', c);
            if (c.size() < 3) { close_all(); return fail(MSNV_EFORMAT, "%s: position id without contig:gene:pos (the reference stops with an IndexError)", hap_paths[i]); }
            std::string code(c[0].first, c[0].second);
            code.push_back(':'); code.append(c[2].first, c[2].second);
            std::vector<int> &lst = wanted[code];
            bool have = false;
            for (int x : lst) have |= x == fi;
            if (!have) lst.push_back(fi);
            p = next;
        }
    }
    if (n_positions) *n_positions = wanted.size();
    uint64_t written = 0;
    if (wanted.empty()) { close_all(); return fail(MSNV_EDOMAIN, "no parse-able data in the *hap_positions.tab files"); }
    std::string code;
    for (int i = 0; i < n_snp; ++i) {
        std::string text;
        if (int rc = read_file(snp_paths[i], text)) { close_all(); return rc; }
        const char *p = text.data(), *end = p + text.size();
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *next = nl ? nl + 1 : end;                    // the line as Python iterates it: with its newline
            // line.split('\t'): fields 0 and 2 (:43-44); the newline stays in the last field
            const char *t1 = (const char *)memchr(p, '\t', (size_t)(next - p));
            const char *t2 = t1 ? (const char *)memchr(t1 + 1, '\t', (size_t)(next - t1 - 1)) : nullptr;
            if (!t2) { close_all(); return fail(MSNV_EFORMAT, "%s: a line has fewer than three tab-separated fields (the reference stops with an IndexError)", snp_paths[i]); }
            const char *t3 = (const char *)memchr(t2 + 1, '\t', (size_t)(next - t2 - 1));
            code.assign(p, t1); code.push_back(':'); code.append(t2 + 1, t3 ? t3 : next);
            auto it = wanted.find(code);
            if (it != wanted.end())
                for (int fi : it->second) { fwrite(p, 1, (size_t)(next - p), files[(size_t)fi]); ++written; }
            p = next;
        }
    }
    int rc = MSNV_OK;
    for (FILE *o : files) if (fclose(o) != 0) rc = fail(MSNV_EIO, "write error on a .pos file: %s", strerror(errno));
    if (n_lines_written) *n_lines_written = written;
    return rc;
}

extern "C" int msnv_snv_allele_freq(msnv_ctx *ctx, const char *pos_path, int32_t min_depth, uint64_t *n_rows, double *ms_kernel) {
    clear_error();
    if (!ctx || !pos_path) return fail(MSNV_EINVAL, "msnv_snv_allele_freq: NULL argument");
    if (int rc = dev_set_device(ctx->device)) return rc;
    if (ms_kernel) *ms_kernel = 0;
    std::string text;
    if (int rc = read_file(pos_path, text)) return rc;
    const std::string out_path = std::string(pos_path) + ".freq";
    FILE *out = fopen(out_path.c_str(), "w");                         // created before the first line is read (:4)
    if (!out) return fail(MSNV_EIO, "Cannot open %s: %s", out_path.c_str(), strerror(errno));
    std::vector<uint32_t> cov, cnt, row_line;
    std::vector<std::string> ids;
    uint32_t S = 0, n_lines = 0;
    std::vector<std::pair<const char *, const char *>> c, cv, al, s;
    const char *p = text.data(), *end = p + text.size();
    int rc = MSNV_OK;
    while (p < end && !rc) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end, *next = nl ? nl + 1 : end;
        while (le > p && py_space(le[-1])) --le;                      // line.rstrip()
        split_char(p, le, '\t', c);
        if (c.size() < 6) { rc = fail(MSNV_EFORMAT, "%s: line %u has fewer than six fields (the reference stops with an IndexError)", pos_path, n_lines + 1); break; }
        split_char(c[4].first, c[4].second, '|', cv);
        if (n_lines == 0) S = (uint32_t)cv.size();
        // one kernel launch over a rectangular table: every line must carry the same number of samples (one run's files do)
        if (cv.size() != S) { rc = fail(MSNV_EFORMAT, "%s: line %u lists %zu coverages, the first line %u", pos_path, n_lines + 1, cv.size(), S); break; }
        for (auto &x : cv) {
            long long v;
            if (!py_int(x.first, x.second, v) || v < 0) { rc = fail(MSNV_EFORMAT, "%s: line %u: coverage is not a non-negative integer", pos_path, n_lines + 1); break; }
            // int(cov) < minDepth is false for a zero coverage only when minDepth <= 0: the reference then divides by zero (:22)
            if (v == 0 && min_depth <= 0) { rc = fail(MSNV_EDOMAIN, "%s: line %u: coverage 0 with minDepth %d (the reference stops with a ZeroDivisionError)", pos_path, n_lines + 1, min_depth); break; }
            cov.push_back((uint32_t)v);
        }
        if (rc) break;
        std::string id(c[0].first, c[0].second);
        id.push_back(':'); id.append(c[1].first, c[1].second); id.push_back(':'); id.append(c[2].first, c[2].second);
        split_char(c[5].first, c[5].second, ',', al);
        for (auto &a : al) {
            split_char(a.first, a.second, '|', s);
            if (s.size() < 2) { rc = fail(MSNV_EFORMAT, "%s: line %u: allele entry without a base (the reference stops with an IndexError)", pos_path, n_lines + 1); break; }
            if (s.size() != (size_t)S + 3 && !(s.size() < 3 && S == 0)) {
                if (s.size() > (size_t)S + 3) rc = fail(MSNV_EFORMAT, "%s: line %u: more allele counts than coverages (the reference stops with an IndexError)", pos_path, n_lines + 1);
                else rc = fail(MSNV_EFORMAT, "%s: line %u: %zu allele counts for %u coverages (rows of one file must have one width here)", pos_path, n_lines + 1, s.size() - 3, S);
                break;
            }
            for (size_t i = 3; i < s.size(); ++i) {
                long long v;                                          // float(s[i]): the counts are digit strings
                if (!py_int(s[i].first, s[i].second, v) || v < 0) { rc = fail(MSNV_EFORMAT, "%s: line %u: allele count is not a non-negative integer", pos_path, n_lines + 1); break; }
                cnt.push_back((uint32_t)v);
            }
            if (rc) break;
            ids.push_back(id + ":" + std::string(s[1].first, s[1].second));
            row_line.push_back(n_lines);
        }
        ++n_lines;
        p = next;
    }
    std::vector<double> freq;
    if (!rc) rc = dev_allele_freq(cov, cnt, row_line, S, (long long)min_depth, ctx->stream, freq, ms_kernel);
    if (!rc) {
        std::string line;
        for (size_t r = 0; r < ids.size(); ++r) {
            line = ids[r];
            for (uint32_t k = 0; k < S; ++k) {
                const double v = freq[r * S + k];
                line.push_back('\t');
                if (v != v) line += "-1"; else py_repr(v, line);      // str(-1) / str(float)
            }
            line.push_back('\n');
            fwrite(line.data(), 1, line.size(), out);
        }
    }
    if (fclose(out) != 0 && !rc) rc = fail(MSNV_EIO, "write error on %s: %s", out_path.c_str(), strerror(errno));
    if (n_rows) *n_rows = ids.size();
    return rc;
}
