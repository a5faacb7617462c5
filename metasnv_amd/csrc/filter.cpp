// metasnv_amd/csrc/filter.cpp -- host side of filter_two (metaSNV_Filtering.py:156-242): parse called_SNPs /
// indiv_called once for all species, batch the lines of species of interest to the device (filter_k.hip), print
// `<species>.filtered.freq` with CPython's repr() of every frequency.
//
// Reference behaviours kept: the species of a line is its contig name up to the first '.' (:172); fields are split
// on whitespace (:184); the output file of a species is created when its first position passes (:199-203) and its
// header is '\t' + the samples of interest (:203); the line id is the first four fields joined by ':' + '>' + ALT +
// ':' + codon tag (:229-231); samples below the coverage cutoff print -1 (an int) (:227).
#include <algorithm>
#include <charconv>
#include <cstring>
#include <map>

#include "dataset.h"
#include "filter.h"

namespace msnv {

// repr(float) of CPython >= 3.1 for a finite double: shortest digits that round-trip (std::to_chars), laid out by
// format_float_short's 'r' rules: exponent form iff decpt <= -4 or decpt > 16, ".0" appended to integers.
void py_repr(double x, std::string &out) {
    if (x != x) { out += "nan"; return; }
    if (x - x != 0) { out += x < 0 ? "-inf" : "inf"; return; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf - 1, x, std::chars_format::scientific);
    *r.ptr = '\0';                                          // to_chars does not terminate
    const char *e = (const char *)memchr(buf, 'e', (size_t)(r.ptr - buf));
    std::string digits;
    bool neg = false;
    for (const char *p = buf; p < e; ++p) { if (*p == '-') neg = true; else if (*p != '.') digits.push_back(*p); }
    int exp10 = atoi(e + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    int decpt = exp10 + 1;
    if (digits == "0") decpt = 1;
    if (neg) out.push_back('-');
    const int nd = (int)digits.size();
    if (decpt <= -4 || decpt > 16) {
        out.push_back(digits[0]);
        if (nd > 1) { out.push_back('.'); out.append(digits, 1, std::string::npos); }
        const int ex = decpt - 1;
        char eb[16];
        snprintf(eb, sizeof eb, "e%c%02d", ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
        out += eb;
    } else if (decpt <= 0) {
        out += "0.";
        out.append((size_t)(-decpt), '0');
        out += digits;
    } else if (decpt >= nd) {
        out += digits;
        out.append((size_t)(decpt - nd), '0');
        out += ".0";
    } else {
        out.append(digits, 0, (size_t)decpt);
        out.push_back('.');
        out.append(digits, (size_t)decpt, std::string::npos);
    }
}

namespace {

struct Fields { const char *b[8]; const char *e[8]; int n; };
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
void split_ws(const char *s, const char *end, Fields &f) {          // str.split() with no argument, first 7 fields
    f.n = 0;
    while (s < end && f.n < 7) {
        while (s < end && is_ws(*s)) ++s;
        if (s >= end) break;
        const char *t = s;
        while (t < end && !is_ws(*t)) ++t;
        f.b[f.n] = s; f.e[f.n] = t; ++f.n;
        s = t;
    }
}

// int(x) for x in s.split('|'); returns false on a token that is not a plain non-negative integer
bool parse_bar_ints(const char *s, const char *e, std::vector<uint32_t> &out, size_t expect) {
    size_t n0 = out.size();
    while (true) {
        if (s >= e || *s < '0' || *s > '9') return false;
        uint64_t v = 0;
        while (s < e && *s >= '0' && *s <= '9') { v = v * 10 + (uint64_t)(*s - '0'); if (v > 0xffffffffull) return false; ++s; }
        out.push_back((uint32_t)v);
        if (s == e) break;
        if (*s != '|') return false;
        ++s;
    }
    return expect == 0 || out.size() - n0 == expect;
}

}  // namespace

int filter_files(msnv_ctx *ctx, const char *const *paths, int n_paths, uint32_t n_samples, const FilterSpecies &sp,
                 double min_cov, double min_prop, const char *out_dir, uint64_t *n_lines_kept, double *ms_kernel) {
    std::map<std::string, uint32_t> sp_index;
    for (size_t i = 0; i < sp.name.size(); ++i) sp_index[sp.name[i]] = (uint32_t)i;
    std::vector<FILE *> outs(sp.name.size(), nullptr);
    auto close_all = [&]() { for (FILE *f : outs) if (f) fclose(f); };
    FilterBatch b;
    b.n_samples = n_samples;
    std::vector<double> freq;
    std::vector<uint8_t> pass;
    std::string text;
    uint64_t kept = 0;
    const size_t BATCH_CELLS = (size_t)32 << 20;                       // 128 MB of u32 per array before a flush

    auto flush = [&]() -> int {
        if (b.row_line.empty()) { b.clear(); return MSNV_OK; }
        if (int rc = dev_filter_batch(b, sp, min_cov, min_prop, ctx->stream, freq, pass, ms_kernel)) return rc;
        for (size_t r = 0; r < b.row_line.size(); ++r) {
            const uint32_t line = b.row_line[r];
            if (!pass[line]) continue;
            const uint32_t s = b.line_species[line];
            if (!outs[s]) {                                            // created when the first position passes (:199-203)
                const std::string path = std::string(out_dir) + "/" + sp.name[s] + ".filtered.freq";
                outs[s] = fopen(path.c_str(), "w");
                if (!outs[s]) return fail(MSNV_EIO, "Cannot open %s", path.c_str());
                text.clear();
                for (const std::string &nm : sp.soi_names[s]) { text.push_back('\t'); text += nm; }
                text.push_back('\n');
                fwrite(text.data(), 1, text.size(), outs[s]);
            }
            if (r == 0 || b.row_line[r - 1] != line) ++kept;
            text.assign(b.row_id[r]);
            const uint32_t n_soi = sp.soi_off[s + 1] - sp.soi_off[s];
            const double *fr = freq.data() + b.row_out[r];
            for (uint32_t i = 0; i < n_soi; ++i) {
                text.push_back('\t');
                if (fr[i] < 0) text += "-1"; else py_repr(fr[i], text);
            }
            text.push_back('\n');
            fwrite(text.data(), 1, text.size(), outs[s]);
        }
        b.clear();
        return MSNV_OK;
    };

    std::vector<char> linebuf;
    for (int pi = 0; pi < n_paths; ++pi) {
        FILE *in = fopen(paths[pi], "r");
        if (!in) { close_all(); return fail(MSNV_EIO, "Cannot open %s", paths[pi]); }
        char *line = nullptr; size_t cap = 0; ssize_t len;
        uint64_t lineno = 0;
        while ((len = getline(&line, &cap, in)) >= 0) {
            ++lineno;
            Fields f;
            split_ws(line, line + len, f);
            if (f.n == 0) { free(line); fclose(in); close_all(); return fail(MSNV_EFORMAT, "%s:%llu: empty line (the reference fails on it)", paths[pi], (unsigned long long)lineno); }
            const char *dot = (const char *)memchr(f.b[0], '.', (size_t)(f.e[0] - f.b[0]));
            const std::string species(f.b[0], dot ? dot : f.e[0]);
            auto it = sp_index.find(species);
            if (it == sp_index.end()) continue;                        // species filter (:175)
            if (f.n < 6) { free(line); fclose(in); close_all(); return fail(MSNV_EFORMAT, "%s:%llu: fewer than 6 fields", paths[pi], (unsigned long long)lineno); }
            const uint32_t li = (uint32_t)b.line_species.size();
            if (!parse_bar_ints(f.b[4], f.e[4], b.cov, n_samples)) {
                free(line); fclose(in); close_all();
                return fail(MSNV_EFORMAT, "%s:%llu: coverage column does not hold %u integers", paths[pi], (unsigned long long)lineno, n_samples);
            }
            b.line_species.push_back(it->second);
            std::string id;
            for (int k = 0; k < 4; ++k) { if (k) id.push_back(':'); id.append(f.b[k], f.e[k]); }
            const uint32_t n_soi = sp.soi_off[it->second + 1] - sp.soi_off[it->second];
            const char *s = f.b[5], *end = f.e[5];
            while (true) {                                             // alleles, comma separated (:211)
                const char *comma = (const char *)memchr(s, ',', (size_t)(end - s));
                const char *ae = comma ? comma : end;
                const char *p1 = (const char *)memchr(s, '|', (size_t)(ae - s));
                const char *p2 = p1 ? (const char *)memchr(p1 + 1, '|', (size_t)(ae - p1 - 1)) : nullptr;
                const char *p3 = p2 ? (const char *)memchr(p2 + 1, '|', (size_t)(ae - p2 - 1)) : nullptr;
                if (!p3 || !parse_bar_ints(p3 + 1, ae, b.cnt, n_samples)) {
                    free(line); fclose(in); close_all();
                    return fail(MSNV_EFORMAT, "%s:%llu: Site coverage and SNP coverage string have uneven length!", paths[pi], (unsigned long long)lineno);   // :217-219
                }
                std::string rid = id;
                rid.push_back('>'); rid.append(p1 + 1, p2); rid.push_back(':'); rid.append(p2 + 1, p3);
                b.row_id.push_back(std::move(rid));
                b.row_line.push_back(li);
                b.row_out.push_back(b.n_out);
                b.n_out += n_soi;
                if (!comma) break;
                s = comma + 1;
            }
            if (b.cnt.size() >= BATCH_CELLS || b.n_out >= BATCH_CELLS / 2) {
                if (int rc = flush()) { free(line); fclose(in); close_all(); return rc; }
            }
        }
        free(line);
        fclose(in);
    }
    int rc = flush();
    close_all();
    if (n_lines_kept) *n_lines_kept = kept;
    return rc;
}

}  // namespace msnv
