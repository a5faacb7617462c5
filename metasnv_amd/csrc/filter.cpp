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

struct msnv_ctx;

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

namespace {
// Batches lines of species of interest, runs the device kernel and prints `<species>.filtered.freq` (shared by the path that
// parses called_SNPs text and the one that reads the resident records).
struct FilterWriter {
    msnv_ctx *ctx; const FilterSpecies &sp; double min_cov, min_prop; const char *out_dir; double *ms_kernel;
    std::vector<FILE *> outs;
    FilterBatch b;
    std::vector<double> freq;
    std::vector<uint8_t> pass;
    std::string text;
    uint64_t kept = 0;
    static constexpr size_t BATCH_CELLS = (size_t)32 << 20;           // 128 MB of u32 per array before a flush
    FilterWriter(msnv_ctx *c, const FilterSpecies &s, uint32_t n_samples, double mc, double mp, const char *od, double *ms)
        : ctx(c), sp(s), min_cov(mc), min_prop(mp), out_dir(od), ms_kernel(ms), outs(s.name.size(), nullptr) { b.n_samples = n_samples; }
    ~FilterWriter() { close_all(); }
    void close_all() { for (FILE *&f : outs) if (f) { fclose(f); f = nullptr; } }
    bool full() const { return b.cnt.size() >= BATCH_CELLS || b.n_out >= BATCH_CELLS / 2; }
    int flush() {
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
    }
};
}  // namespace

int filter_files(msnv_ctx *ctx, const char *const *paths, int n_paths, uint32_t n_samples, const FilterSpecies &sp,
                 double min_cov, double min_prop, const char *out_dir, uint64_t *n_lines_kept, double *ms_kernel) {
    std::map<std::string, uint32_t> sp_index;
    for (size_t i = 0; i < sp.name.size(); ++i) sp_index[sp.name[i]] = (uint32_t)i;
    FilterWriter W(ctx, sp, n_samples, min_cov, min_prop, out_dir, ms_kernel);
    FilterBatch &b = W.b;
    auto close_all = [&]() { W.close_all(); };
    auto flush = [&]() -> int { return W.flush(); };

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
            if (W.full()) {
                if (int rc = flush()) { free(line); fclose(in); close_all(); return rc; }
            }
        }
        free(line);
        fclose(in);
    }
    int rc = flush();
    close_all();
    if (n_lines_kept) *n_lines_kept = W.kept;
    return rc;
}

// filter_two straight from the records of the last pass ("avoids re-parsing called_SNPs text"):
// the same batches, kernel and printer as filter_files, fed from ds.sites / ds.site_samples and the device annotation records
// instead of S-wide text lines.  which = 0: the lines of called_SNPs (population calls), 1: those of indiv_called.
int filter_resident(msnv_dataset &ds, int which, const FilterSpecies &sp, double min_cov, double min_prop, const char *out_dir,
                    const msnv_site_ann *ann, const std::vector<std::string> *gene_names, uint64_t *n_lines_kept, double *ms_kernel) {
    std::map<std::string, uint32_t> sp_index;
    for (size_t i = 0; i < sp.name.size(); ++i) sp_index[sp.name[i]] = (uint32_t)i;
    const size_t S = ds.samples.size();
    FilterWriter W(ds.ctx, sp, (uint32_t)S, min_cov, min_prop, out_dir, ms_kernel);
    FilterBatch &b = W.b;
    static const int order[4] = {0, 1, 3, 2};               // alleles are emitted a, c, t, g (call_vC.cpp:561)
    static const char letter[4] = {'A', 'C', 'G', 'T'};
    std::vector<int32_t> contig_species(ds.names.size(), -1);
    for (size_t c = 0; c < ds.names.size(); ++c) {
        const std::string &nm = ds.names[c];
        auto it = sp_index.find(nm.substr(0, nm.find('.')));
        if (it != sp_index.end()) contig_species[c] = (int32_t)it->second;
    }
    std::string id, tag;
    SiteRowView rows;
    for (size_t i = 0; i < ds.sites.size(); ++i) {
        const msnv_site &s = ds.sites[i];
        if (s.dropped) continue;                              // call_vC.cpp:423: the line never reaches the files
        const int32_t spi = contig_species[(size_t)s.tid];
        if (spi < 0) continue;                                // species filter (metaSNV_Filtering.py:175)
        const uint32_t mask = which ? s.ind_mask : s.pop_mask;
        if (!mask) continue;                                  // no line of this kind for the position
        const msnv_site_sample *ss = rows.row(ds, i, S);
        const msnv_site_ann *an = ann ? &ann[i] : nullptr;
        const bool in_gene = an && an->gene >= 0;
        id.assign(ds.names[(size_t)s.tid]); id.push_back(':');
        id += (in_gene && gene_names && (size_t)an->gene < gene_names->size()) ? (*gene_names)[(size_t)an->gene] : std::string("-");
        id.push_back(':'); id += std::to_string(s.pos + 1); id.push_back(':'); id.push_back((char)s.refchar);
        const uint32_t li = (uint32_t)b.line_species.size();
        const uint32_t n_soi = sp.soi_off[(size_t)spi + 1] - sp.soi_off[(size_t)spi];
        uint32_t n_rows = 0;
        for (int oi = 0; oi < 4; ++oi) {
            const int x = order[oi];
            if (!((mask >> x) & 1u)) continue;
            tag.assign(".");
            if (in_gene) {
                const uint8_t *c = an->codon[x];
                if (!(c[0] & MSNV_ANN_VALID)) return fail(MSNV_EDOMAIN, "no codon for %s:%d", ds.names[(size_t)s.tid].c_str(), s.pos + 1);
                if (c[0] & MSNV_ANN_CIRCULAR) continue;       // the allele vanishes from the line (call_vC.cpp:614-617)
                tag.assign(1, (c[0] & MSNV_ANN_SYNONYMOUS) ? 'S' : 'N');
                tag.push_back('[');
                tag.append(reinterpret_cast<const char *>(c + 2), (size_t)(c[1] & 15));
                tag.push_back('-');
                tag.append(reinterpret_cast<const char *>(c + 5), (size_t)(c[1] >> 4));
                tag.push_back(']');
            }
            if (n_rows == 0) { for (size_t k = 0; k < S; ++k) b.cov.push_back(ss[k].cov); b.line_species.push_back((uint32_t)spi); }
            for (size_t k = 0; k < S; ++k) b.cnt.push_back(ss[k].n[x]);
            std::string rid = id;
            rid.push_back('>'); rid.push_back(letter[x]); rid.push_back(':'); rid += tag;
            b.row_id.push_back(std::move(rid));
            b.row_line.push_back(li);
            b.row_out.push_back(b.n_out);
            b.n_out += n_soi;
            ++n_rows;
        }
        // a population line whose alleles all vanished is written with an empty allele field by the reference, which its own
        // Filtering script cannot split (fewer than 6 fields): the same error as the file path gives
        if (n_rows == 0 && which == 0) return fail(MSNV_EFORMAT, "%s:%d: the called_SNPs line has no allele left (circular gene): fewer than 6 fields", ds.names[(size_t)s.tid].c_str(), s.pos + 1);
        if (W.full()) if (int rc = W.flush()) return rc;
    }
    int rc = W.flush();
    W.close_all();
    if (n_lines_kept) *n_lines_kept = W.kept;
    return rc;
}

}  // namespace msnv
