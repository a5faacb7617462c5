// metasnv_amd/csrc/format.cpp -- text writers in the reference's on-disk formats.
//
//   called_SNPs / indiv_called   call_vC.cpp:635,641-667  (one line per called position)
//   gene column + S/N codon tag  call_vC.cpp:116-199,205-284,567-574,604-633, gene.h
//
// Only called positions reach this file (a few per thousand reference positions); the counts
// come from the device records.  The gene / codon annotation is a per-called-site lookup and
// follows the reference's observable rules, including its quirks.
#include <algorithm>
#include <cstring>
#include <map>

#include "dataset.h"

namespace msnv {

namespace {

// gene.h:28-36,67: anything that is not A/T/C/G/N is stored as 'A'
inline char genome_char(char c) { return (c == 'A' || c == 'T' || c == 'C' || c == 'G' || c == 'N') ? c : 'A'; }

const char *split_tab(const char *s, std::string &tok) {      // call_vC.cpp:92-111 semantics
    while (*s == ' ') ++s;
    const char *e = s;
    while (*e && *e != '\t') ++e;
    tok.assign(s, std::min<size_t>((size_t)(e - s), 10000));
    return *e == '\t' ? e + 1 : e;
}

}  // namespace

int load_annotation(const char *ann_path, const char *fasta_path, Annotation &an) {
    FILE *fg = fopen(ann_path, "r");
    if (!fg) return fail(MSNV_EIO, "Cannot open %s", ann_path);
    char line[10000];
    // ---- gene rows (call_vC.cpp:129-160 index + :237-280 parse).  Rows of one contig must be
    // contiguous; a later block with the same contig name replaces the earlier one (:145).
    if (!fgets(line, sizeof line, fg)) { fclose(fg); an.active = true; return MSNV_OK; }
    std::string cur, tok;
    std::vector<GeneRow> block;
    auto flush = [&]() { if (!cur.empty() || !block.empty()) an.genes[cur] = block; block.clear(); };
    bool first = true;
    while (fgets(line, sizeof line, fg)) {
        size_t l = strlen(line);
        if (l == sizeof line - 1 && line[l - 1] != '\n') { fclose(fg); return fail(MSNV_EDOMAIN, "%s: annotation line longer than 9999 characters", ann_path); }
        std::vector<std::string> f;
        const char *rest = line;
        for (int k = 0; k < 9; ++k) { rest = split_tab(rest, tok); f.push_back(tok); }
        // the contig name is only examined when something follows field 2 (:135-150)
        const bool has_name = f.size() > 3 && (!f[3].empty() || *rest);
        if (has_name) {
            if (first) { cur = f[2]; first = false; }
            else if (f[2] != cur) { flush(); cur = f[2]; }
        }
        if (f[2] != cur) continue;             // :250-253 "Reading wrong gene definition": row ignored
        GeneRow g;
        g.name = f[1];
        g.start = atol(f[6].c_str()) - 1; g.end = atol(f[7].c_str()) - 1;
        std::string st = f[8];
        while (!st.empty() && (st.back() == '\n' || st.back() == '\r')) st.pop_back();
        g.strand = st.empty() ? '\0' : st[0];
        if (g.start > g.end) continue;          // :273-275 "goes around"
        block.push_back(g);
    }
    flush();
    fclose(fg);

    // ---- genome characters exactly as indexGenomeAndGenes reads them (:165-193): every fgets
    // chunk loses its last character; the header is the whole line after '>'.
    if (!fasta_path) { an.active = true; return MSNV_OK; }   // gene rows only (formatter on the gathering rank)
    FILE *fa = fopen(fasta_path, "r");
    if (!fa) return fail(MSNV_EIO, "Cannot open %s", fasta_path);
    std::string name, genome;
    bool skip = false;
    while (fgets(line, sizeof line, fa)) {
        size_t l = strlen(line);
        if (l) line[l - 1] = '\0';
        if (line[0] == '>') {
            if (!genome.empty() && !skip) { an.genome[name] = genome; genome.clear(); }
            name = line + 1;
            skip = an.genes.find(name) == an.genes.end();
        } else if (!skip) {
            for (const char *p = line; *p; ++p) genome.push_back(genome_char(*p));
        }
    }
    an.genome[name] = genome;
    fclose(fa);
    an.active = true;
    return MSNV_OK;
}

namespace {

inline void put_u32(std::string &o, uint32_t v) {
    char b[12]; int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(b[--n]);
}

}  // namespace

// Text of called_SNPs / indiv_called from the device records.  `ann` (optional) holds the gene / codon
// annotation computed on the device (msnv_annotate_sites), one record per site; `gene_names` maps its gene
// index to the annotation's gene name column.
int write_calls_text(msnv_dataset &ds, const char *called_path, const char *indiv_path,
                     const msnv_site_ann *ann, const std::vector<std::string> *gene_names) {
    HostTimerScope ts(HT_FORMAT_WALL);
    FILE *fp = fopen(called_path, "wt");
    if (!fp) return fail(MSNV_EIO, "Cannot open %s", called_path);
    FILE *fi = indiv_path ? fopen(indiv_path, "wt") : nullptr;
    if (indiv_path && !fi) { fclose(fp); return fail(MSNV_EIO, "Cannot open %s", indiv_path); }

    const size_t S = ds.samples.size();
    static const int order[4] = {0, 1, 3, 2};               // alleles are emitted a, c, t, g (:561)
    static const char letter[4] = {'A', 'C', 'G', 'T'};
    std::string pop, ind, head, entry;
    SiteRowView rows;

    for (size_t i = 0; i < ds.sites.size(); ++i) {
        const msnv_site &s = ds.sites[i];
        if (s.dropped) continue;                              // call_vC.cpp:423
        const msnv_site_sample *ss = rows.row(ds, i, S);
        const std::string &cname = ds.names[(size_t)s.tid];
        const msnv_site_ann *an = ann ? &ann[i] : nullptr;
        const bool in_gene = an && an->gene >= 0;

        pop.clear(); ind.clear();
        bool write = false;
        for (int oi = 0; oi < 4; ++oi) {
            const int x = order[oi];
            const bool is_pop = (s.pop_mask >> x) & 1, is_ind = (s.ind_mask >> x) & 1;
            if (!is_pop && !is_ind) continue;
            if (is_pop) write = true;
            entry.clear();
            put_u32(entry, s.n[x]); entry.push_back('|'); entry.push_back(letter[x]); entry.push_back('|');
            if (in_gene) {
                const uint8_t *c = an->codon[x];             // {flags, lengths, old[3], new[3]}
                if (!(c[0] & MSNV_ANN_VALID)) {               // only a rank-local first line that is not the global one gets here
                    fclose(fp); if (fi) fclose(fi);
                    return fail(MSNV_EDOMAIN, "no codon for %s:%d (contig without FASTA record or codon past its end; reference: undefined behaviour)", cname.c_str(), s.pos + 1);
                }
                if (c[0] & MSNV_ANN_CIRCULAR) continue;       // "circular": the allele vanishes (:614-617)
                entry.push_back((c[0] & MSNV_ANN_SYNONYMOUS) ? 'S' : 'N');
                entry.push_back('[');
                entry.append(reinterpret_cast<const char *>(c + 2), (size_t)(c[1] & 15));
                entry.push_back('-');
                entry.append(reinterpret_cast<const char *>(c + 5), (size_t)(c[1] >> 4));
                entry += "]|";
            } else {
                entry += ".|";
            }
            for (size_t k = 0; k < S; ++k) { if (k) entry.push_back('|'); put_u32(entry, ss[k].n[x]); }
            std::string &dst = is_pop ? pop : ind;
            dst.push_back(','); dst += entry;
        }
        if (!write && ind.empty()) continue;
        head.clear();
        head += cname; head.push_back('\t');
        head += (in_gene && gene_names && (size_t)an->gene < gene_names->size()) ? (*gene_names)[(size_t)an->gene] : std::string("-");
        head.push_back('\t');
        if (s.pos < -1) { head.push_back('-'); put_u32(head, (uint32_t)(-((int64_t)s.pos + 1))); }     // (text entry only: snpCall prints atol(field) as it is, call_vC.cpp:499,645)
        else put_u32(head, (uint32_t)(s.pos + 1));
        head.push_back('\t'); head.push_back((char)s.refchar); head.push_back('\t');
        for (size_t k = 0; k < S; ++k) { if (k) head.push_back('|'); put_u32(head, ss[k].cov); }
        head.push_back('\t');
        if (write) {
            fwrite(head.data(), 1, head.size(), fp);
            if (!pop.empty()) fwrite(pop.data() + 1, 1, pop.size() - 1, fp);
            fputc('\n', fp);
        }
        if (!ind.empty() && fi) {
            fwrite(head.data(), 1, head.size(), fi);
            fwrite(ind.data() + 1, 1, ind.size() - 1, fi);
            fputc('\n', fi);
        }
    }
    fclose(fp);
    if (fi) fclose(fi);
    return MSNV_OK;
}

}  // namespace msnv
