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

struct GeneRow { long start, end; std::string name; char strand; };

struct Annotation {
    bool active = false;
    std::map<std::string, std::vector<GeneRow>> genes;      // per contig, file order, start<=end only
    std::map<std::string, std::string> genome;              // per contig: characters as gene.h stores them
};

// gene.h:28-36,67: anything that is not A/T/C/G/N is stored as 'A'
inline char genome_char(char c) { return (c == 'A' || c == 'T' || c == 'C' || c == 'G' || c == 'N') ? c : 'A'; }

const char *split_tab(const char *s, std::string &tok) {      // call_vC.cpp:92-111 semantics
    while (*s == ' ') ++s;
    const char *e = s;
    while (*e && *e != '\t') ++e;
    tok.assign(s, std::min<size_t>((size_t)(e - s), 10000));
    return *e == '\t' ? e + 1 : e;
}

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

inline void put_u32(std::string &o, uint32_t v) {
    char b[12]; int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(b[--n]);
}

void rev_comp(std::string &c) {                             // call_vC.cpp:299-314
    std::string r;
    for (size_t i = c.size(); i-- > 0;) {
        if (c[i] == 'A') r += 'T'; else if (c[i] == 'T') r += 'A'; else if (c[i] == 'C') r += 'G'; else if (c[i] == 'G') r += 'C';
    }
    c = r;
}

char codon_aa(const std::string &c) {                       // gene.h:3-25; unknown -> '\0' (:627)
    static const struct { const char *c; char aa; } T[] = {
        {"TAA",'X'},{"TGA",'X'},{"TAG",'X'},{"GCT",'A'},{"GCC",'A'},{"GCA",'A'},{"GCG",'A'},{"CGT",'R'},{"CGC",'R'},{"CGA",'R'},
        {"CGG",'R'},{"AGA",'R'},{"AGG",'R'},{"AAT",'N'},{"AAC",'N'},{"GAT",'D'},{"GAC",'D'},{"TGT",'C'},{"TGC",'C'},{"CAA",'Q'},
        {"CAG",'Q'},{"GAA",'E'},{"GAG",'E'},{"GGT",'G'},{"GGC",'G'},{"GGA",'G'},{"GGG",'G'},{"CAT",'H'},{"CAC",'H'},{"ATT",'I'},
        {"ATC",'I'},{"ATA",'I'},{"TTA",'L'},{"TTG",'L'},{"CTT",'L'},{"CTC",'L'},{"CTA",'L'},{"CTG",'L'},{"AAA",'K'},{"AAG",'K'},
        {"ATG",'M'},{"TTT",'F'},{"TTC",'F'},{"CCT",'P'},{"CCC",'P'},{"CCA",'P'},{"CCG",'P'},{"TCT",'S'},{"TCC",'S'},{"TCA",'S'},
        {"TCG",'S'},{"AGT",'S'},{"AGC",'S'},{"ACT",'T'},{"ACC",'T'},{"ACA",'T'},{"ACG",'T'},{"TGG",'W'},{"TAT",'Y'},{"TAC",'Y'},
        {"GTA",'V'},{"GTG",'V'},{"GTT",'V'},{"GTC",'V'}};
    for (const auto &e : T) if (c == e.c) return e.aa;
    return '\0';
}

}  // namespace

int write_calls_text(msnv_dataset &ds, const char *called_path, const char *indiv_path, const char *ann_path, const char *fasta_path) {
    Annotation an;
    if (ann_path && fasta_path) if (int rc = load_annotation(ann_path, fasta_path, an)) return rc;   // call_vC.cpp:448

    FILE *fp = fopen(called_path, "wt");
    if (!fp) return fail(MSNV_EIO, "Cannot open %s", called_path);
    FILE *fi = indiv_path ? fopen(indiv_path, "wt") : nullptr;
    if (indiv_path && !fi) { fclose(fp); return fail(MSNV_EIO, "Cannot open %s", indiv_path); }

    const size_t S = ds.samples.size();
    static const int order[4] = {0, 1, 3, 2};               // alleles are emitted a, c, t, g (:561)
    static const char letter[4] = {'A', 'C', 'G', 'T'};
    std::string pop, ind, head, covs, entry;
    int cur_tid = -1;
    const std::vector<GeneRow> *genes = nullptr;
    const std::string *genome = nullptr;
    int rc = MSNV_OK;

    for (size_t i = 0; i < ds.sites.size() && !rc; ++i) {
        const msnv_site &s = ds.sites[i];
        if (s.dropped) continue;                              // call_vC.cpp:423
        const msnv_site_sample *ss = &ds.site_samples[i * S];
        const std::string &cname = ds.names[(size_t)s.tid];
        if (s.tid != cur_tid) {
            cur_tid = s.tid; genes = nullptr; genome = nullptr;
            if (an.active) {
                auto g = an.genes.find(cname);
                if (g != an.genes.end()) {
                    genes = &g->second;
                    auto q = an.genome.find(cname);
                    if (q != an.genome.end()) genome = &q->second;
                }
            }
        }
        const GeneRow *gene = nullptr;
        if (genes) for (const GeneRow &g : *genes) if (g.start <= s.pos && s.pos <= g.end) { gene = &g; break; }   // first in file order

        pop.clear(); ind.clear();
        bool write = false;
        for (int oi = 0; oi < 4 && !rc; ++oi) {
            const int x = order[oi];
            const bool is_pop = (s.pop_mask >> x) & 1, is_ind = (s.ind_mask >> x) & 1;
            if (!is_pop && !is_ind) continue;
            if (is_pop) write = true;
            entry.clear();
            put_u32(entry, s.n[x]); entry.push_back('|'); entry.push_back(letter[x]); entry.push_back('|');
            if (gene) {
                if (!(gene->start < gene->end)) continue;    // "circular": the allele vanishes (:614-617)
                if (!genome) { rc = fail(MSNV_EDOMAIN, "contig %s has genes but no FASTA record (reference: undefined behaviour)", cname.c_str()); break; }
                const int cp = (int)((s.pos - gene->start) % 3);
                const long cs = s.pos - cp;
                if (cs + 2 > (long)genome->size()) { rc = fail(MSNV_EDOMAIN, "codon at %s:%d runs past the contig end (reference: undefined behaviour)", cname.c_str(), s.pos + 1); break; }
                std::string oldc, newc;
                for (long k = cs; k <= cs + 2; ++k) oldc.push_back((size_t)k < genome->size() ? (*genome)[(size_t)k] : 'A');   // gene.h:88 reads zero bits past the end
                newc = oldc;
                newc[(size_t)cp] = letter[x];
                if (gene->strand == '-') { rev_comp(oldc); rev_comp(newc); }
                entry.push_back(codon_aa(newc) == codon_aa(oldc) ? 'S' : 'N');
                entry.push_back('['); entry += oldc; entry.push_back('-'); entry += newc; entry += "]|";
            } else {
                entry += ".|";
            }
            for (size_t k = 0; k < S; ++k) { if (k) entry.push_back('|'); put_u32(entry, ss[k].n[x]); }
            std::string &dst = is_pop ? pop : ind;
            dst.push_back(','); dst += entry;
        }
        if (rc) break;
        if (!write && ind.empty()) continue;
        head.clear();
        head += cname; head.push_back('\t'); head += gene ? gene->name : std::string("-"); head.push_back('\t');
        put_u32(head, (uint32_t)s.pos + 1); head.push_back('\t'); head.push_back((char)s.refchar); head.push_back('\t');
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
    return rc;
}

}  // namespace msnv
