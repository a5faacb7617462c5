// metasnv_amd/csrc/ann_tables.cpp -- host side of the gene / codon annotation (snpCall -g).
//
// The reference answers "which gene covers lP" with a boost::icl split_interval_map per contig
// (call_vC.cpp:83,276-278,567-574; the winner is the first gene in file order, gene.h:139-146) and
// reads codons from a 3-bit packed Genome (gene.h:42-102).  Here the host only parses the files
// (format.cpp: load_annotation) and lays the answers out for the device:
//
//   segments   the gene map flattened to disjoint [gbeg, gend) runs in the dataset's linear
//              position space, each naming the winning gene -> one binary search per site
//   genes      {contig-relative start, contig, strand, start<end}
//   contigs    {first linear position, codon-genome base, codon-genome length (-1: no FASTA record)}
//   codons     the Genome characters as 4-bit codes in gene.h's numbering (A0 T1 C2 G3 N4)
//
// The per-site work (search, codon extraction, reverse complement, amino-acid compare) is the
// kernel msnv_annotate_sites in kernels.hip.
#include <algorithm>
#include <set>

#include "dataset.h"
#include "device.h"

namespace msnv {

// Global gene numbering shared by every rank and by the formatter: contigs in header order, rows in file order.
void ann_gene_names(const Annotation &an, const std::vector<std::string> &contigs, std::vector<std::string> &out) {
    out.clear();
    for (const std::string &c : contigs) {
        auto it = an.genes.find(c);
        if (it == an.genes.end()) continue;
        for (const GeneRow &g : it->second) out.push_back(g.name);
    }
}

static inline uint8_t codon_code(char c) {                  // gene.h:28-36
    switch (c) { case 'T': return 1; case 'C': return 2; case 'G': return 3; case 'N': return 4; default: return 0; }
}

int ann_build(msnv_dataset &ds, const Annotation &an, AnnHost &h) {
    h = AnnHost();
    const size_t C = ds.names.size();
    h.contigs.resize(C);
    int32_t gene_base = 0;
    std::vector<std::pair<long, int>> ev;                   // (position, +k+1 | -(k+1))
    for (size_t tid = 0; tid < C; ++tid) {
        AnnContig &ac = h.contigs[tid];
        ac.goff = 0; ac.cg_base = 0; ac.cg_len = -1;
        auto it = an.genes.find(ds.names[tid]);
        if (it == an.genes.end()) continue;
        const std::vector<GeneRow> &rows = it->second;
        const int32_t base = gene_base;
        gene_base += (int32_t)rows.size();
        for (const GeneRow &g : rows) {
            AnnGene ag;
            ag.start = g.start; ag.contig = (int32_t)tid;
            ag.flags = (g.strand == '-' ? ANN_GENE_MINUS : 0) | (g.start < g.end ? ANN_GENE_LINEAR : 0);
            h.genes.push_back(ag);
        }
        // codon genome of this contig (only contigs with gene rows are kept, call_vC.cpp:176-179)
        auto gq = an.genome.find(ds.names[tid]);
        if (gq != an.genome.end()) {
            const std::string &s = gq->second;
            ac.cg_base = (int64_t)h.codons.size() * 2;
            ac.cg_len = (int64_t)s.size();
            h.codons.resize(h.codons.size() + (s.size() + 1) / 2 + 2, 0);   // +2: the kernel may look 2 codes past the end
            uint8_t *dst = h.codons.data() + ac.cg_base / 2;
            for (size_t i = 0; i < s.size(); ++i) dst[i >> 1] |= (uint8_t)(codon_code(s[i]) << ((i & 1) * 4));
        }
        if (ds.tile_base.empty() || ds.tile_base[tid] == UINT32_MAX) continue;   // contig not in this shard: no sites
        ac.goff = ds.tile_base[tid] * TILE;
        // flatten: sweep the closed intervals, winner = smallest row index among those alive
        const long len = (long)ds.lengths[tid];
        ev.clear();
        for (size_t k = 0; k < rows.size(); ++k) {
            const long a = std::max(0L, rows[k].start), b = std::min(len - 1, rows[k].end);
            if (a > b) continue;
            ev.emplace_back(a, (int)k + 1);
            ev.emplace_back(b + 1, -((int)k + 1));
        }
        std::sort(ev.begin(), ev.end());
        std::set<int> alive;
        size_t e = 0;
        while (e < ev.size()) {
            const long p = ev[e].first;
            while (e < ev.size() && ev[e].first == p) {
                if (ev[e].second > 0) alive.insert(ev[e].second - 1); else alive.erase(-ev[e].second - 1);
                ++e;
            }
            if (alive.empty() || e >= ev.size()) continue;
            const uint32_t gb = ac.goff + (uint32_t)p, ge = ac.goff + (uint32_t)ev[e].first;
            const int32_t gene = base + *alive.begin();
            if (!h.seg_gene.empty() && h.seg_gene.back() == gene && h.seg_end.back() == gb) h.seg_end.back() = ge;
            else { h.seg_beg.push_back(gb); h.seg_end.push_back(ge); h.seg_gene.push_back(gene); }
        }
    }
    // segments must be sorted by gbeg over the whole position space: contigs are laid out in tid order
    for (size_t i = 1; i < h.seg_beg.size(); ++i)
        if (h.seg_beg[i] < h.seg_end[i - 1]) return fail(MSNV_EINVAL, "internal: annotation segments out of order");
    if (h.codons.empty()) h.codons.resize(4, 0);
    return MSNV_OK;
}

}  // namespace msnv
