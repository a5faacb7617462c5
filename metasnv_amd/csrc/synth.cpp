// metasnv_amd/csrc/synth.cpp -- deterministic synthetic workload in the shape of the
// reference's tutorial data (160 in-silico BAMs x 3 refGenomes, README.md:91-100), following
// the value distributions fixed in SURVEY.md section 8(d).  The real tarball is downloaded by
// the reference's CI and is not available offline.
//
// Output = raw BAM alignment records (SAMv1 section 4.2), coordinate sorted, which feed the
// BAM writer (tests), the packer (product path) and the oracle (checker) alike.
#include "msnv_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace msnv {

struct Rng {   // xoshiro256** seeded by splitmix64
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        for (int i = 0; i < 4; ++i) {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

static const char kBases[4] = {'A', 'C', 'G', 'T'};

// Contig table.  Default (contigs_per_species_max <= 1): n_species contigs of contig_len bases, one per species.  Otherwise
// (BASELINE configs[2] / [3] shapes) species k has 1 .. contigs_per_species_max contigs named
// refGenome<k>clus.c<j> whose lengths add up to contig_len (the SPECIES length): random cuts, every contig at least 200 bases.
struct SynthContig { int species; int64_t len; int idx_in_species; };
static std::vector<SynthContig> synth_contig_table(const msnv_synth_params &p) {
    std::vector<SynthContig> v;
    if (p.contigs_per_species_max <= 1) {
        for (int k = 0; k < p.n_species; ++k) v.push_back(SynthContig{k, p.contig_len, 0});
        return v;
    }
    Rng r(p.seed * 7349ull + 99991);
    for (int k = 0; k < p.n_species; ++k) {
        int n = 1 + (int)r.below((uint32_t)p.contigs_per_species_max);
        n = (int)std::max<int64_t>(1, std::min<int64_t>(n, p.contig_len / 400));
        std::vector<int64_t> cuts;
        for (int j = 0; j + 1 < n; ++j) cuts.push_back(200 + (int64_t)(r.uni() * (double)(p.contig_len - 400)));
        cuts.push_back(0); cuts.push_back(p.contig_len);
        std::sort(cuts.begin(), cuts.end());
        int j = 0;
        for (size_t c = 0; c + 1 < cuts.size(); ++c) {
            const int64_t len = cuts[c + 1] - cuts[c];
            if (len < 200 && c + 2 < cuts.size()) { cuts[c + 1] = cuts[c]; continue; }      // merge slivers into the next contig
            if (len > 0) v.push_back(SynthContig{k, len, j++});
        }
    }
    return v;
}

static std::string synth_contig(const msnv_synth_params &p0, int k, int64_t len) {
    msnv_synth_params p = p0;
    p.contig_len = len;
    Rng r(p.seed * 1000003ull + 1001 + (uint64_t)k);
    std::string s((size_t)p.contig_len, 'A');
    for (int64_t i = 0; i < p.contig_len; ++i) s[(size_t)i] = kBases[r.next() >> 62];
    if (p.lowercase_ref) {      // soft-masked stretches (lower-case reference: call_vC.cpp:580)
        int64_t i = 0;
        while (i < p.contig_len) {
            int64_t gap = 200 + r.below(4000), len = 20 + r.below(300);
            i += gap;
            for (int64_t j = i; j < std::min(p.contig_len, i + len); ++j) s[(size_t)j] = (char)tolower(s[(size_t)j]);
            i += len;
        }
        for (int q = 0; q < 3 && p.contig_len > 100; ++q) s[(size_t)r.below((uint32_t)p.contig_len)] = 'N';
    }
    return s;
}

// SNV sites of species k: alt base and the subspecies that carries it
struct SnvSite { int64_t pos; uint8_t alt; uint8_t carrier; };
static std::vector<SnvSite> synth_sites(const msnv_synth_params &p, int k, int species, int64_t len, const char *seq) {
    Rng r(p.seed * 7919ull + 31337 + (uint64_t)k);
    std::vector<SnvSite> v;
    int nsub = species % 3 + 1;
    for (int64_t i = 0; i < len; ++i) {
        if (r.uni() < p.snv_density) {
            uint8_t ref = nt16_of_char((unsigned char)seq[i]);
            uint8_t alt;
            do alt = (uint8_t)(1u << (r.next() >> 62)); while (alt == ref);
            v.push_back(SnvSite{i, alt, (uint8_t)r.below((uint32_t)nsub + 1)});
        }
    }
    return v;
}

static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static void put32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
static void put16(std::vector<uint8_t> &o, uint32_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }

void synth_sample_records(const msnv_synth_params &p, int sample, const std::vector<std::string> &contigs,
                          std::vector<uint8_t> &out) {
    Rng r(p.seed * 2654435761ull + 20240000ull + (uint64_t)sample);
    out.clear();
    uint64_t serial = 0;
    std::vector<uint8_t> codes, quals;
    std::vector<uint32_t> cigar;
    const std::vector<SynthContig> table = synth_contig_table(p);
    // species_per_sample > 0 (configs[2] / [3]): this sample carries exactly that many random species, the others are absent
    std::vector<uint8_t> carried;
    if (p.species_per_sample > 0) {
        carried.assign((size_t)p.n_species, 0);
        Rng rs(p.seed * 104729ull + 777 + (uint64_t)sample);
        int want = std::min(p.species_per_sample, p.n_species), have = 0;
        while (have < want) { const uint32_t k = rs.below((uint32_t)p.n_species); if (!carried[k]) { carried[k] = 1; ++have; } }
    }
    int cur_species = -1, nsub = 1, my_sub = 0;
    double cov = 0.0;
    for (int k = 0; k < (int)table.size(); ++k) {
        const std::string &seq = contigs[(size_t)k];
        const int64_t L = (int64_t)seq.size();
        if (table[(size_t)k].species != cur_species) {           // per species: subspecies of this sample and its coverage
            cur_species = table[(size_t)k].species;
            nsub = cur_species % 3 + 1;
            my_sub = (int)r.below((uint32_t)nsub);
            cov = 0.0;
            if (p.species_per_sample > 0) { if (carried[(size_t)cur_species] && r.uni() >= p.frac_absent) cov = std::exp(std::log(p.mean_cov) + p.sigma_cov * r.normal()); }
            else if (r.uni() >= p.frac_absent) cov = std::exp(std::log(p.mean_cov) + p.sigma_cov * r.normal());
        }
        int64_t n_reads = (int64_t)(cov * (double)L / p.read_len);
        if (L < p.read_len + 8) n_reads = 0;
        if (n_reads == 0 && p.species_per_sample > 0) continue;    // sparse cohorts: nothing to draw for an absent species
        std::vector<SnvSite> sites = synth_sites(p, k, cur_species, L, seq.data());
        std::vector<uint8_t> site_alt((size_t)L, 0), site_car((size_t)L, 0);
        for (const SnvSite &s : sites) { site_alt[(size_t)s.pos] = s.alt; site_car[(size_t)s.pos] = (uint8_t)(s.carrier + 1); }
        if (p.frac_paired > 0) n_reads = (int64_t)((double)n_reads / (1.0 + p.frac_paired));     // a fragment yields two reads
        std::vector<int64_t> starts((size_t)n_reads);
        for (auto &s : starts) s = (int64_t)(r.uni() * (double)(L - p.read_len - 4));
        std::sort(starts.begin(), starts.end());
        // paired-end variant (frac_paired > 0; the default workload is single-end and its bytes do not change): a start
        // becomes a proper pair with probability frac_paired -- second mate 0 .. 2 read lengths downstream, so most pairs
        // overlap on the reference (what mpileup's overlap handling is about); 4 % of the pairs lose the PROPER_PAIR bit (orphans)
        struct Job { int64_t pos, mpos; uint32_t pair_flags; int32_t tlen; uint64_t name; };
        std::vector<Job> jobs;
        jobs.reserve(starts.size());
        for (int64_t st : starts) {
            if (p.frac_paired > 0 && r.uni() < p.frac_paired) {
                const int64_t d = (int64_t)r.below((uint32_t)(2 * p.read_len)), m = std::min<int64_t>(st + d, L - p.read_len - 4);
                const uint32_t proper = r.uni() < 0.04 ? 0u : (uint32_t)BAM_FPROPER_PAIR;
                const int32_t tl = (int32_t)(m + p.read_len - st);
                jobs.push_back(Job{st, m, (uint32_t)BAM_FPAIRED | proper | 0x20u | 0x40u, tl, serial});
                jobs.push_back(Job{m, st, (uint32_t)BAM_FPAIRED | proper | (uint32_t)BAM_FREVERSE | 0x80u, -tl, serial});
                ++serial;
            } else jobs.push_back(Job{st, -1, 0u, 0, serial++});
        }
        if (p.frac_paired > 0) std::stable_sort(jobs.begin(), jobs.end(), [](const Job &a, const Job &b) { return a.pos < b.pos; });
        for (const Job &job : jobs) {
            int64_t pos = job.pos;
            int rl = p.read_len;
            cigar.clear();
            // offset of the indel inside the aligned part: 10 .. rl-21 for the usual read lengths, 2 .. rl-7 for reads below 40 bases
            auto indel_at = [&](int len) { return len >= 40 ? 10 + (int)r.below((uint32_t)(len - 30)) : 2 + (int)r.below((uint32_t)std::max(1, len - 8)); };
            double u = r.uni();
            // CIGAR mix of SURVEY.md 8(d): M only | 5S..M | one I | one D | 3H..M
            int lead_clip = 0, hard = 0, ins_at = -1, ins_len = 0, del_at = -1, del_len = 0;
            if (u < p.frac_clip_reads * 0.8) lead_clip = 5;
            else if (u < p.frac_clip_reads) hard = 3;
            else if (u < p.frac_clip_reads + p.frac_indel_reads * 0.57) { ins_len = 1 + (int)r.below(3); ins_at = indel_at(rl); }
            else if (u < p.frac_clip_reads + p.frac_indel_reads) { del_len = 1 + (int)r.below(3); del_at = indel_at(rl); }
            int l_seq = rl - hard;
            codes.assign((size_t)l_seq, 15); quals.assign((size_t)l_seq, 0);
            int q = 0; int64_t rp = pos;
            auto emit_ref = [&](int n) {
                for (int j = 0; j < n; ++j, ++q, ++rp) {
                    uint8_t c = nt16_of_char((unsigned char)seq[(size_t)rp]);
                    if (c != 1 && c != 2 && c != 4 && c != 8) c = (uint8_t)(1u << (r.next() >> 62));
                    uint8_t car = site_car[(size_t)rp];
                    if (car) {
                        if (car - 1 == my_sub) c = site_alt[(size_t)rp];
                        else if (car - 1 == nsub && r.uni() < 0.3) c = site_alt[(size_t)rp];   // within-sample polymorphism
                    }
                    if (r.uni() < p.error_rate) { uint8_t e; do e = (uint8_t)(1u << (r.next() >> 62)); while (e == c); c = e; }
                    if (r.uni() < 0.0002) c = 15;
                    codes[(size_t)q] = c;
                }
            };
            auto emit_rand = [&](int n) { for (int j = 0; j < n; ++j, ++q) codes[(size_t)q] = (uint8_t)(1u << (r.next() >> 62)); };
            if (hard) cigar.push_back((uint32_t)hard << 4 | C_H);
            if (lead_clip) { cigar.push_back((uint32_t)lead_clip << 4 | C_S); emit_rand(lead_clip); }
            int remain = l_seq - lead_clip;
            if (ins_at >= 0) {
                cigar.push_back((uint32_t)ins_at << 4 | C_M); emit_ref(ins_at);
                cigar.push_back((uint32_t)ins_len << 4 | C_I); emit_rand(ins_len);
                int rest = remain - ins_at - ins_len;
                cigar.push_back((uint32_t)rest << 4 | C_M); emit_ref(rest);
            } else if (del_at >= 0) {
                cigar.push_back((uint32_t)del_at << 4 | C_M); emit_ref(del_at);
                cigar.push_back((uint32_t)del_len << 4 | C_D); rp += del_len;
                int rest = remain - del_at;
                cigar.push_back((uint32_t)rest << 4 | C_M); emit_ref(rest);
            } else {
                cigar.push_back((uint32_t)remain << 4 | C_M); emit_ref(remain);
            }
            for (int j = 0; j < l_seq; ++j)
                quals[(size_t)j] = (uint8_t)(r.uni() < p.frac_lowq ? 2 + r.below(11) : 30 + r.below(11));
            uint32_t flag = r.uni() < 0.5 ? BAM_FREVERSE : 0;
            if (job.pair_flags) flag = job.pair_flags;
            uint32_t mapq = 60;
            double f = r.uni();
            if (f < p.frac_flagged) flag |= BAM_FDUP;
            else if (f < 2 * p.frac_flagged) flag |= BAM_FSECONDARY;
            else if (f < 2.5 * p.frac_flagged) flag |= BAM_FQCFAIL;
            else if (f < 4.5 * p.frac_flagged) mapq = 0;
            // ---- serialise (SAMv1 4.2)
            char name[32];
            int l_name = snprintf(name, sizeof name, "s%dr%llu", sample, (unsigned long long)job.name) + 1;
            // (drawn only when asked for: the default workload's streams stay what they were)
            const bool noseq = p.frac_noseq > 0 && r.uni() < p.frac_noseq;            // SEQ `*`: no bases, no qualities, the CIGAR stays
            const bool aux = p.frac_aux > 0 && r.uni() < p.frac_aux;                  // NM:C MD:Z AS:i behind the qualities
            if (noseq) l_seq = 0;
            uint8_t auxb[24]; uint32_t n_aux = 0;
            if (aux) {
                const uint8_t nm[4] = {'N', 'M', 'C', (uint8_t)r.below(4)};
                memcpy(auxb, nm, 4); n_aux = 4;
                const int md = snprintf((char *)auxb + n_aux, 8, "MDZ%d", rl) + 1; n_aux += (uint32_t)md;
                const uint8_t as[3] = {'A', 'S', 'i'};
                memcpy(auxb + n_aux, as, 3); n_aux += 3;
                const uint32_t score = (uint32_t)rl - r.below(10);
                for (int b = 0; b < 4; ++b) auxb[n_aux++] = (uint8_t)(score >> (8 * b));
            }
            uint32_t bs = 32 + (uint32_t)l_name + 4u * (uint32_t)cigar.size() + (uint32_t)(l_seq + 1) / 2 + (uint32_t)l_seq + n_aux;
            put32(out, bs);
            put32(out, (uint32_t)k); put32(out, (uint32_t)pos);
            out.push_back((uint8_t)l_name); out.push_back((uint8_t)mapq);
            put16(out, (uint32_t)reg2bin(pos, rp > pos ? rp : pos + 1));
            put16(out, (uint32_t)cigar.size()); put16(out, flag);
            put32(out, (uint32_t)l_seq);
            put32(out, job.pair_flags ? (uint32_t)k : 0xffffffffu); put32(out, (uint32_t)job.mpos); put32(out, (uint32_t)job.tlen);
            out.insert(out.end(), name, name + l_name);
            for (uint32_t c : cigar) put32(out, c);
            for (int j = 0; j < l_seq; j += 2) out.push_back((uint8_t)(codes[(size_t)j] << 4 | (j + 1 < l_seq ? codes[(size_t)j + 1] : 0)));
            out.insert(out.end(), quals.begin(), quals.begin() + l_seq);
            out.insert(out.end(), auxb, auxb + n_aux);
        }
    }
}

std::vector<std::string> synth_contigs(const msnv_synth_params &p) {
    std::vector<std::string> v;
    const std::vector<SynthContig> table = synth_contig_table(p);
    for (size_t k = 0; k < table.size(); ++k) v.push_back(synth_contig(p, (int)k, table[k].len));
    return v;
}
int synth_contig_count(const msnv_synth_params &p) { return (int)synth_contig_table(p).size(); }

}  // namespace msnv

using namespace msnv;

extern "C" void msnv_synth_params_default(msnv_synth_params *p) {
    p->n_species = 3; p->contig_len = 300000; p->n_samples = 160; p->read_len = 100;
    p->mean_cov = 10.0; p->sigma_cov = 0.5; p->frac_absent = 0.10; p->snv_density = 0.007;
    p->error_rate = 0.001; p->frac_lowq = 0.10; p->frac_indel_reads = 0.035; p->frac_clip_reads = 0.025;
    p->frac_flagged = 0.01; p->lowercase_ref = 0; p->seed = 1;
    p->frac_paired = 0.0;
    p->contigs_per_species_max = 0; p->species_per_sample = 0;
    p->frac_aux = 0.0; p->frac_noseq = 0.0;
}

extern "C" int msnv_synth_contig_count(const msnv_synth_params *p) { return p ? msnv::synth_contig_count(*p) : 0; }

extern "C" int msnv_synth_reference(const msnv_synth_params *p, char ***names, int64_t **lengths, char ***seqs) {
    clear_error();
    if (!p || !names || !lengths || !seqs) return fail(MSNV_EINVAL, "msnv_synth_reference: NULL argument");
    const std::vector<SynthContig> table = synth_contig_table(*p);
    int n = (int)table.size();
    *names = (char **)calloc((size_t)n + 1, sizeof(char *));
    *seqs = (char **)calloc((size_t)n + 1, sizeof(char *));
    *lengths = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    for (int k = 0; k < n; ++k) {
        char nm[64];
        if (p->contigs_per_species_max <= 1) snprintf(nm, sizeof nm, "refGenome%dclus", k + 1);
        else snprintf(nm, sizeof nm, "refGenome%dclus.c%d", table[(size_t)k].species + 1, table[(size_t)k].idx_in_species + 1);   // species = name up to the first '.'
        (*names)[k] = strdup(nm);
        std::string s = synth_contig(*p, k, table[(size_t)k].len);
        (*seqs)[k] = strdup(s.c_str());
        (*lengths)[k] = (int64_t)s.size();
    }
    return MSNV_OK;
}

extern "C" int msnv_synth_sample(const msnv_synth_params *p, int32_t sample_idx, char *const *seqs,
                                 uint8_t **records, uint64_t *n_bytes) {
    clear_error();
    if (!p || !seqs || !records || !n_bytes) return fail(MSNV_EINVAL, "msnv_synth_sample: NULL argument");
    std::vector<std::string> contigs;
    const int n_contigs = synth_contig_count(*p);
    for (int k = 0; k < n_contigs; ++k) contigs.emplace_back(seqs[k]);
    std::vector<uint8_t> out;
    synth_sample_records(*p, sample_idx, contigs, out);
    *records = (uint8_t *)malloc(out.size() + 1);
    if (!*records) return fail(MSNV_ENOMEM, "out of memory");
    memcpy(*records, out.data(), out.size());
    *n_bytes = out.size();
    return MSNV_OK;
}
