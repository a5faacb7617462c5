// metasnv_amd/csrc/filter.h -- batch layout shared by filter.cpp (host parse / print) and filter_k.hip (kernel).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace msnv {

// samples of interest of every species of interest (metaSNV_Filtering.py:111-145), flattened
struct FilterSpecies {
    std::vector<std::string> name;
    std::vector<std::vector<std::string>> soi_names;    // header of <species>.filtered.freq
    std::vector<uint32_t> soi_off, soi_idx;             // CSR: sample indices in all_samples order
};

// one batch of called_SNPs / indiv_called lines that belong to species of interest
struct FilterBatch {
    uint32_t n_samples = 0;
    std::vector<uint32_t> cov;            // [line][n_samples] site coverages (field 5)
    std::vector<uint32_t> cnt;            // [row][n_samples] allele counts (fields 4.. of an allele entry)
    std::vector<uint32_t> row_line, line_species;
    std::vector<unsigned long long> row_out;   // offset of the row's frequencies in the output
    std::vector<std::string> row_id;      // "contig:gene:pos:ref>ALT:tag"
    uint64_t n_out = 0;
    void clear() { cov.clear(); cnt.clear(); row_line.clear(); line_species.clear(); row_out.clear(); row_id.clear(); n_out = 0; }
};

int dev_filter_batch(const FilterBatch &b, const FilterSpecies &sp, double min_cov, double min_prop, void *stream,
                     std::vector<double> &freq, std::vector<uint8_t> &line_pass, double *ms_kernel);

}  // namespace msnv
