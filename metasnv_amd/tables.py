"""Per-species coverage tables and the split planner -- the Python callers either side of the hot path.

These produce, byte for byte, the files the reference writes with three small scripts:

  species_summary()   <-  src/computeGenomeCoverage.py:13-52   (<bam>.cov + .detail -> .summary)
  collapse_tables()   <-  src/collapse_coverages.py:13-39      (*.summary -> <proj>.all_cov.tab / .all_perc.tab)
  plan_splits()       <-  src/createOptimumSplit.py:46-62      (greedy LPT bins of whole species)

They are pinned by golden files generated from the reference scripts themselves
(tests/golden/python_callers/, made by tests/golden/make_python_goldens.py).  The float text is
Python's own '%f' / str(float), so the host side stays Python on purpose.
"""
import glob
import os


def species_of(contig_name):
    """metaSNV's species id = contig name up to the first '.' (computeGenomeCoverage.py:25)."""
    return contig_name.split('.')[0]


def species_summary(cov_path, detail_path, summary_path):
    """Length-weighted mean coverage and >=1x / >=2x breadth per species."""
    acc = {}                                    # insertion order = first appearance, as a dict in the reference
    with open(cov_path) as cov, open(detail_path) as det:
        cov.readline()                          # "Chromosome Seq_lem Avg_Cov"
        for drow in det:                        # the .detail file ends first: the .cov tail is never reached
            crow = cov.readline().split('\t')
            dcols = drow.split('\t')
            if crow[0] != dcols[0]:
                print("Mismatch in names {} != {}".format(crow[0], dcols[0]))
            a = acc.setdefault(species_of(crow[0]), [0.0, 0.0, 0.0, 0.0])
            length = int(crow[1])
            a[0] += length
            a[1] += float(crow[2]) * length
            a[2] += int(dcols[2])
            a[3] += int(dcols[3])
    with open(summary_path, 'w') as out:
        out.write('TaxId\tAverage_cov\tPercentage_1x\tPercentage_2x\n')
        for sp, (length, wsum, c1, c2) in acc.items():
            out.write('%s\t%f\t%f\t%f\n' % (sp, wsum / length, c1 / length * 100, c2 / length * 100))


def collapse_tables(project_dir):
    """<proj>.all_cov.tab and <proj>.all_perc.tab from cov/*.summary (cells copied verbatim)."""
    name = os.path.basename(project_dir)
    bams, avg, perc = [], {}, {}
    for path in sorted(glob.glob(project_dir + '/cov/*.summary')):
        bam = os.path.basename(path)[:-len('.cov.summary')]
        with open(path) as f:
            next(f, None)
            for line in f:
                tok = line.rstrip().split()
                avg.setdefault(tok[0], {})[bam] = tok[1]
                perc.setdefault(tok[0], {})[bam] = tok[2]
        bams.append(bam)

    def dump(table, label, path):
        with open(path, 'wt') as out:
            out.write('\t' + '\t'.join(bams) + '\n')
            out.write('TaxId\t' + '\t'.join(label for _ in bams) + '\n')
            for sp in sorted(avg):
                out.write(sp + '\t' + '\t'.join(table[sp][b] for b in bams) + '\n')

    dump(avg, 'Average_cov', os.path.join(project_dir, name + '.all_cov.tab'))
    dump(perc, 'Percentage_1x', os.path.join(project_dir, name + '.all_perc.tab'))


def species_coverage(all_cov_tab):
    """{species: coverage summed over the samples} of an all_cov.tab (createOptimumSplit.py:25-40)."""
    cov = {}
    with open(all_cov_tab) as f:
        f.readline(); f.readline()
        for line in f:
            cells = line.rstrip().split('\t')
            s = 0.0
            for c in cells[1:]:
                s += float(c)
            cov[cells[0]] = s
    return cov


def species_weights(all_cov_tab, bed_header):
    """weight = genome length x summed coverage over samples (createOptimumSplit.py:18-44)."""
    length, contigs = {}, {}
    with open(bed_header) as f:
        for line in f:
            sp = species_of(line.split('\t')[0])
            length[sp] = length.get(sp, 0) + int(line.rstrip().split('\t')[2])
            contigs.setdefault(sp, []).append(line)
    cov = {}
    with open(all_cov_tab) as f:
        f.readline(); f.readline()
        for line in f:
            cells = line.rstrip().split('\t')
            s = 0.0
            for c in cells[1:]:
                s += float(c)
            cov[cells[0]] = s
    return [(length[sp] * cov[sp], sp) for sp in length], contigs


def lpt_assign(weighted, n_bins):
    """Greedy longest-processing-time: heaviest species first, each into the lightest bin
    (ties: first bin).  Returns one list of species per bin.  Also the contig -> GPU rank policy."""
    load = [0] * n_bins
    bins = [[] for _ in range(n_bins)]
    for w, sp in sorted(weighted, reverse=True):
        k = load.index(min(load))
        load[k] += w
        bins[k].append(sp)
    return bins


def plan_splits(all_cov_tab, all_perc_tab, bed_header, n_splits, out_prefix):
    """bestsplits/best_split_K files: bed_header lines of the species assigned to bin K."""
    weighted, contigs = species_weights(all_cov_tab, bed_header)
    print('Found {0} genomes'.format(len(weighted)))
    print(n_splits)
    bins = lpt_assign(weighted, n_splits)
    for k, species in enumerate(bins):
        with open('{}_{}'.format(out_prefix, k), 'w') as out:
            for sp in species:
                for line in contigs[sp]:
                    out.write(line)
    return bins
