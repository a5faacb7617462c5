"""profiles/kstats.py FILE [min_us] -- one line per kernel of a rocprofv3 kernel-stats CSV (short names, microseconds)"""
import csv, re, sys
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if float(r['TotalDurationNs']) < mn * 1e3: continue
    m = re.search(r'(msnv_\w+|__amd_rocclr_\w+)', n)
    nm = m.group(1) if m else ('rocprim:' + (re.findall(r'wrapped_(\w+)_config', n) or re.findall(r'detail::(\w+)', n) or ['?'])[0] + ' ' + ('RecCnt' if 'RecCnt' in n else 'u64' if 'unsigned long long' in n else 'u32'))
    print('%-44s %5s %10.1f us total %10.1f avg' % (nm, r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3))
