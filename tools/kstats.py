"""Per-kernel summary of a rocprofv3 --kernel-trace --stats output directory: python tools/kstats.py <dir> [max_rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= lim:
        break
    n = r['Name']; short = n.split('(')[0].replace('void ', '').replace('hrt::', '')
    print("%-62s calls %4s avg %10.1f us  total %9.2f ms  %5s%%" % (short[:62], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
