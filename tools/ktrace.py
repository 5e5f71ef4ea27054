"""Launch-by-launch durations from a rocprofv3 --kernel-trace output directory: python tools/ktrace.py <dir> [substring] [first] [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 80
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
sel = [r for r in rows if sub in r['Kernel_Name']][first:first + count]
for r in sel:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('hrt::', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%10.3f ms  +%8.1f us  q%-3s grid %-8s %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get('Queue_Id', '?'), r.get('Grid_Size', r.get('Grid_Size_X', '?')), n[:70]))
