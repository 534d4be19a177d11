"""The largest device-idle gaps of a rocprofv3 kernel (+ memory-copy) trace with the events around each.  usage: python tools/trace_top_gaps.py <trace dir> [count] [skip first fraction]"""
import csv, glob, os, sys
d = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:64]))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s bytes' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')))))
ev.sort()
first = int(len(ev) * skip)
gaps = sorted(((ev[i][0] - max(e[1] for e in ev[max(0, i - 4):i]), i) for i in range(max(first, 1), len(ev))), reverse=True)[:count]
for g, i in gaps:
    print('gap %.1f us before event %d:' % (g / 1e3, i))
    for j in range(max(0, i - 3), min(len(ev), i + 3)):
        print('    %s %9.1f us  +%8.1f  %s' % ('>>' if j == i else '  ', (ev[j][0] - ev[i][0]) / 1e3, (ev[j][1] - ev[j][0]) / 1e3, ev[j][2]))
