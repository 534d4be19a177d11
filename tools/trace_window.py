"""Events (kernels and memory copies) of a rocprofv3 trace around the n-th launch of a kernel: what the device does in a gap.
usage: python tools/trace_window.py <trace dir> <kernel substring> [n] [microseconds after] """
import csv, glob, os, sys
d, name = sys.argv[1], sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
after = float(sys.argv[4]) if len(sys.argv) > 4 else 1500.0
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:60]))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s bytes' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')))))
ev.sort()
hits = [e for e in ev if name in e[2]]
t0 = hits[min(nth, len(hits) - 1)][1]
for s, e, what in ev:
    if s >= t0 - 60000 and s <= t0 + after * 1000:
        print('%9.1f us  +%7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, what))
