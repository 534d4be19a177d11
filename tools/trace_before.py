import csv, glob, os, sys
d, name = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:70]))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s bytes' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')))))
ev.sort()
idx = [i for i, e in enumerate(ev) if name in e[2]]
i = idx[len(idx) // 2]
t0 = ev[i][0]
for j in range(max(0, i - 14), i + 2):
    s, e, what = ev[j]
    print('%9.1f us  +%7.1f  gap-before %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - ev[j - 1][1]) / 1e3 if j else 0.0, what))
