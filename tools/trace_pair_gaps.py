"""Device-idle gaps between one (previous -> next) kernel pair of a rocprofv3 kernel trace: their distribution and the launches around a few of them.
usage: python tools/trace_pair_gaps.py <trace dir> <previous kernel substring> <next kernel substring> [min gap us] [contexts]"""
import csv, glob, os, sys
d, prev, nxt = sys.argv[1], sys.argv[2], sys.argv[3]
floor_us = float(sys.argv[4]) if len(sys.argv) > 4 else 8.0
contexts = int(sys.argv[5]) if len(sys.argv) > 5 else 4
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], (int(r.get('Grid_Size_X', 0) or 0), int(r.get('Grid_Size_Y', 0) or 0))))
ev.sort()
hits = []
for i in range(1, len(ev)):
    if prev in ev[i - 1][2] and nxt in ev[i][2]:
        g = (ev[i][0] - ev[i - 1][1]) / 1e3
        if g > floor_us: hits.append((g, i))
gs = sorted(g for g, _ in hits)
if not gs:
    print('no such gaps'); sys.exit(0)
print('%d gaps > %.0f us, %.1f ms in total; min %.1f  p25 %.1f  median %.1f  p75 %.1f  p95 %.1f  max %.1f us' % (
    len(gs), floor_us, sum(gs) / 1e3, gs[0], gs[len(gs) // 4], gs[len(gs) // 2], gs[3 * len(gs) // 4], gs[int(0.95 * (len(gs) - 1))], gs[-1]))
edges = [8, 16, 32, 64, 128, 256, 512, 1024, 1e9]
lo = 0.0
for e in edges:
    sel = [g for g in gs if lo < g <= e]
    if sel: print('  (%6.0f, %6.0f] us: n = %4d, %8.2f ms' % (lo, min(e, gs[-1]), len(sel), sum(sel) / 1e3))
    lo = e
hits.sort()
pick = [hits[len(hits) // 2], hits[3 * len(hits) // 4], hits[-1]][:contexts]
for g, i in pick:
    print('gap %.1f us before launch %d:' % (g, i))
    for j in range(max(0, i - 9), min(len(ev), i + 5)):
        print('    %s %9.1f us  +%8.1f  gap-before %7.1f  grid %-12s %s' % ('>>' if j == i else '  ', (ev[j][0] - ev[i][0]) / 1e3, (ev[j][1] - ev[j][0]) / 1e3,
                                                                     (ev[j][0] - ev[j - 1][1]) / 1e3 if j else 0.0, '%dx%d' % ev[j][3], ev[j][2][:70]))
