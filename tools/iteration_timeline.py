"""One EM iteration of a rocprofv3 kernel trace as a timeline of phases: consecutive kernels merged into runs, device-idle gaps above a
threshold listed between them.  usage: python tools/iteration_timeline.py <trace dir> [n-th yt_mix/mix launch to start from] [gap threshold us]"""
import csv, glob, os, sys
d = sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void pgpfa::', '').replace('pgpfa::', '')[:44]))
ev.sort()
marks = [i for i, e in enumerate(ev) if 'yt_mix_kernel' in e[2] or 'mix_slot' in e[2]]
i0, i1 = marks[nth], marks[nth + 1]
t0 = ev[i0][0]
busy = 0.0
run_name, run_start, run_end, run_n, run_busy = None, 0, 0, 0, 0.0
def flush():
    if run_name is not None:
        print('  %9.1f us  %-44s x%-4d busy %8.1f us (span %8.1f)' % ((run_start - t0) / 1e3, run_name, run_n, run_busy / 1e3, (run_end - run_start) / 1e3))
prev_end = ev[i0][0]
gaps = 0.0
for s, e, nm in ev[i0:i1]:
    g = (s - prev_end) / 1e3
    if g > thr:
        flush(); run_name = None
        print('  %9.1f us  ---- idle %.1f us ----' % ((prev_end - t0) / 1e3, g))
        gaps += g
    if nm != run_name:
        flush()
        run_name, run_start, run_n, run_busy = nm, s, 0, 0.0
    run_end = e; run_n += 1; run_busy += e - s
    busy += e - s
    prev_end = max(prev_end, e)
flush()
print('iteration span %.1f ms, kernel time %.1f ms, idle above %.0f us: %.1f ms' % ((ev[i1][0] - t0) / 1e6, busy / 1e6, thr, gaps / 1e3))
