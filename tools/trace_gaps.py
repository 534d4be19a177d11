"""Idle time between consecutive kernels of a rocprofv3 kernel trace, attributed to the (previous -> next) kernel pair: where the device waits
for the host (downloads, decisions, Python) or for launch latency.

usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [min gap in us (default 8)]
"""
import csv, glob, sys, collections
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("pgpfa::", "")[-40:]))
ev.sort()
busy = sum(e - s for s, e, _ in ev) * 1e-6
span = (ev[-1][1] - ev[0][0]) * 1e-6
pairs = collections.defaultdict(lambda: [0, 0.0, []])
small = [0, 0.0]
end = ev[0][1]
prev = ev[0][2]
for s, e, n in ev[1:]:
    gap = (s - end) * 1e-3
    if gap > thr:
        p = pairs[(prev, n)]
        p[0] += 1; p[1] += gap; p[2].append(gap)
    elif gap > 0:
        small[0] += 1; small[1] += gap
    if e > end:
        end, prev = e, n
print("kernels %d, span %.1f ms, busy %.1f ms (%.1f %%)" % (len(ev), span, busy, 100 * busy / span))
print("gaps <= %.0f us: %d, %.1f ms in total" % (thr, small[0], small[1] * 1e-3))
tot = sum(v[1] for v in pairs.values())
print("gaps  > %.0f us: %d, %.1f ms in total; by (previous -> next) kernel:" % (thr, sum(v[0] for v in pairs.values()), tot * 1e-3))
# (median and largest next to the mean: one allocation or cold code object inside the window reads as a per-launch cost otherwise)
for (a, b), (cnt, us, gl) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:30]:
    gl.sort()
    print("  %-40s -> %-40s n=%5d total=%8.2f ms mean=%7.1f median=%7.1f max=%9.1f us" % (a, b, cnt, us * 1e-3, us / cnt, gl[len(gl) // 2], gl[-1]))
