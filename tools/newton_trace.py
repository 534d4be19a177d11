"""The inner Newton-PCG solves of a rocprofv3 kernel trace, kernel by kernel: what a solve's time is made of (per-kernel totals inside the
solves, idle time between their launches) and the timeline of one typical solve.

A solve = the kernels from its first one (pcg_cg_start_kernel of pcg_form 2, else grad_total_kernel) to the step_stats_kernel (step_stats_x32_kernel with pcg_rx32) behind it.
usage: python tools/newton_trace.py <dir with *_kernel_trace.csv> [index of the solve to print, default: the longest of the second half]
"""
import csv, glob, sys, collections
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("pgpfa::", "")
        grid = "%sx%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid))
ev.sort()
names = set(e[2] for e in ev)
first = "pcg_cg_start_kernel" if any(n.startswith("pcg_cg_start_kernel") for n in names) else "grad_total_kernel"
solves, cur = [], None
for e in ev:
    if e[2].startswith(first):
        cur = [e]
    elif cur is not None:
        cur.append(e)
        if e[2].startswith("step_stats_kernel") or e[2].startswith("step_stats_x32_kernel"):
            solves.append(cur)
            cur = None
if not solves:
    sys.exit("no solves found (first kernel %s)" % first)
tot = collections.defaultdict(lambda: [0, 0.0])
gap_tot, span_tot, nk = 0.0, 0.0, 0
for s in solves:
    span_tot += (s[-1][0] - s[0][0]) * 1e-3            # up to the start of step_stats
    end = s[0][0]
    for st, en, n, g in s[:-1]:
        tot[n][0] += 1
        tot[n][1] += (en - st) * 1e-3
        if st > end:
            gap_tot += (st - end) * 1e-3
        end = max(end, en)
        nk += 1
    if s[-1][0] > end:
        gap_tot += (s[-1][0] - end) * 1e-3
print("%d solves, %.2f ms in total (first kernel %s): %d launches, %.2f ms of kernels, %.2f ms idle between them (%.1f us per launch)"
      % (len(solves), span_tot * 1e-3, first, nk, sum(v[1] for v in tot.values()) * 1e-3, gap_tot * 1e-3, gap_tot / max(nk, 1)))
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-46s n=%6d total=%8.2f ms mean=%7.1f us" % (n[-46:], c, us * 1e-3, us / c))
half = solves[len(solves) // 2:]
pick = int(sys.argv[2]) if len(sys.argv) > 2 else solves.index(max(half, key=lambda s: s[-1][0] - s[0][0]))
s = solves[pick]
print("solve %d: %d launches, %.1f us" % (pick, len(s) - 1, (s[-1][0] - s[0][0]) * 1e-3))
end = s[0][0]
for st, en, n, g in s:
    print("  +%8.1f us  gap %6.1f  dur %7.1f  %-40s %s" % ((st - s[0][0]) * 1e-3, (st - end) * 1e-3, (en - st) * 1e-3, n[-40:], g))
    end = max(end, en)
