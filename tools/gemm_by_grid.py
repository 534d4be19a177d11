"""Group the kernels of a rocprofv3 kernel trace by (name, workgroups in x, y): count, mean and total duration.

usage: python tools/gemm_by_grid.py <dir with *_kernel_trace.csv> [name filter]
"""
import csv, glob, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt and flt not in n:
            continue
        key = (n.split("(")[0][-48:], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
               int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = sum(sum(v) for v in rows.values())
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{k[0]:50s} blocks={k[1]:6d}x{k[2]:<5d} n={len(v):5d} mean={sum(v)/len(v):9.1f} us total={sum(v)*1e-3:9.2f} ms ({100*sum(v)/tot:4.1f}%)")
