#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: calls, summed counter value.
usage: pmc_summary.py FETCH_dir WRITE_dir out.json   (separate passes, MI355X_MICROARCH.md 'HBM')"""
import collections, csv, glob, json, os, sys


def load(d):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r['Kernel_Name'].split('(')[0]
                tot[k] += float(r['Counter_Value'])
                cnt[k] += 1
    return tot, cnt


fetch, nf = load(sys.argv[1])
write, nw = load(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    calls = max(nf.get(k, 0), nw.get(k, 0))
    # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; both counters are in KB
    hbm = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
    out[k] = {'calls': calls, 'FETCH_SIZE_KB': fetch.get(k, 0.0), 'WRITE_SIZE_KB': write.get(k, 0.0),
              'hbm_bytes_corrected': hbm, 'hbm_bytes_per_launch': hbm / max(calls, 1)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k in list(out)[:8]:
    print(k, out[k])
