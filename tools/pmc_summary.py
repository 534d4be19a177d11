#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: calls, summed counter value.
usage: pmc_summary.py FETCH_dir WRITE_dir out.json [calibration.json]   (separate passes, MI355X_MICROARCH.md 'HBM')

FETCH_SIZE is doubled for EVERY kernel, whatever its load width.  The guide states the factor for 16-byte-per-lane streaming reads and calls other
widths uncalibrated; tools/probes/pmc_width_probe.hip + tools/pmc_calibrate.py measured them on this chip (profiles/r05_pmc_calibration.json: a 2-GiB
buffer streamed once per shape): FETCH_SIZE / bytes = 0.500 for 4-, 8- and 16-byte loads alike, 0.515 for the (C,d) M-step's 8-byte gather of 55 of every
100 doubles, 0.524 for 64-double runs of rows that are not line-aligned; WRITE_SIZE / bytes = 1.000 for 4-, 8- and 16-byte stores (0.5 for the runtime's
fill kernel).  With a calibration file the factor is 1 / (its 16-byte ratio) instead of the literal 2."""
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


factor = 2.0
if len(sys.argv) > 4:
    cal = json.load(open(sys.argv[4]))
    r16 = [v['fetch_ratio'] for k, v in cal.items() if 'read_kernel' in k and '4u' in k]
    if r16 and r16[0] > 0:
        factor = 1.0 / r16[0]
fetch, nf = load(sys.argv[1])
write, nw = load(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    calls = max(nf.get(k, 0), nw.get(k, 0))
    # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; both counters are in KB
    hbm = (factor * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
    out[k] = {'calls': calls, 'FETCH_SIZE_KB': fetch.get(k, 0.0), 'WRITE_SIZE_KB': write.get(k, 0.0),
              'hbm_bytes_corrected': hbm, 'hbm_bytes_per_launch': hbm / max(calls, 1)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k in list(out)[:8]:
    print(k, out[k])
