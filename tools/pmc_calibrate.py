#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration from tools/probes/pmc_width_probe run under rocprofv3 --pmc (one pass per counter):
counter bytes / true bytes per access shape -> JSON {shape: {'fetch_ratio': .., 'write_ratio': ..}} for tools/pmc_summary.py.
usage: pmc_calibrate.py FETCH_dir WRITE_dir bytes out.json     (bytes: the figure the probe printed)"""
import collections, csv, glob, json, os, sys


def load(d):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            tot[k] += float(r['Counter_Value']); cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


fetch, write = load(sys.argv[1]), load(sys.argv[2])
nbytes = float(sys.argv[3])
used = {'read_b64_gather55': 1.0}          # (every line of the buffer is touched: the bytes that must come from memory are all of them)
out = {}
for k in sorted(set(fetch) | set(write)):
    out[k] = {'fetch_ratio': fetch.get(k, 0.0) * 1024.0 / nbytes, 'write_ratio': write.get(k, 0.0) * 1024.0 / nbytes}
    print('%-36s FETCH_SIZE/bytes %.3f   WRITE_SIZE/bytes %.3f' % (k, out[k]['fetch_ratio'], out[k]['write_ratio']))
json.dump(out, open(sys.argv[4], 'w'), indent=1)
