"""Per-kernel sums of the counters of several rocprofv3 --pmc passes (one directory per pass), the top kernels by wave cycles.
usage: python tools/pmc_table.py <dir> [<dir> ...]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('pgpfa::', '')[-44:]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k][r['Counter_Name']] += 1
names = sorted({n for v in acc.values() for n in v})
order = sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", 0.0)))[:20]
print('counters summed over all launches of the kernel in the pass that collected them (launch counts may differ between passes by none)')
for k in order:
    print(k)
    v = acc[k]
    line = ', '.join('%s %.4g' % (n, v[n]) for n in names if n in v)
    print('   ' + line)
    g = v.get('GRBM_GUI_ACTIVE')
    if g:
        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: elapsed cycles of the kernel's launches = g / 8 (checked against the kernel trace);
        # SQ_VALU_MFMA_BUSY_CYCLES and the instruction counts are sums over the 1024 SIMDs, the LDS counters over the 256 CUs
        el = g / 8.0
        parts = ['elapsed %.3g cycles' % el]
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v: parts.append('matrix cores busy %.0f %% of a SIMD\'s time' % (100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / el))
        if 'SQ_INSTS_VALU' in v: parts.append('vector instructions x 4 cycles %.0f %%' % (100 * 4 * v['SQ_INSTS_VALU'] / 1024 / el))
        if 'SQ_INSTS_MFMA' in v and v['SQ_INSTS_MFMA'] > 0: parts.append('%.1f vector instructions per MFMA' % (v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA']))
        if 'SQ_LDS_IDX_ACTIVE' in v: parts.append('LDS array active %.0f %% of a CU\'s time (bank conflicts %.0f %%)' % (100 * v['SQ_LDS_IDX_ACTIVE'] / 256 / el, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / 256 / el))
        print('   ' + '; '.join(parts))
