import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('void ', '').replace('pgpfa::', '')[:28] for r in rows]
pat = collections.Counter()
gapsum = collections.Counter()
half = len(rows) // 2
for i in range(half, len(rows)):
    if 'copyBuffer' in names[i]:
        key = ' | '.join(names[max(0, i - 3):i]) + '  ==> copy ==>  ' + ' | '.join(names[i + 1:i + 3])
        pat[key] += 1
        gapsum[key] += (int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
for k, n in pat.most_common(12):
    print('%4d x  gap before copy %.0f us mean : %s' % (n, gapsum[k] / n, k))
