"""Summarises a rocprofv3 kernel trace: per-kernel mean duration and mean gap to the previous kernel (steady state)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'][:40]
    dur[n].append((e - s) / 1e3)
    if prev is not None and (s - prev) / 1e3 < 20: gap[n].append((s - prev) / 1e3)
    prev = e
for n in dur:
    if len(dur[n]) < 10: continue
    d = sorted(dur[n]); g = sorted(gap[n]) or [0]
    print('%-42s n=%4d  dur median %7.2f us   gap-before median %6.2f us' % (n, len(d), d[len(d)//2], g[len(g)//2]))
