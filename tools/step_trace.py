#!/usr/bin/env python3
"""One training step out of a rocprofv3 --kernel-trace csv: every launch with its grid (blocks) and duration,
plus totals per kernel.  usage: step_trace.py <kernel_trace.csv> [min_us]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
idx = [i for i, r in enumerate(rows) if 'adam_ema' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
tot = collections.OrderedDict()
t_first, t_last = int(rows[a + 1]['Start_Timestamp']), int(rows[b]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    blocks = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    k = tot.setdefault(n, [0, 0.0])
    k[0] += 1; k[1] += d
    if d >= min_us:
        print('%-44s %6d blocks %8.1f us' % (n[:44], blocks, d))
print('---- per kernel')
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print('%-60s %4d calls %9.1f us' % (n[:60], c, d))
print('sum of kernel time %.2f ms, first start -> last end %.2f ms' % (sum(v[1] for v in tot.values()) / 1e3, (t_last - t_first) / 1e6))
