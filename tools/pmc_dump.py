#!/usr/bin/env python3
"""Per-launch sums of every counter of a rocprofv3 --pmc pass for one kernel (last launch): pmc_dump.py <dir> <kernel-substring>"""
import collections, csv, glob, sys
d, kn = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
k = sorted(agg, key=int)[-1]
print(d, 'dispatch', k, ' '.join('%s=%.4g' % kv for kv in sorted(agg[k].items())))
