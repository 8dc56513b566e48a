#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ pass over tools/one_kernel.py (csv output): per-launch counter sums,
kernel duration, effective clock (GRBM_GUI_ACTIVE/8/duration) and MFMA-busy fraction of the kernel's cycles.
usage: pmc_sq.py <dir> <kernel-name-substring> [out.txt]"""
import collections, csv, glob, sys
d, kn = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r['Kernel_Name']:
            dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
ids = sorted(agg, key=int)[-3:]
for k in ids:
    v, t = agg[k], dur.get(k, 0)
    line = 'dispatch %s: %.1f us' % (k, t / 1e3)
    if 'GRBM_GUI_ACTIVE' in v:
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        line += ', %.0f cycles (%.2f GHz)' % (cyc, cyc / max(t, 1))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            line += ', MFMA busy %.1f %% of SIMD cycles' % (100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc)
        if 'SQ_BUSY_CU_CYCLES' in v:
            line += ', CU busy %.1f %%' % (100 * v['SQ_BUSY_CU_CYCLES'] / 256 / cyc)
    print(line, file=out)
    print('   ' + ', '.join('%s=%.4g' % (a, b) for a, b in sorted(v.items())), file=out)
