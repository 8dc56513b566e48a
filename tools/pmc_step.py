#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc SQ pass over bench.py's training step (csv output): for every kernel of the step, its
launches, mean duration, effective clock (GRBM_GUI_ACTIVE / 8 / duration), MFMA-busy share of all SIMD cycles and CU-busy share.
Counters serialise the launches (one kernel on the chip at a time): durations are the single-stream ones.
usage: pmc_step.py <dir> [out.txt]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        name[r['Dispatch_Id']] = r['Kernel_Name']
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
ids = sorted(agg, key=int)
adam = [i for i in ids if 'adam_ema' in name[i]]
if len(adam) >= 2:                                   # the last whole step
    lo, hi = int(adam[-2]), int(adam[-1])
    ids = [i for i in ids if lo < int(i) <= hi]
per = collections.OrderedDict()
for i in ids:
    n = re.sub(r'\(anonymous namespace\)::', '', name[i])
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    p = per.setdefault(n, dict(n=0, t=0.0, cyc=0.0, mfma=0.0, cu=0.0))
    v = agg[i]
    p['n'] += 1; p['t'] += dur.get(i, 0); p['cyc'] += v.get('GRBM_GUI_ACTIVE', 0) / 8
    p['mfma'] += v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0); p['cu'] += v.get('SQ_BUSY_CU_CYCLES', 0)
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
tot = sum(p['t'] for p in per.values())
print('%-58s %5s %9s %9s %6s %9s %8s' % ('kernel (one training step, counters on)', 'calls', 'total us', 'mean us', 'GHz', 'MFMA busy', 'CU busy'), file=out)
for n, p in sorted(per.items(), key=lambda kv: -kv[1]['t']):
    if p['t'] < 0.002 * tot:
        continue
    cyc = max(p['cyc'], 1.0)
    print('%-58s %5d %9.1f %9.1f %6.2f %8.1f%% %7.1f%%' % (n[:58], p['n'], p['t'] / 1e3, p['t'] / 1e3 / p['n'], cyc / max(p['t'], 1),
                                                       100 * p['mfma'] / 1024 / cyc, 100 * p['cu'] / 256 / cyc), file=out)
print('sum of kernel time %.2f ms' % (tot / 1e6), file=out)
