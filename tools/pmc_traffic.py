#!/usr/bin/env python3
"""Parse rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/one_kernel.py, tools/x3_one.py) and write the HBM bytes per
launch of one kernel as profiles/<out>.json.
usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel-name-substring> <out.json> <algorithmic_bytes> <note>
gfx950 corrections from MI355X_MICROARCH.md: counters are in KiB; FETCH_SIZE reads exactly half the bytes of a wide
coalesced stream (x2); WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics (other widths uncalibrated)."""
import collections
import csv
import glob
import json
import sys

fetch_dir, write_dir, kname, outp, alg, note = sys.argv[1:7]
out = {}
for name, d in (('FETCH_SIZE', fetch_dir), ('WRITE_SIZE', write_dir)):
    vals = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kname in r['Kernel_Name'] and r['Counter_Name'] == name:
                vals[r['Dispatch_Id']].append(float(r['Counter_Value']))
    per_launch = [sum(v) for v in vals.values()]
    out[name + '_KiB_per_launch'] = sum(per_launch) / max(len(per_launch), 1)
    out[name + '_launches'] = len(per_launch)
fetch = out['FETCH_SIZE_KiB_per_launch'] * 1024 * 2      # gfx950: FETCH_SIZE counts 64 B per 128-B request
write = out['WRITE_SIZE_KiB_per_launch'] * 1024
out.update(hbm_read_bytes=fetch, hbm_write_bytes=write, hbm_bytes=fetch + write, algorithmic_bytes=float(alg), kernel=kname, note=note)
json.dump(out, open(outp, 'w'), indent=1)
print(json.dumps(out))
