#!/usr/bin/env python3
"""Per-kernel statistics (calls, total, average, share) from a rocprofv3 rocpd sqlite database,
written in the layout of `rocprofv3 --stats --output-format csv`.
usage: rocpd_stats.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                   f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout)
out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, n, t, avg, mn, mx in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    out.writerow([name[:160], n, t, '%.1f' % avg, '%.2f' % (100.0 * t / tot), mn, mx])
