#!/usr/bin/env python3
"""Sum of one PMC counter per dispatch, averaged over the dispatches of the kernels whose name contains a pattern
(rocprofv3 --pmc pass, rocpd sqlite).  usage: pmc_kernel.py <db> <name-substring>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, counter_value, duration, dispatch_id from pmc_events").fetchall()
per = {}
for n, c, v, dur, did in rows:
    if sys.argv[2] not in n:
        continue
    d = per.setdefault((did, c), [0.0, float(dur), 0])
    d[0] += float(v); d[2] += 1
by = {}
for (did, c), (v, dur, recs) in per.items():
    a = by.setdefault(c, [0, 0.0, 0.0, 0])
    a[0] += 1; a[1] += v; a[2] += dur; a[3] = recs
for c, (calls, v, dur, recs) in by.items():
    print(f"{c}: dispatches {calls} records/dispatch {recs} sum/dispatch {v / calls:.2f} avg_duration_us {dur / calls / 1e3:.2f}")
