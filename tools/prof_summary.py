#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table (text)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(db, skip_first=0):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':112s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{short(name):112s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:9.2f} {a[3]:9.2f} {100*a[1]/tot:6.2f}")
    print(f"TOTAL kernel time {tot/1e3:.3f} ms over {len(rows)} dispatches; wall span {(rows[-1][2]-rows[0][1])/1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
