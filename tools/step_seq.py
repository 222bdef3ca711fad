#!/usr/bin/env python3
"""Kernel sequence of ONE training step (the last complete one) from a rocprofv3 rocpd database (kernel-trace): steps are cut at
the end of each adamw_kernel; prints start offset, duration and name of every dispatch - where the runtime's fill / copy kernels sit."""
import sqlite3
import sys


def main(db, only=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    cuts = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    a, b = cuts[-2], cuts[-1]
    t0 = rows[a][2]
    for n, s, e in rows[a + 1: b + 1]:
        if only and only not in n:
            continue
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {n[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
