#!/usr/bin/env python3
"""Idle time between kernels inside one training step, from a rocprofv3 rocpd database (kernel-trace): steps are cut at the
end of each adamw_kernel; prints per step wall, busy (sum of kernel durations), idle, dispatches and the largest gaps."""
import sqlite3
import sys


def main(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    cuts = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = rows[a + 1: b + 1]
        wall = (seg[-1][2] - rows[a][2]) / 1e3
        busy = sum(e - s for _, s, e in seg) / 1e3
        gaps = []
        prev_end, prev_name = rows[a][2], rows[a][0]
        for n, s, e in seg:
            gaps.append(((s - prev_end) / 1e3, prev_name[:48], n[:48]))
            prev_end, prev_name = max(prev_end, e), n
        gaps.sort(reverse=True)
        print(f"step: wall {wall:9.1f} us  busy {busy:9.1f} us  idle {wall - busy:8.1f} us  dispatches {len(seg)}")
        for g in gaps[:6]:
            print(f"    gap {g[0]:7.1f} us  after {g[1]}  before {g[2]}")


if __name__ == "__main__":
    main(sys.argv[1])
