#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table
(the same columns as rocprofv3's *_kernel_stats.csv).  usage: rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(n, [0, 0, 10 ** 18, 0])
        dt = e - s
        d[0] += 1
        d[1] += dt
        d[2] = min(d[2], dt)
        d[3] = max(d[3], dt)
    tot = sum(v[1] for v in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{short(n)}",{v[0]},{v[1]},{v[1] / v[0]:.0f},{100.0 * v[1] / tot:.2f},{v[2]},{v[3]}')
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
