#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (CSV on stdout):
name, calls, total_ns, avg_ns, min_ns, max_ns, percent.  Usage: rocpd_stats.py results.db [> stats.csv]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for n, c, t, a, mn, mx in rows:
        print(f'"{n}",{c},{t},{a:.1f},{mn},{mx},{100.0 * t / tot:.3f}')


if __name__ == "__main__":
    main(sys.argv[1])
