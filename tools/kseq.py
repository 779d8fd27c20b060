#!/usr/bin/env python3
"""Per-launch-position durations of one kernel in a rocprofv3 rocpd database: the launches whose name contains PATTERN, in start
order, folded modulo PERIOD (= launches of that kernel per bench step): mean / min in microseconds per position.
  python tools/kseq.py run_results.db film_s2_kernel 6"""
import sqlite3
import sys


def main(db, pattern, period):
    c = sqlite3.connect(db)
    rows = [e - s for name, s, e in c.execute("select name, start, end from kernels order by start") if pattern in name]
    period = int(period)
    n = len(rows) // period * period
    rows = rows[len(rows) - n:]
    for i in range(period):
        d = rows[i::period]
        print(f"{pattern}[{i}]: n={len(d)} mean {sum(d)/len(d)/1e3:8.2f} us  min {min(d)/1e3:8.2f} us")


if __name__ == "__main__":
    main(*sys.argv[1:4])
