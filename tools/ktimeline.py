#!/usr/bin/env python3
"""One bench step as a timeline from a rocprofv3 rocpd database: every kernel between two consecutive launches of ANCHOR (default: the
first kernel of a step, stft_fft_kernel), with start offset, duration, and the idle gap before it on the whole device.
  python tools/ktimeline.py run_results.db [anchor] [which step from the end, default 2]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name.replace("void tvc::", "").replace("tvc::", ""))
    return name[:90]


def main(db, anchor="stft_fft_kernel", back=2):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    # steps of the timed batch: anchor-to-anchor spans of 4 ... 12 ms; `back` counts from the last of them
    spans = [(idx[i], idx[i + 1]) for i in range(len(idx) - 1) if 4e6 < rows[idx[i + 1]][1] - rows[idx[i]][1] < 12e6]
    a, b = spans[-int(back)]
    t0 = rows[a][1]
    busy_end = t0
    tot_gap = 0
    for name, s, e in rows[a:b]:
        gap = max(0, s - busy_end)
        tot_gap += gap
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  {short(name)}")
        busy_end = max(busy_end, e)
    print(f"# step {(rows[b][1] - t0) / 1e3:.1f} us, {b - a} kernels, device idle between kernels {tot_gap / 1e3:.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
