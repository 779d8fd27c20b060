#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`*_results.db`) as a per-kernel table
(calls, total / average / min / max duration) — the `--stats` view, as text for profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void tvc::", "").replace("tvc::", "")
    return name[:150]


def main(db, out=None, skip_calls=0):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = [f"# {db}: {len(rows)} kernel dispatches, {tot/1e6:.3f} ms total GPU kernel time",
             f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f}  {short(name)}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
