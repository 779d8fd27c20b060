#!/usr/bin/env python3
"""Which launches do the __amd_rocclr_copyBuffer dispatches of a bench step belong to?  Prints, for a rocprofv3
--kernel-trace database, the histogram of the kernel dispatched right AFTER each copyBuffer (same stream order).

    python tools/copybuffer_origin.py /tmp/p4/.../run_results.db
"""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tables if "kernel_dispatch" in t and "rocpd" in t] or [t for t in tables if t == "kernels"]
try:
    rows = db.execute("select name, start from kernels order by start").fetchall()
except Exception:
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
names = [r[0] for r in rows]
nxt = Counter()
for i, n in enumerate(names[:-1]):
    if "copyBuffer" in n:
        nxt[names[i + 1].split("(")[0][:110]] += 1
for k, v in nxt.most_common(12):
    print(f"{v:6d}  next: {k}")
