#!/usr/bin/env python3
"""HBM traffic of one bench step's FilterNet launches from two rocprofv3 PMC passes
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only), written as the JSON that
bench.py reads for `roofline.traffic`.

    python tools/filter_traffic.py FETCH.db WRITE.db profiles/rNN_filter_traffic_pmc.json

FilterNet's launches of a step are the dispatches from the content/f0 input contraction
(igemm ... EpiSumCond) through the second fused ups.4 kernel (up24s_kernel<U24S<..., true, ...>> / up24_kernel<Up24Cfg<..., true, ...>>).
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 B);
the doubling is re-checked here on the fused ups.4 first-half kernel, whose byte counts are known.
"""
import json
import sqlite3
import sys


def per_dispatch(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, value, duration from counters_collection where counter_name=? order by dispatch_id", (counter,)).fetchall()
    agg = {}
    for did, name, val, dur in rows:
        a = agg.setdefault(did, [name, 0.0, dur])
        a[1] += val
    return [agg[k] for k in sorted(agg)]


def is_up24(name, second):
    """the fused ups.4 kernels: up24_kernel<Up24Cfg<W, D1, D2, SECOND, ...>> (fp32) or up24s_kernel<U24S<W, D1, D2, SECOND, E>> (split)."""
    for tag in ("Up24Cfg", "U24S"):
        if tag in name:
            args = name.split(tag)[-1]
            return ("true" in args) == second
    return False


def filter_segments(disp):
    """[(first, last)] index ranges of FilterNet launches, one per step."""
    segs, start = [], None
    for i, (name, _, _) in enumerate(disp):
        if "EpiSumCond" in name:
            start = i                                   # the last one before the fused ups.4 kernels is FilterNet's input layer
        if start is not None and is_up24(name, second=True):
            segs.append((start, i))
            start = None
    return segs


def main(fetch_db, write_db, out):
    f = per_dispatch(fetch_db, "FETCH_SIZE")
    w = per_dispatch(write_db, "WRITE_SIZE")
    sf, sw = filter_segments(f), filter_segments(w)
    assert sf and len(sf) == len(sw), (len(sf), len(sw))
    k = len(sf) - 1                                       # last (steady-state) step
    fa, fb = sf[k]
    wa, wb = sw[k]
    fetch_kb = sum(v for _, v, _ in f[fa:fb + 1])
    write_kb = sum(v for _, v, _ in w[wa:wb + 1])
    ms = sum(d for _, _, d in f[fa:fb + 1]) / 1e6
    halfA = [x for x in f[fa:fb + 1] if is_up24(x[0], second=False)]
    halfAw = [x for x in w[wa:wb + 1] if is_up24(x[0], second=False)]
    res = {
        "note": "FilterNet launches of one bench step (64 x 4 s, default bench.py workload): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                "separate passes (with --kernel-trace only), summed by tools/filter_traffic.py. FETCH_SIZE is doubled as "
                "MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); calibration: the fused ups.4 first-half "
                "kernel writes x1 = 64*24*96000*4 B = 576000 KB (WRITE_SIZE should report exactly that) and reads cond 576000 KB plus "
                "the low-rate x with halo (~120000 KB), to be compared with 2*FETCH_SIZE.",
        "launches_per_step": fb - fa + 1,
        "fetch_size_kb": fetch_kb,
        "write_size_kb": write_kb,
        "traffic_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
        "kernel_ms_sum_under_pmc": ms,
        "calibration": {"up24_halfA_fetch_kb": halfA[0][1] if halfA else None, "up24_halfA_write_kb": halfAw[0][1] if halfAw else None},
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
