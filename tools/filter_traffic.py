#!/usr/bin/env python3
"""HBM traffic of one bench step's FilterNet launches from two rocprofv3 PMC passes
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only), written as the JSON that
bench.py reads for `roofline.traffic`.

    python tools/filter_traffic.py FETCH.db WRITE.db profiles/rNN_filter_traffic_pmc.json

FilterNet's launches of a step are its content/f0 input contraction (gemm ... EpiSumCond, launched early on the library's side stream) and the
dispatches from downs.0 (down0s_kernel) through the second fused ups.4 kernel (up24s_kernel<U24S<..., true, ...>>).
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
    """[[indices]] of FilterNet's launches, one list per step: its input contraction + everything from downs.0 (down0s_kernel) through the
    second fused ups.4 kernel.  Since round 6 the input contraction is launched EARLY, on the library's side stream (decoder.hip run_decoder),
    so it is no longer adjacent: of the EpiSumCond contractions dispatched since the previous step's last FilterNet launch - SourceNet's
    768 -> 128 and FilterNet's 768 -> 384 - it is the longer one (three times the rows)."""
    segs, start, gemms = [], None, []
    for i, (name, _, dur) in enumerate(disp):
        if "EpiSumCond" in name:
            gemms.append((dur, i))
        if "down0s_kernel" in name and start is None:
            start = i
        if start is not None and is_up24(name, second=True):
            g = max(gemms)[1] if gemms else None
            segs.append(([g] if g is not None and g < start else []) + list(range(start, i + 1)))
            start, gemms = None, []
    return segs


def main(fetch_db, write_db, out):
    f = per_dispatch(fetch_db, "FETCH_SIZE")
    w = per_dispatch(write_db, "WRITE_SIZE")
    sf, sw = filter_segments(f), filter_segments(w)
    assert sf and len(sf) == len(sw), (len(sf), len(sw))
    k = len(sf) - 1                                       # last (steady-state) step
    fsel, wsel = [f[i] for i in sf[k]], [w[i] for i in sw[k]]
    assert len(fsel) == len(wsel), (len(fsel), len(wsel))
    fetch_kb = sum(v for _, v, _ in fsel)
    write_kb = sum(v for _, v, _ in wsel)
    ms = sum(d for _, _, d in fsel) / 1e6
    halfA = [x for x in fsel if is_up24(x[0], second=False)]
    halfAw = [x for x in wsel if is_up24(x[0], second=False)]
    res = {
        "note": "FilterNet launches of one bench step (64 x 4 s, default bench.py workload): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                "separate passes (with --kernel-trace only), summed by tools/filter_traffic.py. FETCH_SIZE is doubled as "
                "MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); calibration: the fused ups.4 first-half "
                "kernel writes x1 = 64*24*96000*4 B = 576000 KB (WRITE_SIZE should report exactly that) and reads cond 576000 KB plus "
                "the low-rate x with halo (~120000 KB), to be compared with 2*FETCH_SIZE.",
        "launches_per_step": len(fsel),
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
