#!/usr/bin/env python3
"""Per-kernel PMC summary from a rocprofv3 rocpd database collected with --pmc.
Prints, per kernel name: calls, avg duration, and per-call averages of every counter, plus
derived ratios (MFMA busy per SIMD-cycle, wait fractions)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void tvc::", "").replace("tvc::", "")[:110]


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration, vgpr_count, accum_vgpr_count, lds_block_size, grid_size, workgroup_size from counters_collection").fetchall()
    disp = {}
    for did, kn, cn, val, dur, vg, ag, lds, gs, wg in rows:
        d = disp.setdefault(did, {"name": kn, "dur": dur, "vgpr": vg, "agpr": ag, "lds": lds, "grid": gs, "wg": wg, "c": defaultdict(float)})
        d["c"][cn] += val
    agg = {}
    for d in disp.values():
        a = agg.setdefault(d["name"], {"n": 0, "dur": 0.0, "c": defaultdict(float), "vgpr": d["vgpr"], "agpr": d["agpr"], "lds": d["lds"]})
        a["n"] += 1
        a["dur"] += d["dur"]
        for k, v in d["c"].items():
            a["c"][k] += v
    lines = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
        n = a["n"]
        cs = {k: v / n for k, v in a["c"].items()}
        dur_us = a["dur"] / n / 1e3
        line = f"{n:5d} {dur_us:9.1f}us vgpr={a['vgpr']}+{a['agpr']} lds={a['lds']} {short(name)}\n      "
        line += " ".join(f"{k}={v:.3g}" for k, v in sorted(cs.items()))
        wc = cs.get("SQ_WAVE_CYCLES")
        if wc:
            extra = []
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in cs:
                    extra.append(f"{k[3:]}/WAVE={cs[k] / wc:.2f}")
            line += "\n      " + " ".join(extra)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
            # MFMA busy is summed over SIMDs (cycles); 1024 SIMDs on the chip; kernel cycles ~ dur * clock
            line += f"  mfma_busy_per_simd_cycle@2.1GHz={cs['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * dur_us * 2100):.3f}"
        lines.append(line)
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
