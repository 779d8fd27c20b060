#!/usr/bin/env python3
"""Diagnostic (needs a library built with TVC_EXTRA_FLAGS=-DS_TRACE): cycle stamps of one wave's walk through the
conv3s slab loop, for the launches of one encoder (or convert) call.  Prints per-slab phase durations.

  TVC_LIB_PATH=... python tools/micro/slab_trace.py [--batch 64 --frames 200 --dec]
"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import build_generator  # noqa: E402
from tinyvc_amd import synth, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--dec", action="store_true")
ap.add_argument("--knn", type=int, default=0, help="trace the coarse kNN pass against an index of this many vectors instead")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = build_generator(dev)
eng = gen.engine(dev)
lib = _lib.load_library()
L = args.frames * 480
wf = synth.synth_wave(args.batch, L, seed=100).to(dev)
tgt = synth.synth_index(args.knn or 10000, seed=4).to(dev)
for _ in range(args.reps):
    out = gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 256))()
fn = getattr(lib, "tvc_debug_trace_knn" if args.knn else ("tvc_debug_trace_dec" if args.dec else "tvc_debug_trace_enc"))
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
tr = np.frombuffer(buf, dtype=np.uint64).reshape(64, 256)
names = {0: "top", 1: "bar1", 2: "lstore", 3: "loads", 4: "bar2", 5: "mfma", 7: "tile_end"}
names9 = {0: "top", 1: "stage_issue", 2: "mfma", 3: "epilogue", 4: "barrier"}
names2 = {0: "top", 1: "mfma_k0", 2: "split+ds_write", 3: "load_issue", 4: "mfma_k1", 5: "barrier"}
for slot in range(64):
    g = tr[slot]
    if (int(g[0]) >> 24) != 0x5452414345:
        continue
    h = int(g[0])
    mtb, kg, taps, scaled, film = (h >> 20) & 15, (h >> 16) & 15, (h >> 12) & 15, (h >> 8) & 15, h & 255
    cin, n = int(g[1]) >> 32, int(g[1]) & 0xffffffff
    grid, ntiles = int(g[2]) >> 32, int(g[2]) & 0xffffffff
    st = [(int(v) >> 8, int(v) & 255) for v in g[4:4 + n]]
    nm = names9 if mtb == 9 else (names2 if mtb == 2 else names)
    if int(g[3]):
        mt, rt = int(g[3]) >> 32, int(g[3]) & 0xffffffff
        print(f"   calibration: {mt} s_memtime ticks in {rt} x 10 ns -> {mt / max(rt, 1) * 100:.0f} MHz")
    print(f"slot {slot}: MTB={mtb} KG={kg} TAPS={taps} SCALED={scaled} FILM={film} Cin={cin} grid={grid} tiles={ntiles} stamps={n}")
    # per phase: time from previous stamp
    acc = {}
    prev = None
    rows = []
    for t, k in st:
        if prev is not None:
            acc.setdefault(nm.get(k, k), []).append(t - prev)
        prev = t
    tot = st[-1][0] - st[0][0] if st else 0
    print("   total ticks", tot, " per-phase mean (ticks, n):", {k: (round(float(np.mean(v)), 1), len(v)) for k, v in acc.items()})
    print("   first 14 stamps deltas:", [(nm.get(k, k), t - st[i - 1][0] if i else 0) for i, (t, k) in enumerate(st[:14])])
