#!/usr/bin/env python3
"""Diagnostic (needs a library built with TVC_EXTRA_FLAGS=-DU24_TRACE=100): cycle stamps of four waves of one workgroup of the two fused
ups.4 kernels over their first tiles - where a tile's cycles go (phases, barrier waits).
  TVC_LIB_PATH=$PWD/lib_u24t.so python tools/micro/u24_trace.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import build_generator  # noqa: E402
from tinyvc_amd import synth, _lib  # noqa: E402

dev = torch.device("cuda", 0)
gen = build_generator(dev)
lib = _lib.load_library()
wf = synth.synth_wave(64, 200 * 480, seed=100).to(dev)
tgt = synth.synth_index(10000, seed=8).to(dev)
for _ in range(3):
    gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 4 * 64))()
lib.tvc_debug_trace_u24.argtypes = [ctypes.c_void_p]
assert lib.tvc_debug_trace_u24(buf) == 0
tr = np.frombuffer(buf, dtype=np.uint64).reshape(2, 4, 64)
names = {9: "scales", 10: "cond issued", 0: "top", 1: "cond+fetch issued", 2: "S1 done", 3: "barrier", 4: "S2 done", 5: "barrier(S4)", 6: "S4 done", 7: "barrier", 8: "deposit done"}
for half in range(2):
    print("half", "AB"[half])
    for slot, w in enumerate((0, 1, 4, 7)):
        n = int(tr[half, slot, 0])
        st = [(int(v) >> 8, int(v) & 255) for v in tr[half, slot, 1:1 + n]]
        # durations per phase id, skipping the first tile
        acc = {}
        tiles = 0
        prev = None
        for t, k in st:
            if k == 0:
                tiles += 1
            if prev is not None and tiles >= 2:
                acc.setdefault(k, []).append(t - prev)
            prev = t
        tot = sum(sum(v) / len(v) for v in acc.values())
        print(f"  wave {w}: " + "  ".join(f"{names[k]} {sum(v) / len(v):.0f}" for k, v in acc.items()) + f"   | tile {tot:.0f} cycles")
