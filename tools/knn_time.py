#!/usr/bin/env python3
"""Time the kNN stage alone: 12 800 queries (64 x 200 frames) against N-vector indices, both storages.
    python tools/knn_time.py [N ...]           (under rocprofv3 --kernel-trace --stats it gives the per-kernel split)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyvc_amd import synth  # noqa: E402
from tinyvc_amd.engine import default_engine  # noqa: E402

dev = torch.device("cuda:0")
eng = default_engine(dev)
src = torch.randn(64, 768, 200, generator=torch.Generator().manual_seed(1)).to(dev)
for N in [int(a) for a in sys.argv[1:]] or [10000, 100000]:
    idx = synth.synth_index(N, seed=5).to(dev)
    for half in (False, True):
        blob, n = eng.knn_prepare(idx.half() if half else idx)
        for _ in range(3):
            eng.knn_topk(src, blob, n)
        eng.profile(True)
        eng.knn_topk(src, blob, n)
        pr = eng.profile_read()
        eng.profile(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.knn_topk(src, blob, n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"kNN 12800 x {N} {'fp16' if half else 'fp32'} storage: {dt * 1e3:.2f} ms  {pr}")
