#!/usr/bin/env python3
"""Timing experiment: the configs[1] batch step as one HIP-graph replay vs eager launches (same pointers every step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import build_generator  # noqa: E402
from tinyvc_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
gen = build_generator(dev)
wf = synth.synth_wave(64, 96000, seed=100).to(dev)
tgt = synth.synth_index(10000, seed=4).to(dev)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(lambda: gen.convert(wf, tgt, 0.0))
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(2):
        out = gen.convert(wf, tgt, 0.0)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize()
graph = timeit(g.replay)
assert torch.isfinite(out).all()
print(f"eager {eager:.3f} ms/step   graph replay {graph:.3f} ms/step")
