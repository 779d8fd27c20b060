"""Sync-to-sync wall time of ONE 64 x 4 s conversion (median of 30) and of back-to-back calls, for the library TVC_LIB_PATH selects."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tinyvc_amd import synth
dev = torch.device("cuda:0")
gen = bench.build_generator(dev)
wf = synth.synth_wave(64, 96000, seed=100).to(dev)
tgt = synth.synth_index(10000, seed=8).to(dev)
for _ in range(5): gen.convert(wf, tgt, 0.0)
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gen.convert(wf, tgt, 0.0)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize(); bb = (time.perf_counter() - t0) / 20
print(f"{os.environ.get('TVC_LIB_PATH', 'default')}: one call sync-to-sync median {ts[15] * 1e3:.3f} ms (min {ts[0] * 1e3:.3f}), back to back {bb * 1e3:.3f} ms")
