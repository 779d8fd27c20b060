"""Does the batch step profit from running as sequential sub-batches (working sets inside the 256 MB memory-side cache)?
64 utterances x 4 s as one call vs 2 x 32 / 4 x 16 / 8 x 8 calls back to back on one stream (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tinyvc_amd import synth
dev = torch.device("cuda:0")
gen = bench.build_generator(dev)
wf = synth.synth_wave(64, 96000, seed=100).to(dev)
tgt = synth.synth_index(10000, seed=8).to(dev)
def run(nb):
    for b0 in range(0, 64, nb):
        gen.convert(wf[b0:b0 + nb], tgt, 0.0)
for nb in (64, 32, 16, 8, 64):
    for _ in range(3): run(nb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(nb)
    torch.cuda.synchronize()
    print(f"sub-batch {nb:3d}: {(time.perf_counter() - t0) * 100:.3f} ms per 64 utterances")
