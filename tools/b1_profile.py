"""One 4 s utterance against a 1 000-vector index (BASELINE configs[0]) 25 times: run under `rocprofv3 --kernel-trace --stats` for the per-kernel split of a B = 1 conversion
(profiles/r04_b1_kernel_stats.txt)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyvc_amd import synth
import bench
dev = torch.device('cuda:0')
gen = bench.build_generator(dev)
wf = synth.synth_wave(1, 96000, seed=1).to(dev)
tgt = synth.synth_index(1000, seed=2).to(dev)
for _ in range(5): gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize()
for _ in range(20): gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize()
