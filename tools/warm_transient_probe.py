"""Per-iteration wall time of the bench step from process start: looks for a start-up transient."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_generator
from tinyvc_amd import synth
from tinyvc_amd.module.tinyvc.feature_retrieval import prepare_reference

dev = torch.device("cuda", 0)
gen = build_generator(dev)
eng = gen.engine(dev)
B, L = 64, 96000
wf = synth.synth_wave(B, L, seed=100).to(dev)
blob, n = prepare_reference(synth.synth_index(10000, seed=4).to(dev))
out = torch.empty(B, L, device=dev)
t_start = time.perf_counter()
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.convert(wf, blob, n, 0.0, None, out=out)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t_start, (time.perf_counter() - t0) * 1e3))
print(" ".join(f"{ms:.1f}" for _, ms in ts))
slow = [(round(t, 2), round(ms, 1)) for t, ms in ts if ms > 22]
print("slow iterations (t_since_start_s, ms):", slow[:40])
