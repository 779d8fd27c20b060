# B = 1 conversion replayed as one HIP graph, A/B of library builds: tools/ab_graph.sh libA.so libB.so ...
for rep in 1 2 3; do
for lib in "$@"; do
  TVC_LIB_PATH=$PWD/$lib python - <<PY 2>/dev/null
import sys, time, torch
sys.path.insert(0, '.')
import bench
from tinyvc_amd import synth
dev = torch.device('cuda:0')
gen = bench.build_generator(dev)
wf = synth.synth_wave(1, 96000, seed=1).to(dev); tgt = synth.synth_index(1000, seed=2).to(dev)
for _ in range(5): gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = gen.convert(wf, tgt, 0.0)
for _ in range(10): g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print("$lib  B=1 graph median %.3f ms  min %.3f" % (ts[25] * 1e3, ts[0] * 1e3))
PY
done; done
