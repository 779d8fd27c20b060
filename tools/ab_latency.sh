# B = 1 and streaming latency A/B of library builds: tools/ab_latency.sh libA.so libB.so ...
for rep in 1 2; do
for lib in "$@"; do
  b1=$(TVC_LIB_PATH=$PWD/$lib python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
from tinyvc_amd import synth
dev = torch.device('cuda:0')
gen = bench.build_generator(dev)
wf = synth.synth_wave(1, 96000, seed=1).to(dev); tgt = synth.synth_index(1000, seed=2).to(dev)
for _ in range(10): gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): gen.convert(wf, tgt, 0.0)
torch.cuda.synchronize(); print("%.3f" % ((time.perf_counter() - t0) * 10))
PY
)
  st=$(TVC_LIB_PATH=$PWD/$lib python bench_stream.py --blocks 120 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['p50_ms'])")
  echo "$lib  B=1 $b1 ms   stream p50 $st ms"
done; done
