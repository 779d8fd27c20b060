"""Does a stage's speed depend on where the scratch workspace sits?  Times the ConvNeXt GEMMs of the
bench workload with the workspace placed at different byte offsets inside one big allocation."""
import os, sys, json
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
need = eng.workspace(B, L, n).numel()
big = torch.empty(need + (64 << 20), dtype=torch.uint8, device=dev)
print("base ptr %x need %.2f GB" % (big.data_ptr(), need / 1e9))
for off in [0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 16 << 20, 33 << 20, 0]:
    eng._ws = big[off:off + need]
    eng.workspace = lambda *a, **k: eng._ws
    for _ in range(2):
        eng.convert(wf, blob, n, 0.0, None, out=out)
    eng.profile(True); eng.profile_read()
    for _ in range(3):
        eng.convert(wf, blob, n, 0.0, None, out=out)
    p = eng.profile_read(); eng.profile(False)
    print(f"off {off:>10d}: c2 {p['cnx.c2_gelu']/3:.2f} c3 {p['cnx.c3_res']/3:.2f} dw {p['cnx.dwconv_ln']/3:.2f} filter {p['filter_net']/3:.2f} knn {p['knn']/3:.2f}")
