#!/usr/bin/env python3
"""Does running the 64-utterance batch as concurrent equal-length sub-batches (the ragged API's lanes) beat one batch?
Two / four groups whose lengths differ by one frame, against the plain B = 64 call.  Prints ms per call."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinyvc_amd import synth
from tinyvc_amd.module.infer import Generator
from tinyvc_amd.module.tinyvc import Decoder, Encoder

dev = "cuda:0"
enc, dec = Encoder(), Decoder()
enc.load_state_dict(synth.synth_state_dict("encoder")); dec.load_state_dict(synth.synth_state_dict("decoder"))
gen = Generator(enc, dec).to(dev)
tgt = synth.synth_index(10000, seed=4).to(dev)
wf = synth.synth_wave(64, 96000, seed=100).to(dev)
angle = synth.synth_angle(64, 200, 5).to(dev)

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print("one batch of 64:        %.3f ms" % timeit(lambda: gen.convert(wf, tgt, 0.0, noise_angle=angle)))
for groups in (2, 4):
    lens = [96000 - 480 * (i % groups) for i in range(64)]
    print("%d concurrent groups:    %.3f ms" % (groups, timeit(lambda: gen.convert(wf, tgt, 0.0, noise_angle=angle, lengths=lens))))
