#!/usr/bin/env python3
"""Extra parity sweep on the GPU box: whole conversions at shapes the test-suite does not hold, against the live oracle."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tinyvc_amd import synth
import oracle.ref_cpu as R
from helpers import state_dicts  # noqa

def rms(x):
    return float(torch.sqrt(torch.mean(x.double() ** 2)))

def main():
    dev = torch.device("cuda", 0)
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    gen = Generator(enc.to(dev).eval(), dec.to(dev).eval()).to(dev)
    worst = 0.0
    for (B, T, N, shift) in [(1, 500, 3000, 2.0), (3, 64, 257, -5.0), (2, 128, 129, 0.0), (1, 8, 64, 7.0), (5, 33, 1000, 0.0)]:
        wf = synth.synth_wave(B, 480 * T, seed=300 + T)
        tgt = synth.synth_index(N, seed=N)
        ang = synth.synth_angle(B, T, 11)
        ref = R.convert(enc_sd, dec_sd, wf, tgt, shift, ang)
        out = gen.convert(wf.to(dev), tgt.to(dev), shift, noise_angle=ang.to(dev)).cpu()
        d = rms(out - ref)
        worst = max(worst, d)
        print(f"B={B} T={T} N={N} shift={shift}: abs rms diff {d:.3e} (wave rms {rms(ref):.3e})", flush=True)
    print("worst", worst)   # grows with utterance length: the oscillator integrates f0

if __name__ == "__main__":
    main()
