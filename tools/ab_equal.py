"""Bit-for-bit comparison of two library builds (run on the GPU box):
    TVC_LIB_PATH=$PWD/libA.so python tools/ab_equal.py dump /tmp/a.npz
    TVC_LIB_PATH=$PWD/libB.so python tools/ab_equal.py dump /tmp/b.npz
    python tools/ab_equal.py cmp /tmp/a.npz /tmp/b.npz
Dumps encoder (ssl, logits, f0), decoder and whole-path outputs at the shapes that pick different kernels / tilings: the bench batch,
one utterance, a streaming block (32 x 28 frames), odd lengths, a 2000-frame utterance and a ragged batch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dump(path):
    from helpers import state_dicts
    from tinyvc_amd import synth
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder

    dev = "cuda:0"
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    gen = Generator(enc, dec).to(dev)
    enc, dec = gen.encoder, gen.decoder
    out = {}
    tgt = synth.synth_index(1000, seed=8).to(dev)
    for B, T in [(64, 200), (1, 200), (32, 28), (3, 65), (2, 33), (5, 1), (1, 2000), (7, 129)]:
        g = torch.Generator().manual_seed(1000 * B + T)
        spec = (torch.rand(B, 961, T, generator=g) * 3.0).to(dev)
        ssl, logits = enc.forward(spec)
        _, f0 = enc.infer(spec)
        tag = "B%d_T%d" % (B, T)
        if B * T <= 4000:
            out["ssl_" + tag], out["logits_" + tag] = ssl.cpu().numpy(), logits.cpu().numpy()
        else:
            out["ssl_" + tag], out["logits_" + tag] = ssl[::7, :, ::3].cpu().numpy(), logits[::7, :, ::3].cpu().numpy()
        out["f0_" + tag] = f0.cpu().numpy()
    for B, T in [(64, 200), (1, 200), (32, 28), (2, 33)]:
        wf = synth.synth_wave(B, 480 * T, seed=7 + B).to(dev)
        angle = synth.synth_angle(B, T, 31).to(dev)
        y = gen.convert(wf, tgt, 1.0, noise_angle=angle)
        out["wave_B%d_T%d" % (B, T)] = y[:: max(1, B // 8)].cpu().numpy()
    torch.manual_seed(4321)                      # the library's own phase draw (noise_angle = None): same seed, same hash, same samples
    out["wave_default_draw"] = gen.convert(synth.synth_wave(3, 480 * 40, seed=9).to(dev), tgt, 0.0).cpu().numpy()
    frames = [33, 7, 50, 200, 3, 129, 21, 64, 12, 250, 65]
    lens = [480 * f - (17 if i % 2 else 0) for i, f in enumerate(frames)]
    Bn, Tmax = len(frames), max(frames)
    wf = torch.zeros(Bn, 480 * Tmax)
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=500 + b)[0]
    angle = synth.synth_angle(Bn, Tmax, 31).to(dev)
    y = gen.convert(wf.to(dev), tgt, -1.5, noise_angle=angle, lengths=lens)
    out["wave_ragged"] = y.cpu().numpy()
    torch.manual_seed(99)
    out["wave_ragged_default_draw"] = gen.convert(wf.to(dev), tgt, -1.5, lengths=lens).cpu().numpy()
    np.savez(path, **out)
    print("dumped", len(out), "tensors to", path)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = 0
        for k in a.files:
            same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True)
            if not same:
                bad += 1
                d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
                print("DIFF ", k, a[k].shape, "max", d.max(), "count", int((d > 0).sum()), "nan", int(np.isnan(b[k]).sum()))
            else:
                print("equal", k, a[k].shape)
        print("ALL EQUAL" if not bad else "%d tensors differ" % bad)
        sys.exit(1 if bad else 0)
