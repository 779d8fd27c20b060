#!/usr/bin/env python3
"""SURVEY cfg5-like robustness check on the GPU box: one very long utterance (T frames), stage parity against the live oracle."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tinyvc_amd import synth
import oracle.ref_cpu as R
from helpers import state_dicts, rms, rel_rms

def main(T=15000, N=2000):
    dev = torch.device("cuda", 0)
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd); dec.load_state_dict(dec_sd)
    gen = Generator(enc.to(dev).eval(), dec.to(dev).eval()).to(dev)
    wf = synth.synth_wave(1, 480 * T, seed=42)
    tgt = synth.synth_index(N, seed=1)
    ang = synth.synth_angle(1, T, 3)
    t0 = time.time()
    out = gen.convert(wf.to(dev), tgt.to(dev), 0.0, noise_angle=ang.to(dev))
    torch.cuda.synchronize()
    print(f"T={T}: gpu convert {time.time() - t0:.2f} s (incl. first-call setup), finite={bool(torch.isfinite(out).all())}, rms={rms(out.cpu()):.4f}", flush=True)
    t0 = time.time()
    ref = R.convert(enc_sd, dec_sd, wf, tgt, 0.0, ang)
    print(f"oracle {time.time() - t0:.1f} s; abs rms diff {rms(out.cpu() - ref):.3e} (grows with length: DESIGN.md section 2), ref rms {rms(ref):.4f}")
    # stage parity that does not integrate over time: the spectrogram and the encoder outputs
    from tinyvc_amd.module.utils import spectrogram
    spec = spectrogram(wf.to(dev), 1920, 480).cpu()
    print("spectrogram rel rms", rel_rms(spec, R.spectrogram(wf)))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 15000)
