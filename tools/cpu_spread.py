#!/usr/bin/env python3
"""How reproducible is the reference's own CPU path?  (test infrastructure; uses the oracle)

The converted waveform is a phase integral of f0, and f0 is a softmax-weighted mean of class frequencies, so any two
fp32 evaluations of the pitch estimator - the SAME ATen code on 1 thread vs N threads included, because MKL / oneDNN
split their reductions differently - drift apart with utterance length.  This tool measures that floor on the host it
runs on, and the sensitivity of the waveform to the two upstream stages nobody can reproduce bit for bit (the FFT
library's rounding, the GEMM summation order):

  python tools/cpu_spread.py [--frames 28 200 500] [--threads N]

Output (one line per length): rms(wave_1thread - wave_Nthreads), the f0 relative difference behind it, the waveform
change caused by a 1.5e-7 relative perturbation of |STFT| (the size of the difference between two correct fp32 FFTs),
and the waveform change when the pitch trunk is evaluated in fp64 (the oracle's own rounding error).
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cpu as R  # noqa: E402
from tinyvc_amd import synth  # noqa: E402

warnings.filterwarnings("ignore")


def rms(a):
    return float(a.double().pow(2).mean().sqrt())


def rel(a, b):
    return rms(a.double() - b.double()) / rms(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[28, 200, 500])
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--index", type=int, default=1000)
    args = ap.parse_args()
    enc_sd, dec_sd = synth.synth_state_dict("encoder"), synth.synth_state_dict("decoder")
    enc64 = {k: v.double() for k, v in enc_sd.items()}
    print(f"host threads {args.threads}; torch {torch.__version__}")
    for T in args.frames:
        wf = synth.synth_wave(1, 480 * T, seed=100)
        tgt = synth.synth_index(args.index, seed=2)
        angle = synth.synth_angle(1, T, 3)
        torch.set_num_threads(args.threads)
        a = R.convert(enc_sd, dec_sd, wf, tgt, 0.0, angle, return_stages=True)
        torch.set_num_threads(1)
        b = R.convert(enc_sd, dec_sd, wf, tgt, 0.0, angle, return_stages=True)
        torch.set_num_threads(args.threads)
        g = torch.Generator().manual_seed(0)
        specp = a["spec"] * (1 + 1.5e-7 * torch.randn(a["spec"].shape, generator=g))
        f0p = R.pitch_decode(R.pitch_logits(enc_sd, specp))
        wp = R.decoder_infer(dec_sd, a["matched"], R.shift_frequency(f0p, 0.0), a["energy"], angle)
        f0x = R.pitch_decode(R.pitch_logits(enc64, a["spec"].double()).float())
        wx = R.decoder_infer(dec_sd, a["matched"], R.shift_frequency(f0x, 0.0), a["energy"], angle)
        print(f"T={T:5d} ({T / 50:g} s): 1 vs {args.threads} threads wave rms diff {rms(a['wave'] - b['wave']):.3e} "
              f"(f0 rel {rel(b['f0'], a['f0']):.2e}, same kNN rows: {torch.equal(a['matched'], b['matched'])}) | "
              f"|STFT|*(1+1.5e-7 n): {rms(wp - a['wave']):.3e} (f0 rel {rel(f0p, a['f0']):.2e}) | "
              f"fp64 pitch trunk: {rms(wx - a['wave']):.3e} (f0 rel {rel(f0x, a['f0']):.2e})")


if __name__ == "__main__":
    main()
