"""Truth-referenced parity for long utterances.

The GPU path and the reference's fp32 CPU path are two fp32 evaluations of the same function; their mutual difference
(the 1e-4 rms gate of north_star) grows with the utterance because the waveform is a phase integral of f0, and the CPU side
itself moves with its thread count (oneDNN reduction orders).  This file measures BOTH against a fixed truth: the oracle
evaluated in fp64 (`.double()` weights and inputs, same noise phases, same index), whose rounding error is ~1e-13.  The
claim is then independent of the host: the GPU's distance from the truth is not larger than the reference arithmetic's own
(one thread = sequential reductions, the reproducible configuration): f0 within 25 %, the waveform (a random walk of the f0
error) within a factor 2, pooled over several utterances.

The speaker index is degenerate (one vector, repeated), so the discrete kNN stage cannot flip a near-tie in one implementation
and not in the other: the comparison is about arithmetic.
"""
import pytest
import torch

from helpers import oracle_one_thread, rms, state_dicts
from tinyvc_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _log(msg):
    print(msg)
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity.log"), "a") as f:
            f.write(msg + "\n")


@pytest.fixture(scope="module")
def gen():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    return Generator(enc, dec).to(DEV)


NB = 4      # utterances pooled per length (wave seeds 100..103): the waveform error is a random walk of the f0 error over the utterance


@pytest.mark.parametrize("T", [200, 500, 1000])
def test_gpu_is_as_close_to_the_fp64_truth_as_the_reference_arithmetic(gen, T):
    """Two claims, measured on NB utterances of T frames each.
    (1) f0 - the quantity whose error the oscillator integrates into phase - is as accurate on the GPU as in the reference's
        fp32 arithmetic: pooled relative error vs the fp64 truth <= 1.25 x the reference's.  T x NB frames: a tight statistic.
    (2) the waveform: pooled rms(GPU - truth) <= 1.5 x pooled rms(reference fp32 - truth) (measured 0.75 / 0.91 / 1.07).  Per utterance the ratio scatters
        between ~0.4 and ~2 in BOTH directions (it is the end point of a random walk: T = 200 measured 0.44, T = 500 1.9 on single
        utterances with the same f0 accuracy), hence the pooling and the factor; every per-utterance figure is logged."""
    from oracle import ref_cpu as R
    enc_sd, dec_sd = state_dicts(0)
    e64 = {k: v.double() for k, v in enc_sd.items()}
    d64 = {k: v.double() for k, v in dec_sd.items()}
    wf = synth.synth_wave(NB, 480 * T, seed=100)
    angle = synth.synth_angle(NB, T, 3)
    # The speaker index holds ONE vector eight times: whichever four rows a search returns, their mean is that vector, exactly.  The
    # kNN stage is discrete (its indices are tested for equality on gap-checked fixtures elsewhere); a near-tie flipped by either fp32
    # implementation would put a step into the waveform that says nothing about arithmetic.  With NB x T queries no random index is
    # free of near-ties (400 seeds tried at T = 500), so the search is taken out of this comparison instead.
    tgt = synth.synth_index(1, seed=11).expand(1, 768, 8).contiguous()
    truth = R.convert(e64, d64, wf.double(), tgt.double(), 0.0, angle.double(), return_stages=True)
    with oracle_one_thread():
        ref = R.convert(enc_sd, dec_sd, wf, tgt, 0.0, angle, return_stages=True)
    out = gen.convert(wf.to(DEV), tgt.to(DEV), 0.0, noise_angle=angle.to(DEV)).cpu()
    eng = gen.engine(DEV)
    _ssl, f0_gpu, _ = eng.encoder(eng.stft_mag(wf.to(DEV)))
    f0_gpu = f0_gpu.cpu()
    assert torch.isfinite(out).all()
    per = [(rms(out[b] - truth["wave"][b]), rms(ref["wave"][b] - truth["wave"][b])) for b in range(NB)]
    e_gpu, e_ref, mutual = rms(out - truth["wave"]), rms(ref["wave"] - truth["wave"]), rms(out - ref["wave"])
    f_gpu = rms(f0_gpu - truth["f0"]) / rms(truth["f0"])
    f_ref = rms(ref["f0"] - truth["f0"]) / rms(truth["f0"])
    _log(f"[truth] T={T} x {NB} utterances : f0 rel vs fp64 truth: GPU {f_gpu:.2e}, reference fp32 (1 thread) {f_ref:.2e} (ratio {f_gpu / f_ref:.2f}); "
         f"waveform rms vs truth: GPU {e_gpu:.3e}, reference {e_ref:.3e} (ratio {e_gpu / e_ref:.2f}); GPU vs reference {mutual:.3e}; "
         "per utterance GPU/reference: " + ", ".join(f"{a:.2e}/{b:.2e}" for a, b in per))
    assert f_gpu <= 1.25 * f_ref, f"T={T}: f0 is {f_gpu:.2e} from the fp64 truth on the GPU, {f_ref:.2e} in the reference's fp32 arithmetic"
    assert e_gpu <= 1.5 * e_ref, f"T={T}: GPU waveform is {e_gpu:.3e} from the fp64 truth, the reference's own fp32 arithmetic {e_ref:.3e}"
    # the triangle inequality bounds the mutual difference: nothing else (a flipped neighbour, a phase slip) is hiding in it
    assert mutual <= e_gpu + e_ref + 1e-7
