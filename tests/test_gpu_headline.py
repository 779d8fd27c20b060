"""Parity at the headline sizes (MI355X): BASELINE.json configs[0] exactly (one 4 s utterance, 1 000-vector index,
T = 200 frames) and a 4-utterance slice of configs[1] (10 000-vector index), against vectors captured from the reference
(tools/gen_golden.py); FilterNet's Downsample / Upsample blocks compared one by one; and the end-to-end error budget
split into its sources.  Everything goes through the C ABI."""
import numpy as np
import pytest
import torch

from helpers import convert_inputs, load_golden, rel_rms, rms, stage, state_dicts
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
def models():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    return enc, dec, Generator(enc, dec).to(DEV)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def close(name, got, g, key, tol):
    """relative rms of `got` (strided like the stored vector) against golden stage `key`"""
    ref, st = stage(g, key)
    got = got.detach().float().cpu()[..., ::st]
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite"
    rr = rel_rms(got, ref)
    _log(f"[headline] {name:40s} rel_rms={rr:.3e} max_abs={float((got - ref).abs().max()):.3e} (tol {tol:.0e})")
    assert rr <= tol, f"{name}: rel rms {rr:.3e} > {tol:.1e}"
    return rr


def _gpu_chain(gen, wf, tgt, shift, angle):
    """The staged path on the GPU with every intermediate kept (what tvc_convert_f32 runs as one call)."""
    from tinyvc_amd.module import utils
    from tinyvc_amd.module.tinyvc import match_features
    eng = gen.engine(DEV)
    w = utils.autopad_waveform(wf.to(DEV))
    spec = utils.spectrogram(w)
    energy = utils.estimate_energy(w)
    ssl, f0, logits = eng.encoder(spec, want_logits=True)
    matched, idx = match_features(ssl, tgt.to(DEV), return_indices=True)
    f0s = eng.shift_frequency(f0, shift)
    wave = eng.decoder(matched, f0s, energy, angle.to(DEV))
    return dict(spec=spec, energy=energy, ssl=ssl, f0=f0, logits=logits, matched=matched, knn_idx=idx, f0s=f0s, wave=wave)


@pytest.mark.parametrize("case", ["convert_cfg1_T200", "convert_cfg2_B4_T200"])
def test_headline_end_to_end_chain(models, case):
    """wav -> wave through every GPU stage, each compared with the reference's own vector, indices bit-exact, and the
    one-call tvc_convert_f32 equal to the staged chain."""
    _enc, _dec, gen = models
    g = load_golden(case)
    wf, tgt, shift, angle = convert_inputs(g)
    st = _gpu_chain(gen, wf, tgt, shift, angle)
    close(f"{case} spectrogram", st["spec"], g, "spec", 1.5e-6)
    close(f"{case} ssl (gpu spec)", st["ssl"], g, "ssl", 1e-5)
    close(f"{case} logits (gpu spec)", st["logits"], g, "logits", 6e-6)
    close(f"{case} f0 (gpu spec)", st["f0"], g, "f0", 1e-5)
    assert torch.equal(st["knn_idx"].cpu(), _t(g["knn_idx"])), "kNN indices of the GPU chain differ from the reference's (gap-checked index)"
    close(f"{case} matched (gpu chain)", st["matched"], g, "matched", 1e-7)
    ref = _t(g["wave"])
    d = rms(st["wave"].cpu() - ref)
    _log(f"[headline] {case}: END-TO-END abs rms diff {d:.3e} (gate 1e-4; wave rms {rms(ref):.3e})")
    assert d <= 1e-4, f"end-to-end rms difference {d:.3e} > 1e-4 at the headline length"
    one = gen.convert(wf.to(DEV), tgt.to(DEV), shift, noise_angle=angle.to(DEV))
    assert torch.equal(one, st["wave"]), "tvc_convert_f32 != the staged chain"


def test_headline_stages_on_reference_inputs():
    """cfg1 fixture: every GPU stage fed with the REFERENCE's input for it, so each number is that stage's own error."""
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder, match_features
    from tinyvc_amd.module import utils
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    gen = Generator(enc, dec).to(DEV)
    eng = gen.engine(DEV)
    g = load_golden("convert_cfg1_T200")
    wf, tgt, shift, angle = convert_inputs(g)
    spec = _t(g["spec"]).to(DEV)
    ssl, f0, logits = eng.encoder(spec, want_logits=True)
    close("cfg1 ssl (ref spec)", ssl, g, "ssl", 1e-5)
    close("cfg1 logits (ref spec)", logits, g, "logits", 3e-6)
    r = close("cfg1 f0 (ref spec)", f0, g, "f0", 2e-6)          # VERDICT r1 gate: f0 rel <= 2e-6
    # PitchEstimator.decode alone, on the reference's logits: top-4 / softmax / expectation with the transcendentals rounded
    # the way ATen's are -> bit-identical on nearly every frame
    f0d = enc.pitch_estimator.decode(_t(g["logits"]).to(DEV))
    same = float((f0d.cpu() == _t(g["f0"])).float().mean())
    _log(f"[headline] cfg1 decode(reference logits): {same * 100:.1f} % of frames bit-identical")
    close("cfg1 decode (ref logits)", f0d, g, "f0", 5e-8)
    assert same >= 0.9
    f0s = eng.shift_frequency(_t(g["f0"]).to(DEV), shift)
    same = float((f0s.cpu() == _t(g["f0s"])).float().mean())
    _log(f"[headline] cfg1 shift_frequency(reference f0): {same * 100:.1f} % of frames bit-identical")
    close("cfg1 shift_frequency (ref f0)", f0s, g, "f0s", 5e-8)
    assert same >= 0.9
    m, idx = match_features(_t(g["ssl"]).to(DEV), tgt.to(DEV), return_indices=True)
    assert torch.equal(idx.cpu(), _t(g["knn_idx"]))
    assert torch.equal(m.cpu(), _t(g["matched"])), "matched rows must be bit-identical once the indices are"
    energy = utils.estimate_energy(utils.autopad_waveform(wf.to(DEV)))      # bit-exact stage (test_front_end)
    content, f0s = _t(g["matched"]).to(DEV), _t(g["f0s"]).to(DEV)
    wave, amps, kern, source = eng.decoder(content, f0s, energy, angle.to(DEV), stages=True)
    close("cfg1 amps (ref in)", amps, g, "amps", 2e-6)
    close("cfg1 kernel (ref in)", kern, g, "kernel", 2e-6)
    close("cfg1 source (ref in)", source, g, "source", 2e-6)
    d = rms(wave.cpu() - _t(g["wave"]))
    _log(f"[headline] cfg1 decoder on reference inputs: abs rms diff {d:.3e} (gate 1e-5)")
    assert d <= 1e-5
    _log(f"[headline] cfg1 f0 rel error on the reference's spectrogram {r:.3e}")


@pytest.mark.parametrize("case", ["convert_T28", "convert_B2_T50", "convert_cfg1_T200"])
def test_filter_net_blocks(models, case):
    """FilterNet's five Downsample outputs and four Upsample outputs (decoder.py:227-232), one by one, against the
    reference's.  Inputs: the reference's matched / f0s, and `source` from the GPU DSP on them (1.6e-7 from the
    reference's).  A 1e-5 relative perturbation of any block fails its gate."""
    _enc, dec, gen = models
    from tinyvc_amd.module import utils
    g = load_golden(case)
    wf, _tgt, _shift, angle = convert_inputs(g)
    eng = dec.engine(DEV)
    if "matched" in g:
        content = _t(g["matched"]).to(DEV)
    else:
        pytest.skip("fixture stores no full matched tensor")
    f0s = _t(g["f0s"]).to(DEV)
    energy = utils.estimate_energy(utils.autopad_waveform(wf.to(DEV)))
    _w, _a, _k, source = eng.decoder(content, f0s, energy, angle.to(DEV), stages=True)
    wave, skips, ups = eng.filter_net(content, f0s, energy, source, blocks=True)
    assert torch.equal(wave, _w), "tvc_filter_net_f32 must reproduce tvc_decoder_f32's waveform bit for bit"
    for i, s in enumerate(skips):
        close(f"{case} downs[{i}] output", s, g, f"skip{i}", 3e-6)
    for i, u in enumerate(ups):
        close(f"{case} ups[{i}] output", u, g, f"up{i}", 3e-6)
    # sensitivity of the gate itself: a 1e-5 relative bump of one block output must trip it
    bumped = skips[2] * (1 + 1e-5)
    ref, st = stage(g, "skip2")
    assert rel_rms(bumped.cpu()[..., ::st], ref) > 3e-6
    # the module-level mirror of FilterNet.forward
    w2 = dec.filter_net(content, f0s, energy, source)
    assert w2.shape == (content.shape[0], 1, wave.shape[1]) and torch.equal(w2[:, 0], wave)


def test_error_budget_at_4s(models):
    """Where the end-to-end difference at T = 200 comes from.  The waveform is a phase integral of f0 = softmax-weighted
    class frequencies, so fp32-level differences upstream of f0 dominate; everything downstream of f0 agrees to 1e-7."""
    _enc, _dec, gen = models
    eng = gen.engine(DEV)
    from tinyvc_amd.module import utils
    g = load_golden("convert_cfg1_T200")
    wf, tgt, shift, angle = convert_inputs(g)
    ref = _t(g["wave"])
    energy = utils.estimate_energy(utils.autopad_waveform(wf.to(DEV)))
    content = _t(g["matched"]).to(DEV)
    a = angle.to(DEV)

    def dec_with_f0(f0):
        return rms(eng.decoder(content, eng.shift_frequency(f0, shift), energy, a).cpu() - ref)

    _s, f0_ref_spec, _l = eng.encoder(_t(g["spec"]).to(DEV))
    _s, f0_gpu_spec, _l = eng.encoder(utils.spectrogram(utils.autopad_waveform(wf.to(DEV))))
    full = rms(gen.convert(wf.to(DEV), tgt.to(DEV), shift, noise_angle=a).cpu() - ref)
    rows = [("decoder only (reference matched, reference f0)", dec_with_f0(_t(g["f0"]).to(DEV))),
            ("+ GPU pitch trunk on the reference spectrogram", dec_with_f0(f0_ref_spec)),
            ("+ GPU |STFT| (= GPU f0, reference matched)", dec_with_f0(f0_gpu_spec)),
            ("whole GPU path", full)]
    for name, v in rows:
        _log(f"[budget] T=200  {name:50s} abs rms diff {v:.3e}")
    _log(f"[budget] f0 rel error: pitch trunk alone {rel_rms(f0_ref_spec.cpu(), _t(g['f0'])):.3e}, with GPU |STFT| {rel_rms(f0_gpu_spec.cpu(), _t(g['f0'])):.3e}")
    assert rows[0][1] <= 1e-5 and full <= 1e-4


def test_nan_and_inf_samples_do_not_fault(models):
    """A float WAV with NaN / Inf samples: the reference propagates NaN through that utterance (torch.topk orders NaN
    first and gathers real rows); here the affected utterance must come back NaN / finite garbage WITHOUT an out-of-bounds
    gather, and the other utterances of the batch must be untouched."""
    _enc, _dec, gen = models
    wf = synth.synth_wave(3, 9600, seed=400).to(DEV)
    tgt = synth.synth_index(300, seed=2).to(DEV)
    angle = synth.synth_angle(3, 20, 9).to(DEV)
    clean = gen.convert(wf, tgt, 0.0, noise_angle=angle)
    bad = wf.clone()
    bad[1, 1000] = float("nan")
    bad[1, 5000] = float("inf")
    out = gen.convert(bad, tgt, 0.0, noise_angle=angle)
    torch.cuda.synchronize()
    assert torch.equal(out[0], clean[0]) and torch.equal(out[2], clean[2])
    assert not torch.isfinite(out[1]).all()
    # kNN alone on an all-NaN query column: indices stay inside the index (torch.topk: NaN first; here rows 0..3)
    from tinyvc_amd.module.tinyvc import match_features
    q = torch.randn(1, 768, 5, generator=torch.Generator().manual_seed(1))
    q[0, :, 2] = float("nan")
    m, idx = match_features(q.to(DEV), tgt, return_indices=True)
    torch.cuda.synchronize()
    assert int(idx.min()) >= 0 and int(idx.max()) < 300
    assert idx[0, 2].tolist() == [0, 1, 2, 3] and torch.isfinite(m[0, :, 2]).all()
    # the pitch decoder on NaN logits
    eng = gen.engine(DEV)
    spec = torch.full((1, 961, 4), float("nan"), device=DEV)
    _ssl, f0, _ = eng.encoder(spec)
    torch.cuda.synchronize()
    assert torch.isnan(f0).all()


def test_ten_seconds_against_the_host_cpu_spread(models):
    """T = 500 (10 s), oracle run live on this box: the GPU-vs-CPU waveform difference next to the CPU's own
    1-thread-vs-all-threads difference on the same input (tools/cpu_spread.py prints the same figures)."""
    import os
    from oracle import ref_cpu as R
    _enc, _dec, gen = models
    enc_sd, dec_sd = state_dicts(0)
    T = 500
    wf = synth.synth_wave(1, 480 * T, seed=100)
    tgt = synth.synth_index(1000, seed=2)
    angle = synth.synth_angle(1, T, 3)
    n = torch.get_num_threads()
    ref = R.convert(enc_sd, dec_sd, wf, tgt, 0.0, angle, return_stages=True)
    torch.set_num_threads(1)
    one = R.convert(enc_sd, dec_sd, wf, tgt, 0.0, angle)
    torch.set_num_threads(n)
    spread = rms(one - ref["wave"])
    out = gen.convert(wf.to(DEV), tgt.to(DEV), 0.0, noise_angle=angle.to(DEV))
    d = rms(out.cpu() - ref["wave"])
    eng = gen.engine(DEV)
    dec_only = rms(eng.decoder(ref["matched"].to(DEV), ref["f0s"].to(DEV), ref["energy"].to(DEV), angle.to(DEV)).cpu() - ref["wave"])
    _log(f"[headline] T=500: GPU vs CPU({n} threads) {d:.3e}; CPU 1 thread vs {n} threads {spread:.3e}; ratio {d / spread:.2f}; "
         f"decoder on the oracle's inputs {dec_only:.3e}")
    assert dec_only <= 1e-6
    # end to end: logged only - the claim that does not depend on this host's thread count is test_gpu_truth.py (GPU and reference
    # arithmetic both measured against the fp64 evaluation of the path); here only that nothing discrete went wrong
    assert d <= 3e-4       # measured 9.8e-5 (the host's own 1-vs-128-thread spread: 1.7e-4)
