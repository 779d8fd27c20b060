#!/usr/bin/env python3
"""Capture golden vectors from the reference itself (build container only).

Imports uthree/tinyvc from /root/reference (read-only), fills its modules with the formula-seeded
synthetic checkpoints of `tinyvc_amd.synth`, runs its own `Generator.convert` /
`StreamInfer.audio_callback`, and writes inputs + stage outputs as small .npz fixtures under
tests/golden/.  Only data is written: no reference source, bytecode or pickled reference classes.

Three third-party imports the reference makes at module scope but never calls on the inference
path (torchaudio, torchfcpe, pyworld: module/utils/f0_estimation.py:5-9) are absent from this
image and are registered as empty stub modules before import.

Usage:  python tools/gen_golden.py            (re-run only when synth.py's formulas change)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from tinyvc_amd import synth  # noqa: E402


def import_reference():
    for name in ("torchaudio", "torchaudio.functional", "torchfcpe", "pyworld"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchaudio"].functional = sys.modules["torchaudio.functional"]
    sys.modules["torchaudio.functional"].resample = None
    sys.modules["torchaudio.functional"].gain = None
    sys.modules["torchfcpe"].spawn_bundled_infer_model = None
    sys.path.insert(0, REF)
    import module.tinyvc as rt
    import module.infer as ri
    import module.utils as ru
    sys.path.remove(REF)
    return rt, ri, ru


def np32(t):
    return t.detach().cpu().numpy()


def build_models(rt, seed=0):
    enc = rt.Encoder().eval()
    dec = rt.Decoder().eval()
    enc.load_state_dict(synth.synth_state_dict("encoder", seed))
    dec.load_state_dict(synth.synth_state_dict("decoder", seed))
    return enc, dec


def capture_convert(rt, ri, ru, enc, dec, wf, tgt, pitch_shift, noise_seed, decim):
    """Run the reference stage by stage *and* end to end; return a dict of numpy arrays."""
    gen = ri.Generator(enc, dec)
    B = wf.shape[0]
    with torch.inference_mode():
        wfp = ru.autopad_waveform(wf)
        spec = ru.spectrogram(wfp)
        energy = ru.estimate_energy(wfp)
        ssl, f0 = enc.infer(spec)
        logits = enc.pitch_estimator(spec)
        matched = rt.match_features(ssl, tgt.expand(B, -1, -1))
        # fp64 similarity gaps of the fixture (so a test knows whether index equality is decidable)
        s64 = ssl.double().transpose(1, 2)
        r64 = tgt.double().expand(B, -1, -1).transpose(1, 2)
        sims64 = (s64 / (s64.norm(dim=2, keepdim=True) + 1e-6)) @ (r64 / (r64.norm(dim=2, keepdim=True) + 1e-6)).transpose(1, 2)
        top5 = torch.topk(sims64, 5, dim=2)
        min_gap = float((top5.values[..., :-1] - top5.values[..., 1:]).min())
        s32 = ssl.transpose(1, 2)
        r32 = tgt.expand(B, -1, -1).transpose(1, 2)
        sims32 = torch.bmm(s32 / (torch.norm(s32, dim=2, keepdim=True) + 1e-6),
                           (r32 / (torch.norm(r32, dim=2, keepdim=True) + 1e-6)).transpose(1, 2))
        knn_idx = torch.topk(sims32, 4, dim=2).indices
        assert torch.equal(knn_idx, top5.indices[..., :4]), "fixture has fp32-undecidable kNN ties"
        f0s = ru.shift_frequency(f0, pitch_shift)
        amps, kern = dec.source_net(matched, f0s, energy)
        harm = rt.decoder.oscillate_harmonics(f0s, dec.frame_size, dec.sample_rate, dec.num_harmonics)
        torch.manual_seed(noise_seed)
        angle = torch.rand(B, 961, spec.shape[2]) * 2 * np.pi - np.pi   # the draw decoder.py:78 makes
        assert torch.equal(angle, synth.synth_angle(B, spec.shape[2], noise_seed))
        torch.manual_seed(noise_seed)
        source = dec.dsp(f0s, amps, kern)
        fn = dec.filter_net
        x = fn.content_in(matched) + fn.f0_in(torch.log(torch.relu(f0s) + 1e-6))
        src = torch.cat([source, energy], dim=1)
        skips = []
        for down in fn.downs:
            src = down(src)
            skips.append(src)
        ups = []
        for up, s in zip(fn.ups, reversed(skips)):
            x = up(x, s)
            ups.append(x)
        wave_staged = fn.output_layer(x).squeeze(1)
        torch.manual_seed(noise_seed)
        wave = gen.convert(wf, tgt.expand(B, -1, -1), pitch_shift)
        assert torch.equal(wave, wave_staged)
    d = dict(pitch_shift=np.float32(pitch_shift), noise_seed=np.int64(noise_seed),
             spec=np32(spec), energy=np32(energy), ssl=np32(ssl), f0=np32(f0), logits=np32(logits),
             knn_idx=np32(knn_idx).astype(np.int64), knn_min_gap64=np.float64(min_gap),
             matched=np32(matched), f0s=np32(f0s), amps=np32(amps), kernel=np32(kern),
             harmonics_d=np32(harm[:, :, ::decim]), source_d=np32(source[:, :, ::decim]),
             noise=np32(source[:, 15]), wave=np32(wave), decim=np.int64(decim))
    for i, s in enumerate(skips):
        d[f"skip{i}_d"] = np32(s[:, :, ::(decim if i < 2 else 1)])
    for i, u in enumerate(ups):
        d[f"up{i}_d"] = np32(u[:, :, ::(decim if i >= 3 else 1)])
    return d


def capture_stream(rt, ri, enc, dec, tgt, blocks, noise_seed, use_pv=False):
    gen = ri.Generator(enc, dec)
    st = ri.StreamInfer(gen, target=tgt, pitch_shift=0.0, block_size=1920, extra_size=3840,
                        use_phase_vocoder=use_pv)
    st.init_buffer()
    outs, shifts = [], []
    for i, blk in enumerate(blocks):
        torch.manual_seed(noise_seed + i)
        assert torch.equal(torch.rand(1, 961, st.input_size // 480) * 2 * np.pi - np.pi,
                           synth.synth_angle(1, st.input_size // 480, noise_seed + i))
        torch.manual_seed(noise_seed + i)
        # recover the SOLA shift the callback chose: recompute it from the same buffers
        sola_before = st.sola_buffer.clone()
        out = st.audio_callback(blk.clone())
        outs.append(out.clone())
        # replay of the reference's own shift search on a second convert with the same seed
        torch.manual_seed(noise_seed + i)
        y = gen.convert(st.input_wav[None], tgt, 0.0).squeeze(0)
        tmp = y[-1920 - 1920 - 1920 - 3840:-3840]
        ci = tmp[None, None, :3840]
        nom = torch.nn.functional.conv1d(ci, sola_before[None, None, :])
        den = torch.sqrt(torch.nn.functional.conv1d(ci ** 2, torch.ones(1, 1, 1920)) + 1e-8)
        shifts.append(int(torch.argmax(nom[0, 0] / den[0, 0])))
    return dict(noise_seed=np.int64(noise_seed), out=np32(torch.stack(outs)), shift=np.array(shifts, dtype=np.int64),
                input_size=np.int64(st.input_size))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    rt, ri, ru = import_reference()
    enc, dec = build_models(rt, seed=0)

    spec = {"encoder": {k: list(v.shape) for k, v in enc.state_dict().items()},
            "decoder": {k: list(v.shape) for k, v in dec.state_dict().items()}}
    with open(os.path.join(OUT, "state_dict_spec.json"), "w") as f:
        json.dump(spec, f, indent=0)

    # case A: the streaming-sized buffer (T=28, B=1), ragged input length (13440-37 -> autopad)
    wf = synth.synth_wave(1, 13440 - 37, seed=11)
    tgt = synth.synth_index(1000, seed=2)
    a = capture_convert(rt, ri, ru, enc, dec, wf, tgt, 0.0, noise_seed=3, decim=7)
    a.update(wave_seed=np.int64(11), wave_len=np.int64(13440 - 37), batch=np.int64(1),
             index_seed=np.int64(2), index_size=np.int64(1000), weight_seed=np.int64(0))
    np.savez_compressed(os.path.join(OUT, "convert_T28.npz"), **a)
    print("convert_T28 kNN min fp64 gap", a["knn_min_gap64"], "wave rms", float(np.sqrt((a["wave"] ** 2).mean())))

    # case B: B=2, T=50, pitch shift +3 semitones, 2500-vector index
    wf = synth.synth_wave(2, 24000, seed=21)
    tgt = synth.synth_index(2500, seed=4)
    b = capture_convert(rt, ri, ru, enc, dec, wf, tgt, 3.0, noise_seed=5, decim=23)
    b.update(wave_seed=np.int64(21), wave_len=np.int64(24000), batch=np.int64(2),
             index_seed=np.int64(4), index_size=np.int64(2500), weight_seed=np.int64(0))
    np.savez_compressed(os.path.join(OUT, "convert_B2_T50.npz"), **b)
    print("convert_B2_T50 kNN min fp64 gap", b["knn_min_gap64"], "wave rms", float(np.sqrt((b["wave"] ** 2).mean())))

    # case C: streaming, 6 callbacks of 1920 samples, default SOLA cross-fade, and phase vocoder
    stream_wave = synth.synth_wave(1, 6 * 1920, seed=31)[0]
    blocks = stream_wave.view(6, 1920)
    tgt = synth.synth_index(1000, seed=2)
    c = capture_stream(rt, ri, enc, dec, tgt, blocks, noise_seed=40)
    c.update(wave_seed=np.int64(31), n_blocks=np.int64(6), index_seed=np.int64(2), index_size=np.int64(1000))
    np.savez_compressed(os.path.join(OUT, "stream_6blocks.npz"), **c)
    print("stream shifts", c["shift"])
    c2 = capture_stream(rt, ri, enc, dec, tgt, blocks[:3], noise_seed=40, use_pv=True)
    c2.update(wave_seed=np.int64(31), n_blocks=np.int64(3), index_seed=np.int64(2), index_size=np.int64(1000))
    np.savez_compressed(os.path.join(OUT, "stream_pv_3blocks.npz"), **c2)
    print("stream(pv) shifts", c2["shift"])


if __name__ == "__main__":
    main()
