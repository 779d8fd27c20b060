"""The entry scripts on a GPU: infer.py, infer_streaming.py (file-driven), extract_index.py with
formula-seeded checkpoints written to a temp dir in the reference's on-disk formats."""
import os

import numpy as np
import pytest
import torch

from tinyvc_amd import audio_io, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("tvc")
    torch.save(synth.synth_state_dict("encoder"), d / "encoder.pt")          # same format as infer.py:34-37 expects
    torch.save(synth.synth_state_dict("decoder"), d / "decoder.pt")
    torch.save(synth.synth_index(300, seed=2), d / "index.pt")               # extract_index.py:58 format
    (d / "inputs").mkdir()
    wave16k = synth.synth_wave(1, 24000, seed=3)[0][::3][:8000] * 0.9       # a 0.5 s "16 kHz" file
    audio_io.save(str(d / "inputs" / "a.wav"), wave16k[None], 16000)
    audio_io.save(str(d / "inputs" / "b.wav"), wave16k.flip(0)[None], 16000)
    audio_io.save(str(d / "target.wav"), synth.synth_wave(1, 24000, seed=9), 24000)
    return d


def test_infer_py_with_index(workdir):
    import infer
    torch.manual_seed(0)
    rc = infer.main(["-i", str(workdir / "inputs"), "-o", str(workdir / "out"), "-encp", str(workdir / "encoder.pt"),
                     "-decp", str(workdir / "decoder.pt"), "-idx", str(workdir / "index.pt"), "-p", "2.0", "-d", "cuda:0"])
    assert rc == 0
    outs = {}
    for name in ("a", "b"):
        y, sr = audio_io.load(str(workdir / "out" / f"{name}.wav"))
        assert sr == 24000 and y.shape == (1, 12000) and torch.isfinite(y).all() and float(y.abs().max()) > 1e-3
        outs[name] = y
    # same computation through the module API, same seed -> identical samples
    gen = infer.load_generator(str(workdir / "encoder.pt"), str(workdir / "decoder.pt"), torch.device("cuda:0"))
    resample = lambda w, a, b: gen.engine("cuda:0").resample(w.to("cuda:0"), a, b)      # the device resampler infer.py uses
    tgt = torch.load(workdir / "index.pt").to("cuda:0")
    wa, sr = audio_io.load(str(workdir / "inputs" / "a.wav"))
    wb, _ = audio_io.load(str(workdir / "inputs" / "b.wav"))
    batch = torch.cat([resample(wa, sr, 24000), resample(wb, sr, 24000)], 0).to("cuda:0")
    torch.manual_seed(0)
    ref = gen.convert(batch, tgt, 2.0).cpu()
    assert torch.equal(ref[0:1], outs["a"]) and torch.equal(ref[1:2], outs["b"])


def test_infer_py_with_target_wav(workdir):
    import infer
    rc = infer.main(["-i", str(workdir / "inputs"), "-o", str(workdir / "out2"), "-encp", str(workdir / "encoder.pt"),
                     "-decp", str(workdir / "decoder.pt"), "-t", str(workdir / "target.wav"), "-d", "cuda:0"])
    assert rc == 0 and os.path.exists(workdir / "out2" / "a.wav")


def test_extract_index(workdir):
    import extract_index
    out = workdir / "built_index.pt"
    rc = extract_index.main(["--dataset-cache", str(workdir / "inputs"), "-encp", str(workdir / "encoder.pt"), "-size", "10",
                             "-o", str(out), "-d", "cuda:0", "--seed", "1"])
    assert rc == 0
    idx = torch.load(out)
    assert idx.shape == (1, 768, 10) and idx.dtype == torch.float32 and torch.isfinite(idx).all()


def test_infer_streaming_file_driven(workdir, capsys):
    import infer_streaming
    audio_io.save(str(workdir / "mic.wav"), synth.synth_wave(1, 24000, seed=5), 24000)
    rc = infer_streaming.main(["-encp", str(workdir / "encoder.pt"), "-decp", str(workdir / "decoder.pt"), "-idx", str(workdir / "index.pt"),
                               "--input-wav", str(workdir / "mic.wav"), "--output-wav", str(workdir / "conv.wav"), "--streams", "4", "-d", "cuda:0"])
    assert rc == 0
    y, sr = audio_io.load(str(workdir / "conv.wav"))
    assert sr == 24000 and y.shape == (1, 12 * 1920) and torch.isfinite(y).all()
    assert "p50" in capsys.readouterr().out


def test_extract_index_reproduces_the_oracle_features_in_order(workdir):
    """index.pt = encoder features of the clips, every 4th frame, concatenated, randomly permuted,
    truncated (reference extract_index.py:43-58).  With --seed the build's file order and permutation are reproducible,
    so the oracle's `encode` on the same clips must give the same columns in the same order."""
    import extract_index
    from oracle import ref_cpu as R
    d = workdir / "clips"
    d.mkdir()
    lens = (24000, 31200 - 77, 19200 + 5)                      # ragged: autopad before the STFT, as the script does
    for i, n in enumerate(lens):
        audio_io.save(str(d / f"{i}.wav"), synth.synth_wave(1, n, seed=40 + i), 24000)      # float32 WAV: read back bit for bit
    out = workdir / "idx_f1.pt"
    size = 30
    rc = extract_index.main(["--dataset-cache", str(d), "-encp", str(workdir / "encoder.pt"), "-size", str(size), "-o", str(out),
                             "-d", "cuda:0", "--seed", "7"])
    assert rc == 0
    got = torch.load(out)
    assert got.shape == (1, 768, size) and got.dtype == torch.float32
    # the same recipe with the oracle's encoder
    enc_sd = synth.synth_state_dict("encoder")
    files = sorted(str(p) for p in d.glob("*.wav"))
    gen = torch.Generator().manual_seed(7)
    order = torch.randperm(len(files), generator=gen).tolist()
    feats, total = [], 0
    for i in order:
        wf, sr = audio_io.load(files[i])
        assert sr == 24000
        z, _f0 = R.encode(enc_sd, wf)
        z = z[:, :, ::4]
        feats.append(z)
        total += z.shape[2]
        if total > size:
            break
    feats = torch.cat(feats, dim=2)
    want = feats.index_select(2, torch.randperm(feats.shape[2], generator=gen))[:, :, :size]
    err = (got.double() - want.double()).pow(2).sum(dim=1).sqrt() / want.double().pow(2).sum(dim=1).sqrt()      # per column
    print(f"[f1] extract_index vs oracle encode: worst column rel error {float(err.max()):.2e}")
    assert float(err.max()) < 1e-5, "a column differs: wrong order, stride or permutation"
    # --half stores the same vectors in fp16, which match_features takes as the fp16 index storage
    rc = extract_index.main(["--dataset-cache", str(d), "-encp", str(workdir / "encoder.pt"), "-size", str(size), "-o", str(workdir / "idx_h.pt"),
                             "-d", "cuda:0", "--seed", "7", "--half"])
    assert rc == 0
    h = torch.load(workdir / "idx_h.pt")
    assert h.dtype == torch.float16 and torch.equal(h, got.half())


@pytest.mark.parametrize("orig,new", [(16000, 24000), (44100, 24000), (48000, 24000), (22050, 24000), (24000, 16000)])
def test_device_resampler_matches_the_host_restatement(orig, new):
    """tvc_resample_f32 (the torchaudio sinc_interp_hann algorithm on the GPU) against tinyvc_amd/resample.py
    on the host.  (torchaudio itself is absent: parity of both with it is unpinned)"""
    from tinyvc_amd.engine import default_engine
    from tinyvc_amd.resample import resample
    eng = default_engine(torch.device("cuda:0"))
    g = torch.Generator().manual_seed(orig)
    x = torch.randn(2, 3, 12345, generator=g) * 0.1
    want = resample(x, orig, new)
    got = eng.resample(x.to("cuda:0"), orig, new).cpu()
    assert got.shape == want.shape
    rel = float((got.double() - want.double()).pow(2).mean().sqrt() / want.double().pow(2).mean().sqrt())
    print(f"[f3] resample {orig} -> {new}: rel rms vs host {rel:.2e}, out len {got.shape[-1]}")
    assert rel < 1e-6
    assert eng.resample(x.to("cuda:0"), new, new).shape == x.shape


@pytest.mark.parametrize("gain_db", [0.0, -6.0, 3.5])
def test_pcm16_conversions_match_the_reference_loop_arithmetic(gain_db):
    """infer_streaming.py:85-94 op for op: int16 -> /32768 -> gain | gain -> *32768 -> astype(int16), bit-exact."""
    from tinyvc_amd.engine import default_engine
    eng = default_engine(torch.device("cuda:0"))
    rng = np.random.default_rng(3)
    pcm = rng.integers(-32768, 32768, size=5000, dtype=np.int16)
    pcm[:4] = [-32768, 32767, 0, -1]
    ratio = np.float32(10 ** (gain_db / 20))
    x = pcm.astype(np.float32) / np.float32(32768)
    if gain_db != 0:
        x = x * ratio
    got = eng.pcm16_to_f32(torch.from_numpy(pcm).to("cuda:0"), gain_db).cpu().numpy()
    assert np.array_equal(got, x.astype(np.float32))
    y = rng.standard_normal(5000).astype(np.float32) * 0.5
    y[:3] = [0.99999, -1.0, 1.7]            # 1.7 * 32768 is outside int16: numpy wraps through int32
    z = y * ratio if gain_db != 0 else y
    want = (z * np.float32(32768)).astype(np.int32).astype(np.int16)
    got = eng.f32_to_pcm16(torch.from_numpy(y).to("cuda:0"), gain_db).cpu().numpy()
    assert np.array_equal(got, want)


def test_infer_py_chunked_mode(workdir):
    """--chunked makes --chunk-size / --buffer-size real (block-wise conversion through the streaming
    converter, bounded memory); --no-chunking True overrides it and gives the whole-file result bit for bit."""
    import infer
    d = workdir / "long_in"
    d.mkdir()
    audio_io.save(str(d / "c.wav"), synth.synth_wave(1, 24000 * 3, seed=77), 24000)
    common = ["-i", str(d), "-encp", str(workdir / "encoder.pt"), "-decp", str(workdir / "decoder.pt"), "-idx", str(workdir / "index.pt"), "-d", "cuda:0"]
    torch.manual_seed(0)
    assert infer.main(common + ["-o", str(workdir / "o_whole")]) == 0
    torch.manual_seed(0)
    assert infer.main(common + ["-o", str(workdir / "o_nc"), "--chunked", "-nc", "True"]) == 0
    torch.manual_seed(0)
    assert infer.main(common + ["-o", str(workdir / "o_chunk"), "--chunked", "-c", "1920", "-b", "4"]) == 0
    whole, _ = audio_io.load(str(workdir / "o_whole" / "c.wav"))
    nc, _ = audio_io.load(str(workdir / "o_nc" / "c.wav"))
    ch, _ = audio_io.load(str(workdir / "o_chunk" / "c.wav"))
    assert torch.equal(whole, nc)
    assert ch.shape == whole.shape == (1, 72000) and torch.isfinite(ch).all()
    r_w, r_c = float(whole[:, 12000:60000].pow(2).mean().sqrt()), float(ch[:, 12000:60000].pow(2).mean().sqrt())
    print(f"[f4] whole-file rms {r_w:.4f}, chunked rms {r_c:.4f}")
    assert abs(r_w - r_c) / r_w < 0.25
