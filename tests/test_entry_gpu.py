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
    from tinyvc_amd.resample import resample
    gen = infer.load_generator(str(workdir / "encoder.pt"), str(workdir / "decoder.pt"), torch.device("cuda:0"))
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
