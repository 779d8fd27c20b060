"""Host-side front door: WAV I/O, resampler, CLI flag surfaces, import shim (no GPU needed)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tinyvc_amd import audio_io
from tinyvc_amd.resample import gain, resample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wav_roundtrip(tmp_path):
    x = torch.randn(1, 1000).clamp(-1, 1) * 0.5
    p = str(tmp_path / "a.wav")
    audio_io.save(p, x, 24000)
    y, sr = audio_io.load(p)
    assert sr == 24000 and torch.equal(x, y)
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "b.wav"), 16000, (x[0].numpy() * 32767).astype(np.int16))
    z, sr = audio_io.load(str(tmp_path / "b.wav"))
    assert sr == 16000 and z.shape == (1, 1000) and float((z - x).abs().max()) < 1e-4
    with pytest.raises(ValueError):
        audio_io.load(str(tmp_path / "c.mp3"))


@pytest.mark.parametrize("orig,new", [(16000, 24000), (48000, 24000), (44100, 24000), (24000, 24000)])
def test_resample_matches_polyphase_reference(orig, new):
    from scipy.signal import resample_poly
    t = np.arange(orig) / orig                      # 1 s
    x = 0.5 * np.sin(2 * np.pi * 440 * t) + 0.25 * np.sin(2 * np.pi * 3000 * t + 1.0)
    y = resample(torch.from_numpy(x).float()[None], orig, new)[0].numpy()
    assert len(y) == math.ceil(len(x) * new / orig)
    g = math.gcd(orig, new)
    ref = resample_poly(x, new // g, orig // g)
    n = min(len(y), len(ref))
    core = slice(n // 10, n - n // 10)              # ignore the edge transients of the two filter designs
    err = np.sqrt(np.mean((y[core] - ref[core]) ** 2)) / np.sqrt(np.mean(ref[core] ** 2))
    assert err < 5e-3, err


def test_gain():
    x = torch.ones(4)
    assert torch.equal(gain(x, 0), x)
    assert abs(float(gain(x, 20)[0]) - 10.0) < 1e-5


@pytest.mark.parametrize("script,flags", [
    ("infer.py", ["-i", "-o", "-encp", "-decp", "-f0-est", "-idx", "-t", "-d", "-p", "-c", "-b", "-nc",
                  "--inputs", "--outputs", "--encoder-path", "--decoder-path", "--index", "--target", "--pitch-shift",
                  "--chunk-size", "--buffer-size", "--no-chunking"]),
    ("infer_streaming.py", ["-encp", "-decp", "-i", "-o", "-l", "-idx", "-p", "-t", "-c", "-e", "-d", "-sr", "-ig", "-og",
                            "-f0-est", "--loopback", "--chunk", "--extra", "--sample-rate", "--input-gain", "--output-gain"]),
    ("extract_index.py", ["--dataset-cache", "-encp", "-size", "-o", "-d", "--stride"]),
])
def test_cli_flag_surface(script, flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    for f in flags:
        assert f in out.stdout, f"{script} lost flag {f}"


def test_module_shim_resolves_reference_imports():
    code = ("import sys; sys.path.insert(0, %r); "
            "from module.tinyvc import Encoder, Decoder, match_features; "
            "from module.infer import Generator, StreamInfer; "
            "from module.utils import spectrogram, shift_frequency, estimate_energy, autopad_waveform; "
            "import tinyvc_amd.module.tinyvc as t; assert Encoder is t.Encoder; print('ok')") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_infer_refuses_cpu(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "infer.py"), "-d", "cpu"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode != 0 and "AMD GPU" in (out.stderr + out.stdout)
