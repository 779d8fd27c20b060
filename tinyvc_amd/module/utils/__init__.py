"""module.utils of the reference, inference-path subset (reference module/utils/__init__.py:1-5;
`estimate_f0` and the dataset/loss/noise helpers are training-side and not provided)."""
import torch
import torch.nn.functional as F

from ...engine import default_engine


def autopad_waveform(wf, frame_size=480):
    """reference module/utils/auto_padding.py:5-11 — zero-pad [B, L] to a multiple of frame_size."""
    rem = wf.shape[1] % frame_size
    return F.pad(wf, (0, frame_size - rem)) if rem else wf


@torch.no_grad()
def spectrogram(wave, n_fft=1920, hop_size=480):
    """reference module/utils/spectrogram.py:8-15 — [B, L] -> [B, 961, L/480] magnitude."""
    if n_fft != 1920 or hop_size != 480:
        raise NotImplementedError("the HIP STFT is specialised for n_fft=1920, hop_size=480")
    out = default_engine(wave.device).stft_mag(wave)
    return out.to(wave.dtype)


@torch.no_grad()
def estimate_energy(wave, frame_size=64):
    """reference module/utils/energy_estimation.py:9-14 — [B, L] -> [B, 1, L]."""
    if frame_size != 64:
        raise NotImplementedError("the HIP energy kernel is specialised for frame_size=64")
    return default_engine(wave.device).energy(wave)


@torch.no_grad()
def shift_frequency(f0, shift):
    """reference module/utils/pitch_shift.py:11-15 — shift f0 [Hz] by `shift` semitones."""
    return default_engine(f0.device).shift_frequency(f0, shift)


__all__ = ["autopad_waveform", "spectrogram", "estimate_energy", "shift_frequency"]
