"""Band-limited sample-rate conversion for the front door of the entry scripts (the reference calls
torchaudio.functional.resample, infer.py:46,63; infer_streaming.py:70).

Same published algorithm and defaults as torchaudio's `sinc_interp_hann` resampler — a Hann-windowed
sinc polyphase filter bank, lowpass_filter_width = 6, rolloff = 0.99 — written from its documentation.
torchaudio is absent from this image, so this step has no pinned parity.  This host
restatement is the checker of the device kernel (`tvc_resample_f32`, frontdoor.hip): the GPU tests require the
two to agree to 1e-6 relative rms on every rate pair the entry scripts meet (measured: bit-identical), and the CPU tests check this file against
scipy.signal.resample_poly on band-limited signals."""
import math

import torch
import torch.nn.functional as F


def _kernel(orig, new, lowpass_filter_width, rolloff, device, dtype):
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, device=device, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, device=device, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    sinc = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t)
    return (sinc * window * (base / orig)).to(dtype), width


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """waveform [..., frames] -> [..., ceil(frames * new / orig)]"""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).float()
    kern, width = _kernel(orig, new, lowpass_filter_width, rolloff, x.device, x.dtype)
    n = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kern, stride=orig)          # [rows, new, ~n/orig]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * n / orig))
    return y[:, :target].reshape(shape[:-1] + (target,))


def gain(waveform, gain_db=1.0):
    """torchaudio.functional.gain: multiply by 10^(dB/20) (infer_streaming.py:89,91)."""
    if gain_db == 0:
        return waveform
    return waveform * (10 ** (gain_db / 20))
