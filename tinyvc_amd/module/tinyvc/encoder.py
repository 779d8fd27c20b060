"""Encoder = SSL feature estimator + F0 classifier on the 961-bin magnitude spectrogram
(reference module/tinyvc/encoder.py:11-116).  Both trunks run in one tvc_encoder_f32 call."""
import torch
import torch.nn as nn

from ... import spec as S
from .._base import HipModule
from .convnext import ConvNeXtLayer, LayerNorm


def _only_default(name, got, want):
    if got != want:
        raise NotImplementedError(f"{name}={got!r}: the HIP kernels are specialised for the reference default {want!r}")


class _Estimator(nn.Module):
    """Common parameter layout: input 1x1, LayerNorm, ConvNeXt stack, output 1x1."""

    def __init__(self, fft_bin, ch, dilations, out_ch):
        super().__init__()
        self.input_layer = nn.Conv1d(fft_bin, ch, 1)
        self.norm = LayerNorm(ch)
        self.mid_layers = nn.Sequential(*[ConvNeXtLayer(ch, dilation=d) for d in dilations])
        self.output_layer = nn.Conv1d(ch, out_ch, 1)

    def _parent(self):
        p = self.__dict__.get("_encoder")
        if p is None:
            raise RuntimeError("stand-alone estimators are not executable; call them through Encoder")
        return p


class PitchEstimator(_Estimator):
    def __init__(self, n_fft=1920, internal_channels=128, num_layers=4, num_classes=512,
                 classes_per_octave=48, min_frequency=20.0):
        _only_default("n_fft", n_fft, S.N_FFT)
        _only_default("internal_channels", internal_channels, S.PITCH_CH)
        _only_default("num_layers", num_layers, S.PITCH_LAYERS)
        _only_default("num_classes", num_classes, S.PITCH_CLASSES)
        # the class -> Hz table uploaded to the decode kernel and its 20 Hz voicing threshold are built from these two
        _only_default("classes_per_octave", classes_per_octave, S.PITCH_CPO)
        _only_default("min_frequency", float(min_frequency), float(S.PITCH_FMIN))
        super().__init__(n_fft // 2 + 1, internal_channels, [1] * num_layers, num_classes)
        self.num_classes = num_classes
        self.classes_per_octave = classes_per_octave
        self.min_frequency = min_frequency

    def forward(self, spec):           # encoder.py:33-39 -> logits [B, 512, T]
        return self._parent().forward(spec)[1]

    def infer(self, spec):             # encoder.py:69-72 -> f0 [B, 1, T]
        return self._parent().infer(spec)[1]

    @torch.no_grad()
    def decode(self, logits, k=4):     # encoder.py:61-67 -> f0 [B, 1, T]
        _only_default("k", k, 4)
        enc = self._parent()
        logits = enc._input_device(logits)
        return enc.engine(logits.device).pitch_decode(logits)

    # small helpers kept for API parity (training-side utilities of the reference, torch ops)
    def freq2id(self, f):              # encoder.py:41-45
        x = self.classes_per_octave * torch.log2(f / self.min_frequency)
        return torch.ceil(torch.clamp(x, 0, self.num_classes - 1)).to(torch.long)

    def id2freq(self, ids):            # encoder.py:48-54
        x = self.min_frequency * (2 ** (ids.to(torch.float) / self.classes_per_octave))
        return torch.where(x <= self.min_frequency, torch.zeros_like(x), x)


class SSLFeatureEstimator(_Estimator):
    def __init__(self, n_fft=1920, internal_channels=384, dilations=(1, 3, 9, 1, 1, 1), ssl_dim=768):
        _only_default("n_fft", n_fft, S.N_FFT)
        _only_default("internal_channels", internal_channels, S.SSL_CH)
        _only_default("dilations", tuple(dilations), S.SSL_DILATIONS)
        _only_default("ssl_dim", ssl_dim, S.SSL_DIM)
        super().__init__(n_fft // 2 + 1, internal_channels, dilations, ssl_dim)

    def forward(self, spec):           # encoder.py:89-94 -> [B, 768, T]
        return self._parent().infer(spec)[0]

    infer = forward                    # encoder.py:96-97


class Encoder(HipModule):
    def __init__(self, n_fft=1920, hop_size=480):
        _only_default("n_fft", n_fft, S.N_FFT)
        _only_default("hop_size", hop_size, S.HOP)
        super().__init__()
        self.n_fft = n_fft
        self.hop_size = hop_size
        self.ssl_feature_estimator = SSLFeatureEstimator(n_fft)
        self.pitch_estimator = PitchEstimator(n_fft)
        for m in (self.ssl_feature_estimator, self.pitch_estimator):
            m.__dict__["_encoder"] = self   # back-reference, not a registered sub-module

    @torch.no_grad()
    def forward(self, spec):           # encoder.py:108-111 -> (ssl, f0 logits)
        spec = self._input_device(spec)
        ssl, _f0, logits = self.engine(spec.device).encoder(spec, want_logits=True)
        return ssl, logits

    @torch.no_grad()
    def infer(self, spec):             # encoder.py:113-116 -> (ssl [B,768,T], f0 [B,1,T])
        spec = self._input_device(spec)
        ssl, f0, _ = self.engine(spec.device).encoder(spec)
        return ssl, f0
