"""Source-filter decoder (reference module/tinyvc/decoder.py:88-266): parameter containers with the
reference's state-dict layout; SourceNet, the harmonic/noise DSP and FilterNet execute in
csrc/decoder.hip."""
import math

import torch
import torch.nn as nn

from ... import spec as S
from .._base import HipModule
from .convnext import ConvNeXtLayer
from .encoder import _only_default


class FiLM(nn.Module):
    def __init__(self, input_channels, condition_channels):
        super().__init__()
        self.to_shift = nn.Conv1d(condition_channels, input_channels, 1)
        self.to_scale = nn.Conv1d(condition_channels, input_channels, 1)


class SourceNet(nn.Module):
    def __init__(self, content_channels=768, channels=128, kernel_size=7, num_layers=3, n_fft=1920,
                 frame_size=480, num_harmonics=14, sample_rate=24000):
        super().__init__()
        self.n_fft, self.frame_size = n_fft, frame_size
        self.num_harmonics, self.sample_rate = num_harmonics, sample_rate
        self.content_channels = content_channels
        self.content_in = nn.Conv1d(content_channels, channels, 1)
        self.energy_in = nn.Conv1d(1, channels, 1)
        self.f0_in = nn.Conv1d(1, channels, 1)
        self.mid_layers = nn.Sequential(*[ConvNeXtLayer(channels, kernel_size) for _ in range(num_layers)])
        self.to_amps = nn.Conv1d(channels, num_harmonics + 1, 1)
        self.to_kernel = nn.Conv1d(channels, n_fft // 2 + 1, 1)

    @torch.no_grad()
    def forward(self, content, f0, energy):    # decoder.py:126-134 -> (amps, kernel)
        dec = self.__dict__["_decoder"]
        return dec.engine(content.device).source_net(content, f0, energy)


class Downsample(nn.Module):
    def __init__(self, input_channels, output_channels, factor=4):
        super().__init__()
        self.factor = factor
        self.down_res = nn.Conv1d(input_channels, output_channels, 1)
        self.c1 = nn.Conv1d(input_channels, input_channels, 3, 1, 1, dilation=1, padding_mode="replicate")
        self.c2 = nn.Conv1d(input_channels, input_channels, 3, 1, 2, dilation=2, padding_mode="replicate")
        self.c3 = nn.Conv1d(input_channels, output_channels, 3, 1, 4, dilation=4, padding_mode="replicate")


class Upsample(nn.Module):
    def __init__(self, input_channels, output_channels, cond_channels, factor=4):
        super().__init__()
        self.factor = factor
        c = input_channels
        self.c1 = nn.Conv1d(c, c, 3, 1, 1, dilation=1, padding_mode="replicate")
        self.c2 = nn.Conv1d(c, c, 3, 1, 3, dilation=3, padding_mode="replicate")
        self.film1 = FiLM(c, cond_channels)
        self.c3 = nn.Conv1d(c, c, 3, 1, 9, dilation=9, padding_mode="replicate")
        self.c4 = nn.Conv1d(c, c, 3, 1, 27, dilation=27, padding_mode="replicate")
        self.film2 = FiLM(c, cond_channels)
        self.c5 = nn.Conv1d(c, output_channels, 1)


class FilterNet(nn.Module):
    def __init__(self, channels=(384, 192, 96, 48, 24), factors=(2, 3, 4, 4, 5), content_channels=768,
                 num_harmonics=14):
        _only_default("channels", tuple(channels), S.FILTER_CHANNELS)
        _only_default("factors", tuple(factors), S.FILTER_FACTORS)
        super().__init__()
        self.content_in = nn.Conv1d(content_channels, channels[0], 1)
        self.f0_in = nn.Conv1d(1, channels[0], 1)
        self.downs = nn.ModuleList([nn.Conv1d(num_harmonics + 3, channels[-1], 3, 1, 1, padding_mode="replicate")])
        for c, n, f in S.filter_down_plan():
            self.downs.append(Downsample(c, n, f))
        self.ups = nn.ModuleList([Upsample(c, n, c, f) for c, n, f in S.filter_up_plan()])
        self.output_layer = nn.Conv1d(channels[-1], 1, 7, 1, 3, padding_mode="replicate")

    @torch.no_grad()
    def forward(self, content, f0, energy, source):    # decoder.py:222-233 -> [B, 1, L]
        dec = self.__dict__["_decoder"]
        return dec.engine(content.device).filter_net(content, f0, energy, source).unsqueeze(1)


class Decoder(HipModule):
    def __init__(self, sample_rate=24000, n_fft=1920, frame_size=480, num_harmonics=14):
        _only_default("sample_rate", sample_rate, S.SAMPLE_RATE)
        _only_default("n_fft", n_fft, S.N_FFT)
        _only_default("frame_size", frame_size, S.HOP)
        _only_default("num_harmonics", num_harmonics, S.NUM_HARMONICS)
        super().__init__()
        self.sample_rate, self.frame_size = sample_rate, frame_size
        self.num_harmonics, self.n_fft = num_harmonics, n_fft
        self.source_net = SourceNet(frame_size=frame_size, sample_rate=sample_rate, n_fft=n_fft)
        self.filter_net = FilterNet()
        self.source_net.__dict__["_decoder"] = self
        self.filter_net.__dict__["_decoder"] = self

    @staticmethod
    def draw_noise_angle(batch, frames, device):
        """The uniform phases the reference draws inside oscillate_noise on every call (decoder.py:78), from torch's generator for
        `device`: pass the result as `noise_angle` to get torch's own draw.  By default (`noise_angle=None`) the library draws the
        phases inside its noise kernel's launch sequence instead (one launch less per call, no [B, 961, T] tensor through torch): a
        counter-based hash of (seed, utterance row, bin, frame), the seed taken from the device's torch generator (seed and offset, read
        and advanced on the host) - repeatable under
        torch.manual_seed, like the reference's draw, and like it not reproducible across devices (a CUDA and a CPU torch.rand with
        the same seed differ too)."""
        u = torch.rand(batch, S.FFT_BIN, frames, device=device)
        if u.is_cuda:       # the three tensor ops of the reference's expression as one in-place launch with the same roundings
            from ...engine import default_engine
            return default_engine(u.device).noise_angle_from_uniform(u)
        return u * 2 * math.pi - math.pi

    @torch.no_grad()
    def infer(self, content, f0, energy, noise_angle=None):
        """decoder.py:253-257 -> [B, L].  `noise_angle` [B,961,T] (extension) injects the noise
        phases, e.g. a seeded CPU draw for parity with the reference's CPU path."""
        content = self._input_device(content)
        return self.engine(content.device).decoder(content, self._input_device(f0), self._input_device(energy),
                                                   None if noise_angle is None else self._input_device(noise_angle))

    @torch.no_grad()
    def dsp(self, f0, amps, kernel, noise_angle=None):
        """decoder.py:259-266 -> source [B, 16, L]."""
        f0 = self._input_device(f0)
        return self.engine(f0.device).dsp(f0, self._input_device(amps), self._input_device(kernel),
                                          None if noise_angle is None else self._input_device(noise_angle))
