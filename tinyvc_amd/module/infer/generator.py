"""Generator = encoder -> kNN match -> pitch shift -> decoder (reference module/infer/generator.py:12-34),
one tvc_convert_f32 call per batch."""
import torch

from .. import utils
from .._base import HipModule
from ..tinyvc import Decoder, Encoder, match_features
from ..tinyvc.feature_retrieval import prepare_reference


class Generator(HipModule):
    def __init__(self, encoder: Encoder, decoder: Decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def _weight_tensors(self):
        sd = dict(self.encoder.state_dict())
        sd.update(self.decoder.state_dict())
        return sd

    @torch.no_grad()
    def encode(self, wf):
        """generator.py:19-23: wf [B, L] -> (features [B,768,T] usable as a kNN index, f0 [B,1,T])."""
        wf = utils.autopad_waveform(self._input_device(wf))
        eng = self.engine(wf.device)
        ssl, f0, _ = eng.encoder(eng.stft_mag(wf))
        return ssl, f0

    @torch.no_grad()
    def convert(self, wf, tgt, pitch_shift, f0_estimation="default", device=None, noise_angle=None, lengths=None):
        """generator.py:26-34: wf [B, L], tgt [1 or B, 768, N] -> converted waveform [B, L'] (L' = L
        padded to a multiple of 480).  `f0_estimation` / `device` are accepted and ignored exactly as
        in the reference.  `noise_angle` [B,961,T] (extension) injects the decoder's noise phases;
        by default the library draws them itself (seeded from torch's generator: Decoder.draw_noise_angle).
        `lengths` (extension): a RAGGED batch - row b of wf holds an utterance of lengths[b] samples, zero-padded behind it.
        Every utterance is converted over its own length (padded to a multiple of 480), exactly as if it were converted
        alone (the reference's loop, infer.py:60-66; GRN and the oscillator's phase run over the whole time axis, so padding
        to a common length would change the results); row b of the result holds it, zeros behind."""
        wf = utils.autopad_waveform(self._input_device(wf))
        tgt = self._input_device(tgt)
        eng = self.engine(wf.device)
        B, L = wf.shape
        if noise_angle is not None:
            noise_angle = self._input_device(noise_angle)
        if lengths is not None:
            lens = [-(-int(n) // 480) * 480 for n in lengths]
            if len(lens) != B or max(lens) > L or min(lens) <= 960:
                raise ValueError("lengths: one entry per row, each in (960, L]")
            if tgt.shape[0] != 1:
                raise ValueError("a ragged batch takes one shared index")
            blob, n = prepare_reference(tgt)
            return eng.convert_ragged(wf, lens, blob, n, pitch_shift, noise_angle)
        if tgt.shape[0] == 1:
            blob, n = prepare_reference(tgt)
            return eng.convert(wf, blob, n, pitch_shift, noise_angle)
        # one index per utterance: staged path
        spec = eng.stft_mag(wf)
        energy = eng.energy(wf)
        z, f0, _ = eng.encoder(spec)
        z = match_features(z, tgt)
        f0 = eng.shift_frequency(f0, pitch_shift)
        return eng.decoder(z, f0, energy, noise_angle)
