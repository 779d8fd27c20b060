"""Thin host-side wrapper around one `tvc_ctx` (one per device and weight set).

PyTorch is used here for device memory (tensors in, tensors out), the current HIP stream and the
scratch workspace; every computation is a call into libtinyvc_hip.so.
"""
import ctypes
import math
import threading

import torch

from . import _lib, spec

_F32 = torch.float32


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check_dev(t, name, device):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise _lib.TinyVCError(
            f"{name} is on {t.device}: tinyvc_amd runs on an AMD GPU only (no CPU fallback). "
            "Move the model and inputs to 'cuda'.")
    if device is not None and t.device != device:
        raise _lib.TinyVCError(f"{name} is on {t.device}, engine is on {device}")


def _prep(t, name, device):
    _check_dev(t, name, device)
    if t.dtype != _F32:
        t = t.float()
    return t.contiguous()


def pitch_class_table():
    """PitchEstimator.id2freq over ids 0..511 (reference encoder.py:48-54), on the host."""
    ids = torch.arange(spec.PITCH_CLASSES).to(torch.float)
    x = spec.PITCH_FMIN * (2 ** (ids / spec.PITCH_CPO))
    x[x <= spec.PITCH_FMIN] = 0
    return x.contiguous()


class Engine:
    """Owns a tvc_ctx on `device`, its weights and a grow-only scratch workspace."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.TinyVCError(f"tinyvc_amd needs a GPU device, got {device}")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.lib = _lib.load_library()
        h = ctypes.c_void_p()
        rc = self.lib.tvc_ctx_create(device.index, ctypes.byref(h))
        if rc != 0 or not h:
            raise _lib.TinyVCError(f"tvc_ctx_create(device={device.index}) failed with {rc}")
        self.ctx = h
        self._ws = None
        self._seed = 0x1234ABCD
        self.weights_key = None

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.tvc_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _ok(self, rc, what):
        if rc != 0:
            msg = self.lib.tvc_last_error(self.ctx)
            raise _lib.TinyVCError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, B, L, N):
        need = ctypes.c_size_t()
        self._ok(self.lib.tvc_workspace_bytes(self.ctx, int(B), int(L), int(max(N, 4)), ctypes.byref(need)), "tvc_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def _wsargs(self, B, L, N=4):
        ws = self.workspace(B, L, N)
        return _ptr(ws), ctypes.c_size_t(ws.numel())

    def set_ragged_batch_frames(self, max_frames):
        """Frames per in-kernel batch of this engine's ragged calls (0 = the default, 80 000); results do not depend on it."""
        self._ok(self.lib.tvc_ctx_set_ragged_batch_frames(self.ctx, int(max_frames)), "tvc_ctx_set_ragged_batch_frames")

    def next_seed(self):
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._seed

    def profile(self, on=True):
        """Bracket stages with hipEvents on the launch stream (see tvc_profile_read): True / 1 = every stage and FilterNet block,
        2 = the `filter_net` region only, False / 0 = off."""
        self._ok(self.lib.tvc_profile_enable(self.ctx, int(on)), "tvc_profile_enable")

    def profile_read(self):
        """{region: milliseconds} summed since the last read; synchronises the recorded events."""
        buf = ctypes.create_string_buffer(8192)
        self._ok(self.lib.tvc_profile_read(self.ctx, buf, len(buf)), "tvc_profile_read")
        out = {}
        for item in buf.value.decode().split(";"):
            if "=" in item:
                k, v = item.split("=")
                out[k] = float(v)
        return out

    # ------------------------------------------------------------------ weights
    def load_weights(self, tensors):
        """tensors: {state_dict key: tensor} for an encoder, a decoder or both."""
        keep = []
        for k, v in tensors.items():
            t = v.detach().to("cpu", _F32).contiguous()
            keep.append(t)
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            self._ok(self.lib.tvc_load_tensor(self.ctx, k.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()), f"tvc_load_tensor({k})")
        tab = pitch_class_table()
        self._ok(self.lib.tvc_set_pitch_table(self.ctx, ctypes.c_void_p(tab.data_ptr()), tab.numel()), "tvc_set_pitch_table")
        self._ok(self.lib.tvc_finalize_weights(self.ctx), "tvc_finalize_weights")

    # ------------------------------------------------------------------ stages
    def stft_mag(self, wav):
        wav = _prep(wav, "wave", self.device)
        B, L = wav.shape
        if L % spec.HOP:
            raise ValueError("waveform length must be a multiple of 480 (autopad_waveform)")
        out = torch.empty(B, spec.FFT_BIN, L // spec.HOP, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, L)
        self._ok(self.lib.tvc_stft_mag_f32(self.ctx, self._stream(), _ptr(wav), _ptr(out), B, L, p, n), "tvc_stft_mag_f32")
        return out

    def energy(self, wav):
        wav = _prep(wav, "wave", self.device)
        B, L = wav.shape
        out = torch.empty(B, 1, L, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, max(L - L % spec.HOP, spec.HOP))
        self._ok(self.lib.tvc_energy_f32(self.ctx, self._stream(), _ptr(wav), _ptr(out), B, L, p, n), "tvc_energy_f32")
        return out

    # ------------------------------------------------------------------ front door (entry scripts)
    def resample(self, wav, orig_freq, new_freq):
        """torchaudio.functional.resample on the device: wav [..., n] -> [..., ceil(n * new / orig)]."""
        orig_freq, new_freq = int(orig_freq), int(new_freq)
        wav = _prep(wav, "wave", self.device)
        if orig_freq == new_freq:
            return wav
        shape = wav.shape
        x = wav.reshape(-1, shape[-1])
        n_out = self.lib.tvc_resample_out_len(x.shape[1], orig_freq, new_freq)
        y = torch.empty(x.shape[0], n_out, dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_resample_f32(self.ctx, self._stream(), _ptr(x), _ptr(y), x.shape[0], x.shape[1], orig_freq, new_freq), "tvc_resample_f32")
        return y.reshape(shape[:-1] + (n_out,))

    def pcm16_to_f32(self, pcm, gain_db=0.0):
        """int16 PCM -> float in [-1, 1) (x / 32768), then torchaudio.functional.gain(gain_db) (infer_streaming.py:85-89)."""
        _check_dev(pcm, "pcm", self.device)
        if pcm.dtype != torch.int16:
            raise ValueError("pcm must be int16")
        pcm = pcm.contiguous()
        y = torch.empty(pcm.shape, dtype=_F32, device=self.device)
        if pcm.numel():
            self._ok(self.lib.tvc_pcm16_to_f32(self.ctx, self._stream(), _ptr(pcm), _ptr(y), pcm.numel(), float(gain_db)), "tvc_pcm16_to_f32")
        return y

    def f32_to_pcm16(self, x, gain_db=0.0):
        """gain(gain_db) -> * 32768 -> int16 (numpy's cast: truncation toward zero) (infer_streaming.py:91-94)."""
        x = _prep(x, "wave", self.device)
        pcm = torch.empty(x.shape, dtype=torch.int16, device=self.device)
        if x.numel():
            self._ok(self.lib.tvc_f32_to_pcm16(self.ctx, self._stream(), _ptr(x), _ptr(pcm), x.numel(), float(gain_db)), "tvc_f32_to_pcm16")
        return pcm

    def encoder(self, spec_t, want_logits=False):
        x = _prep(spec_t, "spec", self.device)
        B, C, T = x.shape
        if C != spec.FFT_BIN:
            raise ValueError(f"spec must have {spec.FFT_BIN} bins, got {C}")
        ssl = torch.empty(B, spec.SSL_DIM, T, dtype=_F32, device=self.device)
        f0 = torch.empty(B, 1, T, dtype=_F32, device=self.device)
        logits = torch.empty(B, spec.PITCH_CLASSES, T, dtype=_F32, device=self.device) if want_logits else None
        p, n = self._wsargs(B, T * spec.HOP)
        self._ok(self.lib.tvc_encoder_f32(self.ctx, self._stream(), _ptr(x), _ptr(ssl), _ptr(f0), _ptr(logits), B, T, p, n), "tvc_encoder_f32")
        return ssl, f0, logits

    def pitch_decode(self, logits):
        """PitchEstimator.decode: logits [B,512,T] -> f0 [B,1,T]."""
        lg = _prep(logits, "logits", self.device)
        B, C, T = lg.shape
        if C != spec.PITCH_CLASSES:
            raise ValueError(f"logits must have {spec.PITCH_CLASSES} classes")
        f0 = torch.empty(B, 1, T, dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_pitch_decode_f32(self.ctx, self._stream(), _ptr(lg), _ptr(f0), B, T), "tvc_pitch_decode_f32")
        return f0

    def knn_prepare(self, index):
        """index: [768, N] or [1, 768, N] -> (prepared blob (1-D float tensor), N).  An fp32 index gets the fp32 storage
        (bit-exact indices on gap-checked inputs); a torch.float16 index gets the fp16 storage (2 B per element: the
        1 M-vector case).  The blob is self-describing: knn_match / convert take either."""
        _check_dev(index, "index", self.device)
        half = index.dtype == torch.float16
        idx = index if half else _prep(index, "index", self.device)
        if idx.dim() == 3:
            if idx.shape[0] != 1:
                raise ValueError("knn_prepare takes one index ([1, 768, N])")
            idx = idx[0]
        if idx.shape[0] != spec.SSL_DIM:
            raise ValueError(f"index must be [768, N], got {tuple(idx.shape)}")
        N = idx.shape[1]
        if N < 4:
            raise RuntimeError("selected index k out of range")  # what torch.topk raises in the reference
        if half:
            rows = idx.t().contiguous()          # [N, 768] half, one vector per row
            blob = torch.empty(self.lib.tvc_knn_prepared_elems_f16(N), dtype=_F32, device=self.device)
            self._ok(self.lib.tvc_knn_prepare_index_f16(self.ctx, self._stream(), _ptr(rows), _ptr(blob), N), "tvc_knn_prepare_index_f16")
            return blob, N
        blob = torch.empty(self.lib.tvc_knn_prepared_elems(N), dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_knn_prepare_index_f32(self.ctx, self._stream(), _ptr(idx.contiguous()), _ptr(blob), N), "tvc_knn_prepare_index_f32")
        return blob, N

    def knn_match(self, src, prepared, N, want_indices=False):
        src = _prep(src, "source", self.device)
        B, C, T = src.shape
        if C != spec.SSL_DIM:
            raise ValueError(f"source must have {spec.SSL_DIM} channels")
        out = torch.empty_like(src)
        idx = torch.empty(B, T, 4, dtype=torch.int64, device=self.device) if want_indices else None
        p, n = self._wsargs(B, T * spec.HOP, N)
        self._ok(self.lib.tvc_knn_match_f32(self.ctx, self._stream(), _ptr(src), _ptr(prepared), N, _ptr(out), _ptr(idx), B, T, p, n), "tvc_knn_match_f32")
        return (out, idx) if want_indices else out

    # ---- index-sharded match (one prepared index shard per rank; merged by parallel.match_features_sharded) ----
    METRICS = {"cos": 0, "IP": 1, "L2": 2}

    def knn_match_general(self, src, index, k, metrics, want_indices=False):
        """match_features for any k in 1..8 and metrics in {'cos', 'IP', 'L2'} on the RAW index [768, N] (plain fp32, csrc/knn_general.hip):
        src [B, 768, T] -> matched [B, 768, T] (, indices [B, T, k] int64 in rank order)."""
        src = _prep(src, "source", self.device)
        index = _prep(index, "index", self.device)
        if metrics not in self.METRICS:
            raise ValueError(f"metrics must be one of {sorted(self.METRICS)}, got {metrics!r}")
        if index.dim() != 2 or index.shape[0] != spec.SSL_DIM or src.dim() != 3 or src.shape[1] != spec.SSL_DIM:
            raise ValueError("knn_match_general: src [B, 768, T], index [768, N]")
        B, _, T = src.shape
        N = index.shape[1]
        if not 1 <= int(k) <= 8:
            raise NotImplementedError("the HIP kernel serves k = 1 ... 8")
        if N < k:
            raise RuntimeError("selected index k out of range")      # what torch.topk raises in the reference
        out = torch.empty(B, spec.SSL_DIM, T, dtype=_F32, device=self.device)
        idx = torch.empty(B, T, int(k), dtype=torch.int64, device=self.device)
        self._ok(self.lib.tvc_knn_match_general_f32(self.ctx, self._stream(), _ptr(src), _ptr(index), N, int(k), self.METRICS[metrics], _ptr(out), _ptr(idx), None,
                                                    B, T, None, 0), "tvc_knn_match_general_f32")
        return (out, idx) if want_indices else out

    def knn_topk(self, src, prepared, N):
        """This shard's top-4 per query: (sims [B,T,4] fp32 descending, idx [B,T,4] int64 local indices)."""
        src = _prep(src, "source", self.device)
        B, C, T = src.shape
        if C != spec.SSL_DIM:
            raise ValueError(f"source must have {spec.SSL_DIM} channels")
        sims = torch.empty(B, T, 4, dtype=_F32, device=self.device)
        idx = torch.empty(B, T, 4, dtype=torch.int64, device=self.device)
        p, n = self._wsargs(B, T * spec.HOP, N)
        self._ok(self.lib.tvc_knn_topk_f32(self.ctx, self._stream(), _ptr(src), _ptr(prepared), N, _ptr(sims), _ptr(idx), B, T, p, n), "tvc_knn_topk_f32")
        return sims, idx

    def knn_gather_slots(self, prepared, N, idx):
        """idx [B,T,4] int64 local indices (negative = not on this shard) -> slots [B,T,4,768] raw rows / zeros."""
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        slots = torch.empty(*idx.shape, spec.SSL_DIM, dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_knn_gather_slots_f32(self.ctx, self._stream(), _ptr(prepared), N, _ptr(idx), _ptr(slots), idx.numel()), "tvc_knn_gather_slots_f32")
        return slots

    def knn_finish(self, slots):
        """slots [B,T,4,768] -> [B,768,T]: mean of the four rows in the single-GPU summation order."""
        slots = _prep(slots, "slots", self.device)
        B, T = slots.shape[0], slots.shape[1]
        out = torch.empty(B, spec.SSL_DIM, T, dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_knn_finish_f32(self.ctx, self._stream(), _ptr(slots), _ptr(out), B, T), "tvc_knn_finish_f32")
        return out

    def shift_frequency(self, f0, semitones):
        f0 = _prep(f0, "f0", self.device)
        out = torch.empty_like(f0)
        if f0.numel():
            self._ok(self.lib.tvc_shift_frequency_f32(self.ctx, self._stream(), _ptr(f0), _ptr(out), f0.numel(), float(semitones)), "tvc_shift_frequency_f32")
        return out

    def noise_angle_from_uniform(self, u):
        """u (fp32, contiguous, on the device) uniform in [0, 1) -> u * 2 * pi - pi in place: the reference's three tensor ops
        (decoder.py:78) as one launch with the same roundings."""
        _check_dev(u, "u", self.device)
        if u.dtype != _F32 or not u.is_contiguous():
            raise ValueError("u must be contiguous fp32")
        if u.numel():
            self._ok(self.lib.tvc_noise_angle_from_uniform_f32(self.ctx, self._stream(), _ptr(u), u.numel()), "tvc_noise_angle_from_uniform_f32")
        return u

    def _angle(self, noise_angle, B, T):
        if noise_angle is None and torch.cuda.is_current_stream_capturing():
            # inside a graph capture a kernel argument is baked in - one seed would replay for every launch of the graph -; torch's
            # generator is graph-safe (its offset advances per replay), so a captured call keeps the reference's torch.rand draw
            return self.noise_angle_from_uniform(torch.rand(B, spec.FFT_BIN, T, device=self.device)), 0
        if noise_angle is None:
            # the library draws the phases itself (tvc_* with noise_angle = NULL: a counter-based hash of (seed, row, bin, frame)).  Its
            # seed comes from THIS DEVICE's torch generator - the one the reference's torch.rand(device=...) draws from -: (seed, philox
            # offset) read on the host, the offset advanced as a draw would advance it.  No device launch, repeatable under
            # torch.manual_seed, untouched by CPU-side draws (module construction) in between.
            g = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
            off = g.get_offset()
            g.set_offset(off + 4)
            z = (g.initial_seed() * 0x9E3779B97F4A7C15 + off * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            return None, z
        a = _prep(noise_angle, "noise_angle", self.device)
        if tuple(a.shape) != (B, spec.FFT_BIN, T):
            raise ValueError(f"noise_angle must be [{B}, {spec.FFT_BIN}, {T}], got {tuple(a.shape)}")
        return a, 0

    def decoder(self, content, f0, energy, noise_angle=None, stages=False):
        content = _prep(content, "content", self.device)
        f0 = _prep(f0, "f0", self.device)
        energy = _prep(energy, "energy", self.device)
        B, C, T = content.shape
        L = T * spec.HOP
        if C != spec.SSL_DIM or tuple(f0.shape) != (B, 1, T) or tuple(energy.shape) != (B, 1, L):
            raise ValueError(f"decoder shapes: content [B,768,T], f0 [B,1,T], energy [B,1,T*480]; got {tuple(content.shape)}, {tuple(f0.shape)}, {tuple(energy.shape)}")
        a, seed = self._angle(noise_angle, B, T)
        wave = torch.empty(B, L, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, L)
        if not stages:
            self._ok(self.lib.tvc_decoder_f32(self.ctx, self._stream(), _ptr(content), _ptr(f0), _ptr(energy), _ptr(a), seed, _ptr(wave), B, T, p, n), "tvc_decoder_f32")
            return wave
        amps = torch.empty(B, spec.NUM_HARMONICS + 1, T, dtype=_F32, device=self.device)
        kern = torch.empty(B, spec.FFT_BIN, T, dtype=_F32, device=self.device)
        source = torch.empty(B, 16, L, dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_decoder_stages_f32(self.ctx, self._stream(), _ptr(content), _ptr(f0), _ptr(energy), _ptr(a), seed, _ptr(wave),
                                                 _ptr(amps), _ptr(kern), _ptr(source), B, T, p, n), "tvc_decoder_stages_f32")
        return wave, amps, kern, source

    def source_net(self, content, f0, energy):
        """SourceNet.forward (decoder.py:126-134) alone: (amps [B,15,T], kernel [B,961,T]) - no DSP, no FilterNet pass."""
        content = _prep(content, "content", self.device)
        f0 = _prep(f0, "f0", self.device)
        energy = _prep(energy, "energy", self.device)
        B, C, T = content.shape
        if C != spec.SSL_DIM or tuple(f0.shape) != (B, 1, T) or tuple(energy.shape) != (B, 1, T * spec.HOP):
            raise ValueError("source_net shapes: content [B,768,T], f0 [B,1,T], energy [B,1,T*480]")
        amps = torch.empty(B, spec.NUM_HARMONICS + 1, T, dtype=_F32, device=self.device)
        kern = torch.empty(B, spec.FFT_BIN, T, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, T * spec.HOP)
        self._ok(self.lib.tvc_decoder_stages_f32(self.ctx, self._stream(), _ptr(content), _ptr(f0), _ptr(energy), None, 0, None,
                                                 _ptr(amps), _ptr(kern), None, B, T, p, n), "tvc_decoder_stages_f32 (SourceNet)")
        return amps, kern

    def filter_net(self, content, f0, energy, source, blocks=False):
        """FilterNet.forward -> wave [B, L]; blocks=True also returns (skips[5], ups[4]): the Downsample / Upsample
        block outputs of decoder.py:227-232 (ups[4] is folded into the output conv, see tvc_filter_net_f32)."""
        content = _prep(content, "content", self.device)
        f0 = _prep(f0, "f0", self.device)
        energy = _prep(energy, "energy", self.device)
        source = _prep(source, "source", self.device)
        B, C, T = content.shape
        L = T * spec.HOP
        if C != spec.SSL_DIM or tuple(f0.shape) != (B, 1, T) or tuple(energy.shape) != (B, 1, L) or tuple(source.shape) != (B, 16, L):
            raise ValueError("filter_net shapes: content [B,768,T], f0 [B,1,T], energy [B,1,T*480], source [B,16,T*480]")
        wave = torch.empty(B, L, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, L)
        skips, ups, sk, up = None, None, None, None
        if blocks:
            ch, fac = spec.FILTER_CHANNELS, spec.FILTER_FACTORS
            dn = [L, L // 5, L // 20, L // 80, L // 240]
            skips = [torch.empty(B, ch[4 - i], dn[i], dtype=_F32, device=self.device) for i in range(5)]
            ups, l = [], T
            for i in range(4):
                l *= fac[i]
                ups.append(torch.empty(B, ch[i + 1], l, dtype=_F32, device=self.device))
            sk = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in skips])
            up = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ups])
        self._ok(self.lib.tvc_filter_net_f32(self.ctx, self._stream(), _ptr(content), _ptr(f0), _ptr(energy), _ptr(source), _ptr(wave),
                                             sk, up, B, T, p, n), "tvc_filter_net_f32")
        return (wave, skips, ups) if blocks else wave

    def dsp(self, f0, amps, kernel, noise_angle=None):
        f0 = _prep(f0, "f0", self.device)
        amps = _prep(amps, "amps", self.device)
        kernel = _prep(kernel, "kernel", self.device)
        B, _, T = f0.shape
        a, seed = self._angle(noise_angle, B, T)
        source = torch.empty(B, 16, T * spec.HOP, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, T * spec.HOP)
        self._ok(self.lib.tvc_dsp_f32(self.ctx, self._stream(), _ptr(f0), _ptr(amps), _ptr(kernel), _ptr(a), seed, _ptr(source), B, T, p, n), "tvc_dsp_f32")
        return source

    def convert(self, wav, prepared, N, pitch_shift, noise_angle=None, out=None):
        wav = _prep(wav, "wave", self.device)
        B, L = wav.shape
        if L % spec.HOP:
            raise ValueError("waveform length must be a multiple of 480 (autopad_waveform)")
        a, seed = self._angle(noise_angle, B, L // spec.HOP)
        wave = out if out is not None else torch.empty(B, L, dtype=_F32, device=self.device)
        p, n = self._wsargs(B, L, N)
        self._ok(self.lib.tvc_convert_f32(self.ctx, self._stream(), _ptr(wav), _ptr(prepared), N, float(pitch_shift), _ptr(a), seed, _ptr(wave), B, L, p, n), "tvc_convert_f32")
        return wave

    def convert_ragged(self, wav, lengths, prepared, N, pitch_shift, noise_angle=None):
        """wav [B, Lmax] (row b holds an utterance of lengths[b] samples, a multiple of 480, zero-padded behind it) -> [B, Lmax]:
        every utterance converted over its OWN length (tvc_convert_ragged_f32: per-utterance lengths inside the kernels)."""
        wav = _prep(wav, "wave", self.device)
        B, Lmax = wav.shape
        if Lmax % spec.HOP:
            raise ValueError("the padded length must be a multiple of 480")
        lens = (ctypes.c_int64 * B)(*[int(x) for x in lengths])
        a, seed = self._angle(noise_angle, B, Lmax // spec.HOP)
        need = ctypes.c_size_t()
        self._ok(self.lib.tvc_workspace_bytes_ragged(self.ctx, B, Lmax, lens, int(max(N, 4)), ctypes.byref(need)), "tvc_workspace_bytes_ragged")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        wave = torch.empty(B, Lmax, dtype=_F32, device=self.device)
        self._ok(self.lib.tvc_convert_ragged_f32(self.ctx, self._stream(), _ptr(wav), Lmax, lens, _ptr(prepared), N, float(pitch_shift), _ptr(a), seed,
                                                 _ptr(wave), B, _ptr(self._ws), ctypes.c_size_t(self._ws.numel())), "tvc_convert_ragged_f32")
        return wave

    def stream_push(self, buf, blocks):
        """buf [S, n] <- (buf[:, m:], blocks [S, m]) in place, one launch (stream.py:69-70's roll + slice assignment)."""
        _check_dev(buf, "buf", self.device)
        blocks = _prep(blocks, "blocks", self.device)
        S, n = buf.shape
        if blocks.shape[0] != S or not buf.is_contiguous() or buf.dtype != _F32:
            raise ValueError("stream_push: buf [S, n] contiguous fp32, blocks [S, m]")
        self._ok(self.lib.tvc_stream_push_f32(self.ctx, self._stream(), _ptr(buf), _ptr(blocks), S, n, blocks.shape[1]), "tvc_stream_push_f32")
        return buf

    def sola(self, y, sola_buf, fade_in, block, use_phase_vocoder=False, want_shift=False):
        """y [S, Ly]; sola_buf [S, 1920] updated in place; returns out [S, block] (and shifts)."""
        y = _prep(y, "y", self.device)
        _check_dev(sola_buf, "sola_buffer", self.device)
        fade_in = _prep(fade_in, "fade_in_window", self.device)
        if not sola_buf.is_contiguous() or sola_buf.dtype != _F32:
            raise ValueError("sola_buffer must be contiguous fp32")
        S, Ly = y.shape
        out = torch.empty(S, block, dtype=_F32, device=self.device)
        shift = torch.empty(S, dtype=torch.int32, device=self.device) if want_shift else None
        self._ok(self.lib.tvc_sola_f32(self.ctx, self._stream(), _ptr(y), _ptr(sola_buf), _ptr(fade_in), _ptr(out), _ptr(shift), S, Ly, int(block), int(bool(use_phase_vocoder))), "tvc_sola_f32")
        return (out, shift) if want_shift else out


# ---------------------------------------------------------------------- shared weightless engines
_default = {}
_lock = threading.Lock()


def default_engine(device):
    """Engine without checkpoint weights, for the free functions of module.utils / match_features."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.TinyVCError(f"tinyvc_amd runs on an AMD GPU only; got a tensor on {device}")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _lock:
        if idx not in _default:
            _default[idx] = Engine(torch.device("cuda", idx))
        return _default[idx]
