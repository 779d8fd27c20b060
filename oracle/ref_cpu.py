"""ORACLE — test infrastructure, not product code.

A CPU restatement (functional, flat state_dict in, tensors out) of the reference's voice-conversion
inference path, written from the op semantics recorded in SURVEY.md §2.2/§8a.  It runs on stock
ATen CPU ops, i.e. on the same third-party arithmetic the reference itself runs on with `-d cpu`.

Pinned: `tests/test_oracle_golden.py` checks every function here against the golden vectors under
`tests/golden/`, which `tools/gen_golden.py` captured from the reference itself
(imported from /root/reference in the build container; the reference ships no tests or fixtures of
its own).

Who may import this: `tests/`, `__graft_entry__.smoke()`, and `bench.py`'s `cpu_baseline` leg
(as the thing timed *beside* the HIP path).  Nothing under `tinyvc_amd/` imports it; the product
path has no CPU fallback.

Shapes follow the reference: waveforms [B, L] @24 kHz, feature maps [B, C, T], T = L / 480.
"""
import math

import torch
import torch.nn.functional as F

SAMPLE_RATE = 24000
N_FFT = 1920
HOP = 480
FFT_BIN = 961
NUM_HARMONICS = 14
FILTER_CHANNELS = (384, 192, 96, 48, 24)
FILTER_FACTORS = (2, 3, 4, 4, 5)
SSL_DILATIONS = (1, 3, 9, 1, 1, 1)


def _fp(x):
    """fp32 like the reference - except that an fp64 tensor stays fp64: evaluating this file on `.double()` weights and inputs
    gives the path's fp64 value, the fixed truth tests/test_gpu_truth.py measures both fp32 implementations against."""
    return x if x.dtype == torch.float64 else x.float()


# --------------------------------------------------------------------------- front end (a1-a3)
def autopad_waveform(wf, frame=HOP):
    """reference module/utils/auto_padding.py:5-11 — zero-extend the tail to a multiple of 480."""
    rem = wf.shape[1] % frame
    if rem:
        wf = F.pad(wf, (0, frame - rem))
    return wf


def spectrogram(wf):
    """reference module/utils/spectrogram.py:8-15 — |STFT| (n_fft 1920, hop 480, periodic Hann,
    centre/reflect), first frame dropped -> [B, 961, L/480]."""
    wf = _fp(wf)
    win = torch.hann_window(N_FFT, dtype=wf.dtype)
    s = torch.stft(wf, N_FFT, HOP, window=win, return_complex=True).abs()
    return s[:, :, 1:]


def estimate_energy(wf):
    """reference module/utils/energy_estimation.py:9-14 — max|x| over 128-sample windows every 64
    samples (pad 32), linearly interpolated back to L -> [B, 1, L]."""
    e = F.max_pool1d(wf.abs().unsqueeze(1), 128, 64, 32)
    return F.interpolate(e, wf.shape[1], mode="linear")


# --------------------------------------------------------------------------- ConvNeXt-v2 (a4-a6)
def layer_norm_c(x, gamma, beta, eps=1e-5):
    """reference module/tinyvc/convnext.py:17-19 — LayerNorm over channels of [B, C, T]."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), gamma, beta, eps).transpose(1, 2)


def grn(x, gamma, beta, eps=1e-6):
    """reference module/tinyvc/convnext.py:31-34 — global response norm over the whole time axis."""
    gx = torch.norm(x, p=2, dim=2, keepdim=True)
    nx = gx / (gx.mean(dim=1, keepdim=True) + eps)
    return gamma * (x * nx) + beta + x


def conv1d_rep(x, w, b, dilation=1, groups=1):
    """Conv1d with padding_mode='replicate' and 'same' length (k odd)."""
    pad = (w.shape[2] - 1) * dilation // 2
    if pad:
        x = F.pad(x, (pad, pad), mode="replicate")
    return F.conv1d(x, w, b, dilation=dilation, groups=groups)


def convnext_layer(sd, p, x, dilation=1):
    """reference module/tinyvc/convnext.py:49-58."""
    h = conv1d_rep(x, sd[p + ".c1.weight"], sd[p + ".c1.bias"], dilation, groups=x.shape[1])
    h = layer_norm_c(h, sd[p + ".norm.gamma"], sd[p + ".norm.beta"])
    h = F.gelu(F.conv1d(h, sd[p + ".c2.weight"], sd[p + ".c2.bias"]))
    h = grn(h, sd[p + ".grn.gamma"], sd[p + ".grn.beta"])
    return F.conv1d(h, sd[p + ".c3.weight"], sd[p + ".c3.bias"]) + x


# --------------------------------------------------------------------------- encoder (a7-a9)
def _estimator_trunk(sd, p, spec, dilations):
    x = F.conv1d(spec, sd[p + ".input_layer.weight"], sd[p + ".input_layer.bias"])
    x = layer_norm_c(x, sd[p + ".norm.gamma"], sd[p + ".norm.beta"])
    for i, d in enumerate(dilations):
        x = convnext_layer(sd, f"{p}.mid_layers.{i}", x, d)
    return F.conv1d(x, sd[p + ".output_layer.weight"], sd[p + ".output_layer.bias"])


def ssl_features(sd, spec):
    """reference module/tinyvc/encoder.py:89-97 -> [B, 768, T]."""
    return _estimator_trunk(sd, "ssl_feature_estimator", spec, SSL_DILATIONS)


def pitch_logits(sd, spec):
    """reference module/tinyvc/encoder.py:33-39 -> [B, 512, T]."""
    return _estimator_trunk(sd, "pitch_estimator", spec, (1, 1, 1, 1))


def pitch_decode(logits, k=4, fmin=20.0, cpo=48):
    """reference module/tinyvc/encoder.py:48-54,61-67 — softmax over the top-4 logits weights
    the class frequencies 20*2^(id/48) (classes at or below 20 Hz count as 0); f0 <= 20 -> 0."""
    top, ids = torch.topk(logits, k, dim=1)
    p = F.softmax(top, dim=1)
    fr = fmin * (2 ** (ids.to(logits.dtype) / cpo))
    fr = torch.where(fr <= fmin, torch.zeros_like(fr), fr)
    f0 = (p * fr).sum(dim=1, keepdim=True)
    return torch.where(f0 <= fmin, torch.zeros_like(f0), f0)


def encoder_infer(sd, spec):
    """reference module/tinyvc/encoder.py:113-116 -> (ssl [B,768,T], f0 [B,1,T])."""
    return ssl_features(sd, spec), pitch_decode(pitch_logits(sd, spec))


# --------------------------------------------------------------------------- kNN match (a10)
def match_features(source, reference, k=4, return_indices=False, metrics="cos", alpha=0.0):
    """reference module/tinyvc/feature_retrieval.py:15-33 - top-k of every source frame against the index under `metrics`
    ('cos' :24-27, 'IP' :20-21, 'L2' :22-23), mean of the k raw index vectors (:30), blended with the input by alpha (:33).
    `reference` may have batch 1 (broadcast; the reference's bmm needs equal batch)."""
    if reference.shape[0] == 1 and source.shape[0] != 1:
        reference = reference.expand(source.shape[0], -1, -1)
    s = source.transpose(1, 2)
    r = reference.transpose(1, 2)
    if metrics == "IP":
        sims = torch.bmm(s, r.transpose(1, 2))
    elif metrics == "L2":
        sims = -torch.cdist(s, r)
    elif metrics == "cos":
        rn = r / (torch.norm(r, dim=2, keepdim=True, p=2) + 1e-6)
        sn = s / (torch.norm(s, dim=2, keepdim=True, p=2) + 1e-6)
        sims = torch.bmm(sn, rn.transpose(1, 2))
    else:
        raise ValueError(metrics)
    best = torch.topk(sims, k, dim=2)
    picked = torch.stack([r[n][best.indices[n]] for n in range(s.shape[0])], dim=0)
    out = picked.mean(dim=2).transpose(1, 2)
    if alpha != 0.0:
        out = out * (1 - alpha) + source * alpha
    if return_indices:
        return out, best.indices, sims
    return out


# --------------------------------------------------------------------------- pitch shift (a11)
def shift_frequency(f0, semitones):
    """reference module/utils/pitch_shift.py:5-15."""
    midi = torch.log2(F.relu(f0 / 440) + 1e-6) * 12 + 69
    midi = midi + semitones
    return 440 * 2 ** ((midi - 69) / 12)


# --------------------------------------------------------------------------- decoder (a12-a20)
def source_net(sd, content, f0, energy):
    """reference module/tinyvc/decoder.py:126-134 -> amps [B,15,T], kernel [B,961,T]."""
    p = "source_net"
    e = F.max_pool1d(energy, HOP, HOP)
    x = (F.conv1d(content, sd[p + ".content_in.weight"], sd[p + ".content_in.bias"])
         + F.conv1d(e, sd[p + ".energy_in.weight"], sd[p + ".energy_in.bias"])
         + F.conv1d(torch.log(F.relu(f0) + 1e-6), sd[p + ".f0_in.weight"], sd[p + ".f0_in.bias"]))
    for i in range(3):
        x = convnext_layer(sd, f"{p}.mid_layers.{i}", x, 1)
    amps = F.elu(F.conv1d(x, sd[p + ".to_amps.weight"], sd[p + ".to_amps.bias"])) + 1.0
    kern = F.elu(F.conv1d(x, sd[p + ".to_kernel.weight"], sd[p + ".to_kernel.bias"])) + 1.0
    return amps, kern


def oscillate_harmonics(f0, num_harmonics=NUM_HARMONICS, fmin=20.0):
    """reference module/tinyvc/decoder.py:24-54 — sines at f0*(1..15); the phase is the running
    sum of f/24000 (ATen's CPU cumsum accumulates fp32 inputs in fp64), wrapped to one cycle,
    gated by the interpolated voiced mask."""
    T = f0.shape[2]
    L = T * HOP
    mul = (torch.arange(num_harmonics + 1) + 1).view(1, -1, 1)
    fs = F.interpolate(f0, L, mode="linear") * mul
    uv = F.interpolate((f0 > fmin).to(f0.dtype), L, mode="linear")
    cyc = torch.cumsum(fs / SAMPLE_RATE, dim=2)
    return torch.sin(2 * math.pi * (cyc % 1)) * uv


def oscillate_noise(kernel, angle):
    """reference module/tinyvc/decoder.py:63-85 with the uniform phase draw made an argument:
    `angle` is what `torch.rand(N, 961, T) * 2*pi - pi` returns there (decoder.py:78)."""
    y = torch.exp(1j * angle) * _fp(kernel)
    y = F.pad(y, [1, 0])
    return torch.istft(y, N_FFT, HOP).unsqueeze(1)


def dsp(f0, amps, kernel, angle):
    """reference module/tinyvc/decoder.py:259-266 -> source [B, 16, L]."""
    harm = oscillate_harmonics(f0) * F.interpolate(amps, scale_factor=HOP, mode="linear")
    return torch.cat([harm, oscillate_noise(kernel, angle)], dim=1)


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def filter_downsample(sd, p, x, factor):
    """reference module/tinyvc/decoder.py:147-157."""
    x = F.interpolate(x, scale_factor=1.0 / factor, mode="linear")
    res = F.conv1d(x, sd[p + ".down_res.weight"], sd[p + ".down_res.bias"])
    h = x
    for name, d in (("c1", 1), ("c2", 2), ("c3", 4)):
        h = conv1d_rep(_lrelu(h), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], d)
    return h + res


def filter_upsample(sd, p, x, c, factor):
    """reference module/tinyvc/decoder.py:173-190."""
    x = F.interpolate(x, scale_factor=factor, mode="linear")
    for (ca, da), (cb, db), film in ((("c1", 1), ("c2", 3), "film1"), (("c3", 9), ("c4", 27), "film2")):
        h = conv1d_rep(_lrelu(x), sd[f"{p}.{ca}.weight"], sd[f"{p}.{ca}.bias"], da)
        h = conv1d_rep(_lrelu(h), sd[f"{p}.{cb}.weight"], sd[f"{p}.{cb}.bias"], db)
        shift = F.conv1d(c, sd[f"{p}.{film}.to_shift.weight"], sd[f"{p}.{film}.to_shift.bias"])
        scale = F.conv1d(c, sd[f"{p}.{film}.to_scale.weight"], sd[f"{p}.{film}.to_scale.bias"])
        x = h * scale + shift + x
    return F.conv1d(x, sd[p + ".c5.weight"], sd[p + ".c5.bias"])


def filter_net(sd, content, f0, energy, source, return_skips=False, return_blocks=False):
    """reference module/tinyvc/decoder.py:222-233 -> [B, 1, L] (return_blocks: also the five Downsample
    outputs `skips` and the five Upsample outputs `ups`, decoder.py:227-232)."""
    p = "filter_net"
    x = (F.conv1d(content, sd[p + ".content_in.weight"], sd[p + ".content_in.bias"])
         + F.conv1d(torch.log(F.relu(f0) + 1e-6), sd[p + ".f0_in.weight"], sd[p + ".f0_in.bias"]))
    s = torch.cat([source, energy], dim=1)
    skips = [conv1d_rep(s, sd[p + ".downs.0.weight"], sd[p + ".downs.0.bias"], 1)]
    down_f = list(reversed(FILTER_FACTORS[1:]))
    for i, f in enumerate(down_f, start=1):
        skips.append(filter_downsample(sd, f"{p}.downs.{i}", skips[-1], f))
    ups = []
    for i, f in enumerate(FILTER_FACTORS):
        x = filter_upsample(sd, f"{p}.ups.{i}", x, skips[len(skips) - 1 - i], f)
        ups.append(x)
    out = conv1d_rep(x, sd[p + ".output_layer.weight"], sd[p + ".output_layer.bias"], 1)
    if return_blocks:
        return out, skips, ups
    if return_skips:
        return out, skips
    return out


def decoder_infer(sd, content, f0, energy, angle):
    """reference module/tinyvc/decoder.py:253-257 -> [B, L]."""
    amps, kern = source_net(sd, content, f0, energy)
    src = dsp(f0, amps, kern, angle)
    return filter_net(sd, content, f0, energy, src).squeeze(1)


# --------------------------------------------------------------------------- orchestration (a21)
def encode(enc_sd, wf):
    """reference module/infer/generator.py:19-23."""
    with torch.inference_mode():
        return encoder_infer(enc_sd, spectrogram(autopad_waveform(wf)))


def convert(enc_sd, dec_sd, wf, tgt, pitch_shift, angle, return_stages=False):
    """reference module/infer/generator.py:26-34.  `angle` [B, 961, T] replaces the in-call
    `torch.rand` (see oscillate_noise)."""
    with torch.inference_mode():
        wf = autopad_waveform(wf)
        spec = spectrogram(wf)
        energy = estimate_energy(wf)
        z, f0 = encoder_infer(enc_sd, spec)
        zm = match_features(z, tgt)
        f0s = shift_frequency(f0, pitch_shift)
        out = decoder_infer(dec_sd, zm, f0s, energy, angle)
        if return_stages:
            return dict(wf=wf, spec=spec, energy=energy, ssl=z, f0=f0, matched=zm, f0s=f0s, wave=out)
        return out


# --------------------------------------------------------------------------- streaming (a22, a23)
def phase_vocoder(a, b, fade_out, fade_in):
    """reference module/infer/stream.py:9-26."""
    n = a.shape[0]
    window = torch.sqrt(fade_out * fade_in)
    fa = torch.fft.rfft(a * window)
    fb = torch.fft.rfft(b * window)
    mag = fa.abs() + fb.abs()
    if n % 2 == 0:
        mag[1:-1] *= 2
    else:
        mag[1:] *= 2
    pa = torch.angle(fa)
    dp = torch.angle(fb) - pa
    dp = dp - 2 * math.pi * torch.floor(dp / 2 / math.pi + 0.5)
    w = 2 * math.pi * torch.arange(n // 2 + 1).to(a) + dp
    t = torch.arange(n).unsqueeze(-1).to(a) / n
    return a * fade_out ** 2 + b * fade_in ** 2 + torch.sum(mag * torch.cos(w * t + pa), -1) * window / n


class StreamState:
    """Buffers of reference StreamInfer (module/infer/stream.py:31-64) for one stream."""

    def __init__(self, block_size=1920, extra_size=0):
        self.block = block_size
        self.cross = 1920
        self.search = 1920
        self.delay = 3840
        self.input_size = max(self.block + self.cross + self.search + 2 * self.delay,
                              self.block + extra_size)
        self.fade_in = torch.sin(math.pi * torch.arange(0, 1, 1 / self.cross) / 2) ** 2
        self.fade_out = 1 - self.fade_in
        self.input_wav = torch.zeros(self.input_size)
        self.sola = torch.zeros(self.cross)


def stream_callback(st, enc_sd, dec_sd, tgt, pitch_shift, block, angle, use_phase_vocoder=False):
    """reference module/infer/stream.py:68-96 -> (out block [1920], sola shift)."""
    with torch.inference_mode():
        st.input_wav = torch.roll(st.input_wav, -st.block)
        st.input_wav[-st.block:] = block
        y = convert(enc_sd, dec_sd, st.input_wav[None], tgt, pitch_shift, angle)[0]
        tmp = y[-st.block - st.cross - st.search - st.delay:-st.delay]
        ci = tmp[None, None, :st.cross + st.search]
        nom = F.conv1d(ci, st.sola[None, None, :])
        den = torch.sqrt(F.conv1d(ci ** 2, torch.ones(1, 1, st.cross)) + 1e-8)
        shift = int(torch.argmax(nom[0, 0] / den[0, 0]))
        tmp = tmp[shift:shift + st.block + st.cross].clone()
        if use_phase_vocoder:
            tmp[:st.cross] = phase_vocoder(st.sola, tmp[:st.cross], st.fade_out, st.fade_in)
        else:
            tmp[:st.cross] = tmp[:st.cross] * st.fade_in + st.sola * st.fade_out
        st.sola = tmp[-st.cross:].clone()
        return tmp[:-st.cross].clone(), shift
