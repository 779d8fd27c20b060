"""Checkpoint contract of the tinyvc inference path.

`encoder_spec()` / `decoder_spec()` return the ordered `{state_dict key: shape}` maps that
`encoder.pt` / `decoder.pt` of the reference hold (constructors at reference
`module/tinyvc/encoder.py:100-106`, `module/tinyvc/decoder.py:236-251`; hyper-parameters are the
constructor defaults there).  The maps are built from the architecture constants below, not copied
from a checkpoint; `tests/test_spec.py` compares them with the key/shape list captured from the
reference (`tests/golden/state_dict_spec.json`).
"""
from collections import OrderedDict

SAMPLE_RATE = 24000
N_FFT = 1920
HOP = 480                      # frame_size
FFT_BIN = N_FFT // 2 + 1       # 961
SSL_DIM = 768
SSL_CH = 384
SSL_DILATIONS = (1, 3, 9, 1, 1, 1)
PITCH_CH = 128
PITCH_LAYERS = 4
PITCH_CLASSES = 512
PITCH_CPO = 48                 # classes per octave
PITCH_FMIN = 20.0
SRC_CH = 128
SRC_LAYERS = 3
NUM_HARMONICS = 14             # oscillator emits NUM_HARMONICS + 1 sinusoids
FILTER_CHANNELS = (384, 192, 96, 48, 24)
FILTER_FACTORS = (2, 3, 4, 4, 5)


def _conv(d, name, cout, cin, k, groups=1):
    d[name + ".weight"] = (cout, cin // groups, k)
    d[name + ".bias"] = (cout,)


def _convnext(d, name, ch, mul=2, k=7):
    _conv(d, name + ".c1", ch, ch, k, groups=ch)
    d[name + ".norm.gamma"] = (ch,)
    d[name + ".norm.beta"] = (ch,)
    _conv(d, name + ".c2", ch * mul, ch, 1)
    d[name + ".grn.beta"] = (1, ch * mul, 1)
    d[name + ".grn.gamma"] = (1, ch * mul, 1)
    _conv(d, name + ".c3", ch, ch * mul, 1)


def encoder_spec():
    d = OrderedDict()
    p = "ssl_feature_estimator"
    _conv(d, p + ".input_layer", SSL_CH, FFT_BIN, 1)
    d[p + ".norm.gamma"] = (SSL_CH,)
    d[p + ".norm.beta"] = (SSL_CH,)
    for i in range(len(SSL_DILATIONS)):
        _convnext(d, f"{p}.mid_layers.{i}", SSL_CH)
    _conv(d, p + ".output_layer", SSL_DIM, SSL_CH, 1)
    p = "pitch_estimator"
    _conv(d, p + ".input_layer", PITCH_CH, FFT_BIN, 1)
    d[p + ".norm.gamma"] = (PITCH_CH,)
    d[p + ".norm.beta"] = (PITCH_CH,)
    for i in range(PITCH_LAYERS):
        _convnext(d, f"{p}.mid_layers.{i}", PITCH_CH)
    _conv(d, p + ".output_layer", PITCH_CLASSES, PITCH_CH, 1)
    return d


def filter_down_plan():
    """(cin, cout, factor) of downs[1:] (reference decoder.py:207-211)."""
    cs = list(reversed(FILTER_CHANNELS[1:]))
    ns = cs[1:] + [FILTER_CHANNELS[0]]
    fs = list(reversed(FILTER_FACTORS[1:]))
    return list(zip(cs, ns, fs))


def filter_up_plan():
    """(cin, cout, factor) of ups[0:] (reference decoder.py:214-219); cond channels == cin."""
    cs = list(FILTER_CHANNELS)
    ns = list(FILTER_CHANNELS[1:]) + [FILTER_CHANNELS[-1]]
    return list(zip(cs, ns, FILTER_FACTORS))


def decoder_spec():
    d = OrderedDict()
    p = "source_net"
    _conv(d, p + ".content_in", SRC_CH, SSL_DIM, 1)
    _conv(d, p + ".energy_in", SRC_CH, 1, 1)
    _conv(d, p + ".f0_in", SRC_CH, 1, 1)
    for i in range(SRC_LAYERS):
        _convnext(d, f"{p}.mid_layers.{i}", SRC_CH)
    _conv(d, p + ".to_amps", NUM_HARMONICS + 1, SRC_CH, 1)
    _conv(d, p + ".to_kernel", FFT_BIN, SRC_CH, 1)
    p = "filter_net"
    _conv(d, p + ".content_in", FILTER_CHANNELS[0], SSL_DIM, 1)
    _conv(d, p + ".f0_in", FILTER_CHANNELS[0], 1, 1)
    _conv(d, p + ".downs.0", FILTER_CHANNELS[-1], NUM_HARMONICS + 3, 3)
    for i, (c, n, _f) in enumerate(filter_down_plan(), start=1):
        q = f"{p}.downs.{i}"
        _conv(d, q + ".down_res", n, c, 1)
        _conv(d, q + ".c1", c, c, 3)
        _conv(d, q + ".c2", c, c, 3)
        _conv(d, q + ".c3", n, c, 3)
    for i, (c, n, _f) in enumerate(filter_up_plan()):
        q = f"{p}.ups.{i}"
        _conv(d, q + ".c1", c, c, 3)
        _conv(d, q + ".c2", c, c, 3)
        _conv(d, q + ".film1.to_shift", c, c, 1)
        _conv(d, q + ".film1.to_scale", c, c, 1)
        _conv(d, q + ".c3", c, c, 3)
        _conv(d, q + ".c4", c, c, 3)
        _conv(d, q + ".film2.to_shift", c, c, 1)
        _conv(d, q + ".film2.to_scale", c, c, 1)
        _conv(d, q + ".c5", n, c, 1)
    _conv(d, p + ".output_layer", 1, FILTER_CHANNELS[-1], 7)
    return d
