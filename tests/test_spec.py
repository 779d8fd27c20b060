"""The checkpoint contract: our key/shape map equals the reference's state_dict (captured to
tests/golden/state_dict_spec.json by tools/gen_golden.py), and the synthetic formula fills it."""
import json
import os

from tinyvc_amd import spec, synth
from helpers import GOLDEN


def test_spec_matches_reference_state_dict():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    for which, fn in (("encoder", spec.encoder_spec), ("decoder", spec.decoder_spec)):
        ours = {k: list(v) for k, v in fn().items()}
        assert list(ours.keys()) == list(ref[which].keys())
        assert ours == ref[which]


def test_param_counts():
    n = lambda d: sum(int(__import__("math").prod(s)) for s in d.values())
    assert n(spec.encoder_spec()) == 4704256
    assert n(spec.decoder_spec()) == 4660553


def test_synth_is_deterministic():
    a = synth.synth_state_dict("decoder", 0)
    b = synth.synth_state_dict("decoder", 0)
    assert all((a[k] == b[k]).all() for k in a)
    assert (synth.synth_wave(1, 1000, 5) == synth.synth_wave(1, 1000, 5)).all()
