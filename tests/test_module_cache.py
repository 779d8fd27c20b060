"""The modules cache their parameter list (walking the tree costs ~240 us per convert call); the cache must drop whenever any
Parameter or sub-module anywhere in the tree is replaced, or an owner would keep running - and replaying captured stream graphs -
on stale weights."""
import torch
import torch.nn as nn

from tinyvc_amd import synth
from tinyvc_amd.module.infer import Generator
from tinyvc_amd.module.tinyvc import Decoder, Encoder


def test_weights_key_follows_every_kind_of_replacement():
    gen = Generator(Encoder(), Decoder())
    k = gen._weights_key()
    assert gen._weights_key() == k                                  # stable while nothing changes
    dec2 = Decoder()
    dec2.load_state_dict(synth.synth_state_dict("decoder"))
    gen.decoder = dec2                                              # a sub-module re-assigned on the owner
    k2 = gen._weights_key()
    assert k2 != k
    sd = {n: v.clone() + 1 for n, v in synth.synth_state_dict("encoder").items()}
    gen.encoder.load_state_dict(sd, assign=True)                    # a child loads with assign=True: new Parameter objects
    k3 = gen._weights_key()
    assert k3 != k2
    gen.encoder.pitch_estimator.norm.gamma = nn.Parameter(torch.ones(128))      # a Parameter re-assigned three levels down
    k4 = gen._weights_key()
    assert k4 != k3
    conv = gen.decoder.filter_net.downs[0]                          # ... and on a stock nn.Conv1d leaf
    conv.weight = nn.Parameter(torch.zeros_like(conv.weight))
    k5 = gen._weights_key()
    assert k5 != k4
    with torch.no_grad():
        conv.weight.add_(1.0)                                       # in-place edit: the version counter
    assert gen._weights_key() != k5
