"""Shared test helpers: golden fixture loading and formula-regenerated inputs."""
import os

import numpy as np
import torch

from tinyvc_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def convert_inputs(g):
    """Inputs of a `convert_*` fixture, rebuilt from the seeds it records."""
    wf = synth.synth_wave(int(g["batch"]), int(g["wave_len"]), seed=int(g["wave_seed"]))
    tgt = synth.synth_index(int(g["index_size"]), seed=int(g["index_seed"]))
    frames = g["wave"].shape[1] // 480
    angle = synth.synth_angle(int(g["batch"]), frames, int(g["noise_seed"]))
    return wf, tgt, float(g["pitch_shift"]), angle


class oracle_one_thread:
    """The oracle's waveform depends on the host's thread count (oneDNN / MKL reduction orders: 2e-4 rms between 1 and 128
    threads at T = 200 on the GPU box's host).  Live comparisons run it on ONE thread - sequential
    reductions, the reproducible setting (the committed fixtures came from the 8-thread build host)."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *exc):
        torch.set_num_threads(self.n)


_SD = {}


def state_dicts(seed=0):
    if seed not in _SD:
        _SD[seed] = (synth.synth_state_dict("encoder", seed), synth.synth_state_dict("decoder", seed))
    return _SD[seed]


def rms(a):
    a = torch.as_tensor(a).double()
    return float(torch.sqrt((a * a).mean()))


def rel_rms(a, ref):
    return rms(torch.as_tensor(a).double() - torch.as_tensor(ref).double()) / max(rms(ref), 1e-30)


def stage(g, name):
    """(tensor, time stride) of a stored stage output.  Round-1 fixtures keep the encoder-side tensors whole and
    decimate the full-rate ones by `decim`; the headline-length fixtures (convert_cfg*) store `<name>_d` with its own
    `<name>_stride` (tools/gen_golden.py)."""
    if name in g:
        return torch.from_numpy(np.asarray(g[name])), 1
    if name + "_stride" in g:
        return torch.from_numpy(np.asarray(g[name + "_d"])), int(g[name + "_stride"])
    dm = int(g["decim"])
    if name.startswith("skip"):
        st = dm if int(name[4:]) < 2 else 1
    elif name.startswith("up"):
        st = dm if int(name[2:]) >= 3 else 1
    else:
        st = dm
    return torch.from_numpy(np.asarray(g[name + "_d"])), st
