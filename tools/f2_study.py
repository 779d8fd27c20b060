#!/usr/bin/env python3
"""SURVEY.md 8f2 tolerance study: incremental streaming on the CPU, no kernels.

The reference's StreamInfer (module/infer/stream.py:54-57,68-72) recomputes the whole 28-frame window for every 4 new
frames.  The incremental variant studied here keeps, per ConvNeXt layer, the layer's input and its GELU output for the
28 frames of the window; a callback recomputes only the newest `4 + halo` frames of every layer from the (just updated)
cache of the layer below, keeps the older frames as they were computed in earlier callbacks, and takes GRN's L2 norm
over the 28 cached frames of that layer ("running GRN sums").  Exact consistency is impossible to buy cheaply: the
frames that change when 4 frames arrive grow by 3 x dilation per layer (6 -> 9 -> 18 -> 45 > 28 after the SSL trunk's third layer), and
GRN couples every frame to every other one.  So the question is how far a truncated halo is from the full recompute, measured
where it matters: the frames the block's output is cut from (frames 8..20 of the window: stream.py:76-83).

Prints, per halo, the rel-rms of the SSL features, of f0, the share of frames whose 4 nearest index vectors are the same, and the
rel-rms of SourceNet's amps / kernel, against the oracle's full recompute of every window (oracle/ref_cpu.py).  Usage:
  python tools/f2_study.py [--blocks 16] [--halos 0,4,8,12,24]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cpu as R  # noqa: E402
from tinyvc_amd import synth  # noqa: E402

W, S = 28, 4            # window and step in frames (13 440 / 1 920 samples)
OUT = slice(8, 20)      # frames the block's output is cut from


class IncTrunk:
    """input 1x1 -> LN -> ConvNeXt layers -> output 1x1 on a sliding window, newest `S + halo` frames per layer."""

    def __init__(self, sd, p, dilations, halo):
        self.sd, self.p, self.dil, self.halo = sd, p, dilations, halo
        self.x = None          # per layer: its input over the window [C, W]
        self.h = None          # per layer: GELU output over the window [2C, W]

    def _layer_cols(self, l, x, cols):
        """(dw conv k7 -> LN -> c2 -> GELU) of layer l for the window columns `cols`, from the layer input x [1, C, W]."""
        sd, p, d = self.sd, f"{self.p}.mid_layers.{l}", self.dil[l]
        xp = F.pad(x, (3 * d, 3 * d), mode="replicate")
        lo, hi = cols.start, cols.stop
        seg = xp[:, :, lo:hi + 6 * d]                                  # columns lo..hi-1 need lo-3d .. hi-1+3d
        h = F.conv1d(seg, sd[p + ".c1.weight"], sd[p + ".c1.bias"], dilation=d, groups=x.shape[1])
        h = R.layer_norm_c(h, sd[p + ".norm.gamma"], sd[p + ".norm.beta"])
        return F.gelu(F.conv1d(h, sd[p + ".c2.weight"], sd[p + ".c2.bias"]))

    def step(self, x0, first):
        """x0 [1, C, W]: the trunk's normalised input over the window (per-frame, already current).  Returns the trunk output."""
        sd = self.sd
        nl = len(self.dil)
        new = slice(0, W) if first else slice(max(0, W - S - self.halo), W)
        if first:
            self.x, self.h = [None] * (nl + 1), [None] * nl
        else:                                                          # the window slides by S frames: so do the caches
            for l in range(nl):
                self.h[l] = torch.roll(self.h[l], -S, dims=2)
        x = x0
        for l in range(nl):
            p = f"{self.p}.mid_layers.{l}"
            hn = self._layer_cols(l, x, new)
            if first:
                self.h[l] = hn
            else:
                self.h[l][:, :, new] = hn
            h = self.h[l]
            gx = torch.norm(h, p=2, dim=2, keepdim=True)               # GRN over the cached window: old frames keep their old values
            nx = gx / (gx.mean(dim=1, keepdim=True) + 1e-6)
            g = sd[p + ".grn.gamma"] * (h * nx) + sd[p + ".grn.beta"] + h
            y = F.conv1d(g, sd[p + ".c3.weight"], sd[p + ".c3.bias"]) + x
            if first or self.x[l + 1] is None:
                self.x[l + 1] = y
            else:                                                      # older frames of the layer output stay as computed earlier
                keep = torch.roll(self.x[l + 1], -S, dims=2)
                keep[:, :, new] = y[:, :, new]
                self.x[l + 1] = keep
            x = self.x[l + 1]
        return F.conv1d(x, sd[self.p + ".output_layer.weight"], sd[self.p + ".output_layer.bias"]) if (self.p + ".output_layer.weight") in sd else x


def trunk_input(sd, p, spec):
    x = F.conv1d(spec, sd[p + ".input_layer.weight"], sd[p + ".input_layer.bias"])
    return R.layer_norm_c(x, sd[p + ".norm.gamma"], sd[p + ".norm.beta"])


def rel(a, b):
    return float((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--halos", default="0,4,8,12,24")
    args = ap.parse_args()
    torch.set_num_threads(1)
    enc_sd, dec_sd = synth.synth_state_dict("encoder"), synth.synth_state_dict("decoder")
    tgt = synth.synth_index(1000, seed=2)
    wave = synth.synth_wave(1, (args.blocks + 7) * 1920, seed=200)[0]
    print(f"# window {W} frames, step {S}; output frames {OUT.start}..{OUT.stop - 1}; {args.blocks} blocks after the window filled; errors vs the full recompute, rel rms over the output frames")
    print(f"{'halo':>5} {'frames/layer':>12} {'ssl':>10} {'f0':>10} {'kNN same':>9} {'amps':>10} {'kernel':>10}   worst block (ssl)")
    for halo in [int(h) for h in args.halos.split(",")]:
        ssl_t = IncTrunk(enc_sd, "ssl_feature_estimator", R.SSL_DILATIONS, halo)
        pit_t = IncTrunk(enc_sd, "pitch_estimator", (1, 1, 1, 1), halo)
        src_t = IncTrunk(dec_sd, "source_net", (1, 1, 1), halo)
        acc = {k: [] for k in ("ssl", "f0", "knn", "amps", "kern")}
        with torch.inference_mode():
            for blk in range(args.blocks + 1):
                buf = wave[blk * 1920: blk * 1920 + W * 480][None]
                spec = R.spectrogram(buf)                                       # (the spectrogram's own edge frames are recomputed either way: 6 of 28)
                energy = R.estimate_energy(buf)
                # full recompute = the reference
                z_ref, f0_ref = R.encoder_infer(enc_sd, spec)
                _m, idx_ref, _s = R.match_features(z_ref, tgt, return_indices=True)
                zm_ref = R.match_features(z_ref, tgt)
                a_ref, k_ref = R.source_net(dec_sd, zm_ref, f0_ref, energy)
                # incremental
                first = blk == 0
                z = ssl_t.step(trunk_input(enc_sd, "ssl_feature_estimator", spec), first)
                f0 = R.pitch_decode(pit_t.step(trunk_input(enc_sd, "pitch_estimator", spec), first))
                _m, idx, _s = R.match_features(z, tgt, return_indices=True)
                zm = R.match_features(z, tgt)
                e = F.max_pool1d(energy, 480, 480)
                p = "source_net"
                x0 = (F.conv1d(zm, dec_sd[p + ".content_in.weight"], dec_sd[p + ".content_in.bias"])
                      + F.conv1d(e, dec_sd[p + ".energy_in.weight"], dec_sd[p + ".energy_in.bias"])
                      + F.conv1d(torch.log(F.relu(f0) + 1e-6), dec_sd[p + ".f0_in.weight"], dec_sd[p + ".f0_in.bias"]))
                xs = src_t.step(x0, first)
                amps = F.elu(F.conv1d(xs, dec_sd[p + ".to_amps.weight"], dec_sd[p + ".to_amps.bias"])) + 1.0
                kern = F.elu(F.conv1d(xs, dec_sd[p + ".to_kernel.weight"], dec_sd[p + ".to_kernel.bias"])) + 1.0
                if first:
                    continue                                                    # the first window is a full computation by construction
                acc["ssl"].append(rel(z[:, :, OUT], z_ref[:, :, OUT]))
                acc["f0"].append(rel(f0[:, :, OUT], f0_ref[:, :, OUT]))
                acc["knn"].append(float((idx[:, OUT].sort(dim=2).values == idx_ref[:, OUT].sort(dim=2).values).all(dim=2).float().mean()))
                acc["amps"].append(rel(amps[:, :, OUT], a_ref[:, :, OUT]))
                acc["kern"].append(rel(kern[:, :, OUT], k_ref[:, :, OUT]))
        mean = lambda v: sum(v) / len(v)
        print(f"{halo:5d} {min(W, S + halo):12d} {mean(acc['ssl']):10.2e} {mean(acc['f0']):10.2e} {mean(acc['knn']):9.2f} {mean(acc['amps']):10.2e} {mean(acc['kern']):10.2e}   {max(acc['ssl']):.2e}")


if __name__ == "__main__":
    main()
