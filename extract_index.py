#!/usr/bin/env python3
"""Build a kNN speaker index (`index.pt`, FloatTensor [1, 768, size]) from a folder of target-speaker
audio with the GPU encoder: the reference's `extract_index.py` (same output format, reference
extract_index.py:43-58): encoder features of each clip, every `--stride`-th frame, concatenated,
randomly permuted along time, truncated to `-size`.

  python extract_index.py --dataset-cache DIR -encp models/encoder.pt -size 2048 -o models/index.pt

DIR holds `*.wav` clips (the reference's preprocess.py cache of `{i}.wav` files works as is).
"""
import argparse
import glob
import os
import sys

import torch

from tinyvc_amd import audio_io
from tinyvc_amd.module import utils
from tinyvc_amd.module.tinyvc import Encoder


def main(argv=None):
    p = argparse.ArgumentParser(description="extract index")
    p.add_argument("--dataset-cache", default="dataset_cache")
    p.add_argument("-encp", "--encoder-path", default="models/encoder.pt")
    p.add_argument("-size", default=2048, type=int)
    p.add_argument("-o", "--output", default="models/index.pt")
    p.add_argument("-d", "--device", default="cuda")
    p.add_argument("--stride", default=4, type=int)
    p.add_argument("--seed", default=None, type=int, help="fix the shuffle (the reference does not seed it)")
    p.add_argument("--half", action="store_true", help="store the index in half precision (matched with the fp16 index storage: 2 B per element)")
    args = p.parse_args(argv)

    device = torch.device(args.device)
    enc = Encoder()
    enc.load_state_dict(torch.load(args.encoder_path, map_location="cpu"))
    enc = enc.eval().to(device)
    files = sorted(glob.glob(os.path.join(args.dataset_cache, "*.wav")))
    if not files:
        sys.exit(f"no *.wav under {args.dataset_cache}")
    gen = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
    order = torch.randperm(len(files), generator=gen).tolist()      # DataLoader(shuffle=True) in the reference
    feats, total = [], 0
    for i in order:
        wf, sr = audio_io.load(files[i])
        wf = enc.engine(device).resample(wf.to(device), sr, 24000).mean(dim=0, keepdim=True)
        spec = utils.spectrogram(utils.autopad_waveform(wf), enc.n_fft, enc.hop_size)
        z, _f0 = enc.infer(spec)
        z = z.cpu()[:, :, ::args.stride]
        feats.append(z)
        total += z.shape[2]
        if total > args.size:
            break
    feats = torch.cat(feats, dim=2)
    perm = torch.randperm(feats.shape[2], generator=gen)
    tgt = feats.index_select(2, perm)[:, :, :args.size].contiguous()
    if args.half:
        tgt = tgt.half()
    print(f"Extracted {tgt.shape[2]} vectors")
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    torch.save(tgt, args.output)
    return 0


if __name__ == "__main__":
    sys.exit(main())
