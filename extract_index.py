#!/usr/bin/env python3
"""Build a kNN speaker index (`index.pt`, FloatTensor [1, 768, size]) from a folder of target-speaker
audio with the GPU encoder: the reference's `extract_index.py` (same output format, reference
extract_index.py:43-58): encoder features of each clip, every `--stride`-th frame, concatenated,
randomly permuted along time, truncated to `-size`.

  python extract_index.py --dataset-cache DIR -encp models/encoder.pt -size 2048 -o models/index.pt

DIR holds `*.wav` clips (the reference's preprocess.py cache of `{i}.wav` files works as is).

Under a launcher (`python -m torch.distributed.run --nproc-per-node N extract_index.py ...`, WORLD_SIZE > 1) the clips the index needs
are encoded by N GPUs: the shuffled file order and the prefix of it that fills `-size` vectors are computed by every rank from the WAV
headers alone (a clip of n samples at 24 kHz contributes ceil(ceil(n / 480) / stride) vectors), the prefix is split over the ranks by
length (tinyvc_amd/parallel.py lpt_split), each rank encodes its clips on cuda:LOCAL_RANK, and ONE gather (RCCL) brings the features to
rank 0, which assembles them in the shuffled order, permutes, truncates and writes the file - the same bytes as the single-GPU run with
the same `--seed`.
"""
import argparse
import glob
import os
import sys

import torch

from tinyvc_amd import audio_io, parallel
from tinyvc_amd.module import utils
from tinyvc_amd.module.tinyvc import Encoder

SAMPLE_RATE = 24000


def encode_clip(enc, device, path, stride):
    """[1, 768, ceil(T / stride)] on the CPU: the encoder's features of one clip, every `stride`-th frame (extract_index.py:47-52)."""
    wf, sr = audio_io.load(path)
    wf = enc.engine(device).resample(wf.to(device), sr, SAMPLE_RATE).mean(dim=0, keepdim=True)
    spec = utils.spectrogram(utils.autopad_waveform(wf), enc.n_fft, enc.hop_size)
    z, _f0 = enc.infer(spec)
    return z.cpu()[:, :, ::stride]


def clip_columns(path, stride, engine=None):
    """Index vectors a clip will contribute, from its header alone: frames at 24 kHz (the resampler's output length), padded to whole
    480-sample frames, every `stride`-th one."""
    frames, sr, _ch = audio_io.info(path)
    if sr != SAMPLE_RATE:
        frames = engine.lib.tvc_resample_out_len(frames, sr, SAMPLE_RATE) if engine is not None else -(-frames * SAMPLE_RATE // sr)
    t = -(-frames // 480)
    return -(-t // stride)


def needed_prefix(cols, order, size):
    """The reference's loop (extract_index.py:47-55) takes clips in shuffled order until MORE than `size` vectors are collected: the number
    of clips of `order` it ends up using."""
    total = 0
    for k, i in enumerate(order):
        total += cols[i]
        if total > size:
            return k + 1
    return len(order)


def sharded_features(cols, order, size, world, rank, encode, device, group=None):
    """The clips of the shuffled order's needed prefix, encoded by `world` ranks and gathered on rank 0 (the job's one exchange): returns
    the features in prefix order there, None elsewhere.  cols[i] = vectors clip i will contribute (header-derived: every rank computes the
    same prefix and the same split on its own), encode(i) -> [1, 768, cols[i]] on the CPU."""
    import torch.distributed as dist
    prefix = order[:needed_prefix(cols, order, size)]
    split = parallel.lpt_split([cols[i] for i in prefix], world)             # positions in `prefix`, per rank
    local = [encode(prefix[k]) for k in split[rank]]
    for k, z in zip(split[rank], local):
        if tuple(z.shape) != (1, 768, cols[prefix[k]]):
            raise RuntimeError(f"clip {prefix[k]}: features {tuple(z.shape)}, its header promised {cols[prefix[k]]} vectors")
    share = [sum(cols[prefix[k]] for k in split[r]) for r in range(world)]
    buf = torch.zeros(768, max(max(share), 1), device=device)
    if local:
        buf[:, :share[rank]] = torch.cat(local, dim=2)[0].to(device)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0, group=group)
    if rank != 0:
        return None
    feats = [None] * len(prefix)
    for r in range(world):
        off = 0
        for k in split[r]:
            n = cols[prefix[k]]
            feats[k] = parts[r][None, :, off:off + n].cpu()
            off += n
    return feats


def assemble(feats_in_order, size, gen, half):
    feats = torch.cat(feats_in_order, dim=2)
    perm = torch.randperm(feats.shape[2], generator=gen)
    tgt = feats.index_select(2, perm)[:, :, :size].contiguous()
    return tgt.half() if half else tgt


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
    p.add_argument("--force-dist", action="store_true", help="run the WORLD_SIZE > 1 path (header-derived prefix, split, RCCL gather) even at world size 1")
    args = p.parse_args(argv)

    world, rank, local_rank = parallel.dist_env()
    sharded = world > 1 or args.force_dist
    device = torch.device(args.device)
    if sharded and device.type == "cuda" and device.index is None:
        device = torch.device("cuda", local_rank)
    if device.type == "cuda" and device.index is not None:
        torch.cuda.set_device(device)
    enc = Encoder()
    enc.load_state_dict(torch.load(args.encoder_path, map_location="cpu"))
    enc = enc.eval().to(device)
    files = sorted(glob.glob(os.path.join(args.dataset_cache, "*.wav")))
    if not files:
        sys.exit(f"no *.wav under {args.dataset_cache}")

    if not sharded:
        gen = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
        order = torch.randperm(len(files), generator=gen).tolist()      # DataLoader(shuffle=True) in the reference
        feats, total = [], 0
        for i in order:
            z = encode_clip(enc, device, files[i], args.stride)
            feats.append(z)
            total += z.shape[2]
            if total > args.size:
                break
        tgt = assemble(feats, args.size, gen, args.half)
    else:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        own_port = "MASTER_PORT" not in os.environ
        if own_port:                                                     # --force-dist without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        own_group = not dist.is_initialized()
        if own_group:
            dist.init_process_group("nccl" if device.type == "cuda" else "gloo", rank=rank, world_size=world,
                                    **({"device_id": device} if device.type == "cuda" else {}))
        try:
            seed = torch.tensor([args.seed if args.seed is not None else int(torch.seed() & 0x7FFFFFFF)], dtype=torch.int64, device=device)
            dist.broadcast(seed, src=0)                                  # an unseeded run: rank 0's draw is everybody's
            gen = torch.Generator().manual_seed(int(seed.item()))
            order = torch.randperm(len(files), generator=gen).tolist()
            eng = enc.engine(device) if device.type == "cuda" else None
            cols = [clip_columns(f, args.stride, eng) for f in files]
            feats = sharded_features(cols, order, args.size, world, rank, lambda i: encode_clip(enc, device, files[i], args.stride), device)
            tgt = assemble(feats, args.size, gen, args.half) if rank == 0 else None
            dist.barrier()
        finally:
            if own_group:
                dist.destroy_process_group()
            if own_port:
                os.environ.pop("MASTER_PORT", None)
        if rank != 0:
            return 0
    print(f"Extracted {tgt.shape[2]} vectors")
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    torch.save(tgt, args.output)
    return 0


if __name__ == "__main__":
    sys.exit(main())
