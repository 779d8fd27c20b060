#!/usr/bin/env python3
"""Offline voice conversion on an MI355X: the reference's `infer.py` entry point (same flags,
reference infer.py:17-29) driving the HIP path.

  python infer.py -i ./inputs/ -o ./outputs/ -encp models/encoder.pt -decp models/decoder.pt \\
                  -t target.wav | -idx models/index.pt   [-p SEMITONES] [-d cuda]

Differences from the reference, all at the edges of the path: files are read/written with the
package's own WAV I/O and resampler (torchaudio is not a dependency; ogg/mp3 need a codec and are
skipped with a message); `-d` defaults to `cuda` and must be a GPU (there is no CPU path);
the whole directory is converted in one call (a ragged batch: every file over its own length); resampling to 24 kHz runs on the GPU
(tvc_resample_f32); an index.pt stored in half precision is matched with the fp16 index storage.

Under a launcher (`python -m torch.distributed.run --nproc-per-node N infer.py ...`, WORLD_SIZE > 1) every rank converts ITS share of the
directory on cuda:LOCAL_RANK and writes its own outputs: the files are split by length (longest-processing-time-first on padded samples,
read from the WAV headers alone, tinyvc_amd/parallel.py lpt_split), every rank computes the same split by itself, and no collective runs at
all - utterances are independent (reference infer.py:60-69 is a loop over files).  `--seed S` (extension) makes a file's noise phases a
function of (S, file name) alone, so its output does not depend on the call it rides in or the rank that converts it (default, as in the
reference: an unseeded draw).

`--chunk-size / --buffer-size / --no-chunking`: the reference parses them and then converts every file
whole (infer.py:27-29,40-41,66), so that is the default here too (identical output).  `--chunked` makes
them real: the file is fed through the streaming converter in blocks of --chunk-size
samples with --buffer-size blocks of extra left context, SOLA-aligned and cross-faded (sin^2, or the
phase vocoder with --phase-vocoder) exactly as StreamInfer does, so memory stays bounded for
hour-long inputs.  `--no-chunking True` overrides `--chunked`.
"""
import argparse
import glob
import os
import sys
import zlib

import torch

from tinyvc_amd import audio_io, parallel, spec
from tinyvc_amd.module.infer import Generator
from tinyvc_amd.module.tinyvc import Decoder, Encoder

SAMPLE_RATE = 24000


def build_parser():
    p = argparse.ArgumentParser(description="tinyvc offline conversion (MI355X)")
    p.add_argument("-i", "--inputs", default="./inputs/")
    p.add_argument("-o", "--outputs", default="./outputs/")
    p.add_argument("-encp", "--encoder-path", default="./models/encoder.pt")
    p.add_argument("-decp", "--decoder-path", default="./models/decoder.pt")
    p.add_argument("-f0-est", "--f0-estimation", default="default")
    p.add_argument("-idx", "--index", default="NONE")
    p.add_argument("-t", "--target", default="target.wav")
    p.add_argument("-d", "--device", default="cuda")
    p.add_argument("-p", "--pitch-shift", default=0.0, type=float)
    p.add_argument("-c", "--chunk-size", default=1920, type=int)
    p.add_argument("-b", "--buffer-size", default=4, type=int)
    p.add_argument("-nc", "--no-chunking", default=False, type=bool)
    p.add_argument("--chunked", action="store_true", help="honour --chunk-size / --buffer-size: block-wise conversion with SOLA cross-fades (bounded memory)")
    p.add_argument("--phase-vocoder", action="store_true", help="--chunked: phase-vocoder cross-fade (StreamInfer's use_phase_vocoder) instead of sin^2")
    p.add_argument("--seed", default=None, type=int, help="a file's noise phases become a function of (seed, file name) alone (default: unseeded, as in the reference)")
    return p


def file_angle(gen, device, seed, path, frames):
    """[1, 961, frames] noise phases of one file under --seed: the reference's own draw (decoder.py:78: torch.rand * 2 pi - pi) from a device
    generator seeded by (seed, crc32 of the file's base name) - the same phases whichever call, row or rank converts the file."""
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 0x9E3779B1 + zlib.crc32(os.path.basename(path).encode())) & 0x7FFFFFFFFFFFFFFF)
    return gen.engine(device).noise_angle_from_uniform(torch.rand(1, spec.FFT_BIN, frames, device=device, generator=g))


def my_share(paths, world, rank):
    """This rank's files (indices into `paths`, ascending): longest-processing-time split on the padded 24 kHz sample counts read from the
    headers.  A file whose header cannot be read costs 0 here; whoever gets it reports the error it raises when it is loaded."""
    if world <= 1:
        return list(range(len(paths)))
    costs = []
    for p_ in paths:
        try:
            frames, sr, _ch = audio_io.info(p_)
            n = -(-frames * SAMPLE_RATE // sr)
            costs.append(-(-n // 480) * 480)
        except (OSError, ValueError):
            costs.append(0)
    return parallel.lpt_split(costs, world)[rank]


def load_generator(encoder_path, decoder_path, device):
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(torch.load(encoder_path, map_location="cpu"))
    dec.load_state_dict(torch.load(decoder_path, map_location="cpu"))
    return Generator(enc.eval(), dec.eval()).to(device)


def load_target(gen, args, device):
    """Speaker target: features of a target utterance, or a prebuilt index.pt [1, 768, N]."""
    if args.index == "NONE":
        wf, sr = audio_io.load(args.target)
        wf = gen.engine(device).resample(wf.to(device), sr, SAMPLE_RATE)
        tgt, _f0 = gen.encode(wf.mean(dim=0, keepdim=True) if wf.shape[0] > 1 else wf)
        return tgt
    return torch.load(args.index, map_location="cpu").to(device)     # fp32 [1,768,N], or half: fp16 index storage


SOLA_MAX_LATENCY = 1920 + 1920 + 3840     # cross-fade + search range + last delay: a block's output lags its input by this minus the SOLA lag (stream.py:76-83)


@torch.no_grad()
def convert_chunked(gen, batch, tgt, pitch_shift, chunk_size, buffer_blocks, use_phase_vocoder=False, noise_angles=None, return_blocks=False):
    """Block-wise conversion of `batch` [B, L] with bounded memory: every utterance is a stream of `chunk_size`-sample
    blocks through BatchedStreamInfer (rolling buffer with `buffer_blocks` blocks of extra context, full convert per
    block, SOLA alignment + cross-fade: reference module/infer/stream.py:68-96).  The streaming output lags its input by
    SOLA_MAX_LATENCY minus the SOLA lag of the block (the lag locks onto one value per stream once the signal is periodic);
    the tail is flushed with silence and every utterance is trimmed by its own median latency, so the result has the input's
    length and lines up with the whole-file conversion.
    `noise_angles(i)` -> [B, 961, T] injects the decoder's noise phases of block i (parity runs against the oracle's
    `stream_callback`; default: drawn on the device per block, as the reference's `torch.rand` is).
    `return_blocks`: also return the untrimmed block outputs [B, nblk, chunk_size] and the SOLA lags [nblk, B]."""
    from tinyvc_amd.module.infer import BatchedStreamInfer
    B, L = batch.shape
    st = BatchedStreamInfer(gen, n_streams=B, target=tgt, pitch_shift=pitch_shift, device=batch.device, block_size=chunk_size,
                            extra_size=buffer_blocks * chunk_size, use_phase_vocoder=use_phase_vocoder, use_graph=True)
    st.init_buffer()
    nblk = -(-(L + SOLA_MAX_LATENCY) // chunk_size)
    padded = torch.zeros(B, nblk * chunk_size, device=batch.device)
    padded[:, :L] = batch
    outs, lags = [], []
    for i in range(nblk):
        outs.append(st.audio_callback(padded[:, i * chunk_size:(i + 1) * chunk_size], noise_angle=noise_angles(i) if noise_angles else None))
        lags.append(st.last_shift.clone())
    lags = torch.stack(lags, dim=0)                                  # [nblk, B]
    stream = torch.cat(outs, dim=1)
    nsig = max(1, -(-L // chunk_size))                               # blocks that carry signal (the flush blocks' lags are noise)
    latency = (SOLA_MAX_LATENCY - lags[:nsig].float().median(dim=0).values.round().long()).clamp(0, SOLA_MAX_LATENCY).tolist()
    out = torch.stack([stream[b, latency[b]:latency[b] + L] for b in range(B)], dim=0)
    if return_blocks:
        return out, torch.stack(outs, dim=1), lags
    return out


def main(argv=None, world=None, rank=None, local_rank=None):
    """`world / rank / local_rank` default to the launcher's environment (WORLD_SIZE / RANK / LOCAL_RANK); tests pass them to run one
    rank's share in-process."""
    args = build_parser().parse_args(argv)
    env = parallel.dist_env()
    world, rank, local_rank = (env[0] if world is None else world), (env[1] if rank is None else rank), (env[2] if local_rank is None else local_rank)
    device = torch.device(args.device)
    if device.type != "cuda":
        sys.exit("infer.py: this build runs on an AMD GPU only; pass -d cuda (the reference's CPU path is not part of it)")
    if world > 1 and device.index is None:
        device = torch.device("cuda", local_rank)            # one process per GPU
    if device.index is not None:
        torch.cuda.set_device(device)
    gen = load_generator(args.encoder_path, args.decoder_path, device)
    tgt = load_target(gen, args, device)
    os.makedirs(args.outputs, exist_ok=True)

    paths = []
    for ext in ("wav", "ogg", "mp3"):
        paths += sorted(glob.glob(os.path.join(args.inputs, "*." + ext)))
    for path in paths:
        if not path.lower().endswith(".wav") and rank == 0:
            print(f"Skipping {path}: no decoder for this container in this build")
    paths = [p_ for p_ in paths if p_.lower().endswith(".wav")]
    paths = [paths[i] for i in my_share(paths, world, rank)]        # WORLD_SIZE > 1: this rank's files; nothing is exchanged between ranks
    tag = f"[rank {rank}/{world}] " if world > 1 else ""
    jobs = []
    for path in paths:
        wf, sr = audio_io.load(path)
        wf = gen.engine(device).resample(wf.to(device), sr, SAMPLE_RATE).mean(dim=0, keepdim=True)       # mono, 24 kHz (infer.py:63-64), on the GPU
        jobs.append((path, wf))

    # files too short for the STFT's reflect padding (torch.stft raises on them in the reference) are reported, not fatal for the rest
    short = [p for p, wf in jobs if wf.shape[1] <= 960]
    for p in short:
        print(f"Skipping {p}: {960} samples or fewer @24 kHz (the STFT needs more)")
    jobs = [(p, wf) for p, wf in jobs if wf.shape[1] > 960]
    if not jobs:
        return 0
    lengths = [wf.shape[1] for _p, wf in jobs]
    print(f"{tag}Converting {len(jobs)} file(s), {min(lengths)} .. {max(lengths)} samples ...")
    outs = [None] * len(jobs)
    if args.chunked and not args.no_chunking:
        for length in sorted(set(lengths)):      # streams of one group advance in lock step: chunked mode batches equal lengths
            rows = [i for i, n in enumerate(lengths) if n == length]
            batch = torch.cat([jobs[i][1][:, :length] for i in rows], dim=0)
            angles = None
            if args.seed is not None:      # one generator per file, drawn from block after block: a stream's phases do not depend on its neighbours
                gens = []
                for i in rows:
                    g = torch.Generator(device=device)
                    g.manual_seed((int(args.seed) * 0x9E3779B1 + zlib.crc32(os.path.basename(jobs[i][0]).encode())) & 0x7FFFFFFFFFFFFFFF)
                    gens.append(g)
                tb = max(args.chunk_size + 1920 + 1920 + 2 * 3840, args.chunk_size + args.buffer_size * args.chunk_size) // 480      # frames of a stream's rolling buffer (BatchedStreamInfer.input_size, stream.py:52-53)
                angles = lambda _i: gen.engine(device).noise_angle_from_uniform(      # noqa: E731
                    torch.cat([torch.rand(1, spec.FFT_BIN, tb, device=device, generator=g) for g in gens], dim=0))
            o = convert_chunked(gen, batch, tgt, args.pitch_shift, args.chunk_size, args.buffer_size, args.phase_vocoder, noise_angles=angles).cpu()
            for i, y in zip(rows, o):
                outs[i] = y
    else:
        # Ragged batches (every file converted over its own length, exactly as if it were alone; the reference's loop converts them one by
        # one, infer.py:60-66).  Files are taken in order of length and cut into calls whose PADDED size stays under a budget, so memory is
        # bounded by the budget - not by (number of files) x (longest file) - and the padding inside a call stays small.
        order = sorted(range(len(jobs)), key=lambda i: lengths[i])
        budget = int(os.environ.get("TVC_INFER_BATCH_SAMPLES", 32 * 1024 * 1024))      # padded samples per call (x 4 B in, x 4 B out, x 8 for the noise phases)
        start = 0
        while start < len(order):
            end = start + 1
            while end < len(order) and (end + 1 - start) * (-(-lengths[order[end]] // 480) * 480) <= budget:
                end += 1
            rows = order[start:end]
            lens = [lengths[i] for i in rows]
            Lmax = -(-max(lens) // 480) * 480
            batch = torch.zeros(len(rows), Lmax, device=device)
            for r, i in enumerate(rows):
                batch[r, :lengths[i]] = jobs[i][1][0]
            angle = None
            if args.seed is not None:      # per-file phases in the padded [rows, 961, Tmax] layout the ragged call takes
                angle = torch.zeros(len(rows), spec.FFT_BIN, Lmax // 480, device=device)
                for r, i in enumerate(rows):
                    f = -(-lengths[i] // 480)
                    angle[r, :, :f] = file_angle(gen, device, args.seed, jobs[i][0], f)[0]
            if len(set(lens)) == 1:
                out = gen.convert(batch[:, :lens[0]], tgt, args.pitch_shift, noise_angle=angle).cpu()
            else:
                out = gen.convert(batch, tgt, args.pitch_shift, noise_angle=angle, lengths=lens).cpu()
            for r, i in enumerate(rows):
                outs[i] = out[r, :-(-lengths[i] // 480) * 480]
            del batch, out
            start = end
    for (path, _wf), y in zip(jobs, outs):
        name = os.path.splitext(os.path.basename(path))[0]
        audio_io.save(os.path.join(args.outputs, f"{name}.wav"), y[None], SAMPLE_RATE)
    return 0


if __name__ == "__main__":
    sys.exit(main())
