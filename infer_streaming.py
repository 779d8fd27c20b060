#!/usr/bin/env python3
"""Real-time voice conversion on an MI355X: the reference's `infer_streaming.py` entry point
(same flags, reference infer_streaming.py:18-35) on the HIP path.

With `pyaudio` installed and audio devices present it runs the reference's capture -> convert ->
playback loop.  GPU nodes have neither, so the same loop can be driven from files:

  python infer_streaming.py -encp enc.pt -decp dec.pt -idx index.pt \\
         --input-wav in.wav --output-wav out.wav [--streams 32]

`--streams S` feeds S copies of the input as S concurrent streams through one BatchedStreamInfer
(one batched convert + one SOLA launch per block) and reports the p50 / p95 block latency against the
real-time budget (chunk / 24 kHz = 80 ms for the default 1920-sample chunk).
"""
import argparse
import sys
import time

import numpy as np
import torch

from tinyvc_amd import audio_io
from tinyvc_amd.module.infer import BatchedStreamInfer, Generator
from tinyvc_amd.module.tinyvc import Decoder, Encoder


def build_parser():
    p = argparse.ArgumentParser(description="realtime inference")
    p.add_argument("-encp", "--encoder-path", default="./models/encoder.pt")
    p.add_argument("-decp", "--decoder-path", default="./models/decoder.pt")
    p.add_argument("-i", "--input", default=0, type=int)
    p.add_argument("-o", "--output", default=0, type=int)
    p.add_argument("-l", "--loopback", default=-1, type=int)
    p.add_argument("-idx", "--index", default="NONE")
    p.add_argument("-p", "--pitch-shift", default=0, type=float)
    p.add_argument("-t", "--target", default="target.wav")
    p.add_argument("-c", "--chunk", default=1920, type=int)
    p.add_argument("-e", "--extra", default=3840, type=int)
    p.add_argument("-d", "--device", default="cuda")
    p.add_argument("-sr", "--sample-rate", default=24000, type=int)
    p.add_argument("-ig", "--input-gain", default=0, type=float)
    p.add_argument("-og", "--output-gain", default=0, type=float)
    p.add_argument("-f0-est", "--f0-estimation", default="default", choices=["default", "fcpe", "dio", "harvest"])
    # file-driven operation (no audio devices on a GPU node)
    p.add_argument("--input-wav", default=None, help="read the microphone signal from this WAV instead of a device")
    p.add_argument("--output-wav", default=None, help="write the converted stream here")
    p.add_argument("--streams", default=1, type=int, help="number of concurrent streams to simulate")
    return p


def int16_blocks_from_wav(path, sample_rate, chunk, engine):
    """What `stream_input.read(chunk)` yields in the reference: int16 mono blocks at `sample_rate` (the file is resampled on the device,
    tvc_resample_f32, like every other resampling of the entry scripts; tinyvc_amd/resample.py is that kernel's host checker, not a second path)."""
    wf, sr = audio_io.load(path)
    wf = engine.resample(wf.mean(dim=0, keepdim=True).to(engine.device), sr, sample_rate)[0].cpu()
    pcm = (wf.clamp(-1, 1) * 32767).to(torch.int16).numpy()
    n = len(pcm) // chunk
    return [pcm[i * chunk:(i + 1) * chunk] for i in range(n)]


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = torch.device(args.device)
    if device.type != "cuda":
        sys.exit("infer_streaming.py: this build runs on an AMD GPU only; pass -d cuda")
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(torch.load(args.encoder_path, map_location="cpu"))
    dec.load_state_dict(torch.load(args.decoder_path, map_location="cpu"))
    gen = Generator(enc.eval(), dec.eval()).to(device)
    S = max(1, args.streams)
    stream = BatchedStreamInfer(gen, n_streams=S, pitch_shift=args.pitch_shift, block_size=args.chunk, device=device,
                                extra_size=args.extra, f0_estimation=args.f0_estimation)
    if args.index == "NONE":
        wf, sr = audio_io.load(args.target)
        wf = gen.engine(device).resample(wf.to(device), sr, 24000)
        tgt, _ = gen.encode(wf.mean(dim=0, keepdim=True))
    else:
        tgt = torch.load(args.index, map_location="cpu").to(device)
    stream.target = tgt
    stream.init_buffer()

    def process(chunk_i16):
        """One pass of the reference's loop body (infer_streaming.py:84-94) for S streams."""
        eng = gen.engine(device)
        x = eng.pcm16_to_f32(torch.from_numpy(np.ascontiguousarray(chunk_i16)).to(device), args.input_gain)      # / 32768, gain: on the GPU
        y = stream.audio_callback(x.expand(S, -1) if x.dim() == 1 else x)
        return eng.f32_to_pcm16(y, args.output_gain).cpu().numpy()                                                # gain, * 32768, int16: on the GPU

    if args.input_wav is None:
        try:
            import pyaudio
        except ImportError:
            sys.exit("pyaudio is not installed: pass --input-wav / --output-wav for file-driven streaming")
        audio = pyaudio.PyAudio()
        sin = audio.open(format=pyaudio.paInt16, rate=args.sample_rate, channels=1, input_device_index=args.input, input=True)
        sout = audio.open(format=pyaudio.paInt16, rate=args.sample_rate, channels=1, output_device_index=args.output, output=True)
        sloop = audio.open(format=pyaudio.paInt16, rate=args.sample_rate, channels=1, output_device_index=args.loopback, output=True) if args.loopback != -1 else None
        print("Converting voice, Ctrl+C to stop conversion")
        while True:
            chunk = np.frombuffer(sin.read(args.chunk), dtype=np.int16)
            out = process(chunk)[0].tobytes()
            sout.write(out)
            if sloop is not None:
                sloop.write(out)

    blocks = int16_blocks_from_wav(args.input_wav, args.sample_rate, args.chunk, gen.engine(device))
    outs, lat = [], []
    for blk in blocks:
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        o = process(blk)
        lat.append(time.perf_counter() - t0)      # includes the device -> host copy of the block, like the reference loop
        outs.append(o[0])
    if args.output_wav:
        pcm = np.concatenate(outs).astype(np.float32) / 32768
        audio_io.save(args.output_wav, torch.from_numpy(pcm)[None], args.sample_rate)
    if len(lat) > 3:
        l = np.sort(np.array(lat[3:])) * 1e3
        budget = args.chunk / args.sample_rate * 1e3
        print(f"{S} stream(s), {len(lat)} blocks: p50 {l[len(l) // 2]:.2f} ms, p95 {l[int(len(l) * 0.95)]:.2f} ms per block (real-time budget {budget:.0f} ms)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
