#!/usr/bin/env python3
"""Streaming latency (BASELINE.json configs[2]): S concurrent real-time streams on one MI355X, each
delivering 1920-sample blocks (80 ms @24 kHz) into StreamInfer's 13 440-sample rolling buffer
(block 1920, extra 3840: reference infer_streaming.py defaults).  One step = one block for every stream
= one batched convert [S, 13440] + one SOLA launch.  Reports p50/p95 wall latency per block with the
blocks already on the device (host<->device copies of 1920 int16 samples are not included).

  python bench_stream.py [--streams 32 --blocks 220 --warmup 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tinyvc_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--blocks", type=int, default=220)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--index", type=int, default=1000)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one HIP-graph replay per block")
    ap.add_argument("--sweep", default="", help="comma-separated stream counts (e.g. 256,512,1024,1536): measure each and report the largest whose p95 stays under 80 ms")
    args = ap.parse_args()
    from bench import build_generator
    if args.sweep:
        from bench import stream_headroom
        dev = torch.device("cuda", 0)
        res = stream_headroom(build_generator(dev), dev, counts=tuple(int(x) for x in args.sweep.split(",")), blocks=args.blocks, warmup=args.warmup, n_index=args.index)
        res["metric"] = "chunk latency, concurrent real-time streams (headroom sweep)"
        print(json.dumps(res))
        return
    from tinyvc_amd.module.infer import BatchedStreamInfer
    dev = torch.device("cuda", 0)
    gen = build_generator(dev)
    S = args.streams
    st = BatchedStreamInfer(gen, n_streams=S, target=synth.synth_index(args.index, seed=2).to(dev), device=dev,
                            block_size=1920, extra_size=3840, use_graph=not args.no_graph)
    st.init_buffer()
    waves = torch.stack([synth.synth_wave(1, args.blocks * 1920, seed=200 + s)[0] for s in range(min(S, 4))])
    waves = waves[torch.arange(S) % waves.shape[0]].to(dev).view(S, args.blocks, 1920)
    lat = []
    for i in range(args.blocks):
        blk = waves[:, i].contiguous()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = st.audio_callback(blk)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    l = np.sort(np.array(lat[args.warmup:])) * 1e3
    res = {"metric": "chunk latency, concurrent real-time streams", "streams": S, "block_samples": 1920, "budget_ms": 80.0,
           "p50_ms": float(l[len(l) // 2]), "p95_ms": float(l[int(len(l) * 0.95)]), "max_ms": float(l[-1]),
           "blocks": len(l), "hip_graph": not args.no_graph,
           "config": {"workload": f"infer_streaming.py {S} concurrent streams, 13440-sample buffer, {args.index}-vector index (BASELINE.json configs[2])"}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
