#!/usr/bin/env python3
"""Headline benchmark: end-to-end voice conversion throughput (BASELINE.json metric).

Workload (BASELINE.json configs[1]): a batch of 64 utterances x 4 s (96 000 samples @24 kHz each,
i.e. "4 s 16 kHz wav" after the 24 kHz resample the reference applies on load), fp32, matched
against a 10 000-vector speaker index, converted by one `Generator.convert` call per step.
Inputs, index and weights are synthetic (tinyvc_amd.synth) and resident in HBM before timing.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; every rank converts its own 64 utterances (weak scaling: utterances are
independent, so the path has no exchange step and a step issues no collective - each rank's
waveforms stay on the GPU that produced them, as with one reference process per device;
`--gather` adds `parallel.gather_waves` (RCCL gather to rank 0) to every step for callers who
want the job's output in one place).  value = 16 kHz-equivalent audio samples converted per
second by the whole job (audio seconds x 16 000 / wall seconds).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tinyvc_amd import synth  # noqa: E402

SR = 24000
FILTER_BYTES_PER_SAMPLE = 87.86e6 / SR      # layer-boundary activation bytes of FilterNet
FILTER_FLOPS_PER_SAMPLE = 2.483e9 / SR       # FilterNet FLOPs per 24 kHz output sample
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_* peak = fp32 vector peak


def measured_filter_traffic(B, L):
    """HBM bytes moved by one step's FilterNet launches, from the committed rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE, collected and corrected as MI355X_MICROARCH.md prescribes); only valid
    for the workload it was measured on."""
    p = os.path.join(ROOT, "profiles", "r01_filter_traffic_pmc.json")
    if B == 64 and L == 96000 and os.path.exists(p):
        return json.load(open(p)).get("traffic_bytes")
    return None


def build_generator(device):
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(synth.synth_state_dict("encoder"))
    dec.load_state_dict(synth.synth_state_dict("decoder"))
    return Generator(enc, dec).to(device).eval()


def cpu_baseline(seconds, n_index, batch=4, reps=3):
    """The oracle (CPU restatement of the reference path, torch CPU ops) on the host cores, on a
    bounded sample of the same workload: `batch` utterances of the same length and index size."""
    from oracle import ref_cpu as R
    enc_sd, dec_sd = synth.synth_state_dict("encoder"), synth.synth_state_dict("decoder")
    L = int(seconds * SR)
    wf = synth.synth_wave(batch, L, seed=100)
    tgt = synth.synth_index(n_index, seed=4)
    angle = synth.synth_angle(batch, L // 480, 3)
    R.convert(enc_sd, dec_sd, wf[:1], tgt, 0.0, angle[:1])          # warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        R.convert(enc_sd, dec_sd, wf, tgt, 0.0, angle)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": batch * seconds * 16000 / t, "unit": "16kHz-samples/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{batch} of the 64 utterances ({seconds:g} s each, {n_index}-vector index), median of {reps} runs, oracle/ref_cpu.py on torch CPU ops"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--index", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="N > 1: also gather every step's waveforms on rank 0 (not part of the path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    B, L = args.batch, int(args.seconds * SR)
    L -= L % 480
    gen = build_generator(dev)
    eng = gen.engine(dev)
    wf = synth.synth_wave(B, L, seed=100 + rank * B).to(dev)
    tgt = synth.synth_index(args.index, seed=4).to(dev)
    from tinyvc_amd.module.tinyvc.feature_retrieval import prepare_reference
    blob, n_idx = prepare_reference(tgt)
    out = torch.empty(B, L, device=dev)
    from tinyvc_amd import parallel

    def step():
        # on-device phase draw (library RNG): the reference draws fresh torch.rand phases per call too
        eng.convert(wf, blob, n_idx, 0.0, None, out=out)
        if world > 1 and args.gather:
            parallel.gather_waves(out, world * B, dst=0)       # optional: the job's output collected on rank 0 (RCCL gather)

    # stage timers: hipEvent pairs recorded by the library on the launch stream.  They are switched on for
    # the warm-up too, so that the context's event pool is populated before the timed region
    # (creating events mid-stream stalls it).
    eng.profile(True)
    step()
    eng.profile_read()
    for _ in range(args.warmup):
        step()
        eng.profile_read()

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert os.environ.get("TVC_BENCH_NOCHECK") or torch.isfinite(out).all(), "non-finite output"

    if rank == 0:
        audio_s = world * B * (L / SR) * args.steps
        value = audio_s * 16000 / dt
        t_filter = prof.get("filter_net", 0.0) / 1e3 / max(args.steps, 1)
        achieved = FILTER_BYTES_PER_SAMPLE * B * L / t_filter / 1e9 if t_filter > 0 else None
        res = {
            "metric": "audio-samples/sec (16 kHz) end-to-end VC",
            "value": value, "unit": "16kHz-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"infer.py {B}-utterance batch fp32, {L / SR:g} s @24 kHz per utterance, {args.index}-vector index (BASELINE.json configs[1])",
                       "global_batch": world * B, "utterance_samples_24k": L, "index_vectors": args.index,
                       "parallelism": f"utterance-dp{world}", "x_realtime": audio_s / dt},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": measured_filter_traffic(B, L),
                         "kernel": "FilterNet Conv1d stack = the filter_net launches of one step (split-precision bf16x3 MFMA everywhere: fused ups.4+output kernels, conv3s for the 48..384-channel levels, conv24s / down0s for the 24-channel ones), hipEvent pair on the launch stream",
                         "algorithmic_bytes_per_launch": FILTER_BYTES_PER_SAMPLE * B * L,
                         "launch_ms": t_filter * 1e3,
                         "fp32_mfma_tflops": FILTER_FLOPS_PER_SAMPLE * B * L / t_filter / 1e12 if t_filter > 0 else None,
                         "fp32_mfma_frac": FILTER_FLOPS_PER_SAMPLE * B * L / t_filter / 1e12 / FP32_MFMA_PEAK_TFLOPS if t_filter > 0 else None},
            "stage_ms_per_step": {k: v / args.steps for k, v in sorted(prof.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(L / SR, args.index)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
