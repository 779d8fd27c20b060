#!/usr/bin/env python3
"""Headline benchmark: end-to-end voice conversion throughput (BASELINE.json metric).

N = 1 (default) runs BASELINE.json configs[1]: a batch of 64 utterances x 4 s (96 000 samples @24 kHz each, i.e. a
"4 s 16 kHz wav" after the 24 kHz resample the reference applies on load), fp32, matched against a 10 000-vector speaker
index, one `Generator.convert` call per step (the shipped module path: autopad check, torch.rand phase draw on the device,
cached prepared index, one tvc_convert_f32).  Inputs, index and weights are synthetic (tinyvc_amd.synth) and resident in
HBM before timing.  The same JSON line also carries configs[2] (32 concurrent real-time streams, p50 / p95 block latency,
measured after the timed region) under "stream".

N > 1 runs configs[3]: 64 utterances per rank (512 at N = 8), a 100 000-vector index replicated per GPU, and an RCCL
gather of every rank's [64, 96000] waveforms to rank 0 INSIDE each step (the path's only exchange); `gather_ms` is
reported separately.  One process per GPU (weak scaling):

  python bench.py [--gpus N --steps K --warmup W]        (N > 1 without WORLD_SIZE: re-executes itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
  python bench.py --force-dist                           (rehearsal: the whole N > 1 branch - RCCL communicator, gather, JSON keys - at world size 1)

Every line carries `parity`: the timed library checked in the same process against the committed reference fixture
tests/golden/convert_cfg2_B4_T200.npz (utterances 0..3 of configs[1], gap-checked index seed 8) - waveform rms and kNN index equality.

value = 16 kHz-equivalent audio samples converted per second by the whole job (audio seconds x 16 000 / wall seconds).
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tinyvc_amd import synth  # noqa: E402

SR = 24000
FILTER_BYTES_PER_SAMPLE = 87.86e6 / SR      # layer-boundary activation bytes of FilterNet per 24 kHz sample
FILTER_FLOPS_PER_SAMPLE = 2.483e9 / SR       # FilterNet FLOPs per 24 kHz output sample
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
F16_MFMA_PEAK_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
FP32_MFMA_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_* peak = fp32 vector peak
SPLIT_PRODUCTS = 3                         # two-term fp16 split: three fp16 part-products per fp32 product


STAGE_STEPS = 5      # untimed steps with every stage bracketed, after the timed region


def measured_filter_traffic(B, L):
    """HBM bytes moved by one step's FilterNet launches, from the newest committed rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE, collected and corrected as MI355X_MICROARCH.md prescribes: tools/filter_traffic.py);
    only valid for the workload and build it was measured on.  Returns (bytes, file name)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_filter_traffic_pmc.json")))
    if B == 64 and L == 96000 and files:
        return json.load(open(files[-1])).get("traffic_bytes"), os.path.basename(files[-1])
    return None, None


def build_generator(device):
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(synth.synth_state_dict("encoder"))
    dec.load_state_dict(synth.synth_state_dict("decoder"))
    return Generator(enc, dec).to(device).eval()


def _median(ts):
    return sorted(ts)[len(ts) // 2]


def cpu_baseline(seconds, n_index, batch=8):
    """The oracle (CPU restatement of the reference path, torch CPU ops) on the host cores, on bounded samples of the
    same workloads.  The thread count is swept ({1, 8, 16, 32, 64, all}; one warm-up + one run each on the
    B = 8 slice of configs[1]): oneDNN / MKL oversubscribe at 128 threads on this size, so "all cores" is not the host's
    best.  `value` = 3 warm-ups + median of 10 at the best setting (BASELINE.md §3's protocol).  Also: configs[0] (one
    4 s utterance, 1 000-vector index, B = 1) and configs[2] with ONE stream (200 blocks, p50 / p95), same thread count."""
    from oracle import ref_cpu as R
    enc_sd, dec_sd = synth.synth_state_dict("encoder"), synth.synth_state_dict("decoder")
    L = int(seconds * SR)
    wf = synth.synth_wave(batch, L, seed=100)
    tgt = synth.synth_index(n_index, seed=8)
    angle = synth.synth_angle(batch, L // 480, 3)
    all_threads = torch.get_num_threads()
    ncpu = os.cpu_count() or all_threads

    def run(b):
        t0 = time.perf_counter()
        R.convert(enc_sd, dec_sd, wf[:b], tgt, 0.0, angle[:b])
        return time.perf_counter() - t0

    sweep = {}
    for th in sorted({t for t in (1, 8, 16, 32, 64, all_threads) if t <= max(all_threads, 1)}):
        torch.set_num_threads(th)
        run(batch)
        sweep[th] = batch * seconds * 16000 / run(batch)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    for _ in range(2):                         # (the sweep's two runs at this setting were the first warm-ups)
        run(batch)
    t = _median([run(batch) for _ in range(10)])
    # the same utterances one at a time (B = 1 calls, what the reference's own loop does, infer.py:60-66): on this host the
    # batched call is often the slower way to spend the cores, so `value` is the better of the two
    for _ in range(3):
        run(1)
    t_seq = _median([run(1) for _ in range(10)])
    v_batched, v_seq = batch * seconds * 16000 / t, seconds * 16000 / t_seq
    # configs[0]: one utterance, 1 000-vector index (the reference's own CPU-runnable case)
    tgt1 = synth.synth_index(1000, seed=2)

    def run1():
        t0 = time.perf_counter()
        R.convert(enc_sd, dec_sd, wf[:1], tgt1, 0.0, angle[:1])
        return time.perf_counter() - t0

    for _ in range(3):
        run1()
    t_cfg1 = _median([run1() for _ in range(10)])
    # configs[2] with one stream: the reference's StreamInfer loop (stream.py:68-96), 20 warm-up + 200 blocks
    st = R.StreamState(block_size=1920, extra_size=3840)
    blocks = synth.synth_wave(1, 220 * 1920, seed=200)[0].view(220, 1920)
    ang = synth.synth_angle(1, st.input_size // 480, 7)
    lat = []
    for i in range(220):
        t0 = time.perf_counter()
        R.stream_callback(st, enc_sd, dec_sd, tgt1, 0.0, blocks[i], ang)
        lat.append(time.perf_counter() - t0)
    lat = sorted(lat[20:])
    torch.set_num_threads(all_threads)
    return {"value": max(v_batched, v_seq), "unit": "16kHz-samples/s", "cores": best, "kind": "port",
            "value_batched_b8": v_batched, "value_one_at_a_time": v_seq,
            "threads_best": best, "host_cpus": ncpu, "threads_sweep": {str(k): v for k, v in sweep.items()},
            "value_1thread": sweep.get(1),
            "cfg1_b1_ms": t_cfg1 * 1e3, "cfg1_b1_value": seconds * 16000 / t_cfg1,
            "cfg3_one_stream": {"p50_ms": lat[len(lat) // 2] * 1e3, "p95_ms": lat[int(len(lat) * 0.95)] * 1e3, "blocks": len(lat), "budget_ms": 80.0},
            "sample": f"{batch} of the 64 utterances ({seconds:g} s each, {n_index}-vector index): thread sweep (1 warm-up + 1 run per setting), then "
                      f"3 warm-ups + median of 10 runs on the best setting = {best} threads of {ncpu} host CPUs (BASELINE.md §3), "
                      f"batched (B = {batch}) and one utterance per call - value = the faster of the two; "
                      f"cfg1_b1 = one 4 s utterance, 1000-vector index, 3 warm-ups + median of 10; cfg3_one_stream = one real-time stream, "
                      f"200 blocks after 20 warm-ups; oracle/ref_cpu.py on torch CPU ops"}


def gpu_side_configs(gen, dev, L):
    """Two more configurations timed on the GPU after the headline region (N = 1 line only):
    `cfg1_b1_ms` = BASELINE configs[0] as one call (one 4 s utterance, 1 000-vector index: latency of a B = 1 convert, median of 30),
    `index100k_ms_per_step` = configs[3]'s per-rank workload (64 x 4 s against a 100 000-vector index, mean of 5 steps)."""
    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        return ts
    wf1 = synth.synth_wave(1, L, seed=1).to(dev)
    tgt1 = synth.synth_index(1000, seed=2).to(dev)
    t1 = timed(lambda: gen.convert(wf1, tgt1, 0.0), 30, 5)
    # the same B = 1 call replayed as ONE HIP graph (~135 launches of 5-50 us: launch-bound when eager)
    out_g, t1g = None, None
    try:
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_g = gen.convert(wf1, tgt1, 0.0)
        t1g = timed(g.replay, 30, 5)
    except Exception as e:                      # a capture failure is reported, not fatal for the headline
        t1g = None
        print(f"bench.py: B = 1 graph capture failed: {e}", file=sys.stderr)
    # ragged batch: 64 utterances of 64 DISTINCT lengths in [3 s, 5 s] (mean 4 s = the equal batch's audio) in one
    # Generator.convert(..., lengths=) call, next to the equal-length step
    frames = [150 + (i * 100) // 63 for i in range(64)]
    lens = [f * 480 for f in frames]
    wfr = synth.synth_wave(64, max(lens), seed=100).to(dev)
    for b, n in enumerate(lens):
        wfr[b, n:] = 0
    tgt10k = synth.synth_index(10000, seed=8).to(dev)
    tr = timed(lambda: gen.convert(wfr, tgt10k, 0.0, lengths=lens), 11, 3)
    wfe = synth.synth_wave(64, L, seed=100).to(dev)
    te = timed(lambda: gen.convert(wfe, tgt10k, 0.0), 11, 3)
    del wfr, wfe
    wf = synth.synth_wave(64, L, seed=1000).to(dev)
    tgt = synth.synth_index(100000, seed=5).to(dev)
    t2 = timed(lambda: gen.convert(wf, tgt, 0.0), 5, 2)
    return {"cfg1_b1_ms": _median(t1) * 1e3, "cfg1_b1_x_realtime": (L / SR) / _median(t1),
            "cfg1_b1_graph_ms": _median(t1g) * 1e3 if t1g else None,
            "ragged64_ms": _median(tr) * 1e3, "ragged64_equal_ms": _median(te) * 1e3, "ragged64_ratio": _median(tr) / _median(te),
            "ragged64_note": "64 utterances of 64 distinct lengths (150..250 frames, sum = 64 x 200 frames) in one convert(lengths=) call vs the equal-length 64 x 200-frame call, sync-to-sync wall time, median of 5",
            "index100k_ms_per_step": sum(t2) / len(t2) * 1e3, "index100k_value": 64 * (L / SR) * 16000 / (sum(t2) / len(t2))}


def stream_headroom(gen, dev, counts=(256, 1024, 2048, 4096), blocks=30, warmup=6, n_index=1000):
    """configs[2]'s headroom, MEASURED: the same per-block pipeline with S = 256 ... 4096 concurrent streams (one batched convert
    [S, 13440] + SOLA per 80 ms block, HIP-graph replay), p50 / p95 wall latency per block; `max_realtime_streams_measured` = the
    largest S tried whose p95 stays inside the 80 ms block period (nothing is extrapolated; the sweep stops at the first S that misses)."""
    import numpy as np
    from tinyvc_amd.module.infer import BatchedStreamInfer
    tgt = synth.synth_index(n_index, seed=2).to(dev)
    base = torch.stack([synth.synth_wave(1, blocks * 1920, seed=200 + s)[0] for s in range(4)]).to(dev)
    out, best = {}, None
    for S in counts:
        try:
            st = BatchedStreamInfer(gen, n_streams=S, target=tgt, device=dev, block_size=1920, extra_size=3840, use_graph=True)
            st.init_buffer()
            waves = base[torch.arange(S, device=dev) % 4].view(S, blocks, 1920)
            lat = []
            for i in range(blocks):
                blk = waves[:, i].contiguous()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                y = st.audio_callback(blk)
                torch.cuda.synchronize(dev)
                lat.append(time.perf_counter() - t0)
            assert torch.isfinite(y).all()
            l = np.sort(np.array(lat[warmup:])) * 1e3
            out[str(S)] = {"p50_ms": float(l[len(l) // 2]), "p95_ms": float(l[min(len(l) - 1, int(len(l) * 0.95))]), "max_ms": float(l[-1]), "blocks": int(len(l))}
            del st, waves
            if out[str(S)]["p95_ms"] < 80.0:
                best = S
            else:
                break
        except Exception as e:                      # (memory: the workspace of S streams) reported, not fatal for the headline
            out[str(S)] = {"error": str(e)[:200]}
            break
    return {"by_streams": out, "max_realtime_streams_measured": best, "budget_ms": 80.0}


def stream_latency(gen, dev, streams=32, blocks=120, warmup=20, n_index=1000):
    """BASELINE.json configs[2]: `streams` concurrent real-time streams, one 1920-sample block (80 ms) per stream and step,
    13 440-sample rolling buffers, HIP-graph replay of the per-block pipeline; wall latency per block, blocks already on
    the device."""
    import numpy as np
    from tinyvc_amd.module.infer import BatchedStreamInfer
    st = BatchedStreamInfer(gen, n_streams=streams, target=synth.synth_index(n_index, seed=2).to(dev), device=dev,
                            block_size=1920, extra_size=3840, use_graph=True)
    st.init_buffer()
    waves = torch.stack([synth.synth_wave(1, blocks * 1920, seed=200 + s)[0] for s in range(4)])
    waves = waves[torch.arange(streams) % 4].to(dev).view(streams, blocks, 1920)
    lat = []
    import gc
    gc.collect()
    gc.disable()      # a latency loop: the collector's pauses (a 10 ms outlier in one run of 200 blocks) are the host interpreter's, not the path's; re-enabled below
    for i in range(blocks):
        blk = waves[:, i].contiguous()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = st.audio_callback(blk)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t0)
    gc.enable()
    assert torch.isfinite(out).all()
    raw = np.array(lat[warmup:]) * 1e3
    l = np.sort(raw)
    worst = np.argsort(raw)[-3:][::-1]
    return {"workload": f"infer_streaming.py {streams} concurrent streams, 13440-sample buffer, {n_index}-vector index (BASELINE.json configs[2])",
            "streams": streams, "p50_ms": float(l[len(l) // 2]), "p95_ms": float(l[int(len(l) * 0.95)]), "max_ms": float(l[-1]),
            "slowest_blocks": [[int(i) + warmup, float(raw[i])] for i in worst],
            "blocks": int(len(l)), "budget_ms": 80.0, "hip_graph": True}


GOLDEN_CFG2 = os.path.join(ROOT, "tests", "golden", "convert_cfg2_B4_T200.npz")


def parity_vs_golden(gen, dev, wf64=None):
    """"RMS(wave_gpu - wave_cpu) and kNN index equality reported with every timing".  One untimed call of the
    module path on the fixture's inputs (utterances 0..3 of configs[1]: wave seeds 100..103, the gap-checked 10 000-vector
    index of seed 8, the fixture's noise phases) against the waveform and the top-4 indices the REFERENCE produced for them
    (tools/gen_golden.py, run next to /root/reference).  With `wf64` (the N = 1 timed batch, whose rows 0..3 ARE those
    utterances) the four are converted inside the full 64-utterance batch, i.e. exactly the timed computation."""
    import numpy as np
    if not os.path.exists(GOLDEN_CFG2):
        return None
    g = np.load(GOLDEN_CFG2)
    nb, L, N = int(g["batch"]), int(g["wave_len"]), int(g["index_size"])
    tgt = synth.synth_index(N, seed=int(g["index_seed"])).to(dev)
    if wf64 is not None and wf64.shape[1] == L and wf64.shape[0] >= nb:
        wf, inside = wf64, f"rows 0..{nb - 1} of the timed {wf64.shape[0]}-utterance batch"
    else:
        wf, inside = synth.synth_wave(nb, L, seed=int(g["wave_seed"])).to(dev), f"a separate {nb}-utterance call (the timed batch is a different workload)"
    B = wf.shape[0]
    angle = synth.synth_angle(B, L // 480, 1234)
    angle[:nb] = synth.synth_angle(nb, L // 480, int(g["noise_seed"]))
    out = gen.convert(wf, tgt, float(g["pitch_shift"]), noise_angle=angle.to(dev))[:nb].double().cpu()
    ref = torch.from_numpy(g["wave"]).double()
    rms = [float(((out[b] - ref[b]) ** 2).mean().sqrt()) for b in range(nb)]
    eng = gen.engine(dev)
    ssl, _, _ = eng.encoder(eng.stft_mag(wf[:nb]))
    from tinyvc_amd.module.tinyvc.feature_retrieval import prepare_reference
    blob, n = prepare_reference(tgt)
    _, idx = eng.knn_match(ssl, blob, n, want_indices=True)
    ref_idx = torch.from_numpy(g["knn_idx"])
    same = idx.cpu() == ref_idx
    # The TIMED call passes no phases: the library draws them inside noise_ifft_kernel<DRAW = true> (a hash of (seed, row, bin, frame)).
    # Same batch once more that way, then with the hash restated on the host (tinyvc_amd/synth.py) injected through the path the fixture
    # comparison above takes: the two must agree sample for sample.  Row 0 and its phases go to the cpu_baseline leg, where the oracle
    # converts the same utterance on them (`drawn_rms_vs_oracle`).
    torch.manual_seed(20260929)
    gdev = torch.cuda.default_generators[dev.index]
    seed = synth.draw_seed(gdev.initial_seed(), gdev.get_offset())
    drawn = gen.convert(wf, tgt, float(g["pitch_shift"]))
    hashed = synth.noise_phase_hash(seed, range(B), L // 480)
    injected = gen.convert(wf, tgt, float(g["pitch_shift"]), noise_angle=hashed.to(dev))
    drawn_equal = bool(torch.equal(drawn, injected))
    global _DRAWN
    _DRAWN = {"wave0": drawn[0].double().cpu(), "angle0": hashed[:1].clone(), "wf0": wf[:1].cpu(), "tgt": tgt.cpu(), "shift": float(g["pitch_shift"])}
    return {"drawn_equals_injected_hash": drawn_equal, "drawn_rms_vs_oracle": None,
            "drawn_note": "the timed call's instantiation: phases drawn in-kernel (noise_angle = None) vs the same batch with the host restatement of the hash injected: torch.equal over the whole batch; drawn_rms_vs_oracle = row 0 against the oracle (one thread) on those phases, filled in by the cpu_baseline leg",
            "fixture": "tests/golden/convert_cfg2_B4_T200.npz (reference PyTorch CPU path, tools/gen_golden.py)", "checked": inside,
            "rms_vs_golden": rms, "rms_vs_golden_max": max(rms), "gate_rms": 1e-4, "ref_wave_rms": float((ref ** 2).mean().sqrt()),
            "knn_idx_equal": bool(same.all()), "knn_idx_mismatches": int((~same).sum()), "knn_queries": int(ref_idx.shape[0] * ref_idx.shape[1]),
            "ok": bool(same.all()) and max(rms) <= 1e-4 and drawn_equal}


_DRAWN = None      # parity_vs_golden -> cpu_baseline: row 0 of the drawn call, its inputs and phases


def drawn_vs_oracle():
    """Row 0 of the call that drew its own phases against the oracle on the same phases (ONE thread: the reproducible oracle)."""
    if _DRAWN is None:
        return None
    from oracle import ref_cpu as R
    enc_sd, dec_sd = synth.synth_state_dict("encoder"), synth.synth_state_dict("decoder")
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ref = R.convert(enc_sd, dec_sd, _DRAWN["wf0"], _DRAWN["tgt"], _DRAWN["shift"], _DRAWN["angle0"])
    finally:
        torch.set_num_threads(n)
    return float(((_DRAWN["wave0"] - ref[0].double()) ** 2).mean().sqrt())


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py --gpus {n}: this box has {have} GPU(s); the {n} ranks are launched anyway and each reports its own device error", file=sys.stderr, flush=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--index", type=int, default=0, help="index vectors (default: 10 000 at N = 1 = configs[1], 100 000 at N > 1 = configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the configs[2] latency measurement (N = 1)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave every rank's waveforms on its own GPU (not configs[3])")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 branch (RCCL communicator, gather, its JSON keys) even at world size 1")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed parity call against the committed reference fixture")
    args = ap.parse_args()
    force_dist = args.force_dist or bool(os.environ.get("TVC_BENCH_FORCE_DIST"))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py rank {rank}: needs GPU {local}, this box has {torch.cuda.device_count()} GPU(s) - "
                         f"--gpus {args.gpus} takes {args.gpus} GPUs on one node (one process per GPU)")
    dist = None
    multi = world > 1 or force_dist            # the N > 1 branch: configs[3]
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rccl_ranks = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:       # --force-dist without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        except Exception as e:
            raise SystemExit(f"bench.py rank {rank}: init_process_group('nccl') failed on {dev}: {e}")
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                     # what the communicator itself says its size is
        rccl_ranks = int(ones.item())
        assert rccl_ranks == world, (rccl_ranks, world)

    B, L = args.batch, int(args.seconds * SR)
    L -= L % 480
    n_index = args.index or (100000 if multi else 10000)
    gather = multi and not args.no_gather
    gen = build_generator(dev)
    eng = gen.engine(dev)
    wf = synth.synth_wave(B, L, seed=(1000 if multi else 100) + rank * B).to(dev)      # SURVEY §8d seeds
    # configs[1]: the gap-checked index of the committed fixture (seed 8: every fp64 top-5 gap > 2e-6, so the reference's indices are
    # THE answer) instead of SURVEY's seed 4 - same size and distribution, and the timed batch's first utterances are the fixture's
    tgt = synth.synth_index(n_index, seed=5 if multi else 8).to(dev)
    dest = torch.empty(world, B, L, device=dev) if gather and rank == 0 else None        # rank 0's landing buffer, allocated once
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + args.warmup + 1 + STAGE_STEPS))] if gather else None
    state = {"i": 0, "out": None}
    from tinyvc_amd import parallel

    def step():
        out = gen.convert(wf, tgt, 0.0)          # one Generator.convert per step (generator.py:26-34)
        if gather:                               # configs[3]: the job's output collected on rank 0 (RCCL gather over xGMI)
            a, b = ev[2 * state["i"]], ev[2 * state["i"] + 1]
            a.record()
            parallel.gather_into(out, dest, dst=0)
            b.record()
            state["i"] += 1
        state["out"] = out

    # stage timers: hipEvent pairs recorded by the library on the launch stream.  They are switched on for
    # the warm-up too, so that the context's event pool is populated before the timed region
    # (creating events mid-stream stalls it).
    timers = not os.environ.get("TVC_BENCH_NOTIMERS")     # diagnostic: run without the library's hipEvent stage timers (no roofline then)
    # During the timed steps only the roofline's region (FilterNet) is bracketed: one hipEvent pair per step.  All 19 regions cost
    # ~0.12 ms of a step (38 event records); the stage split is taken from STAGE_STEPS extra, untimed steps afterwards.
    eng.profile(1 if timers else 0)                       # every region once, so that the context's event pool is populated before the timed steps
    step()
    eng.profile_read()
    eng.profile(2 if timers else 0)
    for _ in range(args.warmup):
        step()
        eng.profile_read()

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # N > 1: the SAME per-rank workload on rank 0 alone (no gather, the other ranks idle at the barrier) - the N = 1 figure the
    # job's value has to be compared with (the default N = 1 line is configs[1], a different index size)
    n1_same = None
    if multi:
        fence()
        if rank == 0:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                gen.convert(wf, tgt, 0.0)
            torch.cuda.synchronize(dev)
            n1_same = B * (L / SR) * 16000 * args.steps / (time.perf_counter() - t0)
            eng.profile_read()                            # discard: these steps' filter_net regions must not be summed into the timed ones
    fence()
    first = state["i"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    last = state["i"]
    prof = eng.profile_read()                             # {"filter_net": ms summed over the timed steps}
    stage_prof = {}
    if timers:
        eng.profile(1)
        for _ in range(STAGE_STEPS):
            step()
        stage_prof = {k: v / STAGE_STEPS for k, v in eng.profile_read().items()}
        fence()
    eng.profile(False)
    gather_ms = None
    if gather:
        gather_ms = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(first, last)) / max(args.steps, 1)
    if multi:
        t = torch.tensor([dt, gather_ms or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gather_max = float(t[0].item()), float(t[1].item())
    out = state["out"]
    assert os.environ.get("TVC_BENCH_NOCHECK") or torch.isfinite(out).all(), "non-finite output"
    if gather and rank == 0:
        assert torch.equal(dest[0], out), "rank 0's own slot of the gathered job output differs from its local result"

    if rank == 0:
        audio_s = world * B * (L / SR) * args.steps
        value = audio_s * 16000 / dt
        # FilterNet's duration = its region on the launch stream + its input contraction, which the library runs on a side stream beside SourceNet / the DSP
        # stage (decoder.hip run_decoder): added in full, as if it were serial - the roofline fractions must not profit from where the launch sits
        t_filter = (prof.get("filter_net", 0.0) + prof.get("filter_net.input@side", 0.0)) / 1e3 / max(args.steps, 1)
        alg_bytes, flops = FILTER_BYTES_PER_SAMPLE * B * L, FILTER_FLOPS_PER_SAMPLE * B * L
        traffic, traffic_src = measured_filter_traffic(B, L)
        hbm = {"achieved": alg_bytes / t_filter / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / t_filter / 1e9 / HBM_PEAK_GBS,
               "algorithmic_bytes_per_launch": alg_bytes,
               "moved_gbs": traffic / t_filter / 1e9 if traffic else None,
               "moved_frac": traffic / t_filter / 1e9 / HBM_PEAK_GBS if traffic else None} if t_filter > 0 else None
        mfma = {"achieved": SPLIT_PRODUCTS * flops / t_filter / 1e12, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": SPLIT_PRODUCTS * flops / t_filter / 1e12 / F16_MFMA_PEAK_TFLOPS,
                "flops_per_launch": flops, "f16_part_products_per_fp32_product": SPLIT_PRODUCTS,
                "fp32_equiv_tflops": flops / t_filter / 1e12, "fp32_equiv_frac_of_fp32_mfma_peak": flops / t_filter / 1e12 / FP32_MFMA_PEAK_TFLOPS} if t_filter > 0 else None
        # which roof binds, from the data: the pipe the kernels issue on (fp16 MFMA, 3 part-products per product) against the
        # bytes that really crossed HBM (PMC); the layer-boundary byte model (north_star's 40 % target) stays in
        # `hbm.frac` / `hbm_layer_boundary_frac` but is never what selects `bound` once blocks are fused.
        roof = None
        if t_filter > 0 and timers:
            moved = hbm["moved_frac"]
            bound = "mfma" if (moved is None or mfma["frac"] >= moved) else "hbm"
            pick = mfma if bound == "mfma" else {"achieved": hbm["moved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": moved}
            roof = {"bound": bound, "achieved": pick["achieved"], "peak": pick["peak"], "unit": pick["unit"], "frac": pick["frac"],
                    "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "FilterNet Conv1d stack = the filter_net launches of one step (two-term fp16 split on v_mfma_f32_32x32x16_f16, three part-products per fp32 product: fused ups.4+output kernels, film_s2 / conv_s2 / conv3s for the 96..384-channel levels, conv48s / conv48p for the 48-channel one, down24f / down0s for the 24-channel ones), hipEvent pair on the launch stream",
                    "launch_ms": t_filter * 1e3,
                    "hbm_layer_boundary_frac": hbm["frac"], "hbm_moved_frac": moved, "mfma_f16_frac": mfma["frac"],
                    "hbm": hbm, "mfma": mfma,
                    "note": "neither roof is above 0.5: the stack is issue/latency-bound between them (DESIGN.md §4)"
                            if max(mfma["frac"], moved or 0.0) < 0.5 else None}
        cfg = "configs[3]" if multi else "configs[1]"
        parity = None if args.no_parity else parity_vs_golden(gen, dev, None if multi else wf)
        res = {
            "metric": "audio-samples/sec (16 kHz) end-to-end VC",
            "value": value, "unit": "16kHz-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (fp32 in / out / accumulate; channel contractions as two-term fp16 splits on v_mfma_f32_32x32x16_f16: 22-bit significands, three part-products per fp32 product)",
            "data": "synthetic",
            "parity": parity,
            "config": {"workload": (f"infer.py {B}-utterance batch fp32, {L / SR:g} s @24 kHz per utterance, {n_index}-vector index (BASELINE.json {cfg})" if not multi else
                                    f"{world * B} synthetic utterances sharded across {world} MI355X ({B} per rank, {L / SR:g} s each), RCCL gather to rank 0, {n_index}-vector index replicated per GPU (BASELINE.json {cfg})"),
                       "global_batch": world * B, "utterance_samples_24k": L, "index_vectors": n_index,
                       "parallelism": f"utterance-dp{world}", "x_realtime": audio_s / dt},
            "roofline": roof,
            "stage_ms_per_step": dict(sorted(stage_prof.items())),
            "stage_ms_note": f"hipEvent pairs around every stage over {STAGE_STEPS} extra steps after the timed region (the timed steps carry the filter_net pair only)",
        }
        if multi:
            res["rccl_ranks"] = rccl_ranks            # all_reduce of ones over the communicator
            res["forced_dist_rehearsal"] = bool(force_dist and world == 1)
            res["n1_same_workload_value"] = n1_same   # rank 0 alone, same 64 x 4 s vs 100 k index, no gather, same process
            res["scaling_efficiency"] = value / (world * n1_same) if n1_same else None
            res["scaling_efficiency_note"] = "value / (n_gpus x n1_same_workload_value): the default N = 1 line is configs[1] (10 k index), not this workload"
            res["gather"] = "rccl gather -> rank 0, inside the step" if gather else "none"
            res["gather_ms"] = gather_ms
            res["gather_ms_max_over_ranks"] = gather_max if gather else None
            res["gather_bytes_per_rank"] = B * L * 4 if gather else 0
        if not multi and not args.no_stream:
            res["stream"] = stream_latency(gen, dev, blocks=220)
            res["stream"].update(stream_headroom(gen, dev))
            if B == 64 and n_index == 10000:
                res.update(gpu_side_configs(gen, dev, L))
        if not multi and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(L / SR, n_index)
            if parity is not None:                       # the oracle leg of the drawn-phase parity (the oracle runs in the cpu_baseline leg only)
                parity["drawn_rms_vs_oracle"] = drawn_vs_oracle()
                # (gate 1.5e-4: this figure is against the oracle run LIVE on this host - the committed fixture rows above carry the 1e-4 gate -,
                # and the oracle's own waveform moves by up to 1.7e-4 with the host CPU / thread count; the bit-exact link
                # drawn == injected is what ties the timed instantiation to the fixture-gated path)
                parity["drawn_rms_vs_oracle_gate"] = 1.5e-4
                parity["ok"] = bool(parity["ok"] and parity["drawn_rms_vs_oracle"] is not None and parity["drawn_rms_vs_oracle"] <= 1.5e-4)
        print(json.dumps(res), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
