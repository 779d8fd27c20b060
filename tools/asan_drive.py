#!/usr/bin/env python3
"""Walks the C ABI's host-side code paths once for tools/asan_host.sh: weight packing (both checkpoints, a reload), index preparation in
fp32 and fp16, whole-batch / ragged / chunked conversion, the streaming converter with graph capture, front-door resampling and PCM
conversion, error paths (bad arguments, a truncated prepared blob).  Values are checked elsewhere; this run only has to finish clean."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyvc_amd import synth  # noqa: E402
from tinyvc_amd.module.infer import BatchedStreamInfer, Generator  # noqa: E402
from tinyvc_amd.module.tinyvc import Decoder, Encoder  # noqa: E402
import infer as infer_cli  # noqa: E402

dev = "cuda:0"
from tinyvc_amd import _lib  # noqa: E402
print("library:", _lib.load_library()._name)
enc, dec = Encoder(), Decoder()
enc.load_state_dict(synth.synth_state_dict("encoder"))
dec.load_state_dict(synth.synth_state_dict("decoder"))
gen = Generator(enc, dec).to(dev)
tgt = synth.synth_index(1000, seed=2).to(dev)
wf = synth.synth_wave(3, 9600 * 2, seed=1).to(dev)
out = gen.convert(wf, tgt, 1.0)
print("convert", tuple(out.shape), bool(torch.isfinite(out).all()))
out = gen.convert(wf, tgt.half(), 0.0)
print("convert, fp16 index", tuple(out.shape))
out = gen.convert(wf, tgt, 0.0, lengths=[19200, 9600, 14400])
print("ragged", tuple(out.shape))
gen.decoder.load_state_dict(synth.synth_state_dict("decoder", seed=1))       # repack
out = gen.convert(wf[:1], tgt, 0.0)
print("after a reload", tuple(out.shape))
o = infer_cli.convert_chunked(gen, wf[:2, :9600], tgt, 0.0, 1920, 4)
print("chunked", tuple(o.shape))
st = BatchedStreamInfer(gen, n_streams=2, target=tgt, pitch_shift=0.0, device=dev, use_graph=True)
st.init_buffer()
for i in range(3):
    y = st.audio_callback(wf[:2, i * 1920:(i + 1) * 1920])
print("stream", tuple(y.shape))
eng = gen.engine(dev)
r = eng.resample(wf[:1, :4800], 24000, 16000)
print("resample", tuple(r.shape))
for bad in (lambda: gen.convert(wf[:, :100], tgt, 0.0), lambda: gen.convert(wf, tgt[:, :, :2], 0.0)):
    try:
        bad()
        print("no error?")
    except Exception as e:  # noqa: BLE001
        print("error path:", type(e).__name__)
# round 6: a foreign blob's header check (a copy passes, a blob of another format version / N is refused), the per-context ragged cap, the
# general match_features kernel's host side, the stream push, the ragged plan with a cap
blob, n = eng.knn_prepare(tgt[0])
src = synth.synth_tensor("q", (1, 768, 20), seed=4).to(dev)
keep = [blob.clone()]      # (kept alive: the registry is keyed by device address, a recycled address inherits its record until tvc_knn_forget)
ok = eng.knn_match(src, keep[0], n)
print("foreign blob (copy)", tuple(ok.shape))
for word, value in ((5, 1), (2, n + 1)):
    bad_blob = blob.clone()
    keep.append(bad_blob)
    bad_blob.view(torch.int32)[word] = value
    try:
        eng.knn_match(src, bad_blob, n)
        print("no error?")
    except Exception as e:  # noqa: BLE001
        print("error path:", type(e).__name__)
eng.set_ragged_batch_frames(30)
out = gen.convert(wf, tgt, 0.0, lengths=[19200, 9600, 14400])
eng.set_ragged_batch_frames(0)
print("ragged with a 30-frame cap", tuple(out.shape))
from tinyvc_amd.module.tinyvc import match_features  # noqa: E402
for k, m in ((1, "cos"), (8, "L2"), (3, "IP")):
    o, i = match_features(src, tgt, k=k, metrics=m, return_indices=True)
print("general match", tuple(o.shape), tuple(i.shape))
buf = torch.zeros(2, 13440, device=dev)
eng.stream_push(buf, wf[:2, :1920])
print("stream push", float(buf[:, -1920:].abs().sum()) > 0)
torch.cuda.synchronize()
print("done")
