"""Ragged batches through the C ABI (tvc_convert_ragged_f32): utterances of different lengths in one call.

The reference converts the files of a directory one by one (infer.py:60-66), and padding a batch to a common length changes results
(GRN normalises over the whole time axis, convnext.py:31-34; the oscillator's phase is a scan over it).  So the contract is: every
utterance of a ragged batch gets exactly the samples its own B = 1 call gives it, and those are within 1e-4 of the oracle."""
import os

import pytest
import torch

from helpers import oracle_one_thread, rms, state_dicts
from oracle import ref_cpu as R
from tinyvc_amd import audio_io, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gen():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from tinyvc_amd.module.infer import Generator
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    enc_sd, dec_sd = state_dicts(0)
    enc, dec = Encoder(), Decoder()
    enc.load_state_dict(enc_sd)
    dec.load_state_dict(dec_sd)
    return Generator(enc, dec).to(DEV)


def test_ragged_batch_equals_one_call_per_utterance_and_the_oracle(gen):
    enc_sd, dec_sd = state_dicts(0)
    frames = [33, 7, 50, 7, 3, 50, 21, 7, 12]                   # three groups of equal length among them, the shortest legal input too
    lens = [480 * f - (17 if i % 2 else 0) for i, f in enumerate(frames)]      # some not yet padded to a frame
    B, Lmax, Tmax = len(frames), 480 * max(frames), max(frames)
    wf = torch.zeros(B, Lmax)
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=500 + b)[0]
    tgt = synth.synth_index(777, seed=8)
    angle = synth.synth_angle(B, Tmax, 31)
    out = gen.convert(wf.to(DEV), tgt.to(DEV), -1.5, noise_angle=angle.to(DEV), lengths=lens)
    assert out.shape == (B, Lmax)
    worst = 0.0
    for b, f in enumerate(frames):
        L = 480 * f
        one = gen.convert(wf[b:b + 1, :lens[b]].to(DEV), tgt.to(DEV), -1.5, noise_angle=angle[b:b + 1, :, :f].contiguous().to(DEV))
        assert one.shape == (1, L)
        assert torch.equal(out[b, :L], one[0]), f"utterance {b} ({f} frames): ragged batch != its own B = 1 call"
        assert not out[b, L:].any(), "the tail of a row is zero-filled"
        with oracle_one_thread():
            ref = R.convert(enc_sd, dec_sd, wf[b:b + 1, :lens[b]], tgt, -1.5, angle[b:b + 1, :, :f])
        d = rms(out[b, :L].cpu() - ref[0])
        worst = max(worst, d)
        assert d <= 1e-4, f"utterance {b}: {d:.3e}"
    print(f"[ragged] {B} utterances of {sorted(set(frames))} frames in one call: each equals its B = 1 call bit for bit; worst rms vs the oracle {worst:.3e}")
    # a second, differently shaped ragged call on the same engine (workspace regrowth, lane reuse)
    out2 = gen.convert(wf[:4].to(DEV), tgt.to(DEV), -1.5, noise_angle=angle[:4].to(DEV), lengths=lens[:4])
    for b in range(4):
        assert torch.equal(out2[b], out[b])


def test_long_utterances_run_as_one_ragged_batch_inside_the_kernels(gen):
    """Utterances share every kernel launch (csrc/ragged.h): per-utterance lengths inside the kernels.  Seven different lengths of >= 128
    frames - odd and even frame counts, one not padded to a frame, a duplicate - mixed with two short ones (another class: its own batch of
    the same call).  Contract unchanged: every utterance equals its own B = 1 call bit for bit."""
    enc_sd, dec_sd = state_dicts(0)
    frames = [200, 131, 256, 17, 145, 128, 200, 9, 173]
    lens = [480 * f - (23 if i % 3 == 1 else 0) for i, f in enumerate(frames)]
    B, Lmax, Tmax = len(frames), 480 * max(frames), max(frames)
    wf = torch.zeros(B, Lmax)
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=700 + b)[0]
    tgt = synth.synth_index(5003, seed=8)              # the two-stage search (N >= 4096)
    angle = synth.synth_angle(B, Tmax, 33)
    out = gen.convert(wf.to(DEV), tgt.to(DEV), 0.75, noise_angle=angle.to(DEV), lengths=lens)
    assert out.shape == (B, Lmax) and torch.isfinite(out).all()
    bad = []
    for b, f in enumerate(frames):
        L = 480 * f
        one = gen.convert(wf[b:b + 1, :lens[b]].to(DEV), tgt.to(DEV), 0.75, noise_angle=angle[b:b + 1, :, :f].contiguous().to(DEV))
        if not torch.equal(out[b, :L], one[0]):
            bad.append((b, f, rms(out[b, :L].cpu() - one[0].cpu())))
        assert not out[b, L:].any(), "the tail of a row is zero-filled"
    assert not bad, f"ragged batch != B = 1 call for (utterance, frames, rms): {bad}"
    with oracle_one_thread():
        ref = R.convert(enc_sd, dec_sd, wf[2:3, :lens[2]], tgt, 0.75, angle[2:3, :, :frames[2]])
    d = rms(out[2, :480 * frames[2]].cpu() - ref[0])
    print(f"[ragged] in-kernel batch of {sorted(f for f in frames if f >= 128)} frames: bit-identical to the B = 1 calls; 256-frame utterance vs the oracle {d:.3e}")
    assert d <= 1e-4
    # the library's own phase draw (noise_angle = None) and a second call on the same engine
    o2 = gen.convert(wf.to(DEV), tgt.to(DEV), 0.75, lengths=lens)
    assert torch.isfinite(o2).all() and not o2[1, 480 * frames[1]:].any()


def test_sixty_four_distinct_lengths_are_one_batch(gen):
    """The bench's ragged64 workload: 64 utterances of 64 different lengths (150 ... 250 frames).  Rows 0, 31 and 63 against their
    B = 1 calls; the whole call must not be slower than a few equal-length steps (it used to be 64 sequential B = 1 conversions)."""
    import time
    frames = [150 + (i * 100) // 63 for i in range(64)]
    lens = [480 * f for f in frames]
    wf = synth.synth_wave(64, max(lens), seed=100)
    for b, n in enumerate(lens):
        wf[b, n:] = 0
    wf = wf.to(DEV)
    tgt = synth.synth_index(10000, seed=8).to(DEV)
    angle = synth.synth_angle(64, max(frames), 5).to(DEV)
    out = gen.convert(wf, tgt, 0.0, noise_angle=angle, lengths=lens)
    for b in (0, 31, 63):
        one = gen.convert(wf[b:b + 1, :lens[b]], tgt, 0.0, noise_angle=angle[b:b + 1, :, :frames[b]].contiguous())
        assert torch.equal(out[b, :lens[b]], one[0]), f"utterance {b}"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        gen.convert(wf, tgt, 0.0, noise_angle=angle, lengths=lens)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"[ragged] 64 distinct lengths: {ms:.2f} ms per call")
    assert ms < 60.0       # (logged above: 7-8 ms measured; a generous bound only - a loaded box must not fail parity runs on wall time)


def test_short_utterances_of_every_class_share_their_launches(gen):
    """Which FiLM kernel a FilterNet level runs depends on the utterance's length (decoder.hip film_conv), so a ragged call is cut into
    classes at 11, 43 and 128 frames and every class is one in-kernel batch.  48 utterances of 48 different lengths from 3 to 140 frames
    (all four classes): each equals its B = 1 call bit for bit, and the call costs a few batch steps, not 48 sequential conversions."""
    import time
    frames = [3 + (i * 137) // 47 for i in range(48)]
    assert len(set(frames)) == 48 and min(frames) == 3 and max(frames) == 140
    lens = [480 * f for f in frames]
    wf = torch.zeros(48, max(lens))
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=900 + b)[0]
    wf = wf.to(DEV)
    tgt = synth.synth_index(1000, seed=2).to(DEV)
    angle = synth.synth_angle(48, max(frames), 7).to(DEV)
    out = gen.convert(wf, tgt, -0.5, noise_angle=angle, lengths=lens)
    bad = []
    for b, f in enumerate(frames):
        one = gen.convert(wf[b:b + 1, :lens[b]], tgt, -0.5, noise_angle=angle[b:b + 1, :, :f].contiguous())
        if not torch.equal(out[b, :lens[b]], one[0]):
            bad.append((b, f))
        assert not out[b, lens[b]:].any()
    assert not bad, f"ragged batch != B = 1 call for (utterance, frames): {bad}"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        gen.convert(wf, tgt, -0.5, noise_angle=angle, lengths=lens)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"[ragged] 48 utterances of 3 ... 140 frames (four classes): {ms:.2f} ms per call")
    assert ms < 60.0       # (logged above; generous bound, see the test before)


def test_ragged_arguments_are_validated(gen):
    from tinyvc_amd._lib import TinyVCError
    tgt = synth.synth_index(64, seed=1).to(DEV)
    wf = torch.zeros(2, 4800, device=DEV)
    with pytest.raises((ValueError, TinyVCError)):
        gen.convert(wf, tgt, 0.0, lengths=[4800, 900])            # torch.stft's reflect padding needs more than 960 samples
    with pytest.raises((ValueError, TinyVCError)):
        gen.convert(wf, tgt, 0.0, lengths=[4800, 9600])           # longer than its row


def test_infer_py_converts_a_directory_of_different_lengths_in_one_call(tmp_path):
    import infer
    d = tmp_path
    torch.save(synth.synth_state_dict("encoder"), d / "encoder.pt")
    torch.save(synth.synth_state_dict("decoder"), d / "decoder.pt")
    torch.save(synth.synth_index(300, seed=2), d / "index.pt")
    (d / "inputs").mkdir()
    lens = {"a": 12000, "b": 7777, "c": 12000, "d": 20011}
    for i, (name, n) in enumerate(lens.items()):
        audio_io.save(str(d / "inputs" / f"{name}.wav"), synth.synth_wave(1, n, seed=60 + i), 24000)
    gen = infer.load_generator(str(d / "encoder.pt"), str(d / "decoder.pt"), torch.device(DEV))
    calls = []
    orig = type(gen).convert

    def spy(self, *a, **k):
        calls.append(k.get("lengths"))
        return orig(self, *a, **k)

    type(gen).convert = spy
    try:
        torch.manual_seed(0)
        rc = infer.main(["-i", str(d / "inputs"), "-o", str(d / "out"), "-encp", str(d / "encoder.pt"), "-decp", str(d / "decoder.pt"),
                         "-idx", str(d / "index.pt"), "-d", DEV])
    finally:
        type(gen).convert = orig
    assert rc == 0
    assert len(calls) == 1 and sorted(calls[0]) == sorted(lens.values()), "the directory must be converted in ONE ragged call"
    for name, n in lens.items():
        y, sr = audio_io.load(str(d / "out" / f"{name}.wav"))
        assert sr == 24000 and y.shape == (1, -(-n // 480) * 480) and torch.isfinite(y).all() and float(y.abs().max()) > 1e-3
    assert os.path.exists(d / "out" / "d.wav")


def test_infer_py_bounds_its_batches_and_skips_files_it_cannot_convert(tmp_path, monkeypatch, capsys):
    """memory must not grow as (number of files) x (longest file).  With a small padded-sample budget the directory is cut
    into several calls in order of length; a file of 960 samples or fewer is reported and skipped instead of aborting the run."""
    import infer
    d = tmp_path
    torch.save(synth.synth_state_dict("encoder"), d / "encoder.pt")
    torch.save(synth.synth_state_dict("decoder"), d / "decoder.pt")
    torch.save(synth.synth_index(300, seed=2), d / "index.pt")
    (d / "inputs").mkdir()
    lens = {"a": 70000, "b": 9000, "c": 64000, "d": 30011, "e": 500, "f": 66000}
    for i, (name, n) in enumerate(lens.items()):
        audio_io.save(str(d / "inputs" / f"{name}.wav"), synth.synth_wave(1, n, seed=80 + i), 24000)
    monkeypatch.setenv("TVC_INFER_BATCH_SAMPLES", str(150000))
    gen = infer.load_generator(str(d / "encoder.pt"), str(d / "decoder.pt"), torch.device(DEV))
    calls = []
    orig = type(gen).convert

    def spy(self, wf, *a, **k):
        calls.append(tuple(wf.shape))
        return orig(self, wf, *a, **k)

    type(gen).convert = spy
    try:
        rc = infer.main(["-i", str(d / "inputs"), "-o", str(d / "out"), "-encp", str(d / "encoder.pt"), "-decp", str(d / "decoder.pt"),
                         "-idx", str(d / "index.pt"), "-d", DEV])
    finally:
        type(gen).convert = orig
    assert rc == 0
    assert "Skipping" in capsys.readouterr().out and not os.path.exists(d / "out" / "e.wav")
    assert len(calls) >= 2 and all(b * l <= 150000 or b == 1 for b, l in calls), calls
    for name, n in lens.items():
        if name == "e":
            continue
        y, sr = audio_io.load(str(d / "out" / f"{name}.wav"))
        assert sr == 24000 and y.shape == (1, -(-n // 480) * 480) and torch.isfinite(y).all() and float(y.abs().max()) > 1e-3


def test_many_utterances_and_several_batches_per_class(gen):
    """1 500 utterances in one call (more than one 1024-wide round of the table scans, lengths uploaded in two kernel-argument chunks)
    and, with the per-batch frame cap lowered, several in-kernel batches per length class: spot rows equal their B = 1 calls."""
    B = 1500
    frames = [3 + (i * 7) % 23 for i in range(B)]                 # 3 ... 25 frames: two classes
    lens = [480 * f for f in frames]
    g = torch.Generator().manual_seed(5)
    wf = 0.2 * torch.randn(B, max(lens), generator=g)
    for b, n in enumerate(lens):
        wf[b, n:] = 0
    wf = wf.to(DEV)
    tgt = synth.synth_index(64, seed=3).to(DEV)
    angle = synth.synth_angle(B, max(frames), 9).to(DEV)
    out = gen.convert(wf, tgt, 0.0, noise_angle=angle, lengths=lens)
    rows = [0, 1, 511, 1023, 1024, 1025, 1499]
    ones = {}
    for b in rows:
        ones[b] = gen.convert(wf[b:b + 1, :lens[b]], tgt, 0.0, noise_angle=angle[b:b + 1, :, :frames[b]].contiguous())[0]
        assert torch.equal(out[b, :lens[b]], ones[b]), f"utterance {b} ({frames[b]} frames)"
        assert not out[b, lens[b]:].any()
    eng = gen.engine(DEV)
    eng.set_ragged_batch_frames(4000)                                       # ~ 5 batches per class (a property of this engine's context)
    try:
        out2 = gen.convert(wf, tgt, 0.0, noise_angle=angle, lengths=lens)
    finally:
        eng.set_ragged_batch_frames(0)
    assert torch.equal(out2, out)


def test_class_boundaries(gen):
    """Frame counts on both sides of every length-class boundary (11, 43, 128 frames: where a FilterNet level changes the kernel it runs)."""
    frames = [10, 11, 42, 43, 127, 128]
    lens = [480 * f for f in frames]
    wf = torch.zeros(len(frames), max(lens))
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=300 + b)[0]
    wf = wf.to(DEV)
    tgt = synth.synth_index(500, seed=2).to(DEV)
    angle = synth.synth_angle(len(frames), max(frames), 11).to(DEV)
    out = gen.convert(wf, tgt, 1.0, noise_angle=angle, lengths=lens)
    for b, f in enumerate(frames):
        one = gen.convert(wf[b:b + 1, :lens[b]], tgt, 1.0, noise_angle=angle[b:b + 1, :, :f].contiguous())
        assert torch.equal(out[b, :lens[b]], one[0]), f"{f} frames"


def test_default_phase_draw_does_not_depend_on_the_batch(gen):
    """noise_angle = None: the library draws the phases itself, a hash of (seed, row, bin, frame) alone (include/tinyvc_hip.h).  So an
    utterance's samples do not change when OTHER rows of the call change length or class, row 0 equals the B = 1 call with the same
    seed, and torch.manual_seed makes the draw repeatable."""
    frames = [50, 7, 131, 20]
    lens = [480 * f for f in frames]
    wf = torch.zeros(4, max(lens))
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=900 + b)[0]
    wf = wf.to(DEV)
    tgt = synth.synth_index(500, seed=8).to(DEV)
    torch.manual_seed(77)
    a = gen.convert(wf, tgt, 0.0, lengths=lens)
    torch.manual_seed(77)
    a2 = gen.convert(wf, tgt, 0.0, lengths=lens)
    assert torch.equal(a, a2), "same torch seed, same call: same samples"
    wf2, lens2 = wf.clone(), list(lens)
    lens2[1] = 480 * 60                                      # row 1 becomes a longer utterance of another class
    wf2[1, :lens2[1]] = synth.synth_wave(1, lens2[1], seed=5)[0].to(DEV)
    torch.manual_seed(77)
    b_ = gen.convert(wf2, tgt, 0.0, lengths=lens2)
    for r in (0, 2, 3):
        assert torch.equal(a[r], b_[r]), f"row {r} changed with another row's length"
    torch.manual_seed(77)
    one = gen.convert(wf[0:1, :lens[0]], tgt, 0.0)
    assert torch.equal(a[0, :lens[0]], one[0]), "row 0 = the B = 1 call with the same seed"
    torch.manual_seed(78)
    c = gen.convert(wf, tgt, 0.0, lengths=lens)
    assert not torch.equal(a, c), "another seed, other phases"


def test_grn_tile_sums_add_up_the_same_way_alone_and_next_to_a_long_utterance(gen):
    """GRN's norm over time is added up from per-tile sums (cnx_s3.h): by every cnx2 workgroup itself up to 16 tiles, by one small
    launch before it when the longest utterance of the batch has more.  A 200-frame utterance takes the first route alone and the second
    next to a 1100-frame one (18 tiles); a 1100-frame utterance alone takes the second.  Same order in both: bit-identical samples."""
    frames = [200, 1100, 131, 1030]
    lens = [480 * f for f in frames]
    wf = torch.zeros(len(frames), max(lens))
    for b, n in enumerate(lens):
        wf[b, :n] = synth.synth_wave(1, n, seed=900 + b)[0]
    tgt = synth.synth_index(2000, seed=8).to(DEV)
    angle = synth.synth_angle(len(frames), max(frames), 41).to(DEV)
    out = gen.convert(wf.to(DEV), tgt, 0.5, noise_angle=angle, lengths=lens)
    for b, f in enumerate(frames):
        one = gen.convert(wf[b:b + 1, :lens[b]].to(DEV), tgt, 0.5, noise_angle=angle[b:b + 1, :, :f].contiguous())
        assert torch.equal(out[b, :lens[b]], one[0]), f"utterance {b} ({f} frames)"
    # an equal-length batch of the long one (B = 2, not ragged) against the same B = 1 call
    two = gen.convert(wf[1:2, :lens[1]].repeat(2, 1).to(DEV), tgt, 0.5, noise_angle=angle[1:2].repeat(2, 1, 1).contiguous())
    assert torch.equal(two[0], out[1, :lens[1]]) and torch.equal(two[1], two[0])
