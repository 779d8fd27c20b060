"""The entry scripts under a launcher (WORLD_SIZE > 1): infer.py converts a directory with every rank taking its share of the files and
writing its own outputs - no collective -, extract_index.py encodes the clips an index needs on all ranks and gathers the features once.
Host logic here (CPU): the length-balanced split, the WAV header reader it is computed from, the prefix rule of the reference's
extract loop.  GPU: both scripts' sharded paths reproduce the single-process outputs byte for byte.
Reference: infer.py:60-69 (a loop over files), extract_index.py:43-58."""
import os
import random

import numpy as np
import pytest
import torch

from tinyvc_amd import audio_io, parallel, synth


def test_lpt_split_covers_every_file_once_and_balances_padded_samples():
    rnd = random.Random(3)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 64, 203):
            costs = [480 * rnd.randint(3, 1500) for _ in range(n)]
            parts = parallel.lpt_split(costs, world)
            assert len(parts) == world
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n)), "every file exactly once"
            assert all(p == sorted(p) for p in parts)
            loads = [sum(costs[i] for i in p) for p in parts]
            if n >= world and n > 0:
                # Graham's bound for LPT, and in practice far inside it; round-robin over this list is worse whenever lengths are skewed
                opt_lb = max(max(costs), sum(costs) / world)
                assert max(loads) <= (4 / 3 - 1 / (3 * world)) * opt_lb + 1e-9
            assert parts == parallel.lpt_split(costs, world), "deterministic: every rank computes the same split on its own"
    # a directory sorted by name with one long file per group of eight: round-robin gives one rank all the long ones
    costs = [480 * (2000 if i % 8 == 0 else 100) for i in range(64)]
    rr = [sum(costs[i] for i in range(r, 64, 8)) for r in range(8)]
    lpt = [sum(costs[i] for i in p) for p in parallel.lpt_split(costs, 8)]
    assert max(lpt) < 0.5 * max(rr)
    assert max(lpt) - min(lpt) <= max(costs)


def test_wav_header_reader_agrees_with_the_loader(tmp_path):
    for i, (n, sr, ch) in enumerate([(12345, 16000, 1), (480, 24000, 2), (7, 44100, 1), (96001, 48000, 1)]):
        p = str(tmp_path / f"f{i}.wav")
        audio_io.save(p, torch.randn(ch, n) * 0.1, sr)
        assert audio_io.info(p) == (n, sr, ch)
        wf, sr2 = audio_io.load(p)
        assert wf.shape == (ch, n) and sr2 == sr
    # 16-bit PCM written by scipy directly, odd-sized chunk in front of the data chunk
    from scipy.io import wavfile
    p = str(tmp_path / "pcm.wav")
    wavfile.write(p, 22050, (np.random.RandomState(0).randn(1001) * 3000).astype(np.int16))
    assert audio_io.info(p) == (1001, 22050, 1)
    with pytest.raises(ValueError):
        (tmp_path / "junk.wav").write_bytes(b"not a wave file at all")
        audio_io.info(str(tmp_path / "junk.wav"))


def test_extract_index_prefix_rule_and_column_count(tmp_path):
    import extract_index
    # the reference's loop stops after the clip that takes the total ABOVE size
    assert extract_index.needed_prefix([10, 10, 10], [0, 1, 2], 20) == 3
    assert extract_index.needed_prefix([10, 10, 10], [0, 1, 2], 19) == 2
    assert extract_index.needed_prefix([10, 10, 10], [2, 0, 1], 5) == 1
    assert extract_index.needed_prefix([1, 1], [1, 0], 100) == 2
    for n, stride in [(24000, 4), (481, 4), (480, 1), (96000, 3)]:
        p = str(tmp_path / f"c{n}_{stride}.wav")
        audio_io.save(p, torch.zeros(1, n), 24000)
        t = -(-n // 480)
        assert extract_index.clip_columns(p, stride) == -(-t // stride)


def test_infer_share_is_a_partition_of_the_directory(tmp_path):
    import infer
    paths = []
    for i, n in enumerate([30000, 5000, 90000, 5000, 61000, 1200, 47000]):
        p = str(tmp_path / f"u{i}.wav")
        audio_io.save(p, torch.zeros(1, n), 16000 if i % 2 else 24000)
        paths.append(p)
    assert infer.my_share(paths, 1, 0) == list(range(7))
    shares = [infer.my_share(paths, 3, r) for r in range(3)]
    assert sorted(i for s in shares for i in s) == list(range(7))
    assert 2 in shares[0] and len(shares[0]) <= 2, "the longest file leads the first rank's share and that share stays short"


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("shard")
    torch.save(synth.synth_state_dict("encoder"), d / "encoder.pt")
    torch.save(synth.synth_state_dict("decoder"), d / "decoder.pt")
    torch.save(synth.synth_index(300, seed=2), d / "index.pt")
    (d / "inputs").mkdir()
    for i, (n, sr) in enumerate([(31000, 24000), (9000, 16000), (52000, 24000), (9000, 16000), (20011, 22050), (70000, 24000), (4800, 24000)]):
        audio_io.save(str(d / "inputs" / f"utt{i}.wav"), synth.synth_wave(1, n, seed=40 + i) * 0.8, sr)
    return d


def _files(d):
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}


@pytest.mark.gpu
def test_infer_py_ranks_reproduce_the_single_process_run_byte_for_byte(workdir):
    """Three ranks' shares, run one after the other on the one GPU of the test box (infer.py exchanges nothing between ranks, so that IS
    the three-rank job), against the unsharded run: same files, same bytes (--seed makes a file's phases its own)."""
    import infer
    base = ["-i", str(workdir / "inputs"), "-encp", str(workdir / "encoder.pt"), "-decp", str(workdir / "decoder.pt"), "-idx", str(workdir / "index.pt"),
            "-p", "-1.0", "-d", "cuda:0", "--seed", "11"]
    assert infer.main(base + ["-o", str(workdir / "one")]) == 0
    for r in range(3):
        assert infer.main(base + ["-o", str(workdir / "three")], world=3, rank=r, local_rank=0) == 0
    one, three = _files(workdir / "one"), _files(workdir / "three")
    assert sorted(one) == [f"utt{i}.wav" for i in range(7)] == sorted(three)
    for f in one:
        assert one[f] == three[f], f
    # and a rank's own outputs are a strict subset: rank 1 alone writes only its share
    assert infer.main(base + ["-o", str(workdir / "r1")], world=3, rank=1, local_rank=0) == 0
    r1 = _files(workdir / "r1")
    assert 0 < len(r1) < 7 and all(one[f] == r1[f] for f in r1)
    # without --seed the run still works (phases drawn by the library, as before)
    assert infer.main(base[:-2] + ["-o", str(workdir / "unseeded")], world=2, rank=0, local_rank=0) == 0


@pytest.mark.gpu
def test_extract_index_sharded_path_reproduces_the_single_process_index(workdir):
    """--force-dist runs the WORLD_SIZE > 1 path at world size 1 on the GPU: header-derived prefix, the split, the RCCL gather (to self)."""
    import extract_index
    for size in (40, 400):
        args = ["--dataset-cache", str(workdir / "inputs"), "-encp", str(workdir / "encoder.pt"), "-size", str(size), "-d", "cuda:0", "--seed", "5"]
        assert extract_index.main(args + ["-o", str(workdir / f"idx_one_{size}.pt")]) == 0
        assert extract_index.main(args + ["-o", str(workdir / f"idx_dist_{size}.pt"), "--force-dist"]) == 0
        a, b = torch.load(workdir / f"idx_one_{size}.pt"), torch.load(workdir / f"idx_dist_{size}.pt")
        assert a.shape == b.shape and a.shape[2] <= size and torch.equal(a, b)
