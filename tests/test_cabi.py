"""The C ABI boundary, without a GPU: libtinyvc_hip.so loads, exports every prototype declared in
include/tinyvc_hip.h, and the ctypes table in tinyvc_amd/_lib.py mirrors the header one to one."""
import os
import re

import pytest

from tinyvc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "tinyvc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"\b(tvc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
    return {name: [a.strip() for a in args.split(",")] if args.strip() != "void" else [] for name, args in protos}


@pytest.fixture(scope="module")
def lib():
    from tinyvc_amd import build
    build.build(verbose=False)          # hipcc cross-compiles for gfx950 without a GPU
    return _lib.load_library()


def test_every_declared_symbol_is_exported(lib):
    decl = header_functions()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in tinyvc_hip.h but not exported"


def test_ctypes_table_matches_header(lib):
    decl = header_functions()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, args in decl.items():
        assert len(args) == len(_lib.SIGNATURES[name][1]), f"{name}: header has {len(args)} parameters"


def test_version_and_null_safety(lib):
    assert lib.tvc_version() == 1
    # header + raw rows [N][768] + the bf16x3 image of the normalised vectors (1.5 floats per value) + inverse norms + their fp16 image
    assert lib.tvc_knn_prepared_elems(1000) == 64 + 1000 * 768 + 768 * 1024 * 3 // 2 + 1024 + 768 * 1024 // 2
    # fp16 storage: header + inverse norms [Npad] + the fp16 image (half a float per value): 2 B per element
    assert lib.tvc_knn_prepared_elems_f16(1000) == 64 + 1024 + 768 * 1024 // 2 + 8
    assert lib.tvc_knn_prepared_elems_f16(1000000) * 4 < 1.55e9
    assert lib.tvc_knn_prepared_elems(0) == 0
    # argument validation happens before any device work
    assert lib.tvc_finalize_weights(None) == -1
    assert lib.tvc_profile_enable(None, 1) == -1
    assert lib.tvc_last_error(None) == b"null ctx"


def test_no_cpu_fallback_in_product_path():
    """The product package never imports the oracle and fails loudly off-GPU."""
    import torch
    from tinyvc_amd.module.tinyvc import Decoder, Encoder
    from tinyvc_amd.module import utils
    with pytest.raises(_lib.TinyVCError):
        Encoder().infer(torch.zeros(1, 961, 4))
    with pytest.raises(_lib.TinyVCError):
        Decoder().infer(torch.zeros(1, 768, 4), torch.zeros(1, 1, 4), torch.zeros(1, 1, 1920))
    with pytest.raises(_lib.TinyVCError):
        utils.spectrogram(torch.zeros(1, 4800))
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "tinyvc_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f"{f} imports the oracle"


def test_host_helpers():
    import torch
    from tinyvc_amd.engine import pitch_class_table
    from tinyvc_amd.module import utils
    t = pitch_class_table()
    assert t.shape == (512,) and t[0] == 0 and abs(float(t[48]) - 40.0) < 1e-4 and float(t[1]) > 20.0
    x = torch.ones(2, 1000)
    y = utils.autopad_waveform(x)
    assert y.shape == (2, 1440) and float(y[:, 1000:].abs().sum()) == 0 and utils.autopad_waveform(y) is y


def test_ragged_plan_is_host_logic(lib):
    """tvc_ragged_plan needs no context and no device: the split of a ragged call into in-kernel batches (length classes at 11 / 43 / 128
    frames - the kernels a FilterNet level runs depend on the utterance's length there -, a frame cap per batch), and its validation."""
    import ctypes

    def plan(frames, lmax_frames=None, cap=0):
        B = len(frames)
        lens = (ctypes.c_int64 * B)(*[480 * f for f in frames])
        out = (ctypes.c_int32 * B)()
        n = ctypes.c_int()
        rc = lib.tvc_ragged_plan(B, 480 * (lmax_frames or max(frames)), lens, cap, out, ctypes.byref(n))
        return rc, list(out), n.value

    rc, rows, n = plan([200, 5, 131, 10, 11, 42, 43, 127, 128, 3])
    assert rc == 0 and n == 4
    cls = lambda f: (f >= 11) + (f >= 43) + (f >= 128)
    frames = [200, 5, 131, 10, 11, 42, 43, 127, 128, 3]
    by_batch = {}
    for f, r in zip(frames, rows):
        by_batch.setdefault(r, set()).add(cls(f))
    assert all(len(v) == 1 for v in by_batch.values()) and len(by_batch) == 4      # one class per batch
    assert rows[0] == rows[2] == rows[8] == 0                                       # the longest class runs first
    rc, rows, n = plan([150] * 7)
    assert rc == 0 and n == 1 and rows == [0] * 7
    assert plan([2, 50])[0] != 0                                                    # 960 samples: too short for the STFT's reflect padding
    assert plan([50, 60], lmax_frames=55)[0] != 0                                   # longer than its row
    assert plan([90000])[0] != 0                                                    # longer than a batch may be
    rc, rows, n = plan([150] * 7, cap=400)
    assert rc == 0 and n == 4 and rows == [0, 0, 1, 1, 2, 2, 3]                     # the cap cuts a class into several batches, in order
    assert plan([150] * 7, cap=-1)[0] != 0
    assert lib.tvc_ctx_set_ragged_batch_frames(None, 400) == -1                     # the cap is a property of a context: no process-wide state
