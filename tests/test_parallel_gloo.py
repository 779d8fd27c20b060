"""The N>1 path on CPU: 2 (and 3, ragged) gloo processes shard a batch of utterances, run a
stand-in per-utterance transform (the real convert needs a GPU) and gather to rank 0; the result
must equal the single-process result in utterance order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tinyvc_amd import parallel


def _fake_convert(w, gain):
    # per-utterance, length-preserving, batch-independent: same contract as Generator.convert
    return torch.tanh(w * gain) + w.flip(1) * 0.25


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        waves = torch.randn(n_items, 960, generator=g)
        out = parallel.convert_sharded(_fake_convert, waves, 1.5, dst=0)
        if rank == 0:
            q.put(out)
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 8), (2, 5), (3, 7)])
def test_sharded_convert_matches_single_process(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    waves = torch.randn(n_items, 960, generator=g)
    assert torch.equal(out, _fake_convert(waves, 1.5))


def test_shard_bounds_cover_everything():
    for n in (1, 5, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
