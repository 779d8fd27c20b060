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


def _worker_into(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # what bench.py does at N > 1 (configs[3]): equal shards gathered into one landing buffer allocated once on rank 0
        dest = torch.full((world, 4, 960), -1.0) if rank == 0 else None
        for step in range(3):
            local = _fake_convert(torch.randn(4, 960, generator=torch.Generator().manual_seed(100 * step + rank)), 1.5)
            got = parallel.gather_into(local, dest, dst=0)
            if rank == 0:
                assert got is dest
                want = torch.stack([_fake_convert(torch.randn(4, 960, generator=torch.Generator().manual_seed(100 * step + r)), 1.5) for r in range(world)])
                assert torch.equal(dest, want)
            else:
                assert got is None
        if rank == 0:
            q.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_into_preallocated_landing_buffer(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_into, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert q.get() == "ok"
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def test_shard_bounds_cover_everything():
    for n in (1, 5, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- index-sharded match: exchange/merge logic on CPU with the oracle standing in for the per-shard HIP calls ----
class _OracleShardEngine:
    """Test stand-in for Engine.knn_topk / knn_gather_slots / knn_finish (those are HIP kernels): the same
    contracts computed with torch CPU ops, so the all_gather / merge / all_reduce path runs under gloo."""

    def __init__(self, shard):          # shard [768, n_local]
        self.shard = shard

    def knn_topk(self, src, prepared, n_local):
        ref = self.shard / (self.shard.norm(dim=0, keepdim=True) + 1e-6)
        q = src / (src.norm(dim=1, keepdim=True) + 1e-6)
        sims = torch.einsum("bkt,kn->btn", q, ref)
        v, i = parallel.merge_topk(sims, torch.arange(n_local).expand_as(sims))
        return v.contiguous(), i.contiguous()

    def knn_gather_slots(self, prepared, n_local, idx):
        rows = self.shard.t()                                   # [n_local, 768]
        out = torch.zeros(*idx.shape, rows.shape[1])
        ok = idx >= 0
        out[ok] = rows[idx[ok]]
        return out

    def knn_finish(self, slots):
        s = ((slots[..., 0, :] + slots[..., 1, :]) + slots[..., 2, :]) + slots[..., 3, :]
        return (s * 0.25).permute(0, 2, 1).contiguous()


def _knn_worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        index = torch.randn(768, n_total, generator=g)
        index[:, 7] = index[:, 2]                               # an exact tie across (or inside) shards: lower index must win
        src = torch.randn(2, 768, 9, generator=g)
        lo, hi = parallel.shard_bounds(n_total, rank, world)
        eng = _OracleShardEngine(index[:, lo:hi])
        out, sel = parallel.match_features_sharded(eng, src, None, hi - lo, lo)
        if rank == 0:
            q.put((out, sel))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 40), (3, 25)])
def test_index_sharded_match_equals_unsharded(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_knn_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, sel = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    index = torch.randn(768, n_total, generator=g)
    index[:, 7] = index[:, 2]
    src = torch.randn(2, 768, 9, generator=g)
    whole = _OracleShardEngine(index)
    v, i = whole.knn_topk(src, None, n_total)
    assert torch.equal(sel, i)
    assert torch.equal(out, whole.knn_finish(whole.knn_gather_slots(None, n_total, i)))


def test_merge_topk_tie_rule():
    sims = torch.tensor([[0.5, 0.9, 0.9, 0.1, 0.9, 0.3]])
    idx = torch.tensor([[10, 7, 3, 1, 5, 2]])
    v, i = parallel.merge_topk(sims, idx)
    assert i.tolist() == [[3, 5, 7, 10]] and torch.equal(v, torch.tensor([[0.9, 0.9, 0.9, 0.5]]))


# ---- extract_index.py under a launcher: the clips of the needed prefix encoded by all ranks, one gather (extract_index.sharded_features) ----
def _fake_encode(i, n):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(1, 768, n, generator=g)


def _index_worker(rank, world, port, cols, order, size, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import extract_index
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def enc(i):
            calls.append(i)
            return _fake_encode(i, cols[i])
        feats = extract_index.sharded_features(cols, order, size, world, rank, enc, torch.device("cpu"))
        q.put((rank, calls, feats))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_extract_index_features_sharded_equal_the_sequential_loop(world):
    import extract_index
    cols = [13, 2, 40, 7, 7, 25, 1, 9, 30, 4]
    order = torch.randperm(len(cols), generator=torch.Generator().manual_seed(3)).tolist()
    size = 70
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_index_worker, args=(r, world, port, cols, order, size, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get() for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the reference's loop (extract_index.py:47-55): clips in shuffled order until more than `size` vectors are there
    ref, total = [], 0
    for i in order:
        ref.append(_fake_encode(i, cols[i]))
        total += cols[i]
        if total > size:
            break
    used = order[:len(ref)]
    encoded = sorted(i for _r, calls, _f in got for i in calls)
    assert encoded == sorted(used), "exactly the clips the sequential loop uses, each encoded by one rank"
    feats = [f for r, _c, f in got if r == 0][0]
    assert all(f is None for r, _c, f in got if r != 0)
    assert len(feats) == len(ref) and all(torch.equal(a, b) for a, b in zip(feats, ref))
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    assert torch.equal(extract_index.assemble(feats, size, g1, False), extract_index.assemble(ref, size, g2, False))
