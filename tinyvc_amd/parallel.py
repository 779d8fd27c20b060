"""Utterance-level data parallelism: one process per GPU, utterances (or streams) are independent
(no cross-utterance state in Generator.convert: GRN/LayerNorm statistics are per batch element,
reference convnext.py:32-33), so ranks share nothing on the data path.  The only exchange is the
final gather of converted waveforms to one rank (RCCL over xGMI when the backend is "nccl")."""
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous split of `n_items` utterances over `world` ranks; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def dist_env():
    """(world, rank, local_rank) as a launcher (`python -m torch.distributed.run`) exports them; (1, 0, 0) without one."""
    import os
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def lpt_split(costs, world: int):
    """Files of different lengths over `world` ranks: longest-processing-time-first (every item, in order of decreasing cost, goes to the
    least loaded rank; ties: the lower item index first, the lower rank first).  Returns one index list per rank, each in ascending item
    order; deterministic, so every rank computes the same split from the same list without talking to the others.  The makespan is
    within 4/3 - 1/(3 world) of the optimum (Graham 1969); round-robin over a directory sorted by name can be off by the longest file."""
    load = [0] * world
    parts = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        load[r] += costs[i]
        parts[r].append(i)
    return [sorted(p) for p in parts]


def gather_waves(local, n_items: int, dst: int = 0, group=None):
    """Collect per-rank outputs [n_local, L] on `dst` as one [n_items, L] tensor in utterance order.
    Ragged shards are padded to the largest shard for the collective and trimmed on arrival."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]
    cap = max(sizes)
    buf = local
    if local.shape[0] < cap:
        buf = torch.zeros(cap, *local.shape[1:], dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf.contiguous(), parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def gather_into(local, dest, dst: int = 0, group=None):
    """Equal shards, no allocation: every rank's `local` [n, L] lands in `dest[r]` on rank `dst` (`dest` [world, n, L],
    allocated once by the caller on `dst`, None elsewhere).  With the "nccl" backend this is RCCL's gather: world-1
    point-to-point receives over xGMI into disjoint slices of one buffer."""
    rank = dist.get_rank(group)
    dist.gather(local, list(dest.unbind(0)) if rank == dst else None, dst=dst, group=group)
    return dest if rank == dst else None


def convert_sharded(convert_fn, waves, *args, dst: int = 0, group=None, **kwargs):
    """Every rank holds the same `waves` [n_items, L] (or builds its shard from the same recipe);
    each converts its own contiguous shard with `convert_fn(shard, *args, **kwargs)` and `dst`
    returns the gathered [n_items, L'] result (None elsewhere)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(waves.shape[0], rank, world)
    out = convert_fn(waves[lo:hi], *args, **kwargs)
    return gather_waves(out, waves.shape[0], dst=dst, group=group)


# ---- a very large speaker index sharded over the GPUs of a node (SURVEY.md 8e variant) --------------------------
def merge_topk(sims, idx, k: int = 4):
    """sims, idx [..., C]: candidates (similarity, global index) -> the k best per query, ordered by similarity
    descending and, for equal similarities, index ascending (the library's tie rule).  Returns (sims, idx) [..., k]."""
    # stable two-key sort: first by index ascending, then (stable) by similarity descending
    order = torch.argsort(idx, dim=-1, stable=True)
    s1, i1 = torch.gather(sims, -1, order), torch.gather(idx, -1, order)
    order = torch.argsort(s1, dim=-1, descending=True, stable=True)
    return torch.gather(s1, -1, order)[..., :k], torch.gather(i1, -1, order)[..., :k]


def match_features_sharded(engine, source, prepared, n_local: int, shard_start: int, group=None):
    """match_features (reference feature_retrieval.py:15-33, k=4, alpha=0, cos) against an index whose vectors
    [shard_start, shard_start + n_local) live on this rank as the prepared blob `prepared`.  Every rank passes the same
    `source` [B,768,T] and gets the same [B,768,T] back.  Exchanges: one all_gather of the local top-4
    (similarity, global index) = 48 B per query and rank, and one all_reduce of the selected raw rows
    [B,T,4,768] in which every slot has exactly one non-zero contributor (so the sum is exact)."""
    world = dist.get_world_size(group)
    sims, idx = engine.knn_topk(source, prepared, n_local)
    gidx = idx + shard_start
    all_s = [torch.empty_like(sims) for _ in range(world)]
    all_i = [torch.empty_like(gidx) for _ in range(world)]
    dist.all_gather(all_s, sims.contiguous(), group=group)
    dist.all_gather(all_i, gidx.contiguous(), group=group)
    _, sel = merge_topk(torch.cat(all_s, dim=-1), torch.cat(all_i, dim=-1))
    local = sel - shard_start
    local = torch.where((local >= 0) & (local < n_local), local, torch.full_like(local, -1))
    slots = engine.knn_gather_slots(prepared, n_local, local)
    dist.all_reduce(slots, op=dist.ReduceOp.SUM, group=group)
    return engine.knn_finish(slots), sel
