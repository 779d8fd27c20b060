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


def convert_sharded(convert_fn, waves, *args, dst: int = 0, group=None, **kwargs):
    """Every rank holds the same `waves` [n_items, L] (or builds its shard from the same recipe);
    each converts its own contiguous shard with `convert_fn(shard, *args, **kwargs)` and `dst`
    returns the gathered [n_items, L'] result (None elsewhere)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(waves.shape[0], rank, world)
    out = convert_fn(waves[lo:hi], *args, **kwargs)
    return gather_waves(out, waves.shape[0], dst=dst, group=group)
