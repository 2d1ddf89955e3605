"""Multi-GPU layout of the batch: instances are independent, so rank r owns the contiguous index
range [r*B/G, (r+1)*B/G) and generates its inputs from the instance ids -- no scatter, no collective inside the
solve.  The only exchange is the final gather of per-instance results (RCCL over xGMI on GPUs, gloo in tests)."""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, B, rank, world):
    """all_gather of row-sharded per-instance results (ragged last shards are padded)."""
    if world == 1:
        return local
    per = (B + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts, dim=0)


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
