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


class ResultGather:
    """The one exchange of the multi-GPU run: every rank contributes its shard's results as ONE block of
    (B/G) x (nV + 2) doubles -- x (nV), fval, and exit flag / iteration count packed into one double (flag * 1024 + iter, both
    exact in fp64) -- and receives all G blocks with one `all_gather_into_tensor` (RCCL over xGMI on GPUs, gloo in tests).
    Send and receive buffers are allocated once, here: nothing is allocated inside a timed step.  Equal shards (weak scaling:
    the same batch on every rank); a ragged global batch uses gather_rows."""

    def __init__(self, per_rank, nV, world, device):
        self.per, self.nV, self.world = per_rank, nV, world
        self.send = torch.zeros((per_rank, nV + 2), dtype=torch.float64, device=device)
        self.recv = torch.zeros((world * per_rank, nV + 2), dtype=torch.float64, device=device) if world > 1 else self.send
        self._tmp = torch.zeros(per_rank, dtype=torch.float64, device=device)

    def pack(self, x, fval, exitflag, it):
        """copies into the preallocated send block (dtype conversions through copy_: no temporaries)"""
        nV = self.nV
        self.send[:, :nV].copy_(x)
        self.send[:, nV].copy_(fval)
        self._tmp.copy_(exitflag)
        self._tmp.mul_(1024.0)
        code = self.send[:, nV + 1]
        code.copy_(it)
        code.add_(self._tmp)
        return self.send

    def gather(self):
        """-> the (G * B/G) x (nV + 2) block of all ranks, rank-major = global instance order of the index-pure shards"""
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv, self.send)
        return self.recv

    @staticmethod
    def unpack(block, nV):
        """-> x, fval, exitflag, iter"""
        code = block[:, nV + 1]                     # flag * 1024 + iter, 0 <= iter < 1024
        fl = torch.floor(code / 1024.0)
        it = code - fl * 1024.0
        return block[:, :nV], block[:, nV], fl.to(torch.int32), it.to(torch.int32)


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
