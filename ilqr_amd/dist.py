"""Multi-GPU plumbing of the batched solver (one process per GPU, torch.distributed).

Trajectories are independent (no cross-trajectory arithmetic anywhere on the path), so the batch
is sharded in contiguous blocks -- rank r owns global trajectories [r*B, (r+1)*B) -- and there is
no data-path collective.  The one exchange is the gather of per-trajectory costs at the end of a
job (RCCL over xGMI on the GPU box; the same code runs over gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_global, rank, world_size):
    """[lo, hi) of the global batch owned by `rank`; equal contiguous blocks."""
    if n_global % world_size:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_global, world_size))
    per = n_global // world_size
    return rank * per, (rank + 1) * per


def gather_costs(local_cost):
    """all_gather of the per-trajectory costs; returns the [world*B] tensor in global order."""
    rank, ws = world()
    if ws == 1:
        return local_cost
    out = torch.empty(local_cost.numel() * ws, dtype=local_cost.dtype, device=local_cost.device)
    dist.all_gather_into_tensor(out, local_cost.contiguous())
    return out


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (elapsed time of the slowest rank)."""
    rank, ws = world()
    if ws == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    rank, ws = world()
    if ws > 1:
        dist.barrier()
