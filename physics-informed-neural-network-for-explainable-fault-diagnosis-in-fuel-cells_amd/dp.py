"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all
(SURVEY.md 2); this is the row-sharding scheme of SURVEY.md 8(e):

  * rows are independent units: rank r holds the contiguous shard [r*N/G, (r+1)*N/G);
  * every rank computes SUMS over its rows already divided by the GLOBAL row count, so ONE
    all_reduce(SUM) of one flat fp32 bucket (weight gradients + loss scalars in its tail)
    per optimizer step reproduces the full-batch gradient; Adam is replicated;
  * dropout masks are keyed by the global row index, so the result does not depend on G;
  * MC-dropout shards rows with no collective at all.

Everything here is device-agnostic torch code so the N>1 path is exercised on CPU with gloo.
"""
import torch
import torch.distributed as dist

LOSS_TAIL = 4   # floats appended to the flat gradient bucket for the loss sums


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank(group=None):
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def shard_bounds(n_rows, r, g):
    """Contiguous shard of rank r of g: [lo, hi)."""
    return (r * n_rows) // g, ((r + 1) * n_rows) // g


def allreduce_bucket(bucket, group=None):
    """In-place SUM of the flat fp32 bucket across ranks (no-op for a single process)."""
    if _active(group):
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return bucket


def allreduce_grads(grad_full, loss_sums, group=None):
    """grad_full: flat fp32 [P + LOSS_TAIL] whose first P entries the kernel filled;
    loss_sums: double[4] raw local sums.  One collective: the loss rides in the bucket tail.
    Returns loss_sums (global) in place."""
    if not _active(group):
        return loss_sums
    tail = grad_full[-LOSS_TAIL:]
    tail.copy_(loss_sums.to(torch.float32))
    allreduce_bucket(grad_full, group)
    loss_sums.copy_(tail.to(torch.float64))
    return loss_sums


def allreduce_sums(sums, group=None):
    """Residual-pass sums (double[32]) of a physics-parameter stage."""
    if _active(group):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def global_count(n_local, device, group=None):
    if not _active(group):
        return int(n_local)
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())
