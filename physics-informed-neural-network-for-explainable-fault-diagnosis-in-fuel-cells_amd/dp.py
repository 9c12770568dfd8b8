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

LOSS_TAIL = 4   # spare floats behind the flat gradient (kept for the C layout; the loss sums travel in fp64, see below)


def _active(group=None, force=False):
    """A collective is issued when there is more than one rank -- or, with force=True, whenever a process group
    exists (the world_size-1 RCCL test pushes the real buffers through the real backend that way)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return force or dist.get_world_size(group) > 1


def rank(group=None):
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def backend(group=None):
    return dist.get_backend(group) if (dist.is_available() and dist.is_initialized()) else "none"


def shard_bounds(n_rows, r, g):
    """Contiguous shard of rank r of g: [lo, hi)."""
    return (r * n_rows) // g, ((r + 1) * n_rows) // g


def allreduce_bucket(bucket, group=None, force=False):
    """In-place SUM of the flat fp32 gradient bucket across ranks (no-op for a single process)."""
    if _active(group, force):
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return bucket


def allreduce_grads(grad_full, group=None, force=False):
    """ONE collective per optimizer step: the flat fp32 gradient [P (+ LOSS_TAIL)], already divided by the global
    row count of the step on every rank.  The loss sums do NOT ride along: they are only read when a log line is
    printed (every 1000 epochs, 01:957-961) and travel then, in fp64 (allreduce_sums) -- narrowing them to the fp32
    bucket lost digits at 1e6+ rows."""
    return allreduce_bucket(grad_full, group, force)


def allreduce_grads_begin(part, group=None, force=False):
    """Start the SUM of one part of the gradient bucket and return its work handle (None for a single process): the two-part
    step of model.train_dnn puts the tail's collective on a side stream under the head's weight-gradient kernels."""
    if _active(group, force):
        return dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def allreduce_sums(sums, group=None, force=False):
    """fp64 sums: the residual-pass sums (double[32]) of a physics-parameter stage, or the loss sums (double[4])."""
    if _active(group, force):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def global_count(n_local, device, group=None):
    if not _active(group):
        return int(n_local)
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def global_max(n_local, device, group=None):
    if not _active(group):
        return int(n_local)
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def batch_schedule(n_local, batch_size, device, group=None):
    """Minibatch schedule of one epoch, identical in LENGTH on every rank: [(lo, hi, n_global_of_batch)].

    Shards differ by a row, so a schedule built from the local row count gives ranks different numbers of batches
    (131072 vs 131073 rows at 65536: 2 vs 3) and they deadlock in the gradient all-reduce.  Every rank therefore runs
    ceil(max_r n_r / batch_size) batches; batch i covers the local rows [i B, (i + 1) B) clipped to the shard -- possibly
    none, in which case the rank contributes a zero gradient -- and is normalised by the global size of batch i
    (one all-reduce of the whole size vector)."""
    n_local = int(n_local)
    n_max = global_max(n_local, device, group)
    if batch_size is None or batch_size >= n_max:
        return [(0, n_local, global_count(n_local, device, group))]
    nb = max(1, -(-n_max // int(batch_size)))
    local = [(min(n_local, i * batch_size), min(n_local, (i + 1) * batch_size)) for i in range(nb)]
    sizes = torch.tensor([hi - lo for lo, hi in local], dtype=torch.int64, device=device)
    if _active(group):
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    return [(lo, hi, int(c)) for (lo, hi), c in zip(local, sizes.tolist())]
