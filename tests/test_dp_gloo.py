"""CPU, world_size 2, gloo: the data-parallel scheme of pinn_amd.dp (row shards, sums divided by the
GLOBAL row count, ONE all_reduce(SUM) of the flat gradient bucket per step, loss sums in fp64 only when
logged, replicated Adam, masks keyed by global row) reproduces the single-process full-batch step, and the
minibatch schedule has the same length on every rank whatever the shard sizes.  The per-rank gradient
engine here is the CPU oracle (test infrastructure); on the GPU the same dp functions wrap the HIP engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pinn_oracle as O

H, NH, N, SEED = 128, 2, 301, 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(ts):
    return torch.cat([t.reshape(-1) for t in ts])


def _local_step(P, x, y, lo, hi, n_global, step):
    """Sum-over-local-rows gradient / n_global + raw loss sums, masks keyed by GLOBAL rows."""
    masks = O.philox_masks_for_net(SEED, step, lo, hi - lo, H, NH, [0.2] * (NH + 1))
    ps = [p.detach().clone().requires_grad_(True) for p in P]
    u, lv = O.mlp_forward(ps, x[lo:hi], [0.2] * (NH + 1), masks)
    yl = y[lo:hi]
    nll = torch.sum(0.5 * torch.exp(-lv) * (yl - u) ** 2 + 0.5 * lv)
    reg = torch.sum(torch.abs(lv))
    loss = (nll + 0.01 * reg) / n_global
    grads = torch.autograd.grad(loss, ps)
    sums = torch.tensor([nll.item(), reg.item(), torch.sum((yl - u) ** 2).item(), 0.0], dtype=torch.float64)
    return _flat(grads), sums


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pinn_amd import dp, synth
    ds = synth.make_dataset(N, (), seed=1)
    x, y = ds[0], ds[1]
    P = O.init_params([8] + [H] * NH + [1], seed=5)
    opt = O.AdamState(P)
    lo, hi = dp.shard_bounds(N, rank, world)
    losses = []
    for step in (1, 2, 3):
        g, sums = _local_step(P, x, y, lo, hi, N, step)
        bucket = torch.cat([g, torch.zeros(dp.LOSS_TAIL)])
        # the two-part form model.train_dnn overlaps with the head's weight-gradient kernels (tail of the bucket first, both
        # started asynchronously, both waited for before Adam) against ONE blocking all-reduce: the same bits
        two = bucket.clone()
        split = sum(p.numel() for p in P[:2 * (NH - 1)])          # [W0 b0 .. W_{h-2} b_{h-2}] | last hidden layer, heads
        works = [dp.allreduce_grads_begin(two[split:], None), dp.allreduce_grads_begin(two[:split], None)]
        for w in works:
            w.wait()
        dp.allreduce_grads(bucket, None)
        assert torch.equal(two, bucket), "two-part all-reduce differs from the single one"
        dp.allreduce_sums(sums, None)
        g = bucket[:-dp.LOSS_TAIL]
        grads, k = [], 0
        for p in P:
            grads.append(g[k:k + p.numel()].reshape(p.shape)); k += p.numel()
        opt.step(P, grads, 0.01)
        losses.append((sums[0].item() + 0.01 * sums[1].item()) / N)
    # residual-stage sums
    s = torch.arange(32, dtype=torch.float64) * (rank + 1)
    dp.allreduce_sums(s, None)
    assert dp.global_count(hi - lo, torch.device("cpu")) == N
    # minibatch schedule: uneven shards (rank 0: 150 rows, rank 1: 151) at batch 50 -> 4 batches on BOTH ranks,
    # the last one empty on rank 0; global sizes 100, 100, 100, 1; a rank without rows still gets the full schedule
    sched = dp.batch_schedule(hi - lo, 50, torch.device("cpu"))
    assert len(sched) == 4 and [c for _, _, c in sched] == [100, 100, 100, 1]
    assert sched[3][:2] == ((150, 150) if rank == 0 else (150, 151))
    sched0 = dp.batch_schedule(0 if rank == 0 else 120, 50, torch.device("cpu"))
    assert len(sched0) == 3 and [c for _, _, c in sched0] == [50, 50, 20]
    assert dp.batch_schedule(hi - lo, None, torch.device("cpu")) == [(0, hi - lo, N)]
    assert dp.batch_schedule(hi - lo, 151, torch.device("cpu")) == [(0, hi - lo, N)]
    q.put((rank, _flat(P).numpy(), losses, s.numpy()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process reference: full batch, same global-row masks
    from pinn_amd import synth
    ds = synth.make_dataset(N, (), seed=1)
    P = O.init_params([8] + [H] * NH + [1], seed=5)
    opt = O.AdamState(P)
    ref_losses = []
    for step in (1, 2, 3):
        masks = O.philox_masks_for_net(SEED, step, 0, N, H, NH, [0.2] * (NH + 1))
        lo, _, grads, _, _ = O.nll_loss_and_grads(P, ds[0], ds[1], [0.2] * (NH + 1), masks)
        opt.step(P, grads, 0.01)
        ref_losses.append(lo.item())
    want = _flat(P).numpy()
    assert np.array_equal(outs[0][1], outs[1][1]), "replicated Adam diverged between ranks"
    np.testing.assert_allclose(outs[0][1], want, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(outs[0][2], ref_losses, rtol=1e-5)
    np.testing.assert_allclose(outs[0][3], np.arange(32) * 3.0)
