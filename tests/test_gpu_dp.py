"""The product data-parallel path on the GPU:

* RCCL itself: `init_process_group("nccl", world_size=1)` in a child process, the real gradient bucket and the real
  32-double stage sums pushed through `dp.allreduce_*` with the world-size guard bypassed (force=True) -- the buffers
  must come back bit-identical and `dp.backend()` must say nccl;
* world_size 2 on ONE card (gloo carries the collectives; RCCL needs one GPU per rank): `PhysicsInformedNN.train_dnn`
  (full batch and minibatches over UNEVEN shards, 400 and 401 rows with 200-row batches: the longer shard dictates three
  batches per epoch and the shorter one's third batch has no rows -- the branch of train_dnn that steps the dropout stream,
  contributes a zero gradient and still joins the all-reduce; round 2's sizes never reached it), `train_lambda` (both variants),
  `train_thermal / hydrogen / oxygen` on row shards: parameters bit-identical across ranks and equal, within the fp32
  reduction-order tolerance, to one process on all rows.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, BATCH, SEED = 801, 200, 3       # shards of 400 and 401 rows: rank 0's third 200-row batch is EMPTY (zero gradient, same collective)
LAYERS = [8, 256, 256, 256, 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dataset():
    from pinn_amd import synth
    ds = synth.make_dataset(N, (), seed=11)
    return ds[0], ds[1], ds[4], ds[5]


def _model(x, y, sx, sy, lo=0, n_global=None):
    import pinn_amd
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(x, y, LAYERS, sx, sy, p=0.2, logvar=True, seed=SEED, row_offset=lo, n_global=n_global)
    m.verbose = False
    return m


def _state(m):
    return m.dnn.flat_params().detach().cpu().numpy().copy(), m._lambdas().detach().cpu().numpy().copy()


def _stages(m):
    m.train_lambda(3, False)
    m.train_lambda(3, True)
    m.train_thermal(3)
    m.train_hydrogen(3)
    m.train_oxygen(3)


def _rccl_worker(port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from pinn_amd import dp
        x, y, sx, sy = _dataset()
        m = _model(x, y, sx, sy)
        m.dnn.train()
        m.train_step_grads(m.x.detach(), m.u.reshape(-1), 0, N)            # fills the flat gradient bucket
        bucket = m.dnn._flat_grad_full
        want = bucket.clone()
        assert not dp._active(None) and dp._active(None, force=True)
        dp.allreduce_grads(bucket, None, force=True)
        sums = torch.arange(32, dtype=torch.float64, device=bucket.device) * 1.0000001 + 1e-9
        want_s = sums.clone()
        dp.allreduce_sums(sums, None, force=True)
        torch.cuda.synchronize()
        # the two-part form of model._dp_step on RCCL itself: the tail's collective started asynchronously on a side stream
        # behind an event, the head's on the compute stream, both waited for before the bucket is read
        import ctypes
        split = int(m._lib.pinn_grad_split(ctypes.byref(m.dnn._net)))
        assert 0 < split < bucket.numel()
        side = torch.cuda.Stream(device=bucket.device)
        ev = torch.cuda.Event(); ev.record()
        works = []
        with torch.cuda.stream(side):
            side.wait_event(ev)
            works.append(dp.allreduce_grads_begin(bucket[split:], None, force=True))
        works.append(dp.allreduce_grads_begin(bucket[:split], None, force=True))
        for w in works:
            assert w is not None
            w.wait()
        torch.cuda.synchronize()
        q.put(("ok", dp.backend(), bool(torch.equal(bucket, want)), bool(torch.equal(sums, want_s)), float(want.abs().sum().item())))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- reported to the parent, which fails the test
        q.put(("error", repr(e)))


@pytest.mark.timeout(600)
def test_rccl_allreduce_of_the_real_buckets_world1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=480)
    p.join(60)
    assert out[0] == "ok", out
    _, backend, grads_same, sums_same, norm = out
    assert backend == "nccl"
    assert grads_same and sums_same
    assert norm > 0.0              # the bucket really held gradients


def _dp_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pinn_amd import dp
        x, y, sx, sy = _dataset()
        lo, hi = dp.shard_bounds(N, rank, world)
        m = _model(x[lo:hi], y[lo:hi], sx, sy, lo=lo, n_global=N)
        m.train_dnn(3)
        a = _state(m)
        m.train_dnn(2, batch_size=BATCH)
        b = _state(m)
        _stages(m)
        c = _state(m)
        # the same two calls with ONE blocking all-reduce per step instead of the two-part overlapped one: bit for bit the same
        m1 = _model(x[lo:hi], y[lo:hi], sx, sy, lo=lo, n_global=N)
        m1.overlap_allreduce = False
        m1.train_dnn(3)
        a1 = _state(m1)
        m1.train_dnn(2, batch_size=BATCH)
        b1 = _state(m1)
        same = bool(np.array_equal(a[0], a1[0]) and np.array_equal(b[0], b1[0]))
        split = int(m._lib.pinn_grad_split(__import__("ctypes").byref(m.dnn._net)))
        q.put(("ok", rank, a, b, c, m.last_loss, same, split))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put(("error", rank, repr(e)))


def _single_process_reference():
    """One process, all rows.  The minibatch epochs are emulated batch by batch (the sharded batch i is the union of
    one row range per shard, not a contiguous range of the full series): gradients of the two ranges are summed on
    the device, then the same Adam step."""
    import ctypes
    from pinn_amd import _lib, dp
    from pinn_amd.model import _ptr, _stream
    x, y, sx, sy = _dataset()
    m = _model(x, y, sx, sy)
    m.train_dnn(3)
    a = _state(m)
    m._adam_m.zero_(); m._adam_v.zero_()
    m.dnn.train()
    xs, ys = m.x.detach(), m.u.reshape(-1)
    shards = [dp.shard_bounds(N, r, 2) for r in range(2)]
    n_batches = -(-max(hi - lo for lo, hi in shards) // BATCH)
    flat = m.dnn.flat_params()
    step = 0
    for epoch in range(2):
        for i in range(n_batches):
            pieces = [(lo + min(hi - lo, i * BATCH), lo + min(hi - lo, (i + 1) * BATCH)) for lo, hi in shards]
            n_norm = sum(e - s for s, e in pieces)
            counter = m._step_counter
            total = torch.zeros_like(m.dnn._flat_grad)
            for s, e in pieces:
                if e > s:
                    m._step_counter = counter              # both ranks draw from the same dropout stream position
                    m.train_step_grads(xs[s:e], ys[s:e], s, n_norm)
                    total += m.dnn._flat_grad
            m._step_counter = counter + 1
            m.dnn._flat_grad.copy_(total)
            step += 1
            _lib.check(m._lib.pinn_adam_step(_ptr(flat), _ptr(m.dnn._flat_grad), _ptr(m._adam_m), _ptr(m._adam_v), flat.numel(),
                                             ctypes.c_float(0.01), step, _stream()), "adam")
    b = _state(m)
    _stages(m)
    c = _state(m)
    return a, b, c


@pytest.mark.timeout(900)
def test_two_ranks_on_one_card_equal_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(o[0] == "ok" for o in outs), outs
    outs.sort(key=lambda o: o[1])
    ref = _single_process_reference()
    for stage, (r0, r1, want) in enumerate(zip(outs[0][2:5], outs[1][2:5], ref)):
        # replicated optimizer: identical inputs on every rank -> bit-identical parameters, no broadcast needed
        assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1]), "ranks diverged after stage %d" % stage
        # against one process on all rows: same sums in a different order (fp32 gradient tolerance of the parity tests)
        # (Adam divides by sqrt(v): an element whose gradient is ~0 turns reduction-order noise into a step of either sign,
        #  so a handful of the 175 362 weights may differ by a fraction of lr = 0.01; everything else agrees closely)
        err = np.abs(r0[0] - want[0])
        tol = 2e-5 + 2e-3 * np.abs(want[0])
        assert (err > tol).mean() <= 1e-4 and err.max() <= 2e-3, "weights, stage %d: %d outliers, max %.3g" % (stage, (err > tol).sum(), err.max())
        np.testing.assert_allclose(r0[1], want[1], rtol=5e-4, atol=1e-7, err_msg="physics parameters, stage %d" % stage)
    assert np.isfinite(outs[0][5]) and outs[0][5] == outs[1][5]
    # the overlapped two-part all-reduce (tail under the head's weight-gradient kernels) against the single blocking one
    assert outs[0][7] > 0 and all(o[6] for o in outs), "two-part all-reduce differs from the blocking one"
