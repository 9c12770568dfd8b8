"""Size-independent properties at BASELINE.json's full size (1 000 000 rows, [8,256,256,256,1]) in every precision
family, through the C ABI -- the regime bench.py times and the small parity cases do not reach: ~30 row tiles per
workgroup in the grid-stride loops, 256 live gradient slabs, a 7.7 GB training stash whose byte offsets pass 2^32.

  (i)   two identical pinn_mlp_train_grads calls are bitwise equal (no float atomics, fixed reduction order);
  (ii)  two 500 000-row shards with n_global = N (dropout keyed by the GLOBAL row, stash offsets below 2^32 in each
        shard) sum to the full-batch gradient: a 32-bit overflow or a tile-bookkeeping slip in the full-size call shows;
  (iii) pinn_mlp_forward and pinn_mc_dropout (T = 4) on the full array agree, on three 4096-row windows (first rows, the
        rows around 558 000 -- where a per-row stash offset would cross 2^32 bytes -- and the ragged tail), with the same
        calls on the windows alone (matching row_offset), and those agree with the oracle at the fp32 tolerance
        (the bf16-mixed family: at its rtol 2e-2 against the fp32 oracle).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pinn_oracle as O

N, H, NH, W = 1_000_000, 256, 3, 4096
PL = [0.2] * (NH + 1)
WINDOWS = [0, 558_000 - W // 2, N - W]
PRECS = [(2, "f32x6"), (0, "fp32"), (1, "bf16")]


@pytest.fixture(scope="module")
def lib():
    from pinn_amd import _lib
    return _lib.load()


@pytest.fixture(scope="module")
def data():
    import hip_helpers as hh
    from pinn_amd import synth
    ds = synth.make_dataset(N, (), seed=2026)
    P = O.init_params([8] + [H] * NH + [1], seed=17)
    return {"P": P, "x": ds[0].contiguous(), "y": ds[1].reshape(-1).contiguous(), "fp": hh.flat_params(P, H, NH).to(hh.dev()),
            "xd": ds[0].to(hh.dev()).contiguous(), "yd": ds[1].reshape(-1).to(hh.dev()).contiguous()}


def _mc(lib, fp, x, drop, T, precision):
    import hip_helpers as hh
    from pinn_amd import _lib
    n = x.shape[0]
    out = torch.empty(3, n, device=hh.dev())
    net = hh.make_net(lib, H, NH, precision)
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(x), n, ctypes.byref(drop), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                   hh.ptr(out[2]), hh.stream()), "pinn_mc_dropout")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("prec,name", PRECS)
def test_train_grads_full_size_deterministic_and_shard_additive(lib, data, prec, name):
    import hip_helpers as hh
    mk = lambda off: hh.dropout_struct(1, PL, seed=4242, stream_id=7, row_offset=off)
    g1, l1 = hh.train_grads(lib, H, NH, data["fp"], data["xd"], data["yd"], mk(0), precision=prec)
    g2, l2 = hh.train_grads(lib, H, NH, data["fp"], data["xd"], data["yd"], mk(0), precision=prec)
    assert torch.isfinite(g1).all() and torch.isfinite(l1).all()
    assert torch.equal(g1, g2) and torch.equal(l1, l2), "%s: the full-size gradient is not bitwise repeatable" % name
    del g2, l2
    cut = N // 2
    ga, la = hh.train_grads(lib, H, NH, data["fp"], data["xd"][:cut], data["yd"][:cut], mk(0), n_global=N, precision=prec)
    gb, lb = hh.train_grads(lib, H, NH, data["fp"], data["xd"][cut:], data["yd"][cut:], mk(cut), n_global=N, precision=prec)
    scale = float(g1.abs().max())
    assert scale > 0
    # fp32 sums in a different order (bf16 family: bf16-rounded stash, same tolerance relative to the largest element)
    assert float((ga + gb - g1).abs().max()) <= 2e-5 * scale, name
    np.testing.assert_allclose((la + lb).cpu().numpy()[:3], l1.cpu().numpy()[:3], rtol=1e-7)      # (each lane adds its ~30 tiles in fp32 first)


@pytest.mark.parametrize("prec,name", PRECS)
def test_forward_and_mc_full_size_windows(lib, data, prec, name):
    import hip_helpers as hh
    seed, stream, T = 99, 1000, 4
    tol = dict(rtol=1e-5, atol=1e-5) if prec != 1 else dict(rtol=2e-2, atol=2e-2)
    drop = hh.dropout_struct(1, PL, seed=seed, stream_id=stream, row_offset=0)
    u, lv = hh.forward(lib, H, NH, data["fp"], data["xd"], drop, precision=prec)
    mc = _mc(lib, data["fp"], data["xd"], hh.dropout_struct(1, [0.4] * 4, seed=seed, stream_id=stream, row_offset=0), T, prec)
    assert torch.isfinite(u).all() and torch.isfinite(lv).all() and torch.isfinite(mc).all()
    for w0 in WINDOWS:
        xs = data["xd"][w0:w0 + W].contiguous()
        dw = hh.dropout_struct(1, PL, seed=seed, stream_id=stream, row_offset=w0)
        uw, lvw = hh.forward(lib, H, NH, data["fp"], xs, dw, precision=prec)
        # the same rows through the same arithmetic: bitwise, wherever the window sits in the grid-stride schedule
        assert torch.equal(uw, u[w0:w0 + W]) and torch.equal(lvw, lv[w0:w0 + W]), "%s: forward window at row %d" % (name, w0)
        mcw = _mc(lib, data["fp"], xs, hh.dropout_struct(1, [0.4] * 4, seed=seed, stream_id=stream, row_offset=w0), T, prec)
        assert torch.equal(mcw, mc[:, w0:w0 + W]), "%s: MC-dropout window at row %d" % (name, w0)
        # and the oracle on the window (same Philox masks: global row = w0 + local row)
        xw = data["x"][w0:w0 + W]
        with torch.no_grad():
            uf, lvf = O.mlp_forward(data["P"], xw, PL, O.philox_masks_for_net(seed, stream, w0, W, H, NH, PL))
        np.testing.assert_allclose(uw.cpu().numpy(), uf.numpy().reshape(-1), **tol)
        np.testing.assert_allclose(lvw.cpu().numpy(), lvf.numpy().reshape(-1), **tol)
        pm, au, eu = O.mc_dropout(data["P"], xw, 0.4, T, lambda t: O.philox_masks_for_net(seed, stream + t, w0, W, H, NH, [0.4] * 4))
        np.testing.assert_allclose(mcw[0].cpu().numpy(), np.asarray(pm).reshape(-1), **tol)
        np.testing.assert_allclose(mcw[1].cpu().numpy(), np.asarray(au).reshape(-1), rtol=max(tol["rtol"], 2e-5), atol=tol["atol"])
        np.testing.assert_allclose(mcw[2].cpu().numpy(), np.asarray(eu).reshape(-1), rtol=max(tol["rtol"], 1e-4), atol=max(tol["atol"], 2e-5))


def test_config4_one_rank_shape(lib):
    """BASELINE configs[3] (8 x MI355X data-parallel training: 1e7 rows, per-GPU minibatch 65 536) as ONE rank sees it, where
    rank 7 of 8 would run: a 65 536-row minibatch normalised by the global batch of 8 x 65 536 = 524 288 rows, its dropout
    rows counted from global row 8 750 000.  The multi-GPU run itself cannot be made here (one GPU per box); the per-rank
    kernel path at that shape can: (i) bitwise repeatable, (ii) two half-batches (row offsets 8 750 000 and + 32 768) add
    up to the minibatch gradient, (iii) a 4096-row window alone, with its global rows, equals the oracle's gradient on
    the same Philox masks (the oracle normalises by the window, the device by n_global: rescaled)."""
    import hip_helpers as hh
    from pinn_amd import synth
    B, NG, OFF, w0 = 65536, 524288, 8_750_000, 30000
    ds = synth.make_dataset(B, (), seed=44)
    x, y = ds[0].contiguous(), ds[1].reshape(-1).contiguous()
    P = O.init_params([8] + [H] * NH + [1], seed=44)
    fp, xd, yd = hh.flat_params(P, H, NH).to(hh.dev()), x.to(hh.dev()), y.to(hh.dev())
    seed, stream = 31337, 12
    mk = lambda off: hh.dropout_struct(1, PL, seed=seed, stream_id=stream, row_offset=off)
    g1, l1 = hh.train_grads(lib, H, NH, fp, xd, yd, mk(OFF), n_global=NG, precision=2)
    g2, l2 = hh.train_grads(lib, H, NH, fp, xd, yd, mk(OFF), n_global=NG, precision=2)
    assert torch.isfinite(g1).all() and torch.equal(g1, g2) and torch.equal(l1, l2)
    cut = B // 2
    ga, la = hh.train_grads(lib, H, NH, fp, xd[:cut], yd[:cut], mk(OFF), n_global=NG, precision=2)
    gb, lb = hh.train_grads(lib, H, NH, fp, xd[cut:], yd[cut:], mk(OFF + cut), n_global=NG, precision=2)
    scale = float(g1.abs().max())
    assert float((ga + gb - g1).abs().max()) <= 2e-5 * scale
    np.testing.assert_allclose((la + lb).cpu().numpy()[:3], l1.cpu().numpy()[:3], rtol=1e-7)
    gw, lw = hh.train_grads(lib, H, NH, fp, xd[w0:w0 + W].contiguous(), yd[w0:w0 + W].contiguous(), mk(OFF + w0), n_global=NG, precision=2)
    masks = O.philox_masks_for_net(seed, stream, OFF + w0, W, H, NH, PL)
    lo, mse, go, _, _ = O.nll_loss_and_grads(P, x[w0:w0 + W], y[w0:w0 + W].reshape(-1, 1), PL, masks)
    ls = lw.cpu().numpy()
    assert abs((ls[0] + 0.01 * ls[1]) / W - lo.item()) <= 2e-5 * abs(lo.item())
    for n, a, b in zip(O.param_names(NH), hh.unflat(gw.cpu() * (NG / W), H, NH), go):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-12, n


WH, WNH, WN = 1024, 4, 262144          # BASELINE configs[4]: [8, 1024 x 4, 1], bench.py's config-5 leg trains on 262 144 rows


@pytest.fixture(scope="module")
def wide():
    import hip_helpers as hh
    from pinn_amd import synth
    ds = synth.make_dataset(WN, (), seed=55)
    P = O.init_params([8] + [WH] * WNH + [1], seed=55)
    return {"P": P, "x": ds[0].contiguous(), "y": ds[1].reshape(-1).contiguous(), "fp": hh.flat_params(P, WH, WNH).to(hh.dev()),
            "xd": ds[0].to(hh.dev()).contiguous(), "yd": ds[1].reshape(-1).to(hh.dev()).contiguous()}


def test_wide_train_grads_bench_size_deterministic_and_shard_additive(lib, wide):
    """[8,1024 x 4,1] at bench size (262 144 rows, four 65 536-row chunks through the layer kernels, 256 x 256 gradient
    blocks over 256 slices): bitwise repeatable, two 131 072-row shards add up to the full gradient."""
    import hip_helpers as hh
    pl = [0.2] * (WNH + 1)
    mk = lambda off: hh.dropout_struct(1, pl, seed=77, stream_id=5, row_offset=off)
    g1, l1 = hh.train_grads(lib, WH, WNH, wide["fp"], wide["xd"], wide["yd"], mk(0), precision=2)
    g2, l2 = hh.train_grads(lib, WH, WNH, wide["fp"], wide["xd"], wide["yd"], mk(0), precision=2)
    assert torch.isfinite(g1).all() and torch.isfinite(l1).all()
    assert torch.equal(g1, g2) and torch.equal(l1, l2), "wide net: the bench-size gradient is not bitwise repeatable"
    del g2, l2
    cut = WN // 2
    ga, la = hh.train_grads(lib, WH, WNH, wide["fp"], wide["xd"][:cut], wide["yd"][:cut], mk(0), n_global=WN, precision=2)
    gb, lb = hh.train_grads(lib, WH, WNH, wide["fp"], wide["xd"][cut:], wide["yd"][cut:], mk(cut), n_global=WN, precision=2)
    scale = float(g1.abs().max())
    assert scale > 0 and float((ga + gb - g1).abs().max()) <= 2e-5 * scale
    np.testing.assert_allclose((la + lb).cpu().numpy()[:3], l1.cpu().numpy()[:3], rtol=1e-7)


def test_wide_forward_and_mc_bench_size_windows(lib, wide):
    """Forward on all 262 144 rows against three windows alone (bitwise) and against the oracle (fp32 tolerance; 512-row
    windows: the oracle's 1024-wide layers are slow on the host), and BASELINE configs[4]'s "1024-way MC-dropout": T = 1024
    on a 4096-row slice bitwise equal to the same call on its first 64 rows alone, those against the oracle on the same
    Philox masks."""
    import hip_helpers as hh
    from pinn_amd import _lib
    pl = [0.2] * (WNH + 1)
    seed, stream, Ww = 123, 40, 512
    u, lv = hh.forward(lib, WH, WNH, wide["fp"], wide["xd"], hh.dropout_struct(1, pl, seed=seed, stream_id=stream, row_offset=0), precision=2)
    assert torch.isfinite(u).all() and torch.isfinite(lv).all()
    for w0 in (0, 65536 - Ww // 2, WN - Ww):            # first rows, across a chunk edge, the tail
        xs = wide["xd"][w0:w0 + Ww].contiguous()
        uw, lvw = hh.forward(lib, WH, WNH, wide["fp"], xs, hh.dropout_struct(1, pl, seed=seed, stream_id=stream, row_offset=w0), precision=2)
        assert torch.equal(uw, u[w0:w0 + Ww]) and torch.equal(lvw, lv[w0:w0 + Ww]), "wide forward window at row %d" % w0
        with torch.no_grad():
            uf, lvf = O.mlp_forward(wide["P"], wide["x"][w0:w0 + Ww], pl, O.philox_masks_for_net(seed, stream, w0, Ww, WH, WNH, pl))
        np.testing.assert_allclose(uw.cpu().numpy(), uf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(lvw.cpu().numpy(), lvf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    T, S, R = 1024, 4096, 64
    net = hh.make_net(lib, WH, WNH, 2)

    def mc(x, off):
        out = torch.empty(3, x.shape[0], device=hh.dev())
        d = hh.dropout_struct(1, [0.4] * (WNH + 1), seed=seed, stream_id=stream, row_offset=off)
        _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(wide["fp"]), hh.ptr(x), x.shape[0], ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                       hh.ptr(out[2]), hh.stream()), "pinn_mc_dropout")
        torch.cuda.synchronize()
        return out
    s0 = 100000
    big = mc(wide["xd"][s0:s0 + S].contiguous(), s0)
    small = mc(wide["xd"][s0:s0 + R].contiguous(), s0)
    assert torch.isfinite(big).all() and torch.equal(small, big[:, :R])
    pm, au, eu = O.mc_dropout(wide["P"], wide["x"][s0:s0 + R], 0.4, T,
                              lambda t: O.philox_masks_for_net(seed, stream + t, s0, R, WH, WNH, [0.4] * (WNH + 1)))
    np.testing.assert_allclose(small[0].cpu().numpy(), np.asarray(pm).reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(small[1].cpu().numpy(), np.asarray(au).reshape(-1), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(small[2].cpu().numpy(), np.asarray(eu).reshape(-1), rtol=1e-4, atol=2e-5)
