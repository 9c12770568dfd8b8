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
