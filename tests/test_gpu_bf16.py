"""GPU parity of the opt-in bf16/fp32-mixed kernels (pinn_net_t.precision = PINN_PREC_BF16).

Two checks per kernel: (1) against the oracle run with the SAME rounding policy (bf16 MFMA inputs,
fp32 everything else) -- tight, only accumulation order differs; (2) against the fp32 oracle (= the
reference's arithmetic) at the mixed-precision tolerance rtol 2e-2 of SURVEY.md 8(c)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pinn_oracle as O


@pytest.fixture(scope="module")
def lib():
    from pinn_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("H,nh,N,mode", [(256, 3, 1000, 0), (256, 3, 777, 1), (128, 3, 333, 1), (128, 1, 64, 1), (256, 5, 130, 1),
                                         (512, 2, 300, 1), (1024, 4, 200, 1)])          # the last two: wide nets (pinn_wide.hip, scheme B1)
def test_forward_bf16(lib, H, nh, N, mode):
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    x = synth.make_dataset(max(N, 2), (), seed=N)[0][:N].contiguous()
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 424242, 9, 1000
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    u, lv = hh.forward(lib, H, nh, fp, xd, drop, precision=1)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode else None
    with torch.no_grad():
        ub, lvb = O.mlp_forward(P, x, pl, masks, bf16=True)
        uf, lvf = O.mlp_forward(P, x, pl, masks)
    # same rounding policy: tight (a bf16 rounding of an activation can flip on a 1-ulp fp32 difference, hence 2e-3)
    np.testing.assert_allclose(u.cpu().numpy(), ub.numpy().reshape(-1), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(lv.cpu().numpy(), lvb.numpy().reshape(-1), rtol=2e-3, atol=2e-3)
    # against the fp32 reference arithmetic: mixed-precision tolerance
    scale = float(uf.abs().max())
    assert float((u.cpu() - uf.reshape(-1)).abs().max()) <= 2e-2 * max(scale, 1.0)
    assert float((lv.cpu() - lvf.reshape(-1)).abs().max()) <= 5e-2


def test_mc_dropout_bf16(lib):
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    H, nh, N, T, p = 256, 3, 300, 16, 0.4
    P = O.init_params([8, H, H, H, 1], seed=1)
    x = synth.make_dataset(N, (), seed=2)[0]
    out = torch.empty(3, N, device=hh.dev())
    net = hh.make_net(lib, H, nh, 1)
    d = hh.dropout_struct(1, [p] * 4, seed=99, stream_id=1000, row_offset=0)
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(xd), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                   hh.ptr(out[2]), hh.stream()), "mc")
    o = out.cpu().numpy()
    mf = lambda t: O.philox_masks_for_net(99, 1000 + t, 0, N, H, nh, [p] * 4)
    # oracle with the same rounding policy
    with torch.no_grad():
        ue, _ = O.mlp_forward(P, x, bf16=True)
        us, lvs = zip(*[O.mlp_forward(P, x, [p] * 4, mf(t), bf16=True) for t in range(T)])
    us = np.array([u.numpy() for u in us]); lvs = np.array([l.numpy() for l in lvs])
    np.testing.assert_allclose(o[0], ue.numpy().reshape(-1), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(o[1], np.sqrt(np.exp(lvs.mean(0))).reshape(-1), rtol=5e-3)
    np.testing.assert_allclose(o[2], np.sqrt(us.var(0)).reshape(-1), rtol=2e-2, atol=2e-3)
    # and the fp32 reference statistics within the mixed-precision band
    pm, au, eu = O.mc_dropout(P, x, p, T, mf)
    assert abs(np.mean(o[2]) / np.mean(eu) - 1) < 2e-2 and abs(np.mean(o[1]) / np.mean(au) - 1) < 2e-2


@pytest.mark.parametrize("H,nh,N,mode", [(256, 3, 1000, 1), (128, 3, 333, 1), (256, 2, 129, 0), (128, 1, 64, 1), (512, 2, 300, 1), (1024, 4, 200, 1)])
def test_train_grads_bf16(lib, H, nh, N, mode):
    """bf16-mixed training step: loss against the same-policy oracle (tight), gradients against the fp32
    reference arithmetic at the mixed-precision tolerance (5e-2 of each tensor's max: d pre-activations are rounded to bf16 before the K = rows contraction)."""
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    ds = synth.make_dataset(N, (), seed=5)
    x, y = ds[0], ds[1].reshape(-1)
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 987654321987, 42, 12345
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    fp, xd, yd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()).contiguous(), y.to(hh.dev()).contiguous()
    grads, loss = hh.train_grads(lib, H, nh, fp, xd, yd, drop, precision=1)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode == 1 else None
    l = loss.cpu().numpy()
    with torch.no_grad():
        ub, lvb = O.mlp_forward(P, x, pl, masks, bf16=True)
        lb = O.aleatoric_loss(ds[1], ub, lvb).item()
    assert abs((l[0] + 0.01 * l[1]) / N - lb) <= 2e-3 * abs(lb) + 2e-4
    lo, mse, go, _, _ = O.nll_loss_and_grads(P, x, ds[1], pl, masks)
    assert abs((l[0] + 0.01 * l[1]) / N - lo.item()) <= 2e-2 * abs(lo.item()) + 2e-3
    got = hh.unflat(grads.cpu(), H, nh)
    for n, g, w in zip(O.param_names(nh), got, go):
        scale = float(w.abs().max()) + 1e-30
        err = float((g - w).abs().max())
        assert err <= 5e-2 * scale, (n, err, scale)
        # and the direction agrees closely
        cos = float((g * w).sum() / (g.norm() * w.norm() + 1e-30))
        assert cos > 0.998, (n, cos)


def test_model_surface_bf16_end_to_end():
    """precision="bf16" through the reference-shaped Python surface: trains, MC-samples, assembles results,
    and stays close to the fp32 path on the same seeds."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(3000, (300,), seed=0)
    outs = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True, seed=3, precision=prec)
        m.verbose = False
        m.train_dnn(1); l0 = m.last_loss
        m.train_dnn(40); l1 = m.last_loss
        assert l1 < l0
        m.train_lambda(10, False); m.train_thermal(10)
        arr = pinn_amd.create_comprehensive_results_array_v2(m, ds, mc_times=16, dropout=0.4)
        assert arr.shape == (3300, 22) and np.all(np.isfinite(arr))
        outs[prec] = (l1, arr)
    # same seeds, same masks: the two precisions follow the same trajectory closely
    assert abs(outs["bf16"][0] - outs["fp32"][0]) < 0.05 * abs(outs["fp32"][0]) + 0.02
    yp32, yp16 = outs["fp32"][1][:, 9], outs["bf16"][1][:, 9]
    assert np.abs(yp32 - yp16).max() < 0.05 * np.abs(yp32).max()
    with pytest.raises(ValueError):
        pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True, precision="fp8")
