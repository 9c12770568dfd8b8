"""GPU parity of the fp32-accurate matrix-core kernels: pinn_net_t.precision = PINN_PREC_F32X6 (every product from two fp16
parts per operand, three MFMAs, fp32 accumulation; packed fp16 training stash) and PINN_PREC_F32X6_G6 (gradients from three
bf16 parts, six MFMAs), fused and wide nets.  The bar is the SAME fp32 tolerance as the exact-fp32 kernels (rtol = atol =
1e-5 against the fp32 oracle), plus float64 referees for the gradients and the range / NaN behaviour at the domain's edge."""
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


# row counts on both sides of the kernel choice: <= 128 * (#CU / 2) rows run the 4-wave / 64-row-tile kernels, more the
# 8-wave / 128-row-tile ones
@pytest.mark.parametrize("H,nh,N,mode", [(256, 3, 1000, 0), (256, 3, 777, 1), (128, 3, 333, 1), (128, 1, 64, 1), (256, 5, 130, 1),
                                         (256, 3, 1, 0), (128, 2, 4097, 0), (256, 3, 20001, 1), (128, 2, 33000, 1)])
def test_forward_x6(lib, H, nh, N, mode):
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    x = synth.make_dataset(max(N, 2), (), seed=N)[0][:N].contiguous()
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 424242, 9, 1000
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    u, lv = hh.forward(lib, H, nh, fp, xd, drop, precision=2)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode else None
    with torch.no_grad():
        uf, lvf = O.mlp_forward(P, x, pl, masks)
    np.testing.assert_allclose(u.cpu().numpy(), uf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), lvf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    # and as close to the exact-fp32 kernels as those are to the oracle
    u0, lv0 = hh.forward(lib, H, nh, fp, xd, drop, precision=0)
    np.testing.assert_allclose(u.cpu().numpy(), u0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), lv0.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_forward_x6_injected_masks(lib):
    import hip_helpers as hh
    from pinn_amd import synth
    H, nh, N = 256, 3, 500
    P = O.init_params([8, H, H, H, 1], seed=5)
    x = synth.make_dataset(N, (), seed=6)[0]
    g = torch.Generator().manual_seed(3)
    widths = [H] * nh + [H // 2]
    masks = [(torch.rand(N, w, generator=g) >= 0.2).numpy() for w in widths]
    bits = hh.pack_mask_bits([masks]).to(hh.dev())
    drop = hh.dropout_struct(2, [0.2] * 4, bits=bits)
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    u, lv = hh.forward(lib, H, nh, fp, xd, drop, precision=2)
    with torch.no_grad():
        uf, lvf = O.mlp_forward(P, x, [0.2] * 4, [torch.from_numpy(m) for m in masks])
    np.testing.assert_allclose(u.cpu().numpy(), uf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), lvf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,T", [(300, 16), (17000, 2)])
def test_mc_dropout_x6(lib, N, T):
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    H, nh, p = 256, 3, 0.4
    P = O.init_params([8, H, H, H, 1], seed=1)
    x = synth.make_dataset(N, (), seed=2)[0]
    out = torch.empty(3, N, device=hh.dev())
    net = hh.make_net(lib, H, nh, 2)
    d = hh.dropout_struct(1, [p] * 4, seed=99, stream_id=1000, row_offset=0)
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(xd), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                   hh.ptr(out[2]), hh.stream()), "mc")
    o = out.cpu().numpy()
    mf = lambda t: O.philox_masks_for_net(99, 1000 + t, 0, N, H, nh, [p] * 4)
    pm, au, eu = O.mc_dropout(P, x, p, T, mf)
    np.testing.assert_allclose(o[0], np.asarray(pm).reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o[1], np.asarray(au).reshape(-1), rtol=1e-4)
    np.testing.assert_allclose(o[2], np.asarray(eu).reshape(-1), rtol=1e-3, atol=2e-6)      # atol: the fp32 forward noise of u_t itself


def _check_grads(got_flat, want_list, H, nh, rtol, atol_scale=1e-6):
    import hip_helpers as hh
    got = hh.unflat(got_flat.cpu(), H, nh)
    for n, g, w in zip(O.param_names(nh), got, want_list):
        w = torch.as_tensor(w)
        scale = float(w.abs().max()) + 1e-30
        err = float((g - w).abs().max())
        assert err <= rtol * scale + atol_scale * scale, (n, err, scale)


@pytest.mark.parametrize("prec", [2, 3])
@pytest.mark.parametrize("H,nh,N,mode", [(256, 3, 1000, 1), (128, 3, 333, 1), (256, 2, 129, 0), (128, 4, 4096, 1), (256, 1, 64, 1),
                                         (128, 1, 200, 1), (256, 3, 17001, 1), (128, 2, 20000, 1),
                                         # H = 256 chooses its forward kernel by row count: four waves per 16-row tile up to 8192 rows
                                         # (nh = 5 and the ragged 8192 + 1 boundary here), one wave per tile in 64-row workgroups up to
                                         # 16 384, 128-row workgroups beyond
                                         (256, 5, 300, 1), (256, 3, 8192, 1), (256, 3, 8193, 1), (256, 4, 12000, 1)])
def test_train_grads_x6_vs_oracle_autograd(lib, H, nh, N, mode, prec):
    """Training step of the split-operand family (forward + NLL kernel, backward kernel, weight-gradient kernels; prec 2 =
    PINN_PREC_F32X6: two fp16 parts / three products everywhere; prec 3 = PINN_PREC_F32X6_G6: gradients from three bf16
    parts / six products): loss and all 14 gradient tensors against torch autograd on the oracle -- the SAME tolerances as
    the exact-fp32 kernels (tests/test_gpu_train.py) -- and against those kernels."""
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    ds = synth.make_dataset(N, (), seed=5)
    x, y = ds[0], ds[1].reshape(-1)
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 987654321987, 42, 12345
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    fp, xd, yd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()).contiguous(), y.to(hh.dev()).contiguous()
    grads, loss = hh.train_grads(lib, H, nh, fp, xd, yd, drop, precision=prec)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode == 1 else None
    lo, mse, go, _, _ = O.nll_loss_and_grads(P, x, ds[1], pl, masks)
    l = loss.cpu().numpy()
    assert abs((l[0] + 0.01 * l[1]) / N - lo.item()) <= 2e-5 * abs(lo.item())
    assert abs(l[2] / N - mse.item()) <= 2e-5 * abs(mse.item())
    _check_grads(grads, go, H, nh, rtol=2e-4)
    # and against the exact-fp32 kernels
    g0, l0 = hh.train_grads(lib, H, nh, fp, xd, yd, drop, precision=0)
    scale = float(g0.abs().max())
    assert float((grads - g0).abs().max()) <= 2e-5 * scale
    np.testing.assert_allclose(l[:3], l0.cpu().numpy()[:3], rtol=2e-6)


def test_train_grads_x6_injected_masks(lib):
    """G4 with the reference's recorded dropout masks through the x6 chain."""
    import hip_helpers as hh
    from conftest import load_golden, params_from_golden, unpack_mask
    g = load_golden("g_net128.npz")
    P = params_from_golden(g)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"]).reshape(-1)
    masks = [unpack_mask(g["mask%d_p0.2_t0" % l], 128 if l < 3 else 64) for l in range(4)]
    bits = hh.pack_mask_bits([masks]).to(hh.dev())
    drop = hh.dropout_struct(2, [0.2] * 4, bits=bits)
    fp, xd, yd = hh.flat_params(P, 128, 3).to(hh.dev()), x.to(hh.dev()), y.to(hh.dev())
    grads, loss = hh.train_grads(lib, 128, 3, fp, xd, yd, drop, precision=2)
    N = x.shape[0]
    l = loss.cpu().numpy()
    total = l[0] / N + 0.01 * l[1] / N
    assert abs(total - float(g["loss_p0.2_t0"])) <= 1e-5 * abs(float(g["loss_p0.2_t0"]))
    _check_grads(grads, [g["grad." + n] for n in O.param_names(3)], 128, 3, rtol=1e-4)


def test_train_grads_x6_deterministic_and_shard_additive(lib):
    """Size-independent properties through the x6 kernels: two identical calls are bitwise equal; two row shards with
    n_global = N (Philox keyed by the global row) sum to the full-batch gradient."""
    import hip_helpers as hh
    from pinn_amd import synth
    H, nh, N = 256, 3, 1536
    P = O.init_params([8, H, H, H, 1], seed=3)
    ds = synth.make_dataset(N, (), seed=9)
    fp = hh.flat_params(P, H, nh).to(hh.dev())
    x, y = ds[0].to(hh.dev()).contiguous(), ds[1].reshape(-1).to(hh.dev()).contiguous()
    mk = lambda off: hh.dropout_struct(1, [0.2] * 4, seed=77, stream_id=5, row_offset=off)
    g1, l1 = hh.train_grads(lib, H, nh, fp, x, y, mk(0), precision=2)
    g2, l2 = hh.train_grads(lib, H, nh, fp, x, y, mk(0), precision=2)
    assert torch.equal(g1, g2) and torch.equal(l1, l2)
    cut = 640
    ga, la = hh.train_grads(lib, H, nh, fp, x[:cut].contiguous(), y[:cut].contiguous(), mk(0), n_global=N, precision=2)
    gb, lb = hh.train_grads(lib, H, nh, fp, x[cut:].contiguous(), y[cut:].contiguous(), mk(cut), n_global=N, precision=2)
    scale = g1.abs().max().item()
    assert (ga + gb - g1).abs().max().item() <= 2e-5 * scale
    np.testing.assert_allclose((la + lb).cpu().numpy()[:3], l1.cpu().numpy()[:3], rtol=1e-6)


def test_mc_dropout_x6_row_shards_match(lib):
    """MC-dropout statistics of a row do not depend on which launch / shard holds it (row_offset keys the masks)."""
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    H, nh, N, T, p = 256, 3, 700, 8, 0.4
    P = O.init_params([8, H, H, H, 1], seed=4)
    x = synth.make_dataset(N, (), seed=6)[0].to(hh.dev()).contiguous()
    fp = hh.flat_params(P, H, nh).to(hh.dev())
    net = hh.make_net(lib, H, nh, 2)

    def run(xs, off):
        out = torch.empty(3, xs.shape[0], device=hh.dev())
        d = hh.dropout_struct(1, [p] * 4, seed=5, stream_id=100, row_offset=off)
        _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(xs), xs.shape[0], ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                       hh.ptr(out[2]), hh.stream()), "mc")
        torch.cuda.synchronize()
        return out
    full = run(x, 0)
    a, b = run(x[:300].contiguous(), 0), run(x[300:].contiguous(), 300)
    assert torch.equal(torch.cat([a, b], dim=1), full)


@pytest.mark.parametrize("H,nh,N,mode", [(512, 2, 300, 1), (1024, 4, 200, 1), (1024, 4, 129, 0), (512, 1, 64, 1)])
def test_forward_wide(lib, H, nh, N, mode):
    """Nets wider than the register-resident chain (BASELINE config 5 = [8, 1024 x 4, 1]): the layer-by-layer kernels
    of pinn_wide.hip against the fp32 oracle, same Philox stream, fp32 tolerance."""
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    x = synth.make_dataset(max(N, 2), (), seed=N)[0][:N].contiguous()
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 424242, 9, 1000
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    u, lv = hh.forward(lib, H, nh, fp, xd, drop, precision=2)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode else None
    with torch.no_grad():
        uf, lvf = O.mlp_forward(P, x, pl, masks)
    np.testing.assert_allclose(u.cpu().numpy(), uf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), lvf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)


def test_mc_dropout_wide(lib):
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    H, nh, N, T, p = 1024, 4, 150, 6, 0.4
    P = O.init_params([8] + [H] * nh + [1], seed=1)
    x = synth.make_dataset(N, (), seed=2)[0]
    out = torch.empty(3, N, device=hh.dev())
    net = hh.make_net(lib, H, nh, 2)
    pl = [p] * (nh + 1)
    d = hh.dropout_struct(1, pl, seed=99, stream_id=1000, row_offset=0)
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(xd), N, ctypes.byref(d), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                   hh.ptr(out[2]), hh.stream()), "mc")
    o = out.cpu().numpy()
    mf = lambda t: O.philox_masks_for_net(99, 1000 + t, 0, N, H, nh, pl)
    pm, au, eu = O.mc_dropout(P, x, p, T, mf)
    np.testing.assert_allclose(o[0], np.asarray(pm).reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o[1], np.asarray(au).reshape(-1), rtol=1e-4)
    np.testing.assert_allclose(o[2], np.asarray(eu).reshape(-1), rtol=1e-3, atol=2e-6)      # atol: the fp32 forward noise of u_t itself


@pytest.mark.parametrize("H,nh,N,mode", [(512, 2, 300, 1), (1024, 4, 200, 1), (512, 1, 129, 0), (2048, 2, 70, 1)])
def test_train_grads_wide_vs_oracle_autograd(lib, H, nh, N, mode):
    """Training step of a wide net (layer-by-layer x6 kernels + blocked weight-gradient kernels) against torch autograd
    on the oracle: loss and all 14 gradient tensors at the tolerances of the fused kernels."""
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    ds = synth.make_dataset(N, (), seed=5)
    x, y = ds[0], ds[1].reshape(-1)
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 987654321987, 42, 12345
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    fp, xd, yd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()).contiguous(), y.to(hh.dev()).contiguous()
    grads, loss = hh.train_grads(lib, H, nh, fp, xd, yd, drop, precision=2)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode == 1 else None
    lo, mse, go, _, _ = O.nll_loss_and_grads(P, x, ds[1], pl, masks)
    l = loss.cpu().numpy()
    assert abs((l[0] + 0.01 * l[1]) / N - lo.item()) <= 2e-5 * abs(lo.item())
    assert abs(l[2] / N - mse.item()) <= 2e-5 * abs(mse.item())
    _check_grads(grads, go, H, nh, rtol=2e-4)


def test_wide_injected_masks(lib):
    """Injected-mask mode of the wide path (the deterministic-parity mode of SURVEY.md 9.4, so far fused kernels only):
    forward, MC-dropout over T passes and the training gradients of [8,512,512,1] replay torch-drawn masks exactly as
    the oracle applies them."""
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    H, nh, N, T, p = 512, 2, 300, 3, 0.3
    P = O.init_params([8] + [H] * nh + [1], seed=11)
    ds = synth.make_dataset(N, (), seed=12)
    x, y = ds[0], ds[1].reshape(-1)
    g = torch.Generator().manual_seed(5)
    widths = [H] * nh + [H // 2]
    per_pass = [[(torch.rand(N, w, generator=g) >= p).numpy() for w in widths] for _ in range(T)]
    bits = hh.pack_mask_bits(per_pass).to(hh.dev())
    pl = [p] * (nh + 1)
    fp, xd, yd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()).contiguous(), y.to(hh.dev()).contiguous()
    drop = hh.dropout_struct(2, pl, bits=bits)
    u, lv = hh.forward(lib, H, nh, fp, xd, drop, precision=2)
    with torch.no_grad():
        uf, lvf = O.mlp_forward(P, x, pl, [torch.from_numpy(m) for m in per_pass[0]])
    np.testing.assert_allclose(u.cpu().numpy(), uf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), lvf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
    out = torch.empty(3, N, device=hh.dev())
    net = hh.make_net(lib, H, nh, 2)
    _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), hh.ptr(fp), hh.ptr(xd), N, ctypes.byref(drop), T, hh.ptr(out[0]), hh.ptr(out[1]),
                                   hh.ptr(out[2]), hh.stream()), "mc")
    pm, au, eu = O.mc_dropout(P, x, p, T, lambda t: [torch.from_numpy(m) for m in per_pass[t]])
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[0], np.asarray(pm).reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o[1], np.asarray(au).reshape(-1), rtol=1e-4)
    np.testing.assert_allclose(o[2], np.asarray(eu).reshape(-1), rtol=1e-3, atol=2e-6)
    grads, loss = hh.train_grads(lib, H, nh, fp, xd, yd, drop, precision=2)
    lo, mse, go, _, _ = O.nll_loss_and_grads(P, x, ds[1], pl, [torch.from_numpy(m) for m in per_pass[0]])
    l = loss.cpu().numpy()
    assert abs((l[0] + 0.01 * l[1]) / N - lo.item()) <= 2e-5 * abs(lo.item())
    _check_grads(grads, go, H, nh, rtol=2e-4)


def test_wide_model_surface():
    """BASELINE config 5's architecture through the reference-shaped Python surface: trains, predicts, MC-samples."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(500, (), seed=3)
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 1024, 1024, 1024, 1024, 1], ds[4], ds[5], p=0.2, logvar=True, seed=1)
    m.verbose = False
    w0 = m.dnn.state_dict()["layers.layer_2.weight"].clone()
    m.train_dnn(3)
    assert not torch.equal(w0, m.dnn.state_dict()["layers.layer_2.weight"])
    u, lv = m.predict(ds[0], ds[4])
    assert u.shape == (500, 1) and np.all(np.isfinite(u)) and np.all(np.isfinite(lv))
    pm, au, eu = pinn_amd.get_MC_samples(m, ds[0], ds[4], mc_times=4, dropout=0.4)
    assert np.all(np.isfinite(pm)) and np.all(au > 0) and np.all(eu > 0)
    with pytest.raises(Exception):
        m.dnn.set_precision("fp32"); m.predict(ds[0], ds[4])          # wide nets: split-operand or bf16-mixed arithmetic only


def test_forward_wide_multi_chunk(lib):
    """More rows than one scratch chunk (65 536): rows on both sides of the chunk edge against the oracle."""
    import hip_helpers as hh
    H, nh, N = 512, 1, 65536 + 300
    P = O.init_params([8, H, 1], seed=9)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, 8, generator=g) * 2 - 1
    fp, xd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev())
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 7, 3, 500
    drop = hh.dropout_struct(1, pl, seed=seed, stream_id=stream, row_offset=row0)
    u, lv = hh.forward(lib, H, nh, fp, xd, drop, precision=2)
    u, lv = u.cpu().numpy(), lv.cpu().numpy()
    for lo, hi in ((0, 200), (65436, N)):
        masks = O.philox_masks_for_net(seed, stream, row0 + lo, hi - lo, H, nh, pl)
        with torch.no_grad():
            uf, lvf = O.mlp_forward(P, x[lo:hi], pl, masks)
        np.testing.assert_allclose(u[lo:hi], uf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(lv[lo:hi], lvf.numpy().reshape(-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("H,nh", [(256, 3), (512, 2)])
def test_weight_outside_the_split_operand_range_is_reported(lib, H, nh):
    """Scheme X3 stores fp16(64 w): one weight of 2000 in a hidden matrix is outside its domain, while the fp32 reference
    (01:389-438) computes a finite output.  The kernels must not return wrong finite numbers quietly: the forward output of
    every row is non-finite, pinn_net_range_status reports PINN_E_RANGE after the forward AND after a training call, the
    record clears when the weight is back in range, and -- fused nets -- exact fp32 matches the oracle on the same weights."""
    import hip_helpers as hh
    from pinn_amd import _lib, synth
    N = 300
    P = O.init_params([8] + [H] * nh + [1], seed=5)
    x = synth.make_dataset(N, (), seed=5)[0]
    y = torch.zeros(N)
    good = hh.flat_params(P, H, nh).to(hh.dev())
    P_bad = [p.clone() for p in P]
    P_bad[2][7, 11] = 2000.0                                   # layers.layer_1.weight
    bad = hh.flat_params(P_bad, H, nh).to(hh.dev())
    xd = x.to(hh.dev())
    net = hh.make_net(lib, H, nh, 2)
    status = lambda: lib.pinn_net_range_status(ctypes.byref(net), hh.stream())
    u, lv = hh.forward(lib, H, nh, good, xd, precision=2)
    assert status() == 0 and bool(torch.isfinite(u).all())
    u, lv = hh.forward(lib, H, nh, bad, xd, precision=2)
    assert status() == -4
    assert not bool(torch.isfinite(u).any()), "a weight outside fp16(64 w) must not give finite outputs quietly"
    g, _ = hh.train_grads(lib, H, nh, bad, xd, y.to(hh.dev()), None, precision=2)
    assert status() == -4
    g, _ = hh.train_grads(lib, H, nh, good, xd, y.to(hh.dev()), None, precision=2)
    assert status() == 0 and bool(torch.isfinite(g).all())
    with torch.no_grad():
        uo, lvo = O.mlp_forward(P_bad, x)
    assert bool(torch.isfinite(uo).all())
    if H <= 256:
        u32, lv32 = hh.forward(lib, H, nh, bad, xd, precision=0)
        np.testing.assert_allclose(u32.cpu().numpy(), uo.numpy().reshape(-1), rtol=2e-5, atol=2e-5)
        fp32net = hh.make_net(lib, H, nh, 0)
        assert lib.pinn_net_range_status(ctypes.byref(fp32net), hh.stream()) == 0


def test_model_check_range_raises():
    """The Python surface turns PINN_E_RANGE into an exception where it synchronises anyway (predict, get_MC_samples,
    train_dnn's log lines), and `set_precision("fp32")` is the way out the message names."""
    import pinn_amd
    from pinn_amd import _lib, synth
    ds = synth.make_dataset(200, (), seed=1)
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 128, 128, 1], ds[4], ds[5], p=0.2, logvar=True)
    m.verbose = False
    m.predict(ds[0], ds[4])
    with torch.no_grad():
        m.dnn.state_dict()["layers.layer_1.weight"][3, 4] = -5000.0
    with pytest.raises(_lib.PinnRangeError):
        m.predict(ds[0], ds[4])
    with pytest.raises(_lib.PinnRangeError):
        pinn_amd.get_MC_samples(m, ds[2], ds[4], mc_times=2, dropout=0.4)
    m.dnn.set_precision("fp32")
    u, lv = m.predict(ds[0], ds[4])
    assert np.all(np.isfinite(u)) and np.all(np.isfinite(lv))
    m.dnn.set_precision("f32x6")
    with pytest.raises(_lib.PinnRangeError):
        m.train_dnn(1)              # (the step itself has run by the time the log line checks: the weights are NaN afterwards)


@pytest.mark.parametrize("prec", [2, 3])
def test_gradient_error_no_worse_than_torch_fp32(lib, prec):
    """How exact is "fp32-accurate"?  Every gradient tensor of [8,256,256,256,1] on 4096 rows against a float64 autograd of
    the oracle, beside torch's own fp32 autograd (the reference's arithmetic) against the same: for the default precision the
    device's rms error must not exceed 2x torch's and its largest error 2x torch's largest, every tensor -- the factor the
    test was written with (round 2 widened it to 3 after a red run; measured now: 0.04-1.5, tools/diag_grad_err.py).
    The opt-in PINN_PREC_F32X6_G6 is held to 3x, for a reason that is the hardware's: its hidden-layer BIAS gradients sit at
    2.3-2.4x, twice the default's, because it issues twice the MFMAs per product and a 16-bit MFMA does not round its 32
    products into the accumulator as one sum -- products below the accumulator's last bits are cut off one by one
    (tools/mfma_round_probe.hip: 32 products of 1.75 / 32 ulp each leave the accumulator unchanged, one product of 1.75 ulp
    rounds it up by 2) -- a bias of constant sign that survives in a sum over rows with no weights in it; the exact-fp32
    MFMA is an fmaf chain and shows 0.8-1.1x.  Slab reduction and bias sums accumulate in float64 in every family."""
    K = 2.0 if prec == 2 else 3.0
    import hip_helpers as hh
    from pinn_amd import synth
    H, nh, N = 256, 3, 4096
    pl = [0.2] * (nh + 1)
    ds = synth.make_dataset(N, (), seed=17)
    x, y = ds[0].contiguous(), ds[1].reshape(-1, 1).contiguous()
    P = O.init_params([8] + [H] * nh + [1], seed=17)
    masks = O.philox_masks_for_net(99, 7, 0, N, H, nh, pl)
    _, _, g32, _, _ = O.nll_loss_and_grads(P, x, y, pl, masks)
    _, _, g64, _, _ = O.nll_loss_and_grads([p.double() for p in P], x.double(), y.double(), pl, masks)
    drop = hh.dropout_struct(1, pl, seed=99, stream_id=7, row_offset=0)
    g, _ = hh.train_grads(lib, H, nh, hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()), y.reshape(-1).to(hh.dev()), drop, precision=prec)
    for n, a, b, c in zip(O.param_names(nh), hh.unflat(g.cpu(), H, nh), g32, g64):
        if a.numel() < 64:
            continue
        a, b, c = a.double().numpy().reshape(-1), b.double().numpy().reshape(-1), c.numpy().reshape(-1)
        rms = lambda e: float(np.sqrt((e ** 2).mean()))
        assert rms(a - c) <= K * rms(b - c) + 1e-9 * rms(c), (n, rms(a - c) / rms(c), rms(b - c) / rms(c))
        assert np.abs(a - c).max() <= K * np.abs(b - c).max() + 1e-8 * np.abs(c).max(), (n, np.abs(a - c).max(), np.abs(b - c).max())


@pytest.mark.parametrize("prec", [2, 3])
def test_train_grads_outlier_rows_against_float64(lib, prec):
    """Dynamic range of one call's gradients: four of 2048 rows carry targets 1e3 away (their d loss / d u is ~1e3 x the
    others'), which sets the common power-of-two scale of the fp16 weight-gradient kernels (PINN_PREC_F32X6) far above the
    typical row.  Every gradient tensor must still match a float64 autograd to 2e-6 of its largest element, and the two
    split schemes must agree element by element."""
    import hip_helpers as hh
    from pinn_amd import synth
    H, nh, N = 256, 3, 2048
    pl = [0.2] * (nh + 1)
    ds = synth.make_dataset(N, (), seed=23)
    x, y = ds[0].contiguous(), ds[1].reshape(-1, 1).clone()
    y[[5, 700, 1333, 2047]] += torch.tensor([[1e3], [-1e3], [3e2], [1e3]])
    P = O.init_params([8] + [H] * nh + [1], seed=23)
    masks = O.philox_masks_for_net(7, 3, 0, N, H, nh, pl)
    _, _, g64, _, _ = O.nll_loss_and_grads([p.double() for p in P], x.double(), y.double(), pl, masks)
    drop = hh.dropout_struct(1, pl, seed=7, stream_id=3, row_offset=0)
    fp = hh.flat_params(P, H, nh).to(hh.dev())
    run = lambda p: hh.unflat(hh.train_grads(lib, H, nh, fp, x.to(hh.dev()), y.reshape(-1).to(hh.dev()), drop, precision=p)[0].cpu(), H, nh)
    got, other = run(prec), run(5 - prec)
    for n, a, b, c in zip(O.param_names(nh), got, other, g64):
        a, b, c = a.double().numpy().reshape(-1), b.double().numpy().reshape(-1), c.numpy().reshape(-1)
        scale = np.abs(c).max()
        assert np.isfinite(a).all() and np.abs(a - c).max() <= 2e-6 * scale, (n, np.abs(a - c).max() / scale)
        assert np.abs(a - b).max() <= 2e-6 * scale, (n, np.abs(a - b).max() / scale)
