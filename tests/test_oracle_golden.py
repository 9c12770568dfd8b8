"""Pin the CPU oracle (oracle/pinn_oracle.py) against vectors produced by the real reference
(oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import pinn_oracle as O
from conftest import ScalerFromArrays, load_golden, params_from_golden, unpack_mask

NAMES = O.LAMBDA_NAMES


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(*[np.array([c]) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_philox_mask_statistics():
    for p in (0.2, 0.4):
        m = O.philox_keep_mask(seed=5, stream=3, row0=1000, n_rows=512, layer_id=1, width=256, p=p)
        assert abs(m.mean() - (1 - p)) < 0.005
        # independent across rows / columns: correlation of neighbours ~ 0
        a = m.astype(np.float64) - m.mean()
        assert abs((a[:, 1:] * a[:, :-1]).mean()) < 0.003
        assert abs((a[1:] * a[:-1]).mean()) < 0.003
    # sharding invariance: a mask of rows [100,164) equals that slice of rows [0,512)
    full = O.philox_keep_mask(7, 9, 0, 512, 2, 128, 0.4)
    part = O.philox_keep_mask(7, 9, 100, 64, 2, 128, 0.4)
    assert np.array_equal(full[100:164], part)
    assert not np.array_equal(O.philox_keep_mask(7, 10, 0, 64, 2, 128, 0.4), full[:64])


def test_philox_mask_unbiased_under_reference_scale():
    """01:405-406: torch's Dropout(p) keeps with probability exactly 1 - p and scales by 1 / (1 - p), so E[mask * scale] = 1.
    The on-chip spec compares a 16-bit draw with round(65536 p): the realised keep probability times the reference's
    scale must be 1 to 2^-17 / (1 - p) (the 8-bit draws of round 2 gave 1.00098 / 1.0026 at p = 0.2 / 0.4), the sample mean of
    a large mask must agree with it within 4 standard errors, and a positive p must never round to "no dropout"."""
    for p in (0.05, 0.2, 0.4, 0.6, 0.9):
        thr = O.dropout_threshold16(p)
        assert abs((1.0 - thr / 65536.0) / (1.0 - p) - 1.0) <= 2.0 ** -17 / (1.0 - p) * 1.0001       # 3.8e-6 at p = 0.2
        m = O.philox_keep_mask(seed=21, stream=4, row0=1 << 20, n_rows=4096, layer_id=2, width=256, p=p)
        se = np.sqrt(p * (1 - p) / m.size)
        assert abs(m.mean() - (1 - p)) < 4 * se
    assert O.dropout_threshold16(0.0) == 0 and O.dropout_threshold16(1e-7) == 1
    assert not O.philox_keep_mask(1, 1, 0, 8, 0, 32, 1.0 - 1e-9).any()


@pytest.mark.parametrize("fname,H", [("g_net128.npz", 128), ("g_net256.npz", 256)])
def test_forward_eval_and_masked(fname, H):
    g = load_golden(fname)
    P = params_from_golden(g)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        u, lv = O.mlp_forward(P, x)
    assert np.array_equal(u.numpy(), g["eval_u"])
    assert np.array_equal(lv.numpy(), g["eval_logvar"])
    for p in (0.2, 0.4):
        for t in range(2):
            tag = "p%.1f_t%d" % (p, t)
            masks = [unpack_mask(g["mask%d_%s" % (l, tag)], H if l < 3 else H // 2) for l in range(4)]
            with torch.no_grad():
                u, lv = O.mlp_forward(P, x, [p] * 4, masks)
            # injected-mask replay of the reference's train-mode forward: bit exact
            assert np.array_equal(u.numpy(), g["sto_u_" + tag])
            assert np.array_equal(lv.numpy(), g["sto_logvar_" + tag])


def test_nll_loss_and_grads():
    g = load_golden("g_net128.npz")
    P = params_from_golden(g)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    masks = [unpack_mask(g["mask%d_p0.2_t0" % l], 128 if l < 3 else 64) for l in range(4)]
    loss, mse, grads, u, lv = O.nll_loss_and_grads(P, x, y, [0.2] * 4, masks)
    assert abs(loss.item() - float(g["loss_p0.2_t0"])) < 1e-7
    for n, gr in zip(O.param_names(3), grads):
        np.testing.assert_allclose(gr.numpy(), g["grad." + n], rtol=1e-5, atol=1e-8)


def _real(g, sx):
    return torch.from_numpy(O.denorm(g["x"], *O.scaler_affine(sx)))


def _lam(vals, grad=False):
    return {n: torch.tensor([v], dtype=torch.float32, requires_grad=grad) for n, v in zip(NAMES, vals)}


def test_denorm_matches_sklearn_path():
    g = load_golden("g_resid.npz")
    sx = ScalerFromArrays(g, "sx.")
    assert np.array_equal(O.denorm(g["x"], *O.scaler_affine(sx)), sx.inverse_transform(g["x"]))


@pytest.mark.parametrize("si", [0, 1])
def test_residual_tuples_and_grads(si):
    g = load_golden("g_resid.npz")
    sx, sy = ScalerFromArrays(g, "sx."), ScalerFromArrays(g, "sy.")
    real = _real(g, sx)
    ymin, yscale = O.scaler_affine(sy)
    u = torch.from_numpy(g["u_eval"])
    y = torch.from_numpy(g["y"])
    lam = _lam(g["s%d.lambdas" % si], grad=True)
    plist = [lam[n] for n in NAMES]
    calls = {"V": lambda: O.net_f_V(real, u, ymin, yscale, lam), "T": lambda: O.net_f_T_simple(real, lam),
             "H": lambda: O.net_f_H(real, lam), "O": lambda: O.net_f_O(real, lam)}
    nkeep = {"V": 9, "T": 3, "H": 4, "O": 5}
    for tag, fn in calls.items():
        res = fn()
        for j in range(nkeep[tag]):
            want = g["s%d.%s.%d" % (si, tag, j)]
            got = res[j].detach().numpy()
            np.testing.assert_allclose(got.reshape(want.shape), want, rtol=2e-6, atol=2e-6, err_msg="%s[%d]" % (tag, j))
        loss = torch.mean(res[0] ** 2)
        gs = torch.autograd.grad(loss, plist, allow_unused=True)
        assert abs(loss.item() - float(g["s%d.%s.loss" % (si, tag)])) <= 2e-6 * abs(loss.item()) + 1e-9
        got = np.array([0.0 if x is None else x.item() for x in gs])
        np.testing.assert_allclose(got, g["s%d.%s.grad" % (si, tag)], rtol=1e-5, atol=1e-9)
        assert np.array_equal(np.array([x is None for x in gs]), g["s%d.%s.grad_none" % (si, tag)])
    # train_lambda(dnn_para=False) variant
    total, phys = O.stage_loss("lambda", real, lam, y=y, u_eval=u, y_min=ymin, y_scale=yscale, u_scal=sy, dnn_para=False)
    gs = torch.autograd.grad(phys, plist, allow_unused=True)
    assert abs(phys.item() - float(g["s%d.Vn.loss" % si])) <= 2e-6 * abs(phys.item())
    np.testing.assert_allclose(np.array([0.0 if x is None else x.item() for x in gs]), g["s%d.Vn.grad" % si], rtol=1e-5, atol=1e-9)
    # Euler thermal model
    n = real.shape[0]
    # the reference runs the DNN on rows [:-1]; eval-mode rows are independent so slice u
    res = O.net_f_T(real, u[:-1], ymin, yscale, lam)
    for j in range(3):
        np.testing.assert_allclose(res[j].detach().numpy(), g["s%d.TE.%d" % (si, j)], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("stage,key,kw", [
    ("lambda", "lambdaF", dict(dnn_para=False)), ("lambda", "lambdaT", dict(dnn_para=True)),
    ("thermal", "thermal", {}), ("hydrogen", "hydrogen", {}), ("oxygen", "oxygen", {})])
def test_stage_trajectories(stage, key, kw):
    g = load_golden("g_traj.npz")
    sx, sy = ScalerFromArrays(g, "sx."), ScalerFromArrays(g, "sy.")
    real = _real(g, sx)
    ymin, yscale = O.scaler_affine(sy)
    u = torch.from_numpy(g["u_eval"])
    y = torch.from_numpy(g["y"])
    ks = [1, 2, 5, 50] + ([1003] if "%s.k1003" % key in g else [])
    lam = O.init_lambdas()
    extra = dict(y=y, u_eval=u, y_min=ymin, y_scale=yscale, u_scal=sy, **kw) if stage == "lambda" else {}
    lam, traj = O.run_stage(stage, max(ks), real, lam, **extra)
    names = O.STAGES[stage][0]
    for k in ks:
        want = g["%s.k%d" % (key, k)]
        for j, n in enumerate(names):
            w = want[NAMES.index(n)]
            # tolerance relative to the value and to the distance travelled from the initial value
            # (lr = 1.0 thermal steps amplify float32 summation-order noise)
            tol = 2e-5 * abs(w) + 2e-6 * abs(O.LAMBDA_INIT[n])
            assert abs(traj[k - 1][j] - w) <= tol, (k, n, traj[k - 1][j], w)


def test_train_dnn_three_steps_with_recorded_masks():
    g = load_golden("g_train.npz")
    P = params_from_golden(g, "w0.")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    opt = O.AdamState(P)
    for s in range(3):
        masks = [unpack_mask(g["mask%d_s%d" % (l, s)], 128 if l < 3 else 64) for l in range(4)]
        _, _, grads, _, _ = O.nll_loss_and_grads(P, x, y, [0.2] * 4, masks)
        opt.step(P, grads, O.steplr(0.01, 0.8, 1000, s))
    for n, p in zip(O.param_names(3), P):
        np.testing.assert_allclose(p.numpy(), g["w3." + n], rtol=2e-4, atol=2e-6)


def test_mc_dropout_recorded_masks():
    g = load_golden("g_mc.npz")
    P = params_from_golden(g)
    x = torch.from_numpy(g["x"])
    mf = lambda t: [unpack_mask(g["mask%d_t%d" % (l, t)], 128 if l < 3 else 64) for l in range(4)]
    pm, au, eu = O.mc_dropout(P, x, 0.4, 4, mf)
    assert np.array_equal(pm, g["pred_mean"])
    np.testing.assert_allclose(au, g["a_u"], rtol=1e-6)
    np.testing.assert_allclose(eu, g["e_u"], rtol=1e-5, atol=1e-7)


def test_moving_average_is_pandas_even_window():
    import pandas as pd
    rng = np.random.default_rng(0)
    for n, w in ((700, 200), (150, 200), (5, 4), (31, 7), (1, 200)):
        a = rng.normal(size=n)
        want = pd.Series(a).rolling(window=w, center=True, min_periods=1).mean().values
        np.testing.assert_allclose(O.moving_average_centered(a, w), want, rtol=1e-12, atol=1e-14)


def test_model_statistics_dict():
    """01:1764-1828 on 700 synthetic test rows (thermal parameters moved off their start values)."""
    g = load_golden("g_stats.npz")
    sx, sy = ScalerFromArrays(g, "sx."), ScalerFromArrays(g, "sy.")
    lam = O.init_lambdas()
    for i, v in enumerate(g["lambda_T"]):
        lam["lambda_T%d" % (i + 1)] = torch.tensor([v], dtype=torch.float32)
    st = O.model_statistics(params_from_golden(g), g["x_test"], g["y_test"], sx, sy, lam, windows=100)
    assert set(st) == {k[5:] for k in g if k.startswith("stat.")}
    for k, v in st.items():
        np.testing.assert_allclose(v, g["stat." + k], rtol=2e-5, err_msg=k)


def test_resid_fixture_weights_reproduce_u_eval():
    """g_resid.npz stores the net behind its `u_eval` (and behind net_f_T's electrochemical term): the oracle forward of
    those weights is the stored eval output bit for bit."""
    g = load_golden("g_resid.npz")
    P = params_from_golden(g)
    with torch.no_grad():
        u, _ = O.mlp_forward(P, torch.from_numpy(g["x"]))
    assert np.array_equal(u.numpy(), g["u_eval"])
