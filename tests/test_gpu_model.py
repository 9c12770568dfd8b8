"""GPU parity of the Python surface (PhysicsInformedNN / get_MC_samples / results array) against
the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pinn_oracle as O
from conftest import ScalerFromArrays, load_golden, unpack_mask

NAMES = O.LAMBDA_NAMES


def _model_from_golden(g, x, y, sx, sy, H=128, prefix="w."):
    import pinn_amd
    m = pinn_amd.PhysicsInformedNN(torch.from_numpy(x), torch.from_numpy(y), [8, H, H, H, 1], sx, sy, p=0.2, logvar=True)
    m.verbose = False
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    missing, unexpected = m.dnn.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("lambda") for k in missing)
    return m


def test_surface_and_state_dict_quirk():
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(300, (), seed=0)
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
    sd = m.dnn.state_dict()
    assert len(sd) == 30                                   # 14 weight/bias tensors + 16 registered lambdas (01:465-528)
    assert dict(m.dnn.named_parameters())["lambda_3"] is m.lambda_4    # the 01:468 registration quirk
    assert sum(p.numel() for k, p in m.dnn.named_parameters() if not k.startswith("lambda")) == 175362
    drops = [n for n, mod in m.dnn.named_modules() if isinstance(mod, torch.nn.Dropout)]
    assert drops == ["layers.dropout_0", "layers.dropout_1", "layers.dropout_2", "var_layers.2"]
    assert abs(m.lambda_2.item() - 2.36682075851268e-06) < 1e-12
    u, lv = m.net_u(m.x)
    assert u.shape == (300, 1) and lv.shape == (300, 1) and u.is_cuda
    res = m.net_f_V(m.X, m.x_scal)
    assert len(res) == 9 and res[0].shape == (300, 1) and res[7] is m.lambda_3
    assert len(m.net_f_T_simple(m.X, m.x_scal)) == 3 and len(m.net_f_H(m.X, m.x_scal)) == 5 and len(m.net_f_O(m.X, m.x_scal)) == 5
    up, lvp = m.predict(ds[0], ds[4])
    assert isinstance(up, np.ndarray) and up.shape == (300, 1)
    with pytest.raises(ValueError):
        pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 100, 100, 1], ds[4], ds[5], p=0.2, logvar=True)


@pytest.mark.parametrize("key,call", [
    ("lambdaF", lambda m, k: m.train_lambda(k, False)), ("lambdaT", lambda m, k: m.train_lambda(k, True)),
    ("thermal", lambda m, k: m.train_thermal(k)), ("hydrogen", lambda m, k: m.train_hydrogen(k)),
    ("oxygen", lambda m, k: m.train_oxygen(k))])
def test_stage_trajectories_golden(key, call):
    """G6: lambdas after k steps of each physics stage == the reference's (Adam + clamp + StepLR order)."""
    g = load_golden("g_traj.npz")
    sx, sy = ScalerFromArrays(g, "sx."), ScalerFromArrays(g, "sy.")
    ks = [1, 2, 5, 50] + ([1003] if "%s.k1003" % key in g else [])
    for k in ks:
        m = _model_from_golden(g, g["x"], g["y"], sx, sy)
        call(m, k)
        got = m._lambda.cpu().numpy().astype(np.float64)
        want = g["%s.k%d" % (key, k)]
        for j, n in enumerate(NAMES):
            tol = 5e-5 * abs(want[j]) + 5e-6 * abs(O.LAMBDA_INIT[n])
            assert abs(got[j] - want[j]) <= tol, (key, k, n, got[j], want[j])


def test_net_f_T_euler_golden():
    """F4 / 01:767-867: all three tuple elements of the fused net_f_T kernel against the reference's own TE.* vectors
    (g_resid.npz carries the weights of the net the reference evaluated), at both parameter sets -- the
    electrochemical term (DNN voltage of row t-1) included -- and the one-row halo under row sharding."""
    g = load_golden("g_resid.npz")
    sx, sy = ScalerFromArrays(g, "sx."), ScalerFromArrays(g, "sy.")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    m = _model_from_golden(g, g["x"], g["y"], sx, sy)
    m.dnn.eval()
    names = ["lambda_1", "lambda_2", "lambda_3", "lambda_4", "lambda_T1", "lambda_T2", "lambda_T3", "lambda_T4", "lambda_T5",
             "lambda_H1", "lambda_H2", "lambda_H3", "lambda_H4", "lambda_O1", "lambda_O2", "lambda_O3", "lambda_O4"]
    for si in (0, 1):
        for n, v in zip(names, g["s%d.lambdas" % si]):
            getattr(m, n).data.fill_(float(v))
        res = m.net_f_T(m.X, sx)
        for j in range(3):
            want = g["s%d.TE.%d" % (si, j)]
            # T_pred is ~1e3..1e5 with the untrained thermal parameters (lambda_T = 10): rtol of the fp32 forward that
            # feeds the electrochemical term; T_out itself is bit-exact
            np.testing.assert_allclose(res[j].cpu().numpy().reshape(want.shape), want, rtol=2e-5, atol=1e-3, err_msg="s%d TE.%d" % (si, j))
        assert np.array_equal(res[2].cpu().numpy().reshape(-1), g["s%d.TE.2" % si].reshape(-1))
        # row shards: the second shard gets the row before it (and its DNN output) as halo -> identical rows
        cut = x.shape[0] // 2 + 1
        u_all = m.net_u(m.x.detach())[0].reshape(-1)
        lo = m.net_f_T(x[:cut], sx)
        hi = m.net_f_T(x[cut:], sx, halo=(x[cut - 1], u_all[cut - 1]))
        for j in range(3):
            both = torch.cat([lo[j], hi[j]], dim=0)
            assert torch.equal(both, res[j]), "shards differ from the full series, element %d" % j
    z = m.net_f_T(x[:1], sx)
    assert all(t.shape == (1, 1) and float(t.abs().sum()) == 0.0 for t in z)          # 01:774-778


def test_train_dnn_three_steps_golden():
    """G7: weights after 3 train_dnn steps with the masks the reference drew."""
    import hip_helpers as hh
    g = load_golden("g_train.npz")
    sc = load_golden("g_traj.npz")
    sx, sy = ScalerFromArrays(sc, "sx."), ScalerFromArrays(sc, "sy.")
    m = _model_from_golden(g, g["x"], g["y"], sx, sy, prefix="w0.")
    per_step = [[unpack_mask(g["mask%d_s%d" % (l, s)], 128 if l < 3 else 64) for l in range(4)] for s in range(3)]
    m.dnn.inject_masks(hh.pack_mask_bits(per_step))
    m.use_graph, m.graph_min_steps = True, 0      # through the replayed-graph path (opt-in, and by default only for calls of >= 200 steps)
    m.train_dnn(3)
    sd = m.dnn.state_dict()
    for n in O.param_names(3):
        np.testing.assert_allclose(sd[n].cpu().numpy(), g["w3." + n], rtol=5e-4, atol=5e-6, err_msg=n)


def test_results_array_golden_and_mat_roundtrip(tmp_path):
    """G9/G10: the [N,22] comprehensive_results array (MC masks replayed) and the .mat contract of scripts 02-05."""
    import hip_helpers as hh
    import scipy.io
    import pinn_amd
    from pinn_amd import synth
    g = load_golden("g_results.npz")
    ds = synth.make_dataset(300, (150, 250), seed=3)
    assert np.array_equal(ds[2].numpy(), g["x_test"])
    m = _model_from_golden(g, ds[0].numpy(), ds[1].numpy(), ds[4], ds[5])
    per_pass = [[unpack_mask(g["mask%d_t%d" % (l, t)], 128 if l < 3 else 64) for l in range(4)] for t in range(3)]
    m.dnn.inject_masks(hh.pack_mask_bits(per_pass))
    arr = pinn_amd.create_comprehensive_results_array_v2(m, ds, mc_times=3, dropout=0.4)
    want = g["results"]
    assert arr.shape == want.shape == (700, 22) and arr.dtype == np.float64
    for c in range(22):
        scale = np.abs(want[:, c]).max() + 1e-30
        tol = 2e-5 if c != 11 else 2e-4          # epistemic std of 3 passes: cancellation-limited
        assert np.abs(arr[:, c] - want[:, c]).max() <= tol * scale, (c, np.abs(arr[:, c] - want[:, c]).max(), scale)
    assert np.array_equal(arr[:, 17], want[:, 17])
    # restored dropout rate and eval mode (01:1468-1473)
    assert all(mod.p == 0.2 for mod in m.dnn.dropout_modules()) and not m.dnn.training
    path = os.path.join(tmp_path, "F01_output.mat")
    scipy.io.savemat(path, {"comprehensive_results": arr})
    back = scipy.io.loadmat(path)["comprehensive_results"]           # what 02:105-114 reads
    assert back.ndim == 2 and back.shape[1] >= 18 and np.array_equal(back, arr)


def test_get_mc_samples_surface():
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(500, (), seed=1)
    torch.manual_seed(1)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
    pm, au, eu = pinn_amd.get_MC_samples(m, ds[2], ds[4], mc_times=32, dropout=0.4)
    assert pm.shape == au.shape == eu.shape == (500,) and pm.dtype == np.float32
    m.dnn.eval()
    u, _ = m.net_u(m.x)
    np.testing.assert_allclose(pm, u.cpu().numpy().reshape(-1), rtol=1e-6, atol=1e-6)
    assert np.all(eu > 0) and np.all(au > 0) and np.all(np.isfinite(eu))


def test_config1_end_to_end_small():
    """BASELINE config 1 in miniature: train_dnn lowers the loss, all five stages run, results assemble."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(2000, (200,), seed=0)
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
    m.verbose = False
    m.train_dnn(1); l0 = m.last_loss
    m.train_dnn(60); l1 = m.last_loss
    assert l1 < l0
    m.train_lambda(20, False); m.train_lambda(20, True); m.train_thermal(20); m.train_hydrogen(20); m.train_oxygen(20)
    assert np.all(np.isfinite(m._lambda.cpu().numpy()))
    arr = pinn_amd.create_comprehensive_results_array_v2(m, ds, mc_times=8, dropout=0.4)
    assert arr.shape == (2200, 22) and np.all(np.isfinite(arr))
    assert set(np.unique(arr[:, 17])) == {0.0, 1.0}


def test_stage_run_persistent_matches_iterated_kernels():
    """the whole-stage persistent kernel (pinn_lambda_stage_run) against the same stage iterated through
    pinn_residuals + pinn_lambda_step (only the order of the row sums differs), all five stage trainers, 300 iterations
    across... (StepLR boundaries are covered by the golden trajectory test, which runs through the persistent path)."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(5000, (), seed=4)

    def run(persistent):
        torch.manual_seed(0)
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 128, 128, 1], ds[4], ds[5], p=0.2, logvar=True, seed=2)
        m.verbose = False
        if not persistent:
            m.stage_run_max_rows = 0
        m.train_lambda(300, False); m.train_lambda(300, True); m.train_thermal(300); m.train_hydrogen(300); m.train_oxygen(300)
        return m._lambda.cpu().numpy().copy(), m.last_loss
    a, la = run(True)
    b, lb = run(False)
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9)
    assert abs(la - lb) <= 1e-5 * abs(lb)


def test_model_statistics_golden():
    """F4: the statistics dict of plot_model_results_detailed_split (01:1764-1828) from the kernels' outputs."""
    import pinn_amd
    from pinn_amd import synth
    g = load_golden("g_stats.npz")
    ds = synth.make_dataset(300, (150, 250), seed=5)
    assert np.array_equal(ds[2].numpy(), g["x_test"])
    m = _model_from_golden(g, ds[0].numpy(), ds[1].numpy(), ds[4], ds[5])
    with torch.no_grad():
        for i, v in enumerate(g["lambda_T"]):
            getattr(m, "lambda_T%d" % (i + 1)).fill_(float(v))
    st = pinn_amd.plot_model_results_detailed_split(m, ds, windows=100)
    assert set(st) == {k[5:] for k in g if k.startswith("stat.")}
    for k, v in st.items():
        np.testing.assert_allclose(v, g["stat." + k], rtol=5e-5, err_msg=k)
    assert not m.dnn.training


def test_checkpoint_roundtrip(tmp_path):
    """F4: save -> load into a fresh model reproduces parameters, physics parameters and the next dropout masks."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(400, (), seed=2)

    def fresh(seed):
        torch.manual_seed(seed)
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 128, 128, 128, 1], ds[4], ds[5], p=0.2, logvar=True)
        m.verbose = False
        return m
    a = fresh(1)
    a.train_dnn(5); a.train_hydrogen(5)
    path = os.path.join(tmp_path, "ck.pt")
    pinn_amd.save_checkpoint(a, path)
    ck = torch.load(path, weights_only=True)                      # tensors only
    assert "dnn.layers.layer_0.weight" in ck and "lambda_H3" in ck
    b = fresh(2)
    assert not torch.equal(a.dnn._flat, b.dnn._flat)
    pinn_amd.load_checkpoint(b, path)
    assert torch.equal(a.dnn.flat_params(), b.dnn.flat_params()) and torch.equal(a._lambdas(), b._lambdas())
    for k, v in a.dnn.state_dict().items():
        assert torch.equal(v, b.dnn.state_dict()[k]), k
    # the restored model continues exactly like the saved one (same dropout stream position)
    a.train_dnn(3); b.train_dnn(3)
    assert torch.equal(a.dnn.flat_params(), b.dnn.flat_params())
    pa = pinn_amd.get_MC_samples(a, ds[2], ds[4], mc_times=4, dropout=0.4)
    pb = pinn_amd.get_MC_samples(b, ds[2], ds[4], mc_times=4, dropout=0.4)
    assert all(np.array_equal(x, y) for x, y in zip(pa, pb))
    c = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, 256, 256, 256, 1], ds[4], ds[5], p=0.2, logvar=True)
    with pytest.raises(ValueError):
        pinn_amd.load_checkpoint(c, path)


def _adam_noise_bands(P0, x, y, bounds, epochs, mask_fn, lr, n_noise=12, seed=0):
    """Float64 referee for a train_dnn trajectory (01:948-955) and, per element, what fp32-grade gradient noise does to it.

    Adam divides by sqrt(v) + eps: an element whose gradient is near eps turns ANY rounding noise of the gradient into a
    visible step, so a fixed band around a reference trajectory (or a comparison of two largest errors) measures luck, not
    arithmetic.  Here the float64 trajectory is re-run `n_noise` times with every gradient tensor perturbed by Gaussian noise
    of the rms the REFERENCE's own fp32 arithmetic shows against float64 on that tensor at that step (torch fp32 autograd at
    the same weights: nothing tuned); the spread of the final weights is the band an fp32-accurate implementation may use,
    element by element.  Returns (w64, w32, spread): float64 trajectory, the fp32 oracle's, per-element rms deviation."""
    P64 = [p.double().clone() for p in P0]
    P32 = [p.clone() for p in P0]
    noisy = [[p.double().clone() for p in P0] for _ in range(n_noise)]
    opt64, opt32, optn = O.AdamState(P64), O.AdamState(P32), [O.AdamState(q) for q in noisy]
    gen = torch.Generator().manual_seed(seed)
    x64, y64 = x.double(), y.double()
    step = 0
    for epoch in range(epochs):
        for s, e in bounds:
            step += 1
            masks, pl = mask_fn(step, s, e)
            _, _, g64, _, _ = O.nll_loss_and_grads(P64, x64[s:e], y64[s:e], pl, masks)
            # the reference's arithmetic at the same weights: its error on each tensor sets that tensor's noise level
            _, _, g32r, _, _ = O.nll_loss_and_grads([p.float() for p in P64], x[s:e], y[s:e], pl, masks)
            sig = [float((a.double() - b).pow(2).mean().sqrt()) for a, b in zip(g32r, g64)]
            for q, on in zip(noisy, optn):
                _, _, gq, _, _ = O.nll_loss_and_grads(q, x64[s:e], y64[s:e], pl, masks)
                on.step(q, [g + sg * torch.randn(g.shape, generator=gen, dtype=torch.float64) for g, sg in zip(gq, sig)], lr)
            opt64.step(P64, g64, lr)
            _, _, g32, _, _ = O.nll_loss_and_grads(P32, x[s:e], y[s:e], pl, masks)
            opt32.step(P32, g32, lr)
    spread = [torch.stack([q[i] - P64[i] for q in noisy]).pow(2).mean(0).sqrt().numpy() for i in range(len(P0))]
    return [p.numpy() for p in P64], [p.double().numpy() for p in P32], spread


@pytest.mark.parametrize("batch_size", [None, 256])
def test_train_dnn_philox_masks_vs_oracle(batch_size):
    """train_dnn with the on-chip Philox masks, full batch and minibatches (BASELINE config 4's shape in miniature): the
    oracle replays the same masks (stream = optimizer step, row offset = first row of the batch) and takes the same Adam
    steps; a minibatch is normalised by its own row count (01:949-955 applied per batch).

    Referee: the same trajectory in float64 (`_adam_noise_bands`).  EVERY element of every tensor must lie within
    K = 8 times the spread that fp32-grade gradient noise induces on that element, plus four fp32 ulps per optimizer step of the
    largest number the update handles (the weight before, after, or the step lr itself: Adam's moments, square root, quotient
    and `p -= ...` each round once in fp32, in the reference as on the device);
    K was fixed before the first run.  The fp32 oracle trajectory -- the
    reference's own arithmetic -- is held to the same band, so a band it cannot meet itself would show.  (Round 2 widened a
    fixed band around the fp32 oracle after a red run; a first float64 version of this test compared largest errors per
    tensor and failed on luck: 7e-8 against 1.5e-8 on one element of var_layers.3.weight whose gradient sits near Adam's eps,
    while every gradient tensor of the device is 0.2-1.3x as far from float64 as torch's, tools/diag_grad_err.py.)"""
    import pinn_amd
    from pinn_amd import synth
    K = 8.0
    N, H, epochs, seed = 600, 128, 2, 77
    ds = synth.make_dataset(N, (), seed=9)
    torch.manual_seed(3)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, H, H, H, 1], ds[4], ds[5], p=0.2, logvar=True, seed=seed)
    m.verbose = False
    names = O.param_names(3)
    sd = m.dnn.state_dict()
    P = [sd[n].detach().cpu().clone() for n in names]
    m.use_graph, m.graph_min_steps = True, 0      # the full-batch case runs through the replayed graph
    m.train_dnn(epochs, batch_size=batch_size)
    bounds = [(0, N)] if batch_size is None else [(s, min(N, s + batch_size)) for s in range(0, N, batch_size)]
    mask_fn = lambda step, s, e: (O.philox_masks_for_net(seed, step, s, e - s, H, 3, [0.2] * 4), [0.2] * 4)
    w64, w32, spread = _adam_noise_bands(P, ds[0], ds[1], bounds, epochs, mask_fn, 0.01)
    got = m.dnn.state_dict()
    n_steps = epochs * len(bounds)
    for n, p0, r64, r32, sp in zip(names, P, w64, w32, spread):
        dev = got[n].cpu().double().numpy()
        big = np.maximum(np.maximum(np.abs(p0.numpy()), np.abs(r64)), 0.01).astype(np.float32)
        band = K * sp + 4 * n_steps * np.spacing(big).astype(np.float64)
        worst = lambda v: float((np.abs(v - r64) / band).max())
        assert worst(r32) <= 1.0, ("the fp32 oracle leaves its own band", n, worst(r32))
        assert worst(dev) <= 1.0, (n, worst(dev), worst(r32))
        assert np.abs(dev - r32).max() <= 2e-3, (n, float(np.abs(dev - r32).max()))      # sanity: the steps happened (lr = 0.01)


@pytest.mark.parametrize("precision,H", [("f32x6", 256), ("f32x6", 128), ("fp32", 128), ("bf16", 256), ("f32x6g6", 128)])
def test_train_dnn_graph_replay_is_bit_identical(precision, H):
    """SURVEY 8(f) F1 / 01:939-955: a full-batch train_dnn call replays ONE captured step (hipGraph; step count, Adam
    coefficients and dropout stream on the device) -- and must leave exactly the weights, Adam moments, loss and
    dropout-stream position that the launch-by-launch path leaves, with on-chip Philox masks and with injected ones
    (the golden three-step test and the float64-referee trajectory test above run through the replay path as well)."""
    import hip_helpers as hh
    import pinn_amd
    from pinn_amd import synth
    N, steps = 700, 9
    ds = synth.make_dataset(N, (), seed=21)

    def run(use_graph, bits):
        torch.manual_seed(5)
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, H, H, H, 1], ds[4], ds[5], p=0.2, logvar=True, seed=31, precision=precision)
        m.verbose = False
        m.use_graph = use_graph
        m.graph_chunk = 2                  # long calls replay several steps per graph launch: exercised here at 9 steps
        m.graph_min_steps = 0
        if bits is not None:
            m.dnn.inject_masks(bits)
        m.train_dnn(steps)
        a = (m.dnn.flat_params().clone(), m._adam_m.clone(), m._adam_v.clone(), m.last_loss, m._step_counter, m.dnn._mask_pass)
        m.train_dnn(3)                     # a second call continues the stream where the first one left it
        return a + (m.dnn.flat_params().clone(),)
    g = torch.Generator().manual_seed(9)
    per_pass = [[(torch.rand(N, w, generator=g) >= 0.2).numpy() for w in (H, H, H, H // 2)] for _ in range(steps + 3)]
    for bits in (None, hh.pack_mask_bits(per_pass)):
        eager, graph = run(False, bits), run(True, bits)
        for k, (a, b) in enumerate(zip(eager, graph)):
            assert (torch.equal(a, b) if torch.is_tensor(a) else a == b), ("element %d differs" % k, precision, bits is not None)
        assert torch.isfinite(eager[0]).all() and eager[4] == steps


@pytest.mark.parametrize("precision,H", [("f32x6", 256), ("fp32", 128), ("bf16", 128), ("f32x6", 512)])
def test_train_step_with_adam_in_the_reduction_is_bit_identical(precision, H):
    """pinn_mlp_train_step (gradients + Adam as one launch sequence, the default of a single-process train_dnn) against
    pinn_mlp_train_grads followed by pinn_adam_step: same weights, moments and loss, bit for bit, full batch and minibatches."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(500, (), seed=23)

    def run(fuse, batch_size):
        torch.manual_seed(7)
        m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, H, H, 1], ds[4], ds[5], p=0.2, logvar=True, seed=5, precision=precision)
        m.verbose = False
        m.use_graph = False
        m.fuse_adam = fuse
        m.train_dnn(4, batch_size=batch_size)
        return m.dnn.flat_params().clone(), m._adam_m.clone(), m._adam_v.clone(), m.last_loss
    for bs in (None, 200):
        a, b = run(True, bs), run(False, bs)
        for k, (u, v) in enumerate(zip(a, b)):
            assert (torch.equal(u, v) if torch.is_tensor(u) else u == v), ("element %d differs" % k, precision, bs)
        assert torch.isfinite(a[0]).all()


def test_reference_main_flow(tmp_path):
    """The reference's `__main__` (01:2055-2201) end to end on synthetic recordings: ingest -> seven trainer calls -> results
    array -> .mat as scripts 02-05 read it (02:105-114), at 0.2 % of the schedule."""
    import importlib.util
    import scipy.io
    spec = importlib.util.spec_from_file_location("reference_main", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples",
                                                                                  "reference_main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = os.path.join(tmp_path, "F01_output.mat")
    arr, _ = mod.main(["--scale", "0.002", "--out", out, "--quiet"])
    assert arr.shape == (3000 + 3 * 400, 22) and np.all(np.isfinite(arr))
    back = scipy.io.loadmat(out)["comprehensive_results"]
    assert np.array_equal(back, arr)
    assert set(np.unique(arr[:, 17])) == {0.0, 1.0, 2.0, 3.0}
    assert np.all(arr[:3000, 17] == 0) and np.all(arr[3000:3400, 17] == 1)
    np.testing.assert_allclose(arr[:, 12], arr[:, 8] - arr[:, 9], rtol=0, atol=1e-12)
