"""GPU parity: fused training step (forward + NLL + backward + wgrad) through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pinn_oracle as O
from conftest import load_golden, params_from_golden, unpack_mask


@pytest.fixture(scope="module")
def lib():
    from pinn_amd import _lib
    return _lib.load()


def _check_grads(got_flat, want_list, H, nh, rtol, atol_scale=1e-6):
    import hip_helpers as hh
    got = hh.unflat(got_flat.cpu(), H, nh)
    for n, g, w in zip(O.param_names(nh), got, want_list):
        w = torch.as_tensor(w)
        scale = float(w.abs().max()) + 1e-30
        err = float((g - w).abs().max())
        assert err <= rtol * scale + atol_scale * scale, (n, err, scale)


def test_golden_loss_and_grads_recorded_masks(lib):
    """G4: the reference's own loss + 14 gradient tensors, dropout masks replayed bit for bit."""
    import hip_helpers as hh
    g = load_golden("g_net128.npz")
    P = params_from_golden(g)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"]).reshape(-1)
    masks = [unpack_mask(g["mask%d_p0.2_t0" % l], 128 if l < 3 else 64) for l in range(4)]
    bits = hh.pack_mask_bits([masks]).to(hh.dev())
    drop = hh.dropout_struct(2, [0.2] * 4, bits=bits)
    fp, xd, yd = hh.flat_params(P, 128, 3).to(hh.dev()), x.to(hh.dev()), y.to(hh.dev())
    grads, loss = hh.train_grads(lib, 128, 3, fp, xd, yd, drop)
    N = x.shape[0]
    l = loss.cpu().numpy()
    total = l[0] / N + 0.01 * l[1] / N
    assert abs(total - float(g["loss_p0.2_t0"])) <= 1e-5 * abs(float(g["loss_p0.2_t0"]))
    _check_grads(grads, [g["grad." + n] for n in O.param_names(3)], 128, 3, rtol=1e-4)


@pytest.mark.parametrize("H,nh,N,mode", [(256, 3, 1000, 1), (128, 3, 333, 1), (256, 2, 129, 0), (128, 4, 4096, 1), (256, 1, 64, 1)])
def test_grads_vs_oracle_autograd(lib, H, nh, N, mode):
    import hip_helpers as hh
    from pinn_amd import synth
    P = O.init_params([8] + [H] * nh + [1], seed=H + nh)
    ds = synth.make_dataset(N, (), seed=5)
    x, y = ds[0], ds[1].reshape(-1)
    pl = [0.2] * (nh + 1)
    seed, stream, row0 = 987654321987, 42, 12345
    drop = hh.dropout_struct(mode, pl, seed=seed, stream_id=stream, row_offset=row0)
    fp, xd, yd = hh.flat_params(P, H, nh).to(hh.dev()), x.to(hh.dev()).contiguous(), y.to(hh.dev()).contiguous()
    grads, loss = hh.train_grads(lib, H, nh, fp, xd, yd, drop)
    masks = O.philox_masks_for_net(seed, stream, row0, N, H, nh, pl) if mode == 1 else None
    lo, mse, go, _, _ = O.nll_loss_and_grads(P, x, ds[1], pl, masks)
    l = loss.cpu().numpy()
    assert abs((l[0] + 0.01 * l[1]) / N - lo.item()) <= 2e-5 * abs(lo.item())
    assert abs(l[2] / N - mse.item()) <= 2e-5 * abs(mse.item())
    _check_grads(grads, go, H, nh, rtol=2e-4)


def test_grads_deterministic_and_shard_additive(lib):
    """Two identical calls are bitwise equal; two row shards with n_global = N sum to the full-batch gradient."""
    import hip_helpers as hh
    from pinn_amd import synth
    H, nh, N = 256, 3, 1536
    P = O.init_params([8, H, H, H, 1], seed=3)
    ds = synth.make_dataset(N, (), seed=9)
    fp = hh.flat_params(P, H, nh).to(hh.dev())
    x, y = ds[0].to(hh.dev()).contiguous(), ds[1].reshape(-1).to(hh.dev()).contiguous()
    mk = lambda off: hh.dropout_struct(1, [0.2] * 4, seed=77, stream_id=5, row_offset=off)
    g1, l1 = hh.train_grads(lib, H, nh, fp, x, y, mk(0))
    g2, l2 = hh.train_grads(lib, H, nh, fp, x, y, mk(0))
    assert torch.equal(g1, g2) and torch.equal(l1, l2)
    cut = 640
    ga, la = hh.train_grads(lib, H, nh, fp, x[:cut].contiguous(), y[:cut].contiguous(), mk(0), n_global=N)
    gb, lb = hh.train_grads(lib, H, nh, fp, x[cut:].contiguous(), y[cut:].contiguous(), mk(cut), n_global=N)
    scale = g1.abs().max().item()
    assert (ga + gb - g1).abs().max().item() <= 2e-5 * scale
    np.testing.assert_allclose((la + lb).cpu().numpy()[:3], l1.cpu().numpy()[:3], rtol=1e-6)
