"""Scratch first-light check on the GPU (not a pytest file)."""
import ctypes, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pinn_amd
from pinn_amd import _lib, layout, synth
import pinn_oracle as O

lib = ctypes.CDLL(_lib.lib_path())
for name, (res, args) in _lib._SIGS.items():
    if hasattr(lib, name):
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
dev = torch.device("cuda:0")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def flat(params, H, nh):
    offs, total = layout.param_offsets(8, H, nh)
    f = torch.zeros(total, dtype=torch.float32)
    for (name, shape, off), p in zip(offs, params):
        f[off:off + p.numel()] = p.reshape(-1)
    return f


for H in (128, 256):
    nh = 3
    P = O.init_params([8, H, H, H, 1], seed=H)
    N = 1000
    ds = synth.make_dataset(N, (), seed=1)
    x = ds[0]
    fp = flat(P, H, nh).to(dev); xd = x.to(dev).contiguous()
    u = torch.empty(N, device=dev); lv = torch.empty(N, device=dev)
    net = _lib.Net(8, H, nh)
    assert lib.pinn_param_count(ctypes.byref(net)) == fp.numel()
    rc = lib.pinn_mlp_forward(ctypes.byref(net), ptr(fp), ptr(xd), N, None, ptr(u), ptr(lv), st())
    torch.cuda.synchronize()
    with torch.no_grad():
        uo, lvo = O.mlp_forward(P, x)
    print("H", H, "rc", rc, "eval max|du|", (u.cpu() - uo.squeeze()).abs().max().item(), "max|dlv|", (lv.cpu() - lvo.squeeze()).abs().max().item(), flush=True)
    # philox stochastic
    d = _lib.Dropout(); d.mode = 1; d.seed = 1234567890123; d.stream = 7; d.row_offset = 5000
    for l in range(nh + 1): d.p[l] = 0.4 if l % 2 else 0.2
    rc = lib.pinn_mlp_forward(ctypes.byref(net), ptr(fp), ptr(xd), N, ctypes.byref(d), ptr(u), ptr(lv), st())
    torch.cuda.synchronize()
    pl = [d.p[l] for l in range(nh + 1)]
    masks = O.philox_masks_for_net(d.seed, 7, 5000, N, H, nh, pl)
    with torch.no_grad():
        uo, lvo = O.mlp_forward(P, x, pl, masks)
    print("  philox rc", rc, "max|du|", (u.cpu() - uo.squeeze()).abs().max().item(), "max|dlv|", (lv.cpu() - lvo.squeeze()).abs().max().item(), flush=True)
    # bits
    bits = np.concatenate([np.packbits(m.astype(np.uint8), axis=-1, bitorder="little") for m in masks], axis=-1)
    bt = torch.from_numpy(np.ascontiguousarray(bits).view(np.int32).copy()).to(dev)
    d2 = _lib.Dropout(); d2.mode = 2; d2.d_bits = bt.data_ptr()
    for l in range(nh + 1): d2.p[l] = pl[l]
    rc = lib.pinn_mlp_forward(ctypes.byref(net), ptr(fp), ptr(xd), N, ctypes.byref(d2), ptr(u), ptr(lv), st())
    torch.cuda.synchronize()
    print("  bits rc", rc, "max|du|", (u.cpu() - uo.squeeze()).abs().max().item(), flush=True)
    # MC
    T = 8
    d.stream = 100
    for l in range(nh + 1): d.p[l] = 0.4
    pm = torch.empty(N, device=dev); au = torch.empty(N, device=dev); eu = torch.empty(N, device=dev)
    rc = lib.pinn_mc_dropout(ctypes.byref(net), ptr(fp), ptr(xd), N, ctypes.byref(d), T, ptr(pm), ptr(au), ptr(eu), st())
    torch.cuda.synchronize()
    mf = lambda t: O.philox_masks_for_net(d.seed, 100 + t, 5000, N, H, nh, [0.4] * (nh + 1))
    pmo, auo, euo = O.mc_dropout(P, x, 0.4, T, mf)
    print("  mc rc", rc, "pm", np.abs(pm.cpu().numpy() - pmo).max(), "au rel", np.abs(au.cpu().numpy() / auo - 1).max(), "eu", np.abs(eu.cpu().numpy() - euo).max(), "eu mean", euo.mean(), flush=True)
    # timing
    N2 = 1 << 20
    x2 = torch.randn(N2, 8, device=dev).clamp(-1, 1).contiguous()
    u2 = torch.empty(N2, device=dev); lv2 = torch.empty(N2, device=dev)
    for mode in (0, 1):
        dd = _lib.Dropout(); dd.mode = mode; dd.seed = 1
        for l in range(nh + 1): dd.p[l] = 0.2
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            lib.pinn_mlp_forward(ctypes.byref(net), ptr(fp), ptr(x2), N2, ctypes.byref(dd), ptr(u2), ptr(lv2), st())
            torch.cuda.synchronize(); dt = time.time() - t0
        macs = 8 * H + (nh - 1) * H * H + H + H * H // 2 + H * H // 8 + H // 4
        print("  fwd mode", mode, "N=%d: %.3f ms  %.1f TFLOP/s" % (N2, dt * 1e3, 2 * macs * N2 / dt / 1e12), flush=True)

# residuals
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g_resid.npz")))
aff = _lib.Affine()
for c in range(8):
    aff.x_min[c] = g["sx.min_"][c]; aff.x_scale[c] = g["sx.scale_"][c]
aff.y_min = g["sy.min_"][0]; aff.y_scale = g["sy.scale_"][0]


class S: pass


sy = S(); sy.feature_range = (-1, 1); sy.data_min_ = g["sy.data_min_"]; sy.data_max_ = g["sy.data_max_"]
sc, mn = O.target_affine(sy); aff.vn_scale = float(sc); aff.vn_min = float(mn)
N = g["x"].shape[0]
xd = torch.from_numpy(g["x"]).to(dev); ud = torch.from_numpy(g["u_eval"]).reshape(-1).to(dev); yd = torch.from_numpy(g["y"]).reshape(-1).to(dev)
for si in (0, 1):
    lam = torch.tensor(g["s%d.lambdas" % si], dtype=torch.float32).to(dev)
    cols = torch.zeros(20, N, device=dev); sums = torch.zeros(32, dtype=torch.float64, device=dev)
    wb = lib.pinn_residuals_workspace_bytes(); work = torch.empty(wb, dtype=torch.uint8, device=dev)
    rc = lib.pinn_residuals(ptr(xd), ptr(ud), ptr(yd), ctypes.byref(aff), ptr(lam), 15, N, ptr(cols), N, ptr(sums), ptr(work), wb, st())
    torch.cuda.synchronize()
    c = cols.cpu().numpy(); s = sums.cpu().numpy()
    for tag, j, ci in (("V", 0, 0), ("V", 1, 1), ("V", 3, 3), ("V", 4, 4), ("V", 5, 5), ("V", 8, 7), ("T", 0, 8), ("T", 1, 9), ("H", 0, 11), ("H", 1, 12), ("H", 2, 13), ("O", 0, 15), ("O", 1, 16), ("O", 2, 17), ("O", 3, 18), ("O", 4, 19)):
        want = g["s%d.%s.%d" % (si, tag, j)].reshape(-1)
        err = np.abs(c[ci] - want).max() / max(1e-30, np.abs(want).max())
        print("  set", si, tag, j, "rel-to-max err %.2e" % err, "exact" if np.array_equal(c[ci], want) else "")
    gV = g["s%d.V.grad" % si]; print("  gradV", 2 * s[1:4] / N, gV[0:3])
    gT = g["s%d.T.grad" % si]; print("  gradT", 2 * s[10:13] / N, gT[[4, 6, 8]])
    gH = g["s%d.H.grad" % si]; print("  gradH", 2 * s[15:18] / N, gH[9:12])
    gO = g["s%d.O.grad" % si]; print("  gradO", 2 * s[21:24] / N, gO[13:16])
    print("  losses", s[0] / N, g["s%d.V.loss" % si], s[9] / N, g["s%d.T.loss" % si], s[14] / N, g["s%d.H.loss" % si], s[20] / N, g["s%d.O.loss" % si], s[4] / N, g["s%d.Vn.loss" % si])
