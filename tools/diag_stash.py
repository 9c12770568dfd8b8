"""Decode the packed training stash of PINN_PREC_F32X6 and compare it, layer by layer, with the fp32 stash of the exact
kernels on the same call: python tools/diag_stash.py H nh N.  (Workspace offsets as plan_workspace lays them out.)"""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pinn_oracle as O
import hip_helpers as hh
from pinn_amd import _lib, synth

H, nh, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
DS, PS, POISON = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (5, H + nh, 0)
lib = _lib.load()
pl = [0.2] * (nh + 1)
ds = synth.make_dataset(N, (), seed=DS)
x, y = ds[0].contiguous(), ds[1].reshape(-1).contiguous()
P = O.init_params([8] + [H] * nh + [1], seed=PS)
fp = hh.flat_params(P, H, nh).to(hh.dev())
drop = hh.dropout_struct(1, pl, seed=99, stream_id=7, row_offset=0)


def run(prec):
    net = hh.make_net(lib, H, nh, prec)
    wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), N)
    work = torch.full((wb,), 0xFF if POISON else 0, dtype=torch.uint8, device=hh.dev())
    grads = torch.zeros(fp.numel(), device=hh.dev()); loss = torch.zeros(4, dtype=torch.float64, device=hh.dev())
    _lib.check(lib.pinn_mlp_train_grads(ctypes.byref(net), hh.ptr(fp), hh.ptr(x.to(hh.dev())), hh.ptr(y.to(hh.dev())), N, N, ctypes.byref(drop),
                                        hh.ptr(grads), hh.ptr(loss), hh.ptr(work), wb, hh.stream()), "train")
    torch.cuda.synchronize()
    return work.cpu().numpy(), grads.cpu().numpy()


t16 = (N + 127) // 128 * 8
assert t16 == (N + 63) // 64 * 4, "pick N so that both kernels pad to the same tile count"
al = lambda b: (b + 255) // 256 * 256
sizes = [nh * t16 * H * 64, t16 * (H // 2) * 64, t16 * (H // 4) * 64] * 2
offs = np.cumsum([0] + [al(s) for s in sizes])
w0, g0 = run(0)
w2, g2 = run(2)
f32 = lambda w, o, n: w[o:o + n].view(np.float32)


def tiled_fp32(w, off, F, layers):
    a = f32(w, off, layers * t16 * F * 64).reshape(layers, t16, F, 16)
    return a.transpose(0, 1, 3, 2).reshape(layers, t16 * 16, F)       # [layer][row][feature]


def packed(w, off, F, layers):
    a = w[off:off + layers * t16 * F * 64].view(np.float16).reshape(layers, t16, F // 32, 2, 16, 4, 8).astype(np.float64)
    v = a[:, :, :, 0] + a[:, :, :, 1]                                   # hi + lo: [layer][t16][g][n][kq][jj]
    out = np.zeros((layers, t16, 16, F))
    for kq in range(4):
        for jj in range(8):
            r, b = jj >> 1, jj & 1
            out[:, :, :, np.arange(F // 32) * 32 + 16 * b + 4 * kq + r] = v[:, :, :, :, kq, jj].transpose(0, 1, 3, 2)
    return out.reshape(layers, t16 * 16, F), a


for name, idx, F, layers in (("h", 0, H, nh), ("v1", 1, H // 2, 1)):
    ref = tiled_fp32(w0, offs[idx], F, layers)
    got, raw = packed(w2, offs[idx], F, layers)
    for l in range(layers):
        d = np.abs(got[l][:N] / 8.0 - ref[l][:N])
        print("stash %s layer %d: max |packed/8 - fp32| = %.3e (max |ref| %.3f), mismatching zeros %d" %
              (name, l, d.max(), np.abs(ref[l][:N]).max(), int(((got[l][:N] == 0) != (ref[l][:N] == 0)).sum())))
# d pre-activations: packed ones are in the rows' normalised units -> rescale by the ratio to the fp32 ones per row
for name, idx, F, layers in (("dpre_h", 3, H, nh), ("dpre_v1", 4, H // 2, 1), ("dpre_v2", 5, H // 4, 1)):
    ref = tiled_fp32(w0, offs[idx], F, layers)
    for l in range(layers):
        got = packed(w2, offs[idx], F, layers)[0][l][:N]
        r = ref[l][:N].astype(np.float64)
        # per-row power-of-two scale: recover from the largest element
        k = np.argmax(np.abs(r), axis=1)
        ratio = got[np.arange(N), k] / r[np.arange(N), k]
        scale = 2.0 ** np.round(np.log2(np.abs(ratio)))
        d = np.abs(got / scale[:, None] - r)
        rel = d.max(axis=1) / np.abs(r).max(axis=1)
        print("%s layer %d: max over rows of (max err / row max) = %.3e, median %.3e; rows with ratio not a power of two: %d" %
              (name, l, rel.max(), np.median(rel), int((np.abs(ratio / scale - 1) > 1e-3).sum())))
offs_p, total = __import__("pinn_amd").layout.param_offsets(8, H, nh)
for (n, shape, off) in offs_p:
    k = int(np.prod(shape)); a, b = g2[off:off + k], g0[off:off + k]
    print("%-24s max |f32x6 - fp32| / max = %.2e" % (n, np.abs(a - b).max() / (np.abs(b).max() + 1e-30)))
# ---- rows whose hidden d pre-activations are off: their du, dz and a few elements
sizes2 = sizes + [t16 * (nh * (H // 32) + H // 64) * 64, t16 * 64, t16 * 64]
offs2 = np.cumsum([0] + [al(s) for s in sizes2])
du = f32(w2, offs2[7], t16 * 64)[:N]; dz = f32(w2, offs2[8], t16 * 64)[:N]
refL = tiled_fp32(w0, offs[3], H, nh)
gotL = packed(w2, offs[3], H, nh)[0]
for l in range(nh - 1, -1, -1):      # (layer 0 is packed like the others since round 3)
    r = refL[l][:N].astype(np.float64); g = gotL[l][:N]
    k = np.argmax(np.abs(r), axis=1); ratio = g[np.arange(N), k] / r[np.arange(N), k]
    scale = 2.0 ** np.round(np.log2(np.abs(ratio) + 1e-300))
    rel = np.abs(g / scale[:, None] - r).max(axis=1) / np.abs(r).max(axis=1)
    bad = np.where(rel > 1e-3)[0]
    print("layer %d bad rows:" % l, bad.tolist()[:8])
    for b in bad[:3]:
        j = np.argsort(np.abs(g[b] / scale[b] - r[b]))[::-1][:4]
        print("   row %d: du %.4e dz %.4e  scale 2^%d  gmax %.4e (global max |du|,|dz| = %.4e)" % (b, du[b], dz[b], int(np.log2(scale[b])), max(abs(du[b]), abs(dz[b])), max(np.abs(du).max(), np.abs(dz).max())))
        for jj in j:
            print("      feature %3d: packed %.6e (/scale %.6e)  ref %.6e   h %.5f" % (jj, g[b, jj], g[b, jj] / scale[b], r[b, jj], tiled_fp32(w0, offs[0], H, nh)[l][b, jj]))
