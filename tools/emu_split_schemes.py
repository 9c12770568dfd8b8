"""CPU emulation (numpy only) of the split-operand schemes for the two gradient products, against float64 -- the check that
preceded the fp16 gradient kernels ("X3 for the gradients"):
  x6  : 3 bf16 parts, 6 products                                   (PINN_PREC_F32X6_G6)
  x3  : 2 fp16 parts (hi, lo), 3 products, per-row normalisation   (backward_pass<X3>)
  x3s : fp16 hi + SCALED lo' = f16((x - hi) * 2048), third product against the other operand's hi * 2^-11, one power-of-two
        scale per call                                             (wgrad kernels, kF16S)
on synthetic operands with a wide spread over rows and features: h = tanh * dropout, d = gradients.
usage: python tools/emu_split_schemes.py"""
import numpy as np

def bf16(x):
    x = np.asarray(x, np.float32); u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)
def f16(x):
    with np.errstate(over='raise'):
        return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
def split3(x):
    h = bf16(x); m = bf16(x - h); l = bf16(x - h - m); return h, m, l
def mm(a, b):   # fp32 accumulate
    return a.astype(np.float32) @ b.astype(np.float32)

def main():
    N, H = 16384, 256
    rng = np.random.default_rng(0)
    # realistic operands: h = tanh * dropout; d with per-row and per-feature spread
    pre = rng.normal(0, 1.2, (N, H))
    keep = rng.random((N, H)) >= 0.2
    h = (np.tanh(pre) * keep / 0.8)
    rowmag = np.exp(rng.normal(0, 2.5, (N, 1))) * 1e-6          # precision * residual / N spread over rows
    featmag = np.exp(rng.normal(0, 2.0, (1, H)))                 # some features 1e-3 below others
    d = rng.normal(0, 1, (N, H)) * rowmag * featmag * (1 - np.tanh(pre) ** 2) * keep / 0.8
    W = rng.normal(0, 1 / 16, (H, H))
    h32, d32, W32 = h.astype(np.float32), d.astype(np.float32), W.astype(np.float32)

    # ---------------- weight gradient dW = d^T h  (K = rows)
    ref = d32.astype(np.float64).T @ h32.astype(np.float64)
    def report(name, got):
        e = np.abs(got - ref)
        rel_max = e.max() / np.abs(ref).max()
        relel = e / np.maximum(np.abs(ref), 1e-300)
        rowmax = np.abs(ref).max(1)
        small = rowmax < np.quantile(rowmax, 0.1)
        print("  %-28s err/max %.2e   median rel/elem %.2e   p99 %.2e   median rel on the 10%% smallest-gradient features %.2e" %
              (name, rel_max, np.median(relel), np.quantile(relel, 0.99), np.median(relel[small])))
    print("wgrad dW = d^T h, K = %d rows" % N)
    report("fp32 matmul", mm(d32.T, h32))
    dh, dm, dl = split3(d32); hh, hm, hl = split3(h32)
    report("x6", mm(dh.T, hh) + (mm(dh.T, hm) + mm(dm.T, hh)) + (mm(dh.T, hl) + mm(dl.T, hh) + mm(dm.T, hm)))
    report("2 bf16 parts (G3)", mm(dh.T, hh) + (mm(dh.T, hm) + mm(dm.T, hh)))
    amax = np.abs(d32).max(); G = np.float32(2.0 ** (14 - np.floor(np.log2(amax))))   # max*G in [2^14, 2^15)
    dg = d32 * G
    Dh = f16(dg); Dl = f16((dg - Dh) * 2048)
    Hh = f16(8 * h32); Hl = f16(8 * h32 - Hh); Hs = f16(Hh / 2048)
    acc = mm(Dh.T, Hh) + mm(Dh.T, Hl) + mm(Dl.T, Hs)
    report("x3s global scale", acc / (8 * G))
    Dl0 = f16(dg - Dh)
    report("x3 plain lo, global scale", (mm(Dh.T, Hh) + mm(Dh.T, Hl) + mm(Dl0.T, Hh)) / (8 * G))
    # per-feature scale (needs a prior pass too)
    fm = np.abs(d32).max(0, keepdims=True); Gf = (2.0 ** (14 - np.floor(np.log2(fm)))).astype(np.float32)
    dgf = d32 * Gf; Dhf = f16(dgf); Dlf = f16((dgf - Dhf) * 2048)
    report("x3s per-feature scale", (mm(Dhf.T, Hh) + mm(Dhf.T, Hl) + mm(Dlf.T, Hs)) / (8 * Gf.T))

    # ---------------- backward product  out[r, o] = sum_k d[r, k] W[k, o]   (per-row normalisation allowed)
    ref = d32.astype(np.float64) @ W32.astype(np.float64)
    print("dgrad out = d W, K = %d features" % H)
    def report2(name, got):
        e = np.abs(got - ref)
        rm = np.abs(ref).max(1, keepdims=True)
        print("  %-28s max err / row max %.2e   median rel/elem %.2e   p99 rel/elem %.2e" % (name, (e / rm).max(), np.median(e / np.abs(ref)), np.quantile(e / np.abs(ref), 0.99)))
    report2("fp32 matmul", mm(d32, W32))
    Wh, Wm, Wl = split3(W32)
    report2("x6", mm(dh, Wh) + (mm(dh, Wm) + mm(dm, Wh)) + (mm(dh, Wl) + mm(dl, Wh) + mm(dm, Wm)))
    rmax = np.abs(d32).max(1, keepdims=True); s = (2.0 ** (3 - np.floor(np.log2(rmax)))).astype(np.float32)   # row max in [8, 16)
    dn = d32 * s
    Nh = f16(dn); Nl = f16(dn - Nh); Ns = f16((dn - Nh) * 2048)
    Fh = f16(64 * W32); Fl = f16(64 * W32 - Fh); Fs = f16(Fh / 2048)
    report2("x3 per-row (today's opt-in)", (mm(Nh, Fh) + mm(Nh, Fl) + mm(Nl, Fh)) / (64 * s))
    report2("x3s per-row", (mm(Nh, Fh) + mm(Nh, Fl) + mm(Ns, Fs)) / (64 * s))

main()
