"""pinn_results_assemble and pinn_residuals (HBM-bound row passes): time per call and GB/s.  python tools/time_results.py [N ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _common as hh
from _common import _lib, lib
for N in (int(a) for a in (sys.argv[1:] or ["1000000", "10000000"])):
    dev = hh.dev()
    x = torch.rand(N, 8, device=dev) * 2 - 1
    y = torch.rand(N, device=dev)
    pm, au, eu = torch.rand(N, device=dev), torch.rand(N, device=dev), torch.rand(N, device=dev)
    cols = torch.rand(_lib.NCOLS, N, device=dev)
    labels = torch.zeros(N, device=dev)
    out = torch.empty(N, 22, dtype=torch.float64, device=dev)
    seg = torch.tensor([N // 2, N // 2 + 150, N], dtype=torch.int64, device=dev)
    aff = _lib.Affine()
    for c in range(8):
        aff.x_scale[c] = 0.01
    aff.y_scale = 0.5
    lam = torch.tensor([0.167897923477715, 2.36682075851268e-06, 2.43414469188443, 1.0] + [10.0] * 5 + [5.0, -1.559, 197.715, 1.2, 2.0, 0.5, 200.0, 1.0],
                       device=dev)
    sums = torch.zeros(_lib.NSUMS, dtype=torch.float64, device=dev)
    work = torch.empty(lib.pinn_residuals_workspace_bytes(), dtype=torch.uint8, device=dev)

    def assemble():
        _lib.check(lib.pinn_results_assemble(hh.ptr(x), hh.ptr(y), ctypes.byref(aff), 0.1, 0.5, 200, hh.ptr(seg), 3, hh.ptr(pm), hh.ptr(au), hh.ptr(eu),
                                             hh.ptr(cols), N, hh.ptr(labels), N, hh.ptr(out), hh.stream()), "assemble")

    def resid_cols():
        _lib.check(lib.pinn_residuals(hh.ptr(x), hh.ptr(y), hh.ptr(y), ctypes.byref(aff), hh.ptr(lam), _lib.RES_ALL, N, hh.ptr(cols), N, None,
                                      hh.ptr(work), work.numel(), hh.stream()), "residuals")

    def resid_sums():
        _lib.check(lib.pinn_residuals(hh.ptr(x), hh.ptr(y), hh.ptr(y), ctypes.byref(aff), hh.ptr(lam), _lib.RES_ALL, N, None, 0, hh.ptr(sums),
                                      hh.ptr(work), work.numel(), hh.stream()), "residuals")

    cache = torch.empty(6 * N, dtype=torch.float32, device=dev)

    def stage(flags, cached):
        def run():
            if cached:
                _lib.check(lib.pinn_residuals_cached(hh.ptr(cache), ctypes.byref(aff), hh.ptr(lam), flags, N, hh.ptr(sums), hh.ptr(work),
                                                     work.numel(), hh.stream()), "cached")
            else:
                _lib.check(lib.pinn_residuals(hh.ptr(x), hh.ptr(y), hh.ptr(y), ctypes.byref(aff), hh.ptr(lam), flags, N, None, 0, hh.ptr(sums),
                                              hh.ptr(work), work.numel(), hh.stream()), "residuals")
        return run

    cases = [("results_assemble", assemble, 84 + 176), ("residuals (all, 20 columns out)", resid_cols, 40 + 80),
             ("residuals (all, sums only)", resid_sums, 40)]
    for tag, flags, nc in (("V", _lib.RES_V, 6), ("T", _lib.RES_T, 4), ("H", _lib.RES_H, 2), ("O", _lib.RES_O, 2)):
        cases.append(("stage %s sums, from the rows" % tag, stage(flags, False), 40))
        _lib.check(lib.pinn_residuals_prepare(hh.ptr(x), hh.ptr(y), hh.ptr(y), ctypes.byref(aff), hh.ptr(lam), flags, N, hh.ptr(cache), hh.stream()), "prep")
        cases.append(("stage %s sums, from the row cache" % tag, stage(flags, True), 4 * nc))
    for name, fn, bytes_row in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("N=%d %-34s %8.1f us  %6.0f GB/s (%d B/row algorithmic)" % (N, name, ms * 1e3, bytes_row * N / ms / 1e6, bytes_row), flush=True)
