#!/usr/bin/env python
"""bench.py -- PINN training samples/s (+ MC-dropout forward-passes/s) on synthetic 1e6 x 8 rows.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: reference architecture [8,256,256,256,1] (p=0.2, variance
head), ROWS_PER_GPU synthetic fuel-cell rows per GPU resident in HBM.
One timed "step" advances EVERY trainer of the reference once over those rows:
   train_dnn step   : fused forward + aleatoric NLL + backward + weight gradients
                      + RCCL all-reduce(SUM) of the flat gradient bucket (N > 1) + Adam
   physics stages   : one iteration each of train_lambda(False), train_lambda(True),
                      train_thermal, train_hydrogen, train_oxygen
                      (pass over the row cache of the stage's parameter-independent terms -> [all-reduce of
                      32 sums] -> Adam + clamp on device; the cache is built once per trainer call, outside the step)
value = rows of all ranks * steps / max-over-ranks wall time (weak scaling: rows per GPU fixed).

Arithmetic (--precision, default f32x6 = the library default): fp32 in, fp32 out, fp32 accumulation;
the matrix products run on the 16-bit matrix cores with split operands: two fp16 parts per operand
(exact power-of-two scales bring them into fp16's range: constants in the forward passes, per row in
the backward chain and -- on the other operand -- in the weight gradients), three cross products -- the
accuracy of an fp32 matmul (same parity tests and tolerances as the exact-fp32 kernels; gradient
tensors as close to a float64 autograd as torch's own fp32 autograd).  The training stash holds the
fp16 fragments themselves (4 B per element, as fp32 did): the weight-gradient kernels read ready operands.
The exact-fp32 kernels (v_mfma_f32_*_f32), the opt-in bf16-mixed ones and the f32x6g6 variant
(gradients from three bf16 parts, six products: round 2's first default) are measured beside it
unless --only.

Extra legs reported in the same JSON line (rank 0):
   mc_dropout   : get_MC_samples-equivalent launch (1 eval + T stochastic passes, on-chip reduce)
   roofline     : the chain (forward + loss kernel, backward kernel) timed with events on the launch
                  stream, together and each alone; algorithmic FLOP per row = 4*M - 32 H (forward
                  2(M - 8H) + dgrad 2(M - 8H), M = 174 400 MAC; the K = 8 input layer is exact f32).
                  Peak = dense 16-bit MFMA peak (2.5 PFLOP/s) / matrix instructions executed per
                  algorithmic multiply-add (forward 3, backward 6: chain 4.5 -> 555.6 TFLOP/s;
                  MC-dropout 3 -> 833.3); "frac_if_priced_at_6_products" keeps round 1's accounting
                  (416.7 TFLOP/s) beside it; exact fp32: 157.3 TFLOP/s.
   cpu_baseline : the CPU oracle's train_dnn step (torch CPU, autograd, torch-bernoulli masks,
                  Adam) on the host cores of this box (N = 1 only): a 1e5-row sample (median of 3)
                  AND the headline's 1e6 rows (1 warm-up on the sample + 2 timed steps), and the
                  MC-dropout unit of work both as the reference counts it (2T predict calls x 2 forwards)
                  and as useful work (1 eval + T stochastic forwards)
   configs      : BASELINE.json's other configurations, each a small leg of its own (rank 0, N = 1):
                  "1" the reference's CPU-runnable case (1e4 + 1e3 rows, train_dnn(100), mc_times = 32) on the
                  GPU and through the CPU oracle; "4_step" one rank's 65 536-row minibatch step with its
                  per-kernel breakdown; "5" the wide net [8,1024x4,1] (262 144 rows, T = 1024 on a row slice);
                  "residuals" the HBM-bound physics row pass in GB/s against the 8 TB/s roof
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, NH = 256, 3
M_MAC = 8 * H + (NH - 1) * H * H + H + H * H // 2 + H * H // 8 + H // 4        # 174 400
CHAIN_FLOP_PER_ROW = 4 * M_MAC - 2 * 2 * 8 * H                                 # fwd 2M + dgrad 2(M - 8H)
STEP_FLOP_PER_ROW = 6 * M_MAC
FWD_FLOP_PER_ROW = 2 * M_MAC
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
# Matrix instructions executed per algorithmic multiply-add ("products"), per phase -- the split-operand schemes of
# pinn_x6_core.h: X3 = two fp16 parts, 3 products; x6 = three bf16 parts, 6 products; the peak an algorithmic FLOP can be
# priced against is the dense 16-bit MFMA peak / products.
# per precision: (dtype string, chain kernel name, forward kernel name, products forward, products backward, products wgrad,
#                 peak of one executed product in TFLOP/s)
PRECISIONS = {
    "f32x6": ("f32 (split-operand matrix products on the 16-bit matrix cores, f32 accumulate, fp32-matmul accuracy: 2 fp16 parts / "
              "3 MFMAs per product in forward, backward and weight gradients)",
              "train_fwd_x3_kernel<256> + train_bwd_kernel<X3,256>", "mlp_x6_kernel<256,MC> (scheme X3)", 3, 3, 3, PEAK_BF16_MFMA_TFLOPS),
    "f32x6g6": ("f32 (as f32x6 with the gradients -- backward chain, weight gradients -- from 3 bf16 parts / 6 MFMAs per product)",
                "train_fwd_x3_kernel<256> + train_bwd_kernel<X6,256>", "mlp_x6_kernel<256,MC> (scheme X3)", 3, 6, 6, PEAK_BF16_MFMA_TFLOPS),
    "fp32": ("f32 (exact: v_mfma_f32_*_f32)", "train_chain_kernel<256>", "mlp_kernel<256,MC>", 1, 1, 1, PEAK_FP32_MFMA_TFLOPS),
    "bf16": ("bf16 MFMA inputs, f32 accumulate/activations/loss/master weights (parity rtol 2e-2)",
             "train_chain_bf16_kernel<256>", "mlp_bf16_kernel<256,MC>", 1, 1, 1, PEAK_BF16_MFMA_TFLOPS),
}
HALF_CHAIN_FLOP_PER_ROW = CHAIN_FLOP_PER_ROW // 2          # forward 2 (M - 8H) = backward 2 (M - 8H): the K = 8 input layer is exact f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000, help="rows per GPU")
    ap.add_argument("--mc-passes", type=int, default=512)
    ap.add_argument("--no-mc", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=100_000)
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the CPU baseline at the headline's full row count (~1 min)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs for BASELINE configs 1, 4, 5 and the residual pass")
    ap.add_argument("--precision", default="f32x6", choices=sorted(PRECISIONS), help="arithmetic of the headline measurement")
    ap.add_argument("--only", action="store_true", help="skip the extra legs in the other precisions")
    ap.add_argument("--no-bf16", action="store_true", help="skip the extra bf16/fp32-mixed leg")
    return ap.parse_args()


def _cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            return next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        return ""


def cpu_baseline(rows, full_rows, mc_passes):
    """CPU oracle (the reference's arithmetic, restated; test infrastructure used here only as the timed baseline):
    train_dnn step on `rows` rows (median of 3 after a warm-up) and on `full_rows` rows (2 timed steps), and the
    MC-dropout forwards at `full_rows` rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pinn_oracle as O
    from pinn_amd import synth
    P = O.init_params([8, H, H, H, 1], seed=0)
    gen = torch.Generator().manual_seed(0)

    def step_time(n, reps, warm):
        ds = synth.make_dataset(n, (), seed=0)
        x, y = ds[0], ds[1]
        opt = O.AdamState(P)
        times = []
        for it in range(warm + reps):
            t0 = time.perf_counter()
            masks = [(torch.rand(n, w, generator=gen) >= 0.2) for w in (H, H, H, H // 2)]   # bernoulli draws, as the reference pays
            _, _, grads, _, _ = O.nll_loss_and_grads(P, x, y, [0.2] * 4, masks)
            opt.step(P, grads, 0.01)
            times.append(time.perf_counter() - t0)
        t = sorted(times[warm:])
        return t[len(t) // 2], x

    t_small, _ = step_time(rows, 3, 1)
    out = {"value": rows / t_small, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "train_dnn step (fwd+NLL+autograd bwd+Adam, torch-bernoulli masks) of oracle/pinn_oracle.py on %d rows, "
                     "median of 3 after 1 warm-up; os.cpu_count()=%d; %s" % (rows, os.cpu_count(), _cpu_model_name()),
           "rows": rows}
    if full_rows and full_rows > rows:
        t_full, x = step_time(full_rows, 2, 0)            # (the sample above was the warm-up)
        out["full_size"] = {"rows": full_rows, "samples_per_s": full_rows / t_full, "seconds_per_step": t_full,
                            "sample": "the headline's row count, median of 2 steps"}
        with torch.no_grad():
            t0 = time.perf_counter()
            O.mlp_forward(P, x)
            t_eval = time.perf_counter() - t0
            t0 = time.perf_counter()
            masks = [(torch.rand(full_rows, w, generator=gen) >= 0.4) for w in (H, H, H, H // 2)]
            O.mlp_forward(P, x, [0.4] * 4, masks)
            t_st = time.perf_counter() - t0
        T = mc_passes
        # 01:1442-1464: T eval predict() + T stochastic predict(), each = net_u + a discarded net_f_V (second DNN forward)
        out["mc"] = {"rows": full_rows, "passes": T, "eval_forward_s": t_eval, "stochastic_forward_s": t_st,
                     "useful_fwd_passes_per_s": full_rows * T / (t_eval + T * t_st),
                     "reference_faithful_fwd_passes_per_s": full_rows * T / (2 * T * t_eval + 2 * T * t_st),
                     "note": "passes/s counts the T stochastic row-passes get_MC_samples returns statistics of; the reference "
                             "executes 4T forwards for them (2T predict calls x 2), useful work is T + 1; extrapolated from one "
                             "eval and one stochastic forward at %d rows" % full_rows}
        out["mc_fwd_passes_per_s"] = out["mc"]["useful_fwd_passes_per_s"]
    return out


def _events(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps       # ms


def leg_config1(with_cpu):
    """BASELINE configs[0] -- the reference's own CPU-runnable case: 1e4 normal (+ 1e3 fault) synthetic rows,
    [8,256,256,256,1], train_dnn(100) (01:2143 at 2.5 % of its 4001 epochs), get_MC_samples(mc_times=32) on all
    1.1e4 rows (01:2156): wall time of the public surface on the GPU, and of the CPU oracle doing the same work."""
    import pinn_amd
    from pinn_amd import synth
    ds = synth.make_dataset(10_000, (1000,), seed=0)
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, H, H, H, 1], ds[4], ds[5], p=0.2, logvar=True, seed=0)
    m.verbose = False
    m.train_dnn(2); pinn_amd.get_MC_samples(m, ds[2][:256], ds[4], mc_times=2, dropout=0.4); torch.cuda.synchronize()     # warm-up
    t0 = time.perf_counter(); m.train_dnn(100); torch.cuda.synchronize(); t_train = time.perf_counter() - t0
    t0 = time.perf_counter(); pinn_amd.get_MC_samples(m, ds[2], ds[4], mc_times=32, dropout=0.4); t_mc = time.perf_counter() - t0
    # the reference's real call is train_dnn(4001) / (8001): 2000 steps of the same call as a replayed hipGraph (opt-in: a replayed
    # step is a few us slower than a launched one on this runtime), next to the default launch-by-launch path
    m.use_graph = True
    t0 = time.perf_counter(); m.train_dnn(2000); torch.cuda.synchronize(); t_long = time.perf_counter() - t0
    m.use_graph = False
    t0 = time.perf_counter(); m.train_dnn(500); torch.cuda.synchronize(); t_eager = time.perf_counter() - t0
    n_tr, n_te = ds[0].shape[0], ds[2].shape[0]
    r = {"workload": "BASELINE configs[0]: %d train rows, 100 train_dnn epochs, mc_times=32 on %d rows" % (n_tr, n_te),
         "gpu": {"train_dnn_100_s": t_train, "train_samples_per_s": 100 * n_tr / t_train, "us_per_step": t_train / 100 * 1e6,
                 "us_per_step_2000_steps_graph_replay": t_long / 2000 * 1e6, "us_per_step_launch_by_launch": t_eager / 500 * 1e6,
                 "mc32_s": t_mc, "mc_fwd_passes_per_s": 32 * n_te / t_mc}}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pinn_oracle as O
        P = O.init_params([8, H, H, H, 1], seed=0)
        opt = O.AdamState(P)
        gen = torch.Generator().manual_seed(0)
        x, y = ds[0], ds[1]
        t0 = time.perf_counter()
        for _ in range(100):
            masks = [(torch.rand(n_tr, w, generator=gen) >= 0.2) for w in (H, H, H, H // 2)]
            _, _, grads, _, _ = O.nll_loss_and_grads(P, x, y, [0.2] * 4, masks)
            opt.step(P, grads, 0.01)
        c_train = time.perf_counter() - t0
        xt = ds[2]
        with torch.no_grad():
            t0 = time.perf_counter()
            O.mlp_forward(P, xt)
            for _ in range(32):
                O.mlp_forward(P, xt, [0.4] * 4, [(torch.rand(n_te, w, generator=gen) >= 0.4) for w in (H, H, H, H // 2)])
            c_useful = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(64):                      # the reference: 32 eval predict() + 32 stochastic predict(), 2 forwards each
                O.mlp_forward(P, xt)
            for _ in range(64):
                O.mlp_forward(P, xt, [0.4] * 4, [(torch.rand(n_te, w, generator=gen) >= 0.4) for w in (H, H, H, H // 2)])
            c_faithful = time.perf_counter() - t0
        r["cpu_oracle"] = {"cores": torch.get_num_threads(), "train_dnn_100_s": c_train, "train_samples_per_s": 100 * n_tr / c_train,
                           "mc32_useful_s": c_useful, "mc32_reference_faithful_s": c_faithful,
                           "mc_fwd_passes_per_s_useful": 32 * n_te / c_useful, "mc_fwd_passes_per_s_reference_faithful": 32 * n_te / c_faithful}
    return r


def leg_config4_step(dev):
    """BASELINE configs[3], one rank's share of a step: a 65 536-row minibatch (of a 524 288-row global batch over 8
    ranks) through train_dnn's step -- chain, weight gradients, slab reduction, Adam -- with the per-kernel times and
    what is left between them (launch gaps; the all-reduce of the 0.70 MB bucket is not part of a 1-GPU run)."""
    import ctypes
    import pinn_amd
    from pinn_amd import _lib, synth
    from pinn_amd.model import _ptr, _stream
    rows, n_global = 65_536, 524_288
    ds = synth.make_dataset(rows, (), seed=3)
    torch.manual_seed(0)
    m = pinn_amd.PhysicsInformedNN(ds[0], ds[1], [8, H, H, H, 1], ds[4], ds[5], p=0.2, logvar=True, seed=0, n_global=n_global)
    m.verbose = False
    m.dnn.train()
    lib = m._lib
    xd, yd, flat = m.x.detach(), m.u.reshape(-1), m.dnn.flat_params()
    work = m._workspace(rows)
    drop = m.dnn.dropout_struct(3, 0)
    loss = torch.empty(4, dtype=torch.float64, device=dev)

    def phase(ph):
        return lambda: _lib.check(lib.pinn_mlp_train_grads_phases(
            ctypes.byref(m.dnn._net), _ptr(flat), _ptr(xd), _ptr(yd), rows, n_global, ctypes.byref(drop), _ptr(m.dnn._flat_grad),
            _ptr(loss), _ptr(work), work.numel(), _stream(), ph), "phases")

    def adam():
        _lib.check(lib.pinn_adam_step(_ptr(flat), _ptr(m.dnn._flat_grad), _ptr(m._adam_m), _ptr(m._adam_v), flat.numel(), 1e-9, 1, _stream()),
                   "adam")

    def step():
        m.train_step_grads(xd, yd, 0, n_global)
        adam()
    ms = {"chain": _events(phase(1), 20), "wgrad": _events(phase(2), 20), "finalize": _events(phase(4), 20), "adam": _events(adam, 20)}
    ms_step = _events(step, 50, warm=5)
    ksum = sum(ms.values())
    return {"workload": "BASELINE configs[3], one rank: %d-row minibatch of a %d-row global batch, train_dnn step" % (rows, n_global),
            "ms_per_step": ms_step, "samples_per_s_per_gpu": rows / ms_step * 1e3, "kernels_ms": ms, "kernel_sum_ms": ksum,
            "step_minus_kernel_sum_ms": ms_step - ksum,       # (each phase timed alone carries its own weight re-pack launch: can be < 0)
            "mfma_frac_of_step": STEP_FLOP_PER_ROW * rows / (ms_step * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 6.0)}


def leg_config5(dev):
    """BASELINE configs[4] on one GPU: wide net [8,1024,1024,1024,1024,1] (M = 3 810 560 MAC/row), 262 144 rows:
    training-gradient call (chain + weight gradients + reduction) and MC-dropout with T = 1024 on a 65 536-row slice
    (512 row tiles: every CU busy; ~3 s),
    f32x6 arithmetic; MFMA fraction = executed matrix FLOPs against the dense 16-bit peak of 2.5 PFLOP/s."""
    import ctypes
    from pinn_amd import _lib, layout
    lib = _lib.load()
    Hw, nhw, rows, mc_rows, T = 1024, 4, 262_144, 65_536, 1024
    Mw = 8 * Hw + (nhw - 1) * Hw * Hw + Hw + Hw * Hw // 2 + Hw * Hw // 8 + Hw // 4
    offs, total = layout.param_offsets(8, Hw, nhw)
    g = torch.Generator().manual_seed(1)
    fp = torch.zeros(total)
    for name, shape, off in offs:
        n = 1
        for d in shape:
            n *= d
        fan_in = shape[1] if len(shape) == 2 else Hw
        fp[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) / (fan_in ** 0.5)
    fp = fp.to(dev)
    x = torch.rand(rows, 8, device=dev) * 2 - 1
    y = torch.rand(rows, device=dev)
    net = _lib.Net(8, Hw, nhw, _lib.PREC_F32X6, None)
    packed = torch.empty(lib.pinn_packed_bytes(ctypes.byref(net)), dtype=torch.uint8, device=dev)
    net.d_packed = packed.data_ptr()
    d = _lib.Dropout()
    d.mode = _lib.DROP_PHILOX
    for l in range(nhw + 1):
        d.p[l] = 0.2
    d.seed, d.stream, d.row_offset, d.d_bits = 1, 2, 0, None
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    wb = lib.pinn_train_workspace_bytes(ctypes.byref(net), rows)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    grads = torch.empty(total, device=dev)
    loss = torch.zeros(4, dtype=torch.float64, device=dev)

    def phase(ph):
        return lambda: _lib.check(lib.pinn_mlp_train_grads_phases(ctypes.byref(net), P(fp), P(x), P(y), rows, rows, ctypes.byref(d), P(grads),
                                                                  P(loss), P(work), wb, st(), ph), "wide train")
    t_all = _events(phase(7), 3, warm=1)
    t_chain, t_wg = _events(phase(1), 3, warm=1), _events(phase(2), 3, warm=1)
    u, lv = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    t_fwd = _events(lambda: _lib.check(lib.pinn_mlp_forward(ctypes.byref(net), P(fp), P(x), rows, ctypes.byref(d), P(u), P(lv), st()), "wide fwd"),
                    3, warm=1)
    out = torch.empty(3, mc_rows, device=dev)
    for l in range(nhw + 1):
        d.p[l] = 0.4
    xs = x[:mc_rows].contiguous()
    mc = lambda: _lib.check(lib.pinn_mc_dropout(ctypes.byref(net), P(fp), P(xs), mc_rows, ctypes.byref(d), T, P(out[0]), P(out[1]), P(out[2]), st()),
                            "wide mc")
    t_mc = _events(mc, 1, warm=0)
    peak = PEAK_BF16_MFMA_TFLOPS
    del work
    # executed matrix FLOPs: 3 products per MAC (scheme X3) in the forward layers, the backward layers and the weight gradients;
    # *_priced_at_6: round 1's convention (every product priced as six), kept so that the rounds compare
    alg = 2.0 * Mw
    tf = lambda flop, ms: flop / (ms * 1e-3) / 1e12
    return {"workload": "BASELINE configs[4] on one GPU: [8,1024x4,1] (M = %d MAC/row), %d rows; MC-dropout T = %d on %d rows" % (Mw, rows, T, mc_rows),
            "products_per_mac": {"forward": 3, "backward": 3, "wgrad": 3},
            "train": {"ms": t_all, "samples_per_s": rows / t_all * 1e3, "chain_ms": t_chain, "wgrad_ms": t_wg,
                      "mfma_frac": tf((3 + 3 + 3) * alg * rows, t_all) / peak, "mfma_frac_priced_at_6": tf(18 * alg * rows, t_all) / peak},
            "forward": {"ms": t_fwd, "algorithmic_TFLOPs": tf(alg * rows, t_fwd), "mfma_frac": tf(3 * alg * rows, t_fwd) / peak,
                        "mfma_frac_priced_at_6": tf(6 * alg * rows, t_fwd) / peak},
            "mc_dropout": {"seconds": t_mc * 1e-3, "fwd_passes_per_s": mc_rows * T / (t_mc * 1e-3),
                           "mfma_frac": tf(3 * alg * mc_rows * (T + 1), t_mc) / peak,
                           "mfma_frac_priced_at_6": tf(6 * alg * mc_rows * (T + 1), t_mc) / peak},
            "peak_TFLOPs": peak, "workspace_GB": wb / 1e9}


def leg_residuals(dev):
    """The HBM-bound part of the path: one iteration of a physics-parameter stage = a pass over the
    stage's row cache (pinn_residuals_cached: 24 B/row for the voltage model) and over the raw rows (pinn_residuals:
    40 B/row), at 1e6 and 1e7 rows -- algorithmic bytes / time against the 8 TB/s HBM3E peak."""
    import ctypes
    from pinn_amd import _lib
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    aff = _lib.Affine()
    for c in range(8):
        aff.x_scale[c] = 0.01
    aff.y_scale = 0.5
    lam = torch.tensor([0.167897923477715, 2.36682075851268e-06, 2.43414469188443, 1.0] + [10.0] * 5 + [5.0, -1.559, 197.715, 1.2, 2.0, 0.5, 200.0, 1.0],
                       device=dev)
    sums = torch.zeros(_lib.NSUMS, dtype=torch.float64, device=dev)
    work = torch.empty(lib.pinn_residuals_workspace_bytes(), dtype=torch.uint8, device=dev)
    res = []
    for n in (1_000_000, 10_000_000):
        x = torch.rand(n, 8, device=dev) * 2 - 1
        y = torch.rand(n, device=dev)
        cache = torch.empty(6 * n, dtype=torch.float32, device=dev)
        _lib.check(lib.pinn_residuals_prepare(P(x), P(y), P(y), ctypes.byref(aff), P(lam), _lib.RES_V, n, P(cache), st()), "prepare")
        t_c = _events(lambda: _lib.check(lib.pinn_residuals_cached(P(cache), ctypes.byref(aff), P(lam), _lib.RES_V, n, P(sums), P(work), work.numel(),
                                                                   st()), "cached"), 50, warm=3)
        t_r = _events(lambda: _lib.check(lib.pinn_residuals(P(x), P(y), P(y), ctypes.byref(aff), P(lam), _lib.RES_V, n, None, 0, P(sums), P(work),
                                                            work.numel(), st()), "residuals"), 50, warm=3)
        for name, t, b in (("residuals_cached_kernel (V stage, 24 B/row)", t_c, 24), ("residuals_kernel (V stage from the rows, 40 B/row)", t_r, 40)):
            gbps = b * n / (t * 1e-3) / 1e9
            res.append({"kernel": name, "rows": n, "us": t * 1e3, "bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s",
                        "frac": gbps / 8000.0})
        del x, y, cache
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    # one rank per GPU; PINN_DIST_BACKEND=gloo lets several ranks share one card to rehearse the N > 1 path on a 1-GPU box
    backend = os.environ.get("PINN_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pinn_amd
    from pinn_amd import _lib, synth

    rows = args.rows
    n_global = rows * world
    # synthetic rows: every rank draws its own shard (seed = rank) and all ranks share rank 0's scalers
    ds0 = synth.make_dataset(4096, (), seed=12345)
    sx, sy = ds0[4], ds0[5]
    Xp, Up = synth.synth_rows(rows, seed=1000 + rank)
    x = torch.from_numpy(sx.transform(Xp).astype("float32"))
    y = torch.from_numpy(sy.transform(Up).astype("float32"))
    torch.manual_seed(0)                               # identical initial weights on every rank
    model = pinn_amd.PhysicsInformedNN(x, y, [8, H, H, H, 1], sx, sy, p=0.2, logvar=True, seed=0,
                                       row_offset=rank * rows, n_global=n_global, precision=args.precision)
    model.verbose = False
    model.dnn.train()
    lib = model._lib
    import ctypes
    from pinn_amd.model import _ptr, _stream
    from pinn_amd import dp

    xd, yd = model.x.detach(), model.u.reshape(-1)
    flat = model.dnn.flat_params()
    aff = model._affine(sx)
    u_eval = torch.zeros(rows, device=dev)
    adam_l = torch.zeros(2 * _lib.NLAMBDA, device=dev)
    loss_l = torch.zeros(2, device=dev)
    stages = [(_lib.STAGE_LAMBDA_PM, _lib.RES_V, 1e-3), (_lib.STAGE_LAMBDA_F, _lib.RES_V, 1e-3),
              (_lib.STAGE_THERMAL, _lib.RES_T, 1.0), (_lib.STAGE_HYDROGEN, _lib.RES_H, 1e-1), (_lib.STAGE_OXYGEN, _lib.RES_O, 1e-2)]
    step_no = [0]

    def one_step():
        step_no[0] += 1
        # ---- train_dnn step (01:949-955): gradients + all-reduce (N > 1: in two parts, the tail under the head's weight-gradient
        #      kernels -- model._dp_step) + Adam
        loss = model._dp_step(xd, yd, model.row_offset, n_global, True)
        _lib.check(lib.pinn_adam_step(_ptr(flat), _ptr(model.dnn._flat_grad), _ptr(model._adam_m), _ptr(model._adam_v),
                                      flat.numel(), 0.01, step_no[0], _stream()), "adam")
        # ---- one iteration of each physics-parameter stage (01:1008-1055, 1107-1151, 1354-1391, 1204-1274)
        lam = model._lambdas()
        for stage, flags, lr in stages:
            _lib.check(lib.pinn_residuals_cached(_ptr(stage_cache[flags]), ctypes.byref(aff), _ptr(lam), flags, rows, _ptr(model._sums),
                                                 _ptr(model._res_work), model._res_work.numel(), _stream()), "residuals_cached")
            dp.allreduce_sums(model._sums, model._group)
            _lib.check(lib.pinn_lambda_step(stage, _ptr(model._sums), n_global, aff.vn_scale, lr, step_no[0], _ptr(lam), _ptr(adam_l),
                                            _ptr(loss_l), _stream()), "lambda_step")
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def time_events(fn, reps):
        fn(); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record(); torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps     # ms

    def max_over_ranks(seconds):
        if world > 1:
            t = torch.tensor([seconds], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return seconds

    loss_buf = torch.empty(4, dtype=torch.float64, device=dev)

    def measure(precision, warmup, steps):
        """Timed steps, per-phase kernel times (events on the launch stream) and the MC-dropout launch in one precision."""
        model.dnn.set_precision(precision)
        dtype, chain_kernel, fwd_kernel, pf, pb, pw, peak1 = PRECISIONS[precision]
        model.dnn.train()
        for _ in range(warmup):
            one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one_step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        r = {"precision": precision, "dtype": dtype, "elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3,
             "train_samples_per_s": n_global * steps / elapsed, "final_loss": float((loss[0] + 0.01 * loss[1]).item() / n_global)}
        work = model._workspace(rows)
        drop = model.dnn.dropout_struct(7, model.row_offset)

        def phase(ph):
            return lambda: _lib.check(lib.pinn_mlp_train_grads_phases(
                ctypes.byref(model.dnn._net), _ptr(flat), _ptr(xd), _ptr(yd), rows, n_global, ctypes.byref(drop), _ptr(model.dnn._flat_grad),
                _ptr(loss_buf), _ptr(work), work.numel(), _stream(), ph), "phases")
        reps = max(3, min(10, steps))
        ms_chain, ms_wgrad, ms_reduce = time_events(phase(1), reps), time_events(phase(2), reps), time_events(phase(4), reps)
        ms_fwd, ms_bwd = time_events(phase(8), reps), time_events(phase(16), reps)        # (not separate kernels in the fp32 / bf16 families)
        achieved = CHAIN_FLOP_PER_ROW * rows / (ms_chain * 1e-3) / 1e12
        # executed matrix FLOP of the chain per algorithmic FLOP: forward half x pf, backward half x pb
        chain_peak = peak1 / ((pf + pb) / 2.0)
        wg = FWD_FLOP_PER_ROW * rows / (ms_wgrad * 1e-3) / 1e12
        step_exec = (HALF_CHAIN_FLOP_PER_ROW * (pf + pb) + FWD_FLOP_PER_ROW * pw) * rows       # executed MFMA FLOP of a whole step
        r["roofline"] = {"kernel": chain_kernel, "bound": "mfma", "achieved": achieved, "peak": chain_peak, "unit": "TFLOP/s",
                         "frac": achieved / chain_peak, "traffic": None, "flop_per_row": CHAIN_FLOP_PER_ROW, "ms": ms_chain,
                         "products_per_mac": {"forward": pf, "backward": pb, "wgrad": pw},
                         "frac_if_priced_at_6_products": achieved / (peak1 / 6.0),
                         "note": "achieved = algorithmic FLOP / time.  peak = dense 16-bit MFMA peak / matrix instructions issued per "
                                 "algorithmic product (products_per_mac), so frac is the utilisation of the matrix unit by the instructions "
                                 "actually issued.  Round 1 issued six per product (peak 416.7; its frac 0.354 corresponds to "
                                 "frac_if_priced_at_6_products here).  The same arithmetic from three instructions raises value and lowers "
                                 "this frac: compare rounds on value or on frac_if_priced_at_6_products.  The chip is power-limited in these "
                                 "kernels (DESIGN.md 3): a pure stream of these MFMAs sustains 58-66 % of the nominal peak",
                         "wgrad": {"ms": ms_wgrad, "achieved": wg, "frac": wg / (peak1 / pw)},
                         "reduce_ms": ms_reduce,
                         "step_mfma_frac": step_exec / (r["ms_per_step"] * 1e-3) / 1e12 / peak1,
                         "hbm_algorithmic_GBps": 36.0 * rows / (r["ms_per_step"] * 1e-3) / 1e9}
        if precision.startswith("f32x6"):
            fa = HALF_CHAIN_FLOP_PER_ROW * rows / (ms_fwd * 1e-3) / 1e12
            ba = HALF_CHAIN_FLOP_PER_ROW * rows / (ms_bwd * 1e-3) / 1e12
            r["roofline"]["kernels"] = {
                chain_kernel.split(" + ")[0]: {"ms": ms_fwd, "achieved": fa, "peak": peak1 / pf, "frac": fa / (peak1 / pf)},
                chain_kernel.split(" + ")[1]: {"ms": ms_bwd, "achieved": ba, "peak": peak1 / pb, "frac": ba / (peak1 / pb)}}
        if not args.no_mc:
            T = args.mc_passes
            for m in model.dnn.dropout_modules():
                m.p = 0.4
            model.dnn.train()
            model.mc_dropout(xd[:4096], 2); barrier()
            t0 = time.perf_counter()
            pm, au, eu = model.mc_dropout(xd, T, row_offset=model.row_offset)
            barrier()
            mc_s = max_over_ranks(time.perf_counter() - t0)
            mc_tf = FWD_FLOP_PER_ROW * rows * (T + 1) / mc_s / 1e12
            r["mc_dropout"] = {"metric": "mc_dropout_fwd_passes_per_s", "value": n_global * T / mc_s, "unit": "fwd-passes/s",
                               "rows_per_gpu": rows, "passes": T, "seconds": mc_s,
                               "roofline": {"kernel": fwd_kernel, "bound": "mfma", "achieved": mc_tf, "peak": peak1 / pf, "unit": "TFLOP/s",
                                            "frac": mc_tf / (peak1 / pf), "products_per_mac": pf,
                                            "frac_if_priced_at_6_products": mc_tf / (peak1 / 6.0),
                                            "hbm_algorithmic_GBps": 44.0 * rows / mc_s / 1e9},
                               "e_u_mean": float(eu.mean().item())}
            for m in model.dnn.dropout_modules():
                m.p = 0.2
        return r

    # once per trainer call in the real schedule (weights are frozen within a physics stage): the eval forward that feeds
    # net_f_V, and the parameter-independent half of every row for each residual model (pinn_residuals_prepare)
    model.dnn.eval(); u_eval.copy_(model.dnn(xd)[0].reshape(-1)); model.dnn.train()
    stage_cache = {}
    for flags in (_lib.RES_V, _lib.RES_T, _lib.RES_H, _lib.RES_O):
        stage_cache[flags] = torch.empty(6 * rows, dtype=torch.float32, device=dev)
        _lib.check(lib.pinn_residuals_prepare(_ptr(xd), _ptr(u_eval), _ptr(yd), ctypes.byref(aff), _ptr(model._lambdas()), flags, rows,
                                              _ptr(stage_cache[flags]), _stream()), "residuals_prepare")
    head = measure(args.precision, args.warmup, args.steps)
    out = {
        "metric": "pinn_train_samples_per_s", "value": head["train_samples_per_s"], "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": head["dtype"], "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: %d synthetic rows x 8 features per GPU, PINN [8,256,256,256,1] + variance head, "
                               "step = train_dnn (fwd+NLL+bwd+wgrad+allreduce+Adam) + one iteration of each of the 5 physics stages"
                               % rows,
                   "rows_per_gpu": rows, "global_rows": n_global, "parallelism": "dp%d" % world, "precision": args.precision,
                   "final_loss": head["final_loss"]},
        "roofline": head["roofline"],
    }
    if "mc_dropout" in head:
        out["mc_dropout"] = head["mc_dropout"]
    # HBM bytes per launch of the chain kernel: NOT measured by this run -- read from the newest committed rocprofv3 PMC
    # summary (FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes at this workload size by
    # tools/collect_profiles.sh); null when the row count or the kernel differs
    import glob
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary_x6.json" if args.precision.startswith("f32x6") else "pmc_summary_v2.json")))
    pmc = cand[-1] if cand else ""
    if rows == 1_000_000 and pmc:
        try:
            # one entry per kernel of the chain ("train_fwd_x3_kernel<256> + train_bwd_kernel<X6,256>": two launches per step)
            keys = [part.strip().split("<")[0] + "<" + ("pinn::x6::" + part.split("<")[1].split(",")[0] + "," if "bwd" in part else "")
                    for part in head["roofline"]["kernel"].split("+")]
            table = json.load(open(pmc))
            ks = [[v for n, v in table.items() if key in n][0] for key in keys]
            # gfx950 correction of the guide's HBM section: FETCH_SIZE counts 16-B/lane streaming reads at half their bytes
            out["roofline"]["traffic"] = sum((2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0 for k in ks)
            out["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT) + " (a committed profile, not this run): (2 x FETCH_SIZE + WRITE_SIZE) " \
                "KB x 1024, per launch, N=1e6 (separate --pmc passes; the factor 2 is the guide's gfx950 correction for 16-B/lane reads and an " \
                "upper bound here: the layer-0 activation re-reads are dword loads); by design 3.84 KB/row of stash written, 3.84 KB/row of d " \
                "pre-activations written, the stash read back once -- see roofline_hbm_design -- not the 36 B/row the algorithm needs"
        except Exception:
            pass

    # ------------------------------------------------------------------ extra legs: the same measurement in the other precisions
    if not args.only:
        extras = [("exact_fp32", "fp32")] + ([] if args.no_bf16 else [("bf16_mixed", "bf16")]) + [("f32x6_g6", "f32x6g6")]
        for name, prec in extras:
            if prec == args.precision:
                continue
            e = measure(prec, max(1, args.warmup), args.steps)
            o = {"dtype": e["dtype"], "train_samples_per_s": e["train_samples_per_s"], "ms_per_step": e["ms_per_step"],
                 "chain_ms": e["roofline"]["ms"], "chain_frac_of_peak": e["roofline"]["frac"], "peak_TFLOPs": e["roofline"]["peak"],
                 "wgrad_ms": e["roofline"]["wgrad"]["ms"]}
            if "mc_dropout" in e:
                o["mc_fwd_passes_per_s"] = e["mc_dropout"]["value"]
                o["mc_seconds"] = e["mc_dropout"]["seconds"]
            out[name] = o
        model.dnn.set_precision(args.precision)

    if world > 1:
        # the same step with ONE blocking all-reduce of the whole bucket instead of the overlapped two-part one
        model.overlap_allreduce = False
        for _ in range(max(1, args.warmup)):
            one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        out["ms_per_step_blocking_allreduce"] = max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3
        model.overlap_allreduce = True
    out["config"]["backend"] = dist.get_backend() if world > 1 else "none (single process)"
    out["config"]["world_size"] = world
    out["config"]["rows_per_rank"] = [rows] * world
    # the HBM roof next to the MFMA one: DESIGN bytes of a step (not the 36 B/row the algorithm needs) -- the activation
    # stash (3.84 KB/row) and the d pre-activations (3.84 KB/row) written by the chain, the stash read back by it and both
    # read by the weight-gradient kernels -- over the same times; which roof binds is the larger fraction
    r = out["roofline"]
    chain_bytes, wgrad_bytes = (36 + 3 * 3840 + 3 * 256 + 2 * 8) * rows, (2 * 3840 + 32 + 8) * rows      # (+ v2 and du, dz across the kernel seam)
    out["roofline_hbm_design"] = {
        "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
        "chain": {"bytes": chain_bytes, "achieved": chain_bytes / (r["ms"] * 1e-3) / 1e9, "frac": chain_bytes / (r["ms"] * 1e-3) / 1e9 / 8000.0},
        "wgrad": {"bytes": wgrad_bytes, "achieved": wgrad_bytes / (r["wgrad"]["ms"] * 1e-3) / 1e9,
                  "frac": wgrad_bytes / (r["wgrad"]["ms"] * 1e-3) / 1e9 / 8000.0},
        "step": {"bytes": chain_bytes + wgrad_bytes, "achieved": (chain_bytes + wgrad_bytes) / (out["ms_per_step"] * 1e-3) / 1e9,
                 "frac": (chain_bytes + wgrad_bytes) / (out["ms_per_step"] * 1e-3) / 1e9 / 8000.0},
        "note": "design bytes: stash written once (h and d pre as packed fp16 fragments, 3.84 KB/row each + heads), h read back by the chain, "
                "both read once by the weight-gradient kernels; algorithmic bytes are 36 B/row (roofline.hbm_algorithmic_GBps).  A pure "
                "LDS-DMA streaming read sustains 6.3 TB/s on this part (profiles/r03/hbm_read_probe.txt): the practical roof for wgrad"}
    if rank == 0 and world == 1 and not args.no_configs:
        del stage_cache
        model._work.clear()
        torch.cuda.empty_cache()
        out["configs"] = {"4_step": leg_config4_step(dev), "5": leg_config5(dev), "residuals": leg_residuals(dev),
                          "1": leg_config1(not args.no_cpu)}
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_rows, 0 if args.no_cpu_full else rows, args.mc_passes)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
