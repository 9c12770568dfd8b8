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
the matrix products run on the bf16 matrix cores with every operand split into three bf16 parts
(hi + mid + lo = the fp32 value exactly) and six cross products -- the accuracy of an fp32 matmul
(same parity tests and tolerances as the exact-fp32 kernels).  The exact-fp32 kernels
(v_mfma_f32_*_f32) and the opt-in bf16-mixed ones are measured beside it ("exact_fp32",
"bf16_mixed" objects) unless --only.

Extra legs reported in the same JSON line (rank 0):
   mc_dropout   : get_MC_samples-equivalent launch (1 eval + T stochastic passes, on-chip reduce)
   roofline     : dominant kernel (forward+backward chain) timed alone with events on the launch
                  stream; algorithmic FLOP per row = 4*M - 4096 (forward 2M + dgrad 2(M - 8H)),
                  M = 174 400 MAC.  Peak: f32x6 executes 6 bf16 MFMA FLOP per algorithmic FLOP, so
                  its ceiling is the dense bf16 peak / 6 = 416.7 TFLOP/s; exact fp32: 157.3 TFLOP/s.
   cpu_baseline : the CPU oracle's train_dnn step (torch CPU, autograd, torch-bernoulli masks,
                  Adam) on a bounded row sample, host cores of this box (N = 1 only)
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
# per precision: (dtype string, chain kernel name, forward kernel name, peak in algorithmic TFLOP/s)
PRECISIONS = {
    "f32x6": ("f32 (products as 3x bf16-split operands, 6 bf16 MFMAs each, f32 accumulate: fp32-accurate)",
              "train_chain_x6_kernel<256>", "mlp_x6_kernel<256,MC>", PEAK_BF16_MFMA_TFLOPS / 6.0),
    "f32x6g3": ("f32 (as f32x6; weight gradients from 2 bf16 parts / 3 products)",
                "train_chain_x6_kernel<256>", "mlp_x6_kernel<256,MC>", PEAK_BF16_MFMA_TFLOPS / 6.0),
    "fp32": ("f32 (exact: v_mfma_f32_*_f32)", "train_chain_kernel<256>", "mlp_kernel<256,MC>", PEAK_FP32_MFMA_TFLOPS),
    "bf16": ("bf16 MFMA inputs, f32 accumulate/activations/loss/master weights (parity rtol 2e-2)",
             "train_chain_bf16_kernel<256>", "mlp_bf16_kernel<256,MC>", PEAK_BF16_MFMA_TFLOPS),
}


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
    ap.add_argument("--precision", default="f32x6", choices=sorted(PRECISIONS), help="arithmetic of the headline measurement")
    ap.add_argument("--only", action="store_true", help="skip the extra legs in the other precisions")
    ap.add_argument("--no-bf16", action="store_true", help="skip the extra bf16/fp32-mixed leg")
    return ap.parse_args()


def cpu_baseline(rows):
    """CPU oracle train_dnn step (the reference's arithmetic, restated) on `rows` rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pinn_oracle as O
    from pinn_amd import synth
    ds = synth.make_dataset(rows, (), seed=0)
    x, y = ds[0], ds[1]
    P = O.init_params([8, H, H, H, 1], seed=0)
    opt = O.AdamState(P)
    gen = torch.Generator().manual_seed(0)
    times = []
    for it in range(6):                                  # 1 warm-up + 5 timed repetitions, median
        t0 = time.perf_counter()
        masks = [(torch.rand(rows, w, generator=gen) >= 0.2) for w in (H, H, H, H // 2)]   # bernoulli draws, as the reference pays
        _, _, grads, _, _ = O.nll_loss_and_grads(P, x, y, [0.2] * 4, masks)
        opt.step(P, grads, 0.01)
        times.append(time.perf_counter() - t0)
    best = sorted(times[1:])[len(times[1:]) // 2]
    # stochastic forward (MC-dropout unit of work)
    t0 = time.perf_counter()
    with torch.no_grad():
        for t in range(2):
            masks = [(torch.rand(rows, w, generator=gen) >= 0.4) for w in (H, H, H, H // 2)]
            O.mlp_forward(P, x, [0.4] * 4, masks)
    fwd = (time.perf_counter() - t0) / 2
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": rows / best, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "train_dnn step (fwd+NLL+autograd bwd+Adam, torch-bernoulli masks) of oracle/pinn_oracle.py on %d rows, "
                      "median of 5 after 1 warm-up; os.cpu_count()=%d; %s" % (rows, os.cpu_count(), model_name),
            "mc_fwd_passes_per_s": rows / fwd}


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
        # ---- train_dnn step (01:949-955)
        loss = model.train_step_grads(xd, yd, model.row_offset, n_global)
        dp.allreduce_grads(model.dnn._flat_grad_full, loss, model._group)
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
        dtype, chain_kernel, fwd_kernel, peak = PRECISIONS[precision]
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
        achieved = CHAIN_FLOP_PER_ROW * rows / (ms_chain * 1e-3) / 1e12
        wg = FWD_FLOP_PER_ROW * rows / (ms_wgrad * 1e-3) / 1e12
        r["roofline"] = {"kernel": chain_kernel, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": None, "flop_per_row": CHAIN_FLOP_PER_ROW, "ms": ms_chain,
                         "wgrad": {"ms": ms_wgrad, "achieved": wg, "frac": wg / (PEAK_BF16_MFMA_TFLOPS / 3.0 if precision == "f32x6g3" else peak)},
                         "reduce_ms": ms_reduce,
                         "step_mfma_frac": STEP_FLOP_PER_ROW * rows / (r["ms_per_step"] * 1e-3) / 1e12 / peak,
                         "hbm_algorithmic_GBps": 36.0 * rows / (r["ms_per_step"] * 1e-3) / 1e9}
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
                               "roofline": {"kernel": fwd_kernel, "bound": "mfma", "achieved": mc_tf, "peak": peak, "unit": "TFLOP/s",
                                            "frac": mc_tf / peak, "hbm_algorithmic_GBps": 44.0 * rows / mc_s / 1e9},
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
    # HBM bytes per launch of the chain kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
    # in separate runs at this workload size); null when the row count or the kernel differs
    pmc = os.path.join(ROOT, "profiles", "r01", "pmc_summary_x6.json" if args.precision.startswith("f32x6") else "pmc_summary_v2.json")
    if rows == 1_000_000 and os.path.exists(pmc):
        try:
            key = head["roofline"]["kernel"].split("<")[0]
            k = [v for n, v in json.load(open(pmc)).items() if key in n][0]
            # gfx950 correction of the guide's HBM section: FETCH_SIZE counts 16-B/lane streaming reads at half their bytes
            out["roofline"]["traffic"] = (2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0
            out["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT) + ": (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024, per launch, N=1e6 " \
                "(separate --pmc passes; the factor 2 is the guide's gfx950 correction for 16-B/lane reads and an upper bound here: the " \
                "layer-0 activation re-reads are dword loads); by design ~3.6 GB of stash reads + 7.7 GB of stash writes, not the 36 B/row"
        except Exception:
            pass

    # ------------------------------------------------------------------ extra legs: the same measurement in the other precisions
    if not args.only:
        extras = [("exact_fp32", "fp32")] + ([] if args.no_bf16 else [("bf16_mixed", "bf16")])
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

    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_rows)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
