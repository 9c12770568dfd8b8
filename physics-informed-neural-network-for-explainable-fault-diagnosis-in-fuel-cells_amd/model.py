"""`DNN` and `PhysicsInformedNN` -- the reference's model surface on the gfx950 HIP library.

Mirrors 01_train_pinn_multiphysics_model.py (cited 01:<line>): same constructor, attributes
(`.dnn`, `.lambda_*`, `.x`, `.u`, `.X`, `.x_scal`, `.u_scal`), methods (`net_u`, `net_f_V`,
`net_f_T_simple`, `net_f_T`, `net_f_H`, `net_f_O`, `aleatoric_loss`, `train_dnn`, `train_lambda`,
`train_thermal`, `train_hydrogen`, `train_oxygen`, `predict`) and tuple orders, so the
reference's `__main__` (01:2139-2158) and its scripts 02-05 run unchanged on top.

What is underneath is different: every numeric step is a call through the C ABI of
include/pinn_hip.h (ctypes, raw device pointers, the current HIP stream).  torch supplies
device memory, streams and `torch.distributed`; there is no CPU fallback.

Extensions (keyword-only, all optional): `precision` ("f32x6", default: fp32-accurate matrix math on the bf16
matrix cores, same parity tolerances as "fp32"; "fp32": exact fp32 MFMA; "bf16" = bf16 MFMA inputs with fp32
accumulation, rtol ~2e-2 -- see DNN.set_precision), `seed` (Philox dropout seed), `row_offset` /
`n_global` (this process holds rows [row_offset, row_offset+N) of an n_global-row series:
data-parallel training with one all-reduce(SUM) of the flat gradient per step), and
`train_dnn(..., batch_size=)` for minibatches.
"""
import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, layout
from . import dp as _dp

LAMBDA_NAMES = ["lambda_1", "lambda_2", "lambda_3", "lambda_4",
                "lambda_T1", "lambda_T2", "lambda_T3", "lambda_T4", "lambda_T5",
                "lambda_H1", "lambda_H2", "lambda_H3", "lambda_H4",
                "lambda_O1", "lambda_O2", "lambda_O3", "lambda_O4"]
# initial values: 01:453-456, 477-481, 497-500, 514-517
LAMBDA_INIT = [0.167897923477715, 2.36682075851268e-06, 2.43414469188443, 1.0,
               10.0, 10.0, 10.0, 10.0, 10.0,
               5.0, -1.559, 197.715, 1.20,
               2.0, 0.5, 200.0, 1.0]


def _device():
    if not torch.cuda.is_available():
        raise _lib.PinnError("pinn_amd needs a ROCm GPU (gfx950): torch.cuda.is_available() is False and there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class DNN(torch.nn.Module):
    """01:389-438.  Same module tree / state_dict keys as the reference; the 14 weight and
    bias tensors are views into ONE flat float32 device buffer that the kernels read."""

    def __init__(self, p, logvar, layers, seed=0, precision="f32x6"):
        super().__init__()
        self.depth = len(layers) - 1
        self.p = p
        self.logvar = logvar
        self.activation = torch.nn.Tanh
        self.n_in, self.hidden, self.n_hidden = layout.check_arch(layers)
        dev = _device()
        self._lib = _lib.load()
        self._net = _lib.Net(self.n_in, self.hidden, self.n_hidden)
        offs, total = layout.param_offsets(self.n_in, self.hidden, self.n_hidden)
        assert self._lib.pinn_param_count(ctypes.byref(self._net)) == total
        self._packed = None
        self.set_precision(precision)
        self._offsets = offs
        self._flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._flat_grad_full = torch.zeros(total + _dp.LOSS_TAIL, dtype=torch.float32, device=dev)
        self._flat_grad = self._flat_grad_full[:total]

        layer_list = []
        for i in range(self.depth - 1):
            layer_list.append(("layer_%d" % i, torch.nn.Linear(layers[i], layers[i + 1])))
            layer_list.append(("activation_%d" % i, self.activation()))
            layer_list.append(("dropout_%d" % i, torch.nn.Dropout(p=self.p)))
        self.layers = torch.nn.Sequential(OrderedDict(layer_list))
        self.predict = torch.nn.Linear(layers[-2], layers[-1])
        self.var_layers = torch.nn.Sequential(
            torch.nn.Linear(layers[-2], layers[-2] // 2), torch.nn.Tanh(), torch.nn.Dropout(p=self.p),
            torch.nn.Linear(layers[-2] // 2, layers[-2] // 4), torch.nn.Tanh(),
            torch.nn.Linear(layers[-2] // 4, layers[-1]))
        # move the torch-default initial values into the flat buffer and alias the Parameters onto it
        mods = dict(self.named_modules())
        self._views = []
        for name, shape, off in offs:
            mname, pname = name.rsplit(".", 1)
            mod = mods[mname]
            init = getattr(mod, pname).detach()
            n = init.numel()
            view = self._flat[off:off + n].view(shape)
            view.copy_(init)
            par = torch.nn.Parameter(view)
            par.grad = self._flat_grad[off:off + n].view(shape)
            setattr(mod, pname, par)
            self._views.append((mod, pname, off, n, shape))
        self.seed = int(seed)
        self._fwd_counter = 0
        # parity-test hook: int32 device tensor [n_passes, N, words] of bit-packed keep-masks;
        # when set, stochastic passes replay these masks instead of drawing Philox ones
        self._mask_bits = None
        self._mask_pass = 0

    def set_precision(self, precision):
        """"fp32": exact fp32 matrix math (v_mfma_f32_*_f32; hidden <= 256 only).  "f32x6" (default): fp32-ACCURATE matrix
        math on the 16-bit matrix cores from split operands (two fp16 parts, three products, fp32 accumulation; gradients
        under exact power-of-two scales) -- same parity tolerances as "fp32", 2-3x faster.  "f32x6g6": as "f32x6" with the
        gradients from three bf16 parts / six products (24-bit operands, fp32's exponent range per element; ~20 % slower).
        "bf16": bf16 MFMA inputs, fp32 accumulate / activations / loss / master weights; rtol ~2e-2."""
        codes = {"fp32": _lib.PREC_FP32, "bf16": _lib.PREC_BF16, "f32x6": _lib.PREC_F32X6, "f32x6g6": _lib.PREC_F32X6_G6}
        if precision not in codes:
            raise ValueError("precision must be 'fp32', 'f32x6', 'f32x6g6' or 'bf16'")
        self.precision = precision
        if precision == "fp32":
            self._net = _lib.Net(self.n_in, self.hidden, self.n_hidden, _lib.PREC_FP32, None)
            return
        probe = _lib.Net(self.n_in, self.hidden, self.n_hidden, codes[precision], None)
        nbytes = self._lib.pinn_packed_bytes(ctypes.byref(probe))
        if self._packed is None or self._packed.numel() < nbytes:
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=_device())
        self._net = _lib.Net(self.n_in, self.hidden, self.n_hidden, codes[precision], self._packed.data_ptr())

    def check_range(self):
        """Raise PinnRangeError when the last launches met a weight (or, in training, a gradient) outside the domain of the
        split-operand precisions -- |w| >= 1023.5 in a hidden or variance-head matrix, which the fp32 reference (01:389-438)
        computes without trouble.  Synchronises the stream: called where the host waits anyway (log lines, results)."""
        rc = self._lib.pinn_net_range_status(ctypes.byref(self._net), _stream())
        if rc == _lib.E_RANGE:
            raise _lib.PinnRangeError(
                "a weight or gradient left the range of precision %r (|w| must stay below 1023.5 in the hidden and variance-head "
                "matrices); the outputs of the last calls are inf / NaN there.  dnn.set_precision('fp32') has no such limit%s"
                % (self.precision, "" if self.hidden <= 256 else " but does not support this width"))
        _lib.check(rc, "pinn_net_range_status")

    # -- flat buffer <-> Parameter aliasing ------------------------------------------------
    def flat_params(self):
        """The flat parameter buffer; re-gathers a Parameter that user code re-pointed elsewhere."""
        base = self._flat.data_ptr()
        for mod, pname, off, n, shape in self._views:
            par = getattr(mod, pname)
            if par.data_ptr() != base + 4 * off or not par.is_contiguous():
                view = self._flat[off:off + n].view(shape)
                view.copy_(par.detach().to(self._flat.device, torch.float32))
                par.data = view
        return self._flat

    def dropout_modules(self):
        return [m for _, m in self.named_modules() if isinstance(m, torch.nn.Dropout)]

    def dropout_struct(self, stream_id, row_offset=0, mode=None, p_override=None):
        d = _lib.Dropout()
        mods = self.dropout_modules()
        d.mode = _lib.DROP_PHILOX if mode is None else mode
        for l, m in enumerate(mods):
            d.p[l] = float(m.p if p_override is None else p_override)
        d.seed = self.seed
        d.stream = int(stream_id) & 0xFFFFFFFF
        d.row_offset = int(row_offset)
        d.d_bits = None
        if self._mask_bits is not None and mode is None:
            d.mode = _lib.DROP_BITS
            d.d_bits = self._mask_bits[self._mask_pass].data_ptr()
        return d

    def inject_masks(self, bits):
        """Replay recorded keep-masks (tests only). bits: int32 [n_passes, N, words] or None."""
        self._mask_bits = None if bits is None else bits.to(self._flat.device).contiguous()
        self._mask_pass = 0

    def forward(self, x, row_offset=0):
        """(out [N,1], logvar [N,1]) -- eval: dropout off; train: on-chip Philox masks (01:421-438)."""
        x = x.detach().to(self._flat.device, torch.float32).contiguous()
        n = x.shape[0]
        u = torch.empty(n, 1, device=x.device, dtype=torch.float32)
        lv = torch.empty(n, 1, device=x.device, dtype=torch.float32)
        drop = None
        if self.training and any(m.p > 0 for m in self.dropout_modules()):
            self._fwd_counter += 1
            drop = self.dropout_struct(0x80000000 + self._fwd_counter, row_offset)
            if self._mask_bits is not None:
                self._mask_pass += 1
        rc = self._lib.pinn_mlp_forward(ctypes.byref(self._net), _ptr(self.flat_params()), _ptr(x), n,
                                        ctypes.byref(drop) if drop is not None else None, _ptr(u), _ptr(lv), _stream())
        _lib.check(rc, "pinn_mlp_forward")
        if not self.logvar:
            lv = torch.zeros_like(u)
        return u, lv


class PhysicsInformedNN():
    """01:441-1410."""

    def __init__(self, X, u, layers, x_scal, u_scal, p, logvar, *, seed=0, row_offset=0, n_global=None, process_group=None,
                 precision="f32x6"):
        dev = _device()
        self._lib = _lib.load()
        self.x = X[:, 0:].clone().detach().float().to(dev).contiguous().requires_grad_(True)
        self.u = u.clone().detach().float().to(dev).contiguous()
        self.u_scal = u_scal
        self.x_scal = x_scal
        self.X = X
        self.row_offset = int(row_offset)
        self.n_local = int(self.x.shape[0])
        self.n_global = int(n_global) if n_global is not None else self.n_local
        self._group = process_group
        # 17 physics parameters: one device vector, each nn.Parameter a 1-element view of it
        self._lambda = torch.tensor(LAMBDA_INIT, dtype=torch.float32, device=dev)
        for i, name in enumerate(LAMBDA_NAMES):
            setattr(self, name, torch.nn.Parameter(self._lambda[i:i + 1]))
        self.dnn = DNN(p, logvar, layers, seed=seed, precision=precision)
        # registration order and the `lambda_3` <- lambda_4 overwrite of 01:465-468 are kept for state_dict parity
        self.dnn.register_parameter("lambda_1", self.lambda_1)
        self.dnn.register_parameter("lambda_2", self.lambda_2)
        self.dnn.register_parameter("lambda_3", self.lambda_3)
        self.dnn.register_parameter("lambda_3", self.lambda_4)
        for name in LAMBDA_NAMES[4:]:
            self.dnn.register_parameter(name, getattr(self, name))
        self._step_counter = 0
        self._work = {}
        self._aff_cache = {}
        n = self.dnn._flat.numel()
        self._adam_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self._adam_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._sums = torch.zeros(_lib.NSUMS, dtype=torch.float64, device=dev)
        self._res_work = torch.empty(self._lib.pinn_residuals_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.verbose = True

    # ------------------------------------------------------------------ helpers
    def _lambdas(self):
        base = self._lambda.data_ptr()
        for i, name in enumerate(LAMBDA_NAMES):
            par = getattr(self, name)
            if par.data_ptr() != base + 4 * i:
                self._lambda[i:i + 1].copy_(par.detach().reshape(1).to(self._lambda.device, torch.float32))
                par.data = self._lambda[i:i + 1]
        return self._lambda

    def _affine(self, x_scal):
        key = id(x_scal)
        if key not in self._aff_cache:
            a = _lib.Affine()
            mn = np.asarray(x_scal.min_, dtype=np.float64).reshape(-1)
            sc = np.asarray(x_scal.scale_, dtype=np.float64).reshape(-1)
            if mn.size != 8:
                raise ValueError("x scaler must have 8 features")
            for c in range(8):
                a.x_min[c], a.x_scale[c] = mn[c], sc[c]
            a.y_min = float(np.asarray(self.u_scal.min_, dtype=np.float64).reshape(-1)[0])
            a.y_scale = float(np.asarray(self.u_scal.scale_, dtype=np.float64).reshape(-1)[0])
            # 01:1017-1022: float32 tensors
            lo, hi = float(self.u_scal.feature_range[0]), float(self.u_scal.feature_range[1])
            dmin = torch.tensor(np.asarray(self.u_scal.data_min_), dtype=torch.float32)
            dmax = torch.tensor(np.asarray(self.u_scal.data_max_), dtype=torch.float32)
            scale_y = (hi - lo) / (dmax - dmin + 1e-12)
            min_y = lo - dmin * scale_y
            a.vn_scale, a.vn_min = float(scale_y), float(min_y)
            self._aff_cache[key] = (a, x_scal)      # keep the scaler alive so id() stays unique
        return self._aff_cache[key][0]

    def _dev_rows(self, X):
        if X is self.X or X is self.x:
            return self.x.detach()
        return X.detach().to(self.x.device, torch.float32).contiguous()

    def _residuals(self, X, x_scal, flags, u=None, y=None, cols=True, sums=False):
        xd = self._dev_rows(X)
        n = xd.shape[0]
        c = torch.empty(_lib.NCOLS, n, device=xd.device, dtype=torch.float32) if cols else None
        rc = self._lib.pinn_residuals(_ptr(xd), _ptr(u), _ptr(y), ctypes.byref(self._affine(x_scal)), _ptr(self._lambdas()),
                                      flags, n, _ptr(c), n, _ptr(self._sums) if sums else None,
                                      _ptr(self._res_work), self._res_work.numel(), _stream())
        _lib.check(rc, "pinn_residuals")
        return c

    @staticmethod
    def _col(c, name):
        return c[_lib.C[name]].unsqueeze(1)

    # ------------------------------------------------------------------ model functions
    def net_u(self, x):
        prediction, log_var = self.dnn(x, self.row_offset if x.shape[0] == self.n_local else 0)
        return prediction, log_var

    def net_f_V(self, X, x_scal):
        """01:724-765 -> (f, V_act, V_ohmic, V_conc, E_nerst, V_out_est*5, i, il, V_out*5)."""
        xd = self._dev_rows(X)
        u, _ = self.net_u(xd)                       # in the caller's train/eval mode, detached (01:733-734)
        c = self._residuals(xd, x_scal, _lib.RES_V, u=u.reshape(-1))
        g = lambda n: self._col(c, n)
        return g("FV"), g("VACT"), g("VOHM"), g("VCONC"), g("ENERNST"), g("VEST5"), g("I"), self.lambda_3, g("VOUT5")

    def net_f_T_simple(self, X, x_scal):
        """01:869-914 -> (f_T, T_out_predicted, T_out_real).  (The reference's unused DNN forward is not run.)"""
        c = self._residuals(X, x_scal, _lib.RES_T)
        return self._col(c, "FT"), self._col(c, "TPRED"), self._col(c, "TOUT")

    def net_f_H(self, X, x_scal):
        """01:621-722 -> (f_H2, actual_excess_ratio, target_excess_ratio, I_total, I_threshold)."""
        c = self._residuals(X, x_scal, _lib.RES_H)
        return self._col(c, "FH"), self._col(c, "ACTH"), self._col(c, "TGTH"), self._col(c, "ITOT"), self.lambda_H3

    def net_f_O(self, X, x_scal):
        """01:535-619 -> (f_O2, actual_excess_ratio, target_excess_ratio, Q_O2_theoretical_slpm, o2_flow_actual)."""
        c = self._residuals(X, x_scal, _lib.RES_O)
        return self._col(c, "FO"), self._col(c, "ACTO"), self._col(c, "TGTO"), self._col(c, "QO2"), self._col(c, "O2FLOW")

    def net_f_T(self, X, x_scal, halo=None):
        """01:767-867: Euler energy balance row t-1 -> t -> (f_T, T_out_predicted_full, T_out_real_full), one fused
        kernel (pinn_net_f_t).  The DNN runs in the caller's train / eval mode on the rows, as the reference runs it on
        X[:-1] (01:826-830).  `halo` = (x_row [8], u) of the row before X[0] when X is a row shard that does not start
        the series (device or host tensors; not in the reference, which has no sharding)."""
        xd = self._dev_rows(X)
        n = xd.shape[0]
        dev = xd.device
        if n < 2 and halo is None:
            z = torch.zeros(n, 1, device=dev)
            return z, z.clone(), z.clone()
        u = self.net_u(xd)[0].reshape(-1).contiguous() if n > 0 else None
        xh = uh = None
        if halo is not None:
            xh = torch.as_tensor(halo[0], dtype=torch.float32).reshape(8).to(dev).contiguous()
            uh = torch.as_tensor(halo[1], dtype=torch.float32).reshape(1).to(dev).contiguous()
        out = torch.empty(3, n, device=dev, dtype=torch.float32)
        rc = self._lib.pinn_net_f_t(_ptr(xd), _ptr(u), _ptr(xh), _ptr(uh), ctypes.byref(self._affine(x_scal)), _ptr(self._lambdas()), n,
                                    _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream())
        _lib.check(rc, "pinn_net_f_t")
        return out[0].unsqueeze(1), out[1].unsqueeze(1), out[2].unsqueeze(1)

    def aleatoric_loss(self, gt, pred_y, logvar):
        """01:916-927."""
        precision = torch.exp(-logvar)
        loss = torch.mean(0.5 * precision * (gt - pred_y) ** 2 + 0.5 * logvar)
        return loss + 0.01 * torch.mean(torch.abs(logvar))

    # ------------------------------------------------------------------ trainers
    def _log(self, *a):
        if self.verbose and _dp.rank(self._group) == 0:
            print(*a)

    def _workspace(self, n_rows):
        key = (n_rows, self.dnn.precision)          # the tile padding of the stash depends on the kernels used
        if key not in self._work:
            self._work.clear()
            wb = self._lib.pinn_train_workspace_bytes(ctypes.byref(self.dnn._net), n_rows)
            if wb == 0:
                raise _lib.PinnError("pinn_train_workspace_bytes rejected the network")
            self._work[key] = torch.empty(wb, dtype=torch.uint8, device=self.x.device)
        return self._work[key]

    def train_step_grads(self, x, y, row_offset, n_global, between=None, adam=None):
        """One fused forward + aleatoric_loss + backward on rows (x, y): fills the flat gradient
        (already divided by n_global) and returns the raw loss sums double[4] on the device.
        `between` (data-parallel overlap): called once the TAIL of the gradient -- last hidden layer and heads,
        [pinn_grad_split, end) -- is final, before the head's weight-gradient kernels are launched."""
        n = x.shape[0]
        work = self._workspace(n)
        self._step_counter += 1
        training = self.dnn.training and any(m.p > 0 for m in self.dnn.dropout_modules())
        drop = self.dnn.dropout_struct(self._step_counter, row_offset) if training else None
        if drop is not None and self.dnn._mask_bits is not None:
            self.dnn._mask_pass += 1
        loss = torch.empty(4, dtype=torch.float64, device=x.device)

        def run(phases):
            rc = self._lib.pinn_mlp_train_grads_phases(ctypes.byref(self.dnn._net), _ptr(self.dnn.flat_params()), _ptr(x), _ptr(y), n,
                                                       int(n_global), ctypes.byref(drop) if drop else None,
                                                       _ptr(self.dnn._flat_grad), _ptr(loss), _ptr(work), work.numel(), _stream(), phases)
            _lib.check(rc, "pinn_mlp_train_grads")
        if adam is not None:      # (lr, step): the optimizer step rides in the gradient reduction's launch (single process, no collective in between)
            rc = self._lib.pinn_mlp_train_step(ctypes.byref(self.dnn._net), _ptr(self.dnn.flat_params()), _ptr(x), _ptr(y), n, int(n_global),
                                               ctypes.byref(drop) if drop else None, _ptr(self.dnn._flat_grad), _ptr(loss), _ptr(work),
                                               work.numel(), _ptr(self._adam_m), _ptr(self._adam_v), adam[0], adam[1], _stream())
            _lib.check(rc, "pinn_mlp_train_step")
        elif between is None:
            run(_lib.PHASE_ALL)
        else:
            run(_lib.PHASE_CHAIN | _lib.PHASE_WGRAD_TAIL | _lib.PHASE_REDUCE_TAIL)
            between()
            run(_lib.PHASE_WGRAD_HEAD | _lib.PHASE_REDUCE_HEAD)
        return loss

    # Data-parallel step with the gradient all-reduce in two parts: the tail of the bucket (last hidden layer +
    # heads, 0.43 of 0.70 MB for the reference net) is summed on a side stream while the head's weight-gradient kernels run;
    # both parts are complete before Adam.  Two ranks sum the same pairs of floats either way: bit-identical to one blocking
    # all-reduce (tests/test_gpu_dp.py); overlap_allreduce = False keeps the single collective.
    overlap_allreduce = True

    def _dp_step(self, xb, yb, row_offset, n_norm, has_rows):
        full = self.dnn._flat_grad_full
        split = int(self._lib.pinn_grad_split(ctypes.byref(self.dnn._net))) if self.overlap_allreduce else 0
        if not (_dp._active(self._group) and split > 0):
            if has_rows:
                loss = self.train_step_grads(xb, yb, row_offset, n_norm)
            else:       # a rank without rows in this batch: zero gradient, same dropout-stream position, same collective
                self._step_counter += 1
                full.zero_()
                loss = torch.zeros(4, dtype=torch.float64, device=self.x.device)
            _dp.allreduce_grads(full, self._group)
            return loss
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.x.device)
        works = []

        def tail_ready():
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._side_stream):
                self._side_stream.wait_event(ev)
                works.append(_dp.allreduce_grads_begin(full[split:], self._group))
        if has_rows:
            loss = self.train_step_grads(xb, yb, row_offset, n_norm, between=tail_ready)
        else:
            self._step_counter += 1
            full.zero_()
            loss = torch.zeros(4, dtype=torch.float64, device=self.x.device)
            tail_ready()
        works.append(_dp.allreduce_grads_begin(full[:split], self._group))
        for w in works:
            if w is not None:
                w.wait()              # the current stream waits for the collective (NCCL); gloo: the host does
        return loss

    # train_dnn at the reference's data sizes (1e3-1e4 rows, 12 002 steps: 01:2143-2147) is a chain of short dependent launches
    # (five per step).  With one process and full batches the step can be captured ONCE as a hipGraph and replayed: step count,
    # Adam coefficients and dropout stream live on the device (pinn_dropout_t.d_step_counter, pinn_mlp_train_step_dev), so nothing
    # in the captured sequence changes from step to step.  Same kernels, same arguments otherwise: bit-identical to the
    # launch-by-launch path (tests/test_gpu_model.py).  Opt-in: since the step's launches all sit on one stream, a replayed
    # step is 3-8 us SLOWER than a launched one on this runtime (92-103 against 87-95 us at 500 .. 4200 rows); what the graph
    # buys is a free host thread.
    use_graph = False

    def _train_dnn_replay(self, epochs, x, y, n_norm, log):
        """Steps 2 .. epochs of a full-batch train_dnn call as replays of one captured step (step 1 ran launch by launch:
        it also settles every lazy launch attribute before the capture)."""
        dev = x.device
        n = x.shape[0]
        flat, grad = self.dnn.flat_params(), self.dnn._flat_grad
        coeffs = np.empty((epochs, 2), dtype=np.float32)
        ss, bs = ctypes.c_float(), ctypes.c_float()
        for k in range(epochs):             # entry k: optimizer step k + 1, lr of epoch k (StepLR(1000, 0.8), 01:939-940)
            self._lib.pinn_adam_coeffs(ctypes.c_float(0.01 * 0.8 ** (k // 1000)), k + 1, ctypes.byref(ss), ctypes.byref(bs))
            coeffs[k] = (ss.value, bs.value)
        d_coeffs = torch.from_numpy(coeffs).to(dev)
        counter = torch.ones(1, dtype=torch.int32, device=dev)          # one step done
        work = self._workspace(n)
        loss = torch.empty(4, dtype=torch.float64, device=dev)
        training = self.dnn.training and any(m.p > 0 for m in self.dnn.dropout_modules())
        # step k of the call draws stream S0 + k and mask pass M0 + k - 1 (S0, M0: the positions before the call, as train_step_grads
        # counts them); the kernels add the device counter = k - 1 to what the struct holds
        drop = self.dnn.dropout_struct(self._step_counter, self.row_offset, mode=None if training else _lib.DROP_NONE)
        if training and self.dnn._mask_bits is not None:
            drop.d_bits = self.dnn._mask_bits[self.dnn._mask_pass - 1].data_ptr()
        drop.d_step_counter = counter.data_ptr()
        net = self.dnn._net

        one_call = net.precision in (_lib.PREC_F32X6, _lib.PREC_F32X6_G6)       # the reduction's launch applies the Adam step too

        def step():
            if one_call:
                _lib.check(self._lib.pinn_mlp_train_step_dev(ctypes.byref(net), _ptr(flat), _ptr(x), _ptr(y), n, int(n_norm), ctypes.byref(drop),
                                                             _ptr(grad), _ptr(loss), _ptr(work), work.numel(), _ptr(self._adam_m),
                                                             _ptr(self._adam_v), _ptr(d_coeffs), _stream()), "pinn_mlp_train_step_dev")
                return
            _lib.check(self._lib.pinn_mlp_train_grads(ctypes.byref(net), _ptr(flat), _ptr(x), _ptr(y), n, int(n_norm), ctypes.byref(drop),
                                                      _ptr(grad), _ptr(loss), _ptr(work), work.numel(), _stream()), "pinn_mlp_train_grads")
            _lib.check(self._lib.pinn_adam_step_dev(_ptr(flat), _ptr(grad), _ptr(self._adam_m), _ptr(self._adam_v), flat.numel(),
                                                    _ptr(d_coeffs), _ptr(counter), _stream()), "pinn_adam_step_dev")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        # a graph launch costs ~8 us of idle GPU between two replays (a fifteenth of a step at 1e4 rows): long calls replay
        # kChunk steps per launch wherever no log line falls inside the chunk (a line reads the loss of ITS step)
        kChunk = self.graph_chunk
        chunk = None
        if epochs - 1 >= 4 * kChunk:
            chunk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(chunk):
                for _ in range(kChunk):
                    step()
        epoch = 1
        while epoch < epochs:
            next_log = (epoch + 999) // 1000 * 1000          # the first epoch >= this one that prints a line
            if chunk is not None and epoch + kChunk <= epochs and next_log >= epoch + kChunk - 1:
                chunk.replay()
                epoch += kChunk
            else:
                graph.replay()
                epoch += 1
            if (epoch - 1) % 1000 == 0:
                log(epoch - 1, loss)
        torch.cuda.current_stream().synchronize()
        self._step_counter += epochs - 1
        if training and self.dnn._mask_bits is not None:
            self.dnn._mask_pass += epochs - 1
        self._keep_alive = (d_coeffs, counter, work, loss, graph, chunk)      # until the next call: the stream may still be replaying
        return loss

    def train_dnn(self, nIter, batch_size=None):
        """01:929-964: Adam(lr=0.01) + StepLR(1000, 0.8) over the 14 weight/bias tensors,
        full batch (or `batch_size`-row minibatches: one optimizer step each)."""
        for param in self.dnn.parameters():
            param.requires_grad = True
        for n in LAMBDA_NAMES[:4]:
            getattr(self, n).requires_grad = False
        self._adam_m.zero_(); self._adam_v.zero_()
        self._log('================== DNN training ==================')
        self._log('  Epoch |    Loss    |    MSE     |    LR    ')
        self.dnn.train()
        x, y = self.x.detach(), self.u.reshape(-1)
        n = self.n_local
        flat, grad = self.dnn.flat_params(), self.dnn._flat_grad
        step = 0
        loss_sums = None
        n_norm = max(1, self.n_global)
        # the schedule has the same length on every rank (empty batches included): one gradient all-reduce per entry
        batches = _dp.batch_schedule(n, batch_size, self.x.device, self._group)

        def log_line(epoch, sums):
            self.dnn.check_range()
            ls = _dp.allreduce_sums(sums.clone(), self._group).cpu().numpy()      # fp64, only when a line is printed
            lr_next = 0.01 * 0.8 ** ((epoch + 1) // 1000)
            self._log(f' {epoch:5d}  | {(ls[0] + 0.01 * ls[1]) / n_norm:10.3e} | {ls[2] / n_norm:10.3e} | {lr_next:8.1e}')
        replay = (self.use_graph and nIter >= max(2, self.graph_min_steps) and len(batches) == 1 and batches[0][:2] == (0, n) and n > 0 and not _dp._active(self._group)
                  and self.dnn.hidden <= 256)
        for epoch in range(nIter):
            lr = 0.01 * 0.8 ** (epoch // 1000)
            for (s, e, n_norm) in batches:
                xb, yb = (x, y) if (s, e) == (0, n) else (x[s:e], y[s:e])
                step += 1
                if self.fuse_adam and not _dp._active(self._group) and e > s:      # one process: gradients and Adam in one launch sequence
                    loss_sums = self.train_step_grads(xb, yb, self.row_offset + s, n_norm, adam=(lr, step))
                    continue
                loss_sums = self._dp_step(xb, yb, self.row_offset + s, n_norm, e > s)
                rc = self._lib.pinn_adam_step(_ptr(flat), _ptr(grad), _ptr(self._adam_m), _ptr(self._adam_v), flat.numel(),
                                              lr, step, _stream())
                _lib.check(rc, "pinn_adam_step")
            if epoch % 1000 == 0 and loss_sums is not None:
                log_line(epoch, loss_sums)
            if replay:
                loss_sums = self._train_dnn_replay(nIter, x, y, n_norm, log_line)
                break
        if loss_sums is not None:
            self.dnn.check_range()
            ls = _dp.allreduce_sums(loss_sums.clone(), self._group).cpu().numpy()
            self.last_loss = (ls[0] + 0.01 * ls[1]) / n_norm
            self._log(f'DNN training done, final loss: {self.last_loss:.3e}')

    def _run_lambda_stage(self, stage, nIter, flags, lr0, gamma, need_u, log_fn):
        self.dnn.eval()
        dev = self.x.device
        x, y = self.x.detach(), self.u.reshape(-1)
        aff = self._affine(self.x_scal)
        u = None
        if need_u:      # DNN weights are frozen during the stage: one eval forward serves every iteration
            u = self.dnn(x)[0].reshape(-1)
        adam = torch.zeros(2 * _lib.NLAMBDA, dtype=torch.float32, device=dev)
        loss = torch.zeros(2, dtype=torch.float32, device=dev)
        lam = self._lambdas()
        if self._group is None and self.n_global == self.n_local and 0 < self.n_local <= self.stage_run_max_rows and nIter > 0:
            # the reference's data sizes: the whole stage in ONE launch (persistent workgroup)
            n_log = (nIter + 999) // 1000
            log = torch.zeros(n_log, _lib.STAGE_LOG_FLOATS, dtype=torch.float32, device=dev)
            wb = self._lib.pinn_lambda_stage_workspace_bytes(self.n_local)
            work = torch.empty(wb, dtype=torch.uint8, device=dev)
            rc = self._lib.pinn_lambda_stage_run(stage, flags, _ptr(x), _ptr(u), _ptr(y), ctypes.byref(aff), self.n_local, lr0, gamma, 1000, 0,
                                                 nIter, _ptr(lam), _ptr(adam), _ptr(loss), _ptr(log), 1000, _ptr(self._sums), _ptr(work), wb,
                                                 _stream())
            _lib.check(rc, "pinn_lambda_stage_run")
            rows = log.cpu().numpy()
            for k in range(n_log):
                self._lambda_log_view = rows[k, 3:20]
                log_fn(1000 * k, rows[k, 0:2], rows[k, 20:20 + 32].astype("float64"), float(rows[k, 2]))
            self._lambda_log_view = None
            self.last_loss = float(loss[0].item())
            return
        # large series / row shards: the parameter-independent half of every row once, then per iteration a pass over that
        # 8-24 B/row cache -> [all-reduce of 32 sums] -> Adam + clamp on the device
        cache = None
        if nIter > 0 and self.n_local > 0:
            cache = torch.empty(6 * self.n_local, dtype=torch.float32, device=dev)
            rc = self._lib.pinn_residuals_prepare(_ptr(x), _ptr(u), _ptr(y), ctypes.byref(aff), _ptr(lam), flags, self.n_local, _ptr(cache),
                                                  _stream())
            _lib.check(rc, "pinn_residuals_prepare")
        for epoch in range(nIter):
            lr = lr0 * gamma ** (epoch // 1000)
            if cache is not None:
                rc = self._lib.pinn_residuals_cached(_ptr(cache), ctypes.byref(aff), _ptr(lam), flags, self.n_local, _ptr(self._sums),
                                                     _ptr(self._res_work), self._res_work.numel(), _stream())
            else:       # a rank without rows still takes part in the all-reduce
                rc = self._lib.pinn_residuals(_ptr(x), _ptr(u), _ptr(y), ctypes.byref(aff), _ptr(lam), flags, self.n_local,
                                              None, 0, _ptr(self._sums), _ptr(self._res_work), self._res_work.numel(), _stream())
            _lib.check(rc, "pinn_residuals")
            _dp.allreduce_sums(self._sums, self._group)
            rc = self._lib.pinn_lambda_step(stage, _ptr(self._sums), self.n_global, aff.vn_scale, lr, epoch + 1, _ptr(lam),
                                            _ptr(adam), _ptr(loss), _stream())
            _lib.check(rc, "pinn_lambda_step")
            if epoch % 1000 == 0:
                log_fn(epoch, loss.cpu().numpy(), self._sums.cpu().numpy(), lr0 * gamma ** ((epoch + 1) // 1000))
        self.last_loss = float(loss[0].item()) if nIter > 0 else None

    graph_chunk = 8                 # train_dnn steps per graph launch in long replayed calls
    fuse_adam = True                # single process: the Adam step rides in the gradient reduction's launch (False: two calls, bit-identical)
    graph_min_steps = 200           # shorter calls launch their steps one by one: a capture costs ~0.5 ms, a replayed step is no faster
    stage_run_max_rows = 32768      # <= _lib.STAGE_RUN_MAX_ROWS; larger series iterate the multi-workgroup kernels
    _lambda_log_view = None

    def _freeze_all_but(self, live):
        for n in LAMBDA_NAMES:
            getattr(self, n).requires_grad = n in live

    def train_lambda(self, nIter, dnn_para=False):
        """01:966-1058."""
        for param in self.dnn.parameters():
            param.requires_grad = dnn_para
        self._freeze_all_but(LAMBDA_NAMES[:4])
        self._log('================ voltage parameter training ================')
        self._log('  Epoch | total loss | phys loss |   l1    |    l2     |   l3   |    LR    ')

        def log(epoch, loss, sums, lr):
            l = self._lambda_log_view if self._lambda_log_view is not None else self._lambda.cpu().numpy()
            self._log(f' {epoch:5d}  | {loss[0]:9.3e} | {loss[1]:10.3e} | {l[0]:7.4f} | {l[1]:9.2e} | {l[2]:6.3f} | {lr:8.1e}')
        self._run_lambda_stage(_lib.STAGE_LAMBDA_F if dnn_para else _lib.STAGE_LAMBDA_PM, nIter, _lib.RES_V, 1e-3, 0.8, True, log)

    def train_thermal(self, nIter):
        """01:1060-1151."""
        for param in self.dnn.parameters():
            param.requires_grad = False
        self._freeze_all_but(LAMBDA_NAMES[4:9])
        self._log('---------------- thermal parameter training ----------------')
        self._log(' Epoch |   Loss    |  MAE(C)  |   T1    |   T2   |   T3   |   T4   |    T5    |    LR   |')

        def log(epoch, loss, sums, lr):
            l = self._lambda_log_view if self._lambda_log_view is not None else self._lambda.cpu().numpy()
            mae = sums[_lib.S["FT_ABS"]] / self.n_global
            self._log(f' {epoch:3d}   | {loss[0]:9.3e} | {mae:8.2f} | {l[4]:7.4f} | {l[5]:6.3f} | {l[6]:6.3f} | {l[7]:6.2f} | {l[8]:6.2f} |{lr:8.1e}')
        self._run_lambda_stage(_lib.STAGE_THERMAL, nIter, _lib.RES_T, 1.0, 0.8, False, log)

    def train_hydrogen(self, nIter):
        """01:1305-1399."""
        for param in self.dnn.parameters():
            param.requires_grad = False
        self._freeze_all_but(LAMBDA_NAMES[9:13])
        self._log('================ hydrogen parameter training ================')
        self._log(' Epoch |   Loss    |   actual   |   target   |   H1    |   H2   |   H3   |   H4   |    LR    ')

        def log(epoch, loss, sums, lr):
            l = self._lambda_log_view if self._lambda_log_view is not None else self._lambda.cpu().numpy()
            self._log(f' {epoch:3d}   | {loss[0]:9.3e} | {sums[_lib.S["ACTH"]] / self.n_global:10.3f} | '
                      f'{sums[_lib.S["TGTH"]] / self.n_global:10.3f} | {l[9]:7.4f} | {l[10]:6.3f} | {l[11]:6.3f} | {l[12]:6.2f} | {lr:8.1e}')
        self._run_lambda_stage(_lib.STAGE_HYDROGEN, nIter, _lib.RES_H, 1e-1, 0.9, False, log)

    def train_oxygen(self, nIter):
        """01:1153-1303."""
        for param in self.dnn.parameters():
            param.requires_grad = False
        self._freeze_all_but(LAMBDA_NAMES[13:17])
        self._log('================ oxygen parameter training ================')
        self._log(' Epoch |   Loss    |   actual   |   target   |   O1    |   O2   |   O3   |   O4   |    LR    ')

        def log(epoch, loss, sums, lr):
            l = self._lambda_log_view if self._lambda_log_view is not None else self._lambda.cpu().numpy()
            self._log(f' {epoch:3d}   | {loss[0]:6.3e} | {sums[_lib.S["ACTO"]] / self.n_global:6.3f} | '
                      f'{sums[_lib.S["TGTO"]] / self.n_global:6.3f} | {l[13]:7.2f} | {l[14]:6.3f} | {abs(l[15]):6.1f}A | {l[16]:5.2f} | {lr:8.1e}')
        self._run_lambda_stage(_lib.STAGE_OXYGEN, nIter, _lib.RES_O, 1e-2, 0.9, False, log)

    def predict(self, X, x_scal):
        """01:1401-1410: (u [N,1], log_var [N,1]) numpy, in the caller's train/eval mode.
        (The reference's discarded `net_f_V` call, 01:1407, is not run.)"""
        x = self._dev_rows(X)
        u, log_var = self.net_u(x)
        self.dnn.check_range()
        return u.detach().cpu().numpy(), log_var.detach().cpu().numpy()

    # ------------------------------------------------------------------ MC-dropout (used by mc.get_MC_samples)
    def mc_dropout(self, X, mc_times, row_offset=0):
        """1 eval pass + mc_times stochastic passes in one persistent launch -> three [N] device tensors."""
        x = self._dev_rows(X)
        n = x.shape[0]
        out = torch.empty(3, n, device=x.device, dtype=torch.float32)
        self._mc_calls = getattr(self, "_mc_calls", 0) + 1
        drop = self.dnn.dropout_struct(0xC0000000 + (self._mc_calls << 16), row_offset)
        rc = self._lib.pinn_mc_dropout(ctypes.byref(self.dnn._net), _ptr(self.dnn.flat_params()), _ptr(x), n, ctypes.byref(drop),
                                       int(mc_times), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream())
        _lib.check(rc, "pinn_mc_dropout")
        return out[0], out[1], out[2]
