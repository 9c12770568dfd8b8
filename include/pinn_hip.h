/* pinn_hip.h -- C ABI of the MI355X (gfx950) PINN training + MC-dropout hot path.
 *
 * The reference (ZhendongS/Physics-Informed-Neural-Network-...-Fuel-Cells) has no FFI or
 * plugin interface: its hot path is the Python object surface of
 * 01_train_pinn_multiphysics_model.py (cited as 01:<line>).  Each entry point below
 * replaces the torch/sklearn/numpy work underneath one group of those methods; the
 * Python classes in the package keep the reference's names and signatures on top
 * (INTEGRATION.md shows the ctypes binding a maintainer of the reference would add).
 *
 * Conventions: every pointer named d_* is DEVICE memory owned by the caller (e.g. a torch
 * tensor's data_ptr()); every call is asynchronous on `stream` (a hipStream_t passed as
 * void*), allocates nothing, never synchronises the device and returns 0 on success or a
 * negative PINN_E_* / positive hipError_t code.  Nothing throws across the ABI.
 * One host thread per process/GPU drives the library; it is not re-entrant per workspace.
 */
#ifndef PINN_HIP_H
#define PINN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PINN_ABI_VERSION 2   /* 2: pinn_dropout_t.d_step_counter, pinn_adam_step_dev, pinn_net_range_status; precision code 3 = F32X6_G6 */

/* error codes (negative; positive values are hipError_t) */
#define PINN_OK 0
#define PINN_E_ARG (-1)        /* null pointer / negative size / bad flag */
#define PINN_E_ARCH (-2)       /* network shape not supported by the fused kernels */
#define PINN_E_WORKSPACE (-3)  /* workspace too small */
#define PINN_E_RANGE (-4)      /* pinn_net_range_status: a weight or gradient left the domain of the split-operand kernels */

/* ---- physics parameters: float[17] on the device, order of 01:453-517 ------------------ */
enum {
  PINN_L1 = 0, PINN_L2, PINN_L3, PINN_L4,            /* voltage: r, io, il, (unused) */
  PINN_LT1, PINN_LT2, PINN_LT3, PINN_LT4, PINN_LT5,  /* thermal */
  PINN_LH1, PINN_LH2, PINN_LH3, PINN_LH4,            /* hydrogen */
  PINN_LO1, PINN_LO2, PINN_LO3, PINN_LO4,            /* oxygen */
  PINN_NLAMBDA = 17
};

/* ---- MinMaxScaler affine maps (host struct, passed by pointer, copied at call time) ----
 * x_phys = (x_n - x_min[c]) / x_scale[c]  : sklearn inverse_transform (01:542, 629, 726, 879)
 *   evaluated as two float64-operand steps each rounded to float32, like numpy does in place.
 * y likewise for the DNN output (01:735).  vn_scale / vn_min: the float32 pair train_lambda
 * builds at 01:1017-1022 to map the physics voltage back to normalised units. */
typedef struct pinn_affine {
  double x_min[8];
  double x_scale[8];
  double y_min;
  double y_scale;
  float vn_scale;
  float vn_min;
} pinn_affine_t;

/* ---- residual pass: net_f_V / net_f_T_simple / net_f_H / net_f_O  (01:724-765, 869-914,
 *      621-722, 535-619) + the stage losses mean(f^2) and their lambda gradients ---------- */
#define PINN_RES_V 1u
#define PINN_RES_T 2u
#define PINN_RES_H 4u
#define PINN_RES_O 8u
#define PINN_RES_ALL 15u

/* per-row output columns (column-major: d_cols[c * ld + row]) */
enum {
  PINN_C_FV = 0, PINN_C_VACT, PINN_C_VOHM, PINN_C_VCONC, PINN_C_ENERNST, PINN_C_VEST5, PINN_C_I, PINN_C_VOUT5,
  PINN_C_FT, PINN_C_TPRED, PINN_C_TOUT,
  PINN_C_FH, PINN_C_ACTH, PINN_C_TGTH, PINN_C_ITOT,
  PINN_C_FO, PINN_C_ACTO, PINN_C_TGTO, PINN_C_QO2, PINN_C_O2FLOW,
  PINN_NCOLS = 20
};

/* reduced sums (double[PINN_NSUMS], sums over rows -- divide by the GLOBAL row count) */
enum {
  PINN_S_FV2 = 0, PINN_S_FV_D1, PINN_S_FV_D2, PINN_S_FV_D3, /* sum f_V^2, sum f_V * df_V/dlambda_k        */
  PINN_S_YV2, PINN_S_YV_D1, PINN_S_YV_D2, PINN_S_YV_D3,     /* sum (y-Vn)^2, sum (y-Vn) * df_V/dlambda_k   */
  PINN_S_YU2,                                               /* sum (y-u)^2  (data loss, 01:1033)           */
  PINN_S_FT2, PINN_S_FT_D1, PINN_S_FT_D3, PINN_S_FT_D5, PINN_S_FT_ABS,
  PINN_S_FH2, PINN_S_FH_D1, PINN_S_FH_D2, PINN_S_FH_D3, PINN_S_ACTH, PINN_S_TGTH,
  PINN_S_FO2, PINN_S_FO_D1, PINN_S_FO_D2, PINN_S_FO_D3, PINN_S_ACTO, PINN_S_TGTO,
  PINN_NSUMS = 32
};

/* Workspace (bytes) pinn_residuals needs for its per-workgroup partial sums. */
size_t pinn_residuals_workspace_bytes(void);

/* One pass over n_rows normalised rows.
 *   d_x      [n_rows, 8] row-major float32 (normalised inputs)
 *   d_u      [n_rows] DNN mean output, normalised (needed iff flags & PINN_RES_V), else NULL
 *   d_y      [n_rows] normalised target or NULL (then the YV / YU sums are 0)
 *   d_lambda [17] float32
 *   d_cols   NULL, or PINN_NCOLS columns of leading dimension ld >= n_rows
 *   d_sums   NULL, or double[PINN_NSUMS]: overwritten with this call's sums (deterministic:
 *            per-workgroup partials in d_work, then a fixed-order final reduction)
 */
int pinn_residuals(const float* d_x, const float* d_u, const float* d_y, const pinn_affine_t* aff,
                   const float* d_lambda, unsigned flags, long long n_rows,
                   float* d_cols, long long ld, double* d_sums, void* d_work, size_t work_bytes,
                   void* stream);

/* ---- physics-parameter stage step: grads from sums -> Adam -> clamp (01:1036-1047 and the
 *      three sibling trainers).  Runs on the device so a stage needs no host round trip. ---- */
enum { PINN_STAGE_LAMBDA_PM = 0, /* train_lambda(dnn_para=False): loss mean((y-Vn)^2)+mean((y-u)^2) */
       PINN_STAGE_LAMBDA_F = 1,  /* train_lambda(dnn_para=True):  loss mean(f_V^2)+mean((y-u)^2)    */
       PINN_STAGE_THERMAL = 2, PINN_STAGE_HYDROGEN = 3, PINN_STAGE_OXYGEN = 4 };

/* d_adam: float[2*17] first/second moments (caller zeroes at stage start), d_loss: float[2]
 * (total, physics) written for logging.  `step` is the 1-based Adam step of this stage,
 * `lr` the StepLR-scheduled rate of this epoch, n_global the global row count. */
int pinn_lambda_step(int stage, const double* d_sums, long long n_global, float vn_scale,
                     float lr, int step, float* d_lambda, float* d_adam, float* d_loss, void* stream);

/* ---- a whole physics-parameter stage in ONE launch (01:999-1055, 1098-1151, 1191-1274, 1344-1391).
 * n_iters iterations, epochs first_epoch .. first_epoch + n_iters - 1, of: residual pass (exactly one of the PINN_RES_*
 * flags) over all n_rows rows -> gradients of the stage's scalars -> Adam(lr = lr0 * gamma^(epoch / lr_step), the
 * StepLR schedule) -> clamp; the arithmetic of pinn_residuals + pinn_lambda_step, run by a single persistent
 * workgroup, so an iteration costs no launch.  For one process holding ALL rows (n_global == n_rows) and
 * n_rows <= PINN_STAGE_RUN_MAX_ROWS; larger or sharded series iterate the two calls above.
 * d_adam as in pinn_lambda_step (carried across calls); d_loss float[2] of the last iteration; d_log: NULL, or
 * float[n_logged][PINN_STAGE_LOG_FLOATS] with one row per epoch divisible by log_every (first_epoch must be):
 * [0] total loss, [1] physics loss, [2] lr of the following epoch, [3..19] the 17 parameters after the step,
 * [20..51] the PINN_NSUMS sums of that epoch (float); d_sums: NULL or double[PINN_NSUMS] of the last iteration. */
#define PINN_STAGE_RUN_MAX_ROWS 65536
#define PINN_STAGE_LOG_FLOATS 64
int pinn_lambda_stage_run(int stage, unsigned flags, const float* d_x, const float* d_u, const float* d_y, const pinn_affine_t* aff,
                          long long n_rows, double lr0, double gamma, int lr_step, int first_epoch, int n_iters, float* d_lambda,
                          float* d_adam, float* d_loss, float* d_log, int log_every, double* d_sums, void* d_work, size_t work_bytes,
                          void* stream);
size_t pinn_lambda_stage_workspace_bytes(long long n_rows);   /* d_work: the rows' parameter-independent terms, computed once per call */

/* The same split for any row count and for row shards (one process per GPU): within one trainer call x, u (the eval forward)
 * and y are fixed, so everything of a row that does not depend on the stage's parameters -- the float64 de-normalisation,
 * powf / expf of the Nernst terms, the flow ratios (A4-A7's parameter-free half) -- is computed once into d_cache
 * (float[6 * n_rows], i.e. pinn_lambda_stage_workspace_bytes(n_rows)) by pinn_residuals_prepare; every iteration then is
 * pinn_residuals_cached (8-24 B/row read) -> [all-reduce of d_sums] -> pinn_lambda_step.  flags: exactly one PINN_RES_*;
 * d_sums / d_work as for pinn_residuals (same sums, same layout; sums of other stages are 0). */
int pinn_residuals_prepare(const float* d_x, const float* d_u, const float* d_y, const pinn_affine_t* aff,
                           const float* d_lambda, unsigned flags, long long n_rows, float* d_cache, void* stream);
int pinn_residuals_cached(const float* d_cache, const pinn_affine_t* aff, const float* d_lambda, unsigned flags,
                          long long n_rows, double* d_sums, void* d_work, size_t work_bytes, void* stream);

/* net_f_T (01:767-867): the Euler energy-balance thermal model, row t-1 -> t, one fused pass (HBM-bound: 36 B/row read,
 * 12 B/row written).  d_u = the DNN's eval-mode output on the same rows (normalised units; row t uses u[t-1], as the
 * reference runs the net on X[:-1]).  Row 0 of the series has no predecessor: T_pred = T_out (01:857).  Under row
 * sharding a rank passes the LAST row of the previous shard (8 floats) and its DNN output (1 float) as d_x_halo /
 * d_u_halo (device pointers; both NULL on the shard that starts the series) -- a one-row halo, no collective.
 * Outputs: the three tuple elements (f_T = T_out - T_pred, T_pred, T_out), float[n_rows] each.  Reads lambda_T1..T4. */
int pinn_net_f_t(const float* d_x, const float* d_u, const float* d_x_halo, const float* d_u_halo, const pinn_affine_t* aff,
                 const float* d_lambda, long long n_rows, float* d_f, float* d_t_pred, float* d_t_real, void* stream);

/* ---- the network ------------------------------------------------------------------------
 * Architecture [n_in=8, hidden x n_hidden, 1] + variance head hidden -> hidden/2 -> hidden/4 -> 1
 * (01:389-438).  Parameters live in ONE flat float32 device buffer in state_dict order,
 * each tensor in torch layout [out, in] row-major:
 *   W_0 b_0 ... W_{h-1} b_{h-1}  W_p b_p  Wv_0 bv_0  Wv_1 bv_1  Wv_2 bv_2
 * The fused kernels support hidden in {128, 256} (hidden % 128 == 0, <= 256), 1 <= n_hidden <= 8.
 */
#define PINN_PREC_FP32 0  /* exact fp32 matrix math (v_mfma_f32_*_f32); parity with the reference at fp32 tolerance */
#define PINN_PREC_BF16 1  /* bf16 MFMA inputs, fp32 accumulate / activations / loss / master weights (rtol ~2e-2)    */
#define PINN_PREC_F32X6 2 /* fp32-ACCURATE matrix math on the 16-bit matrix cores from split operands, fp32 accumulation; same
                             tolerances as PINN_PREC_FP32, 2-3x faster.  Every product -- forward,
                             MC-dropout, backward chain, weight gradients -- from two fp16 parts per operand, three MFMAs
                             (gradients under exact power-of-two scales: per row in the backward chain, one per call in the
                             weight gradients); gradient tensors come out as close to a float64 autograd as torch's own fp32
                             autograd does.  Wide nets (layer-by-layer kernels): the same schemes.  What the Python
                             surface uses by default.  (The name is round 1's, when every product took six MFMAs.) */
#define PINN_PREC_F32X6_G6 3 /* as F32X6 with the gradients (backward chain, weight gradients) from three bf16 parts per operand,
                                six MFMAs per product: 24-bit operands and fp32's full exponent range for every element
                                (F32X6's weight gradients keep full relative precision down to 2^-29 of the call's largest
                                d pre-activation, an absolute 2^-36 of it below).  ~20 % slower; the conservative choice. */

typedef struct pinn_net {
  int n_in;       /* 8 */
  int hidden;     /* H */
  int n_hidden;   /* number of H-wide hidden layers (3 in the reference, 01:2139) */
  int precision;  /* PINN_PREC_* */
  void* d_packed; /* every precision but PINN_PREC_FP32: device scratch of pinn_packed_bytes(net) bytes; every call re-packs the
                     bf16 weight copies from d_params into it (stateless), NULL for fp32 */
} pinn_net_t;

size_t pinn_packed_bytes(const pinn_net_t* net);             /* 0 for fp32 / unsupported shapes */

/* Domain of PINN_PREC_F32X6 / _G6 (the fp32 reference has no such limit; PINN_PREC_FP32 and _BF16 neither): the matrix
 * operands are fp16 parts under fixed power-of-two scales, so every weight of the hidden x hidden and variance-head
 * matrices must satisfy |w| < 1023.5 (fp16(64 w) finite), and the row-normalised gradients of PINN_PREC_F32X6's backward
 * chain must stay below 65504 (they do while 16 x the column abs-sums of those matrices do).  Outside it the kernels
 * produce inf / NaN -- never silently wrong finite numbers -- and record the fact in d_packed: every call that packs the
 * weights rewrites the record, pinn_mlp_train_grads adds its gradient check.  This query reads it back: it WAITS for
 * `stream` (the one entry point that synchronises) and returns PINN_OK, PINN_E_RANGE, or an argument / HIP error.
 * Call it where the host synchronises anyway (logging, fetching results); on PINN_E_RANGE switch the net to PINN_PREC_FP32. */
int pinn_net_range_status(const pinn_net_t* net, void* stream);

long long pinn_param_count(const pinn_net_t* net);          /* floats in the flat buffer, <0 on error */

/* dropout source */
enum { PINN_DROP_NONE = 0,   /* eval mode: identity                                              */
       PINN_DROP_PHILOX = 1, /* on-chip Philox4x32-10, keyed (seed, stream, global row, layer, f) */
       PINN_DROP_BITS = 2 }; /* injected bit-packed keep-masks (parity tests)         */

typedef struct pinn_dropout {
  int mode;
  float p[9];                 /* drop probability of dropout module l (hidden 0..n_hidden-1, then var head) */
  unsigned long long seed;    /* PHILOX */
  unsigned stream;            /* PHILOX: optimizer step / first pass index */
  long long row_offset;       /* global index of local row 0 (data-parallel shards) */
  const unsigned* d_bits;     /* BITS: [n_passes][n_rows][words] uint32, bit f of module l at
                                 word offset l*(H/32) (+ f/32), words = n_hidden*H/32 + H/64 */
  unsigned* d_step_counter;   /* NULL, or (pinn_mlp_train_grads on the fused nets, hidden <= 256) a device counter of completed
                                 optimizer steps: the call draws PHILOX stream `stream + *d_step_counter` (BITS: pass
                                 *d_step_counter of d_bits) and adds 1 to the counter when its gradients are final -- so ONE
                                 captured launch sequence (a hipGraph) can be replayed step after step; see pinn_adam_step_dev */
} pinn_dropout_t;

/* DNN.forward (01:421-438): d_u, d_logvar [n_rows]. */
int pinn_mlp_forward(const pinn_net_t* net, const float* d_params, const float* d_x, long long n_rows,
                     const pinn_dropout_t* drop, float* d_u, float* d_logvar, void* stream);

/* get_MC_samples (01:1413-1491) as one persistent launch: 1 eval pass + T stochastic passes
 * per row tile, reduced on chip.  d_pred_mean, d_a_u, d_e_u [n_rows] (normalised units). */
int pinn_mc_dropout(const pinn_net_t* net, const float* d_params, const float* d_x, long long n_rows,
                    const pinn_dropout_t* drop, int n_passes,
                    float* d_pred_mean, float* d_a_u, float* d_e_u, void* stream);

/* train_dnn forward + aleatoric_loss + backward (01:949-953) on a row shard.
 *   d_grads  [pinn_param_count] : SUM over local rows of d(loss_row)/dparam, already divided by
 *            n_global (so an all-reduce(SUM) over shards gives the full-batch gradient)
 *   d_loss   double[4]: sum_rows nll term, sum |logvar|, sum (y-u)^2, (spare) -- raw sums
 *   workspace from pinn_train_workspace_bytes(net, n_rows); d_grads and d_work 16-byte aligned (PINN_E_ARG otherwise)
 */
size_t pinn_train_workspace_bytes(const pinn_net_t* net, long long n_rows);
int pinn_mlp_train_grads(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                         long long n_rows, long long n_global, const pinn_dropout_t* drop,
                         float* d_grads, double* d_loss, void* d_work, size_t work_bytes, void* stream);

/* The same call restricted to a subset of its kernel launches, so a benchmark can bracket one
 * kernel with events (bench.py's roofline leg).  phases = PINN_PHASE_ALL is pinn_mlp_train_grads. */
#define PINN_PHASE_CHAIN 1u   /* forward + loss + backward chain kernel (writes the activation stash) */
#define PINN_PHASE_WGRAD 2u   /* the per-layer weight-gradient kernels (read the stash)              */
#define PINN_PHASE_REDUCE 4u  /* fixed-order slab reduction -> d_grads, d_loss                       */
#define PINN_PHASE_ALL 7u
/* PINN_PREC_F32X6 / _G6 on the fused nets (hidden <= 256): the chain is two kernels, forward (+ loss) and backward; either
 * alone (the other precisions and the wide nets treat these two bits like PINN_PHASE_CHAIN) */
#define PINN_PHASE_CHAIN_FWD 8u
#define PINN_PHASE_CHAIN_BWD 16u
/* Two-part form of the weight-gradient and reduction phases, for overlapping the data-parallel all-reduce with the rest of
 * the step: the flat gradient splits at pinn_grad_split(net) floats into a HEAD [0, split) -- the input layer and every hidden
 * layer but the last -- and a TAIL [split, n) -- the last hidden layer, the predict head and the variance head, whose d
 * pre-activations the backward chain finishes first.  _TAIL / _HEAD run the weight-gradient kernels (the slab reduction) of
 * that part only; PINN_PHASE_WGRAD / _REDUCE are both parts.  pinn_grad_split is 0 (everything is "tail") where the
 * precision's kernels do not split (PINN_PREC_BF16 on the fused nets). */
#define PINN_PHASE_WGRAD_TAIL 32u
#define PINN_PHASE_WGRAD_HEAD 64u
#define PINN_PHASE_REDUCE_TAIL 128u
#define PINN_PHASE_REDUCE_HEAD 256u
long long pinn_grad_split(const pinn_net_t* net);
int pinn_mlp_train_grads_phases(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                                long long n_rows, long long n_global, const pinn_dropout_t* drop,
                                float* d_grads, double* d_loss, void* d_work, size_t work_bytes, void* stream,
                                unsigned phases);

/* torch.optim.Adam defaults (01:939): flat vectors of n floats; step is 1-based. */
int pinn_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, long long n,
                   float lr, int step, void* stream);

/* The same step with its two scalars read on the DEVICE, for a captured (hipGraph) training step that is replayed with
 * nothing but device state changing (train_dnn at the reference's data sizes is launch-bound: 01:939-955, 12 002 steps of
 * ~1e4 rows).  d_coeffs: float[2 * n_steps], entry k = the coefficients of step k + 1 as pinn_adam_coeffs gives them (the host
 * arithmetic of pinn_adam_step, so both paths are bit-identical); d_step_counter: the counter pinn_mlp_train_grads advanced
 * (pinn_dropout_t.d_step_counter): this call applies entry *d_step_counter - 1. */
void pinn_adam_coeffs(float lr, int step, float* step_size, float* bc2_sqrt);      /* host only */
int pinn_adam_step_dev(float* d_params, const float* d_grads, float* d_m, float* d_v, long long n,
                       const float* d_coeffs, const unsigned* d_step_counter, void* stream);

/* pinn_mlp_train_grads + pinn_adam_step_dev as one launch sequence with the optimizer step applied by the gradient reduction's
 * own launch: one full train_dnn step (01:949-954) with nothing but device state changing, for capture and replay.  Arguments
 * as for the two calls (drop->d_step_counter is required; d_params is updated in place; d_grads receives the gradients the
 * step applied).  Bit-identical to the two calls.  PINN_PREC_F32X6 / _G6 on the fused nets (hidden <= 256), PINN_E_ARCH otherwise. */
int pinn_mlp_train_step_dev(const pinn_net_t* net, float* d_params, const float* d_x, const float* d_y,
                            long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                            double* d_loss, void* d_work, size_t work_bytes, float* d_m, float* d_v,
                            const float* d_coeffs, void* stream);
/* The same with the step's two scalars by value (pinn_adam_step's lr and 1-based step): pinn_mlp_train_grads + pinn_adam_step
 * as one launch sequence for callers that launch step by step.  Bit-identical to the two calls.  Every precision and width. */
int pinn_mlp_train_step(const pinn_net_t* net, float* d_params, const float* d_x, const float* d_y,
                        long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                        double* d_loss, void* d_work, size_t work_bytes, float* d_m, float* d_v, float lr, int step,
                        void* stream);

/* ---- results assembly: create_comprehensive_results_array_v2 (01:1877-2010) -----------------------------------
 * Fills d_out = float64 [n_rows, 22] row-major (the `comprehensive_results` layout scripts 02-05 read):
 *   0-7 inputs and 8 target, de-normalised like sklearn's inverse_transform on float32 (aff->x_*, aff->y_*; 01:1916-1917);
 *   9 = (pred_mean - mc_min) / (mc_scale + 1e-12), 10 / 11 = a_u, e_u / (mc_scale + 1e-12) (float64, 01:1925-1936),
 *   each smoothed by a centred moving average of `window` rows with pandas' even-window semantics, separately inside
 *   every segment [d_seg_end[k-1], d_seg_end[k]) (ascending exclusive ends, the last == n_rows; n_segments == 0: one
 *   segment; 01:1830-1872, 01:1971-1986); 12 = col 8 - col 9; 13-16 = f_V, f_T, f_H2, f_O2; 17 = d_labels (NULL: 0);
 *   18-21 = 5*V_est, T_pred, H2 and O2 excess ratios -- taken from d_cols as pinn_residuals(PINN_RES_ALL) wrote them.
 * d_pred_mean / d_a_u / d_e_u: pinn_mc_dropout's outputs.  1 <= window <= 1024; d_x and d_out 16-byte aligned. */
int pinn_results_assemble(const float* d_x, const float* d_y, const pinn_affine_t* aff, double mc_min, double mc_scale,
                          int window, const long long* d_seg_end, int n_segments, const float* d_pred_mean,
                          const float* d_a_u, const float* d_e_u, const float* d_cols, long long ld,
                          const float* d_labels, long long n_rows, double* d_out, void* stream);

int pinn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PINN_HIP_H */
