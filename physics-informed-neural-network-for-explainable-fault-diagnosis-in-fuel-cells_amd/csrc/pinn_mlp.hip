// pinn_mlp.hip -- DNN.forward (01:421-438) and get_MC_samples (01:1413-1491) for gfx950.
//
// K2 mlp_forward_kernel : eval / stochastic forward, persistent over 128-row tiles.
// K3 mc_dropout_kernel  : 1 eval pass + T stochastic passes per tile inside one persistent
//                         launch; running sums of u, u^2, logvar stay in registers, masks come
//                         from on-chip Philox keyed by (seed, pass, global row, layer, feature);
//                         HBM traffic is 32 B in + 12 B out per row regardless of T.
// Both are MFMA-bound (v_mfma_f32_16x16x4_f32, exact fp32); see pinn_mlp_core.h for the layout.
// 64-row workgroup tiles, two workgroups per CU (persistent grid = 2 x #CU).
#include "pinn_mlp_core.h"

namespace pinn {

template <int H, bool MC, bool kBits>
__global__ __launch_bounds__(kThreads, 2) void mlp_kernel(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds_w[2 * kChunkBytes];
  __shared__ ChunkDesc tab[kMaxChunks];
  __shared__ __attribute__((aligned(16))) float small[kMaxSmall];
  ParamLayout L{a.H, a.nh};
  const int n_chunks = (a.nh - 1) * (H / 32) + H / 32 + H / 64;
  if (threadIdx.x == 0) build_forward_chunks(tab, L, 0);
  load_small_params(small, a.params, L);
  Pipe pipe;
  pipe.params = a.params; pipe.tab = tab; pipe.lds = lds_w; pipe.n = n_chunks;
  pipe.prime();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long lrow = tile * kTileRows + wave * kWaveRows + (lane & 15);
    const bool valid = lrow < a.n_rows;
    const long long srow = valid ? lrow : a.n_rows - 1;
    const long long grow = a.drop.row_offset + lrow;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2];
    const f32x4 xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    RowCtx c{lane, lane >> 4, grow, srow, a.n_rows, 0u, a.drop.mode};
    const StashPtrs st{};
    f32x4 v2[H / 64];
    if (!MC) {
      float u, z;
      forward_pass<H, false, kBits>(a.params, small, L, pipe, a.drop, c, xa, xb, st, u, z, v2);
      if (valid && lane < 16) {
        a.o0[lrow] = u;
        a.o1[lrow] = logf(softplus_f32(z) + 1e-6f);
      }
    } else {
      // pass -1: eval (dropout off) -> pred_mean (01:1442-1445, 1480); passes 0..T-1 stochastic
      float u_eval = 0.f, mean = 0.f, m2 = 0.f, sl = 0.f;
#pragma unroll 1
      for (int t = -1; t < a.n_passes; ++t) {
        c.mode = (t < 0) ? PINN_DROP_NONE : a.drop.mode;
        c.pass = (unsigned)(t < 0 ? 0 : t);
        float u, z;
        forward_pass<H, false, kBits>(a.params, small, L, pipe, a.drop, c, xa, xb, st, u, z, v2);
        if (t < 0) {
          u_eval = u;
        } else {
          welford_update(mean, m2, u - u_eval, 1.0f / (float)(t + 1));      // population variance of the passes: m2 / T
          sl += logf(softplus_f32(z) + 1e-6f);
        }
      }
      if (valid && lane < 16) {
        const float inv_t = 1.0f / (float)a.n_passes;
        const float var = m2 * inv_t;
        a.o0[lrow] = u_eval;
        a.o1[lrow] = expf(0.5f * (sl * inv_t));   // sqrt(exp(mean_t logvar_t)), 01:1483
        a.o2[lrow] = sqrtf(var);                  // population std over passes, 01:1486
      }
    }
  }
}

static int convert_drop(const pinn_net_t* net, const pinn_dropout_t* in, DropDev* out) {
  out->mode = PINN_DROP_NONE;
  out->bits = nullptr; out->words = 0; out->nb = net->hidden / 32;
  out->seed_lo = out->seed_hi = 0; out->stream = 0; out->row_offset = 0; out->step_counter = nullptr;
  for (int l = 0; l < kMaxDrop; ++l) { out->thr[l] = 0; out->scale[l] = 1.0f; }
  if (!in) return PINN_OK;
  if (in->mode < PINN_DROP_NONE || in->mode > PINN_DROP_BITS) return PINN_E_ARG;
  out->mode = in->mode;
  out->row_offset = in->row_offset;
  if (in->mode == PINN_DROP_NONE) return PINN_OK;
  for (int l = 0; l <= net->n_hidden; ++l) {
    const float p = in->p[l];
    if (!(p >= 0.0f && p < 1.0f)) return PINN_E_ARG;
    double t = floor((double)p * 65536.0 + 0.5);
    if (p > 0.0f && t < 1.0) t = 1.0;      // a positive p never rounds to "no dropout"
    out->thr[l] = (unsigned)(t < 0 ? 0 : (t > 65536.0 ? 65536.0 : t));
    out->scale[l] = 1.0f / (float)(1.0 - (double)p);
  }
  out->seed_lo = (unsigned)(in->seed & 0xFFFFFFFFull);
  out->seed_hi = (unsigned)(in->seed >> 32);
  out->stream = in->stream;
  if (in->mode == PINN_DROP_BITS) {
    if (!in->d_bits) return PINN_E_ARG;
    out->bits = in->d_bits;
    out->words = net->n_hidden * (net->hidden / 32) + net->hidden / 64;
  }
  return PINN_OK;
}

static int check_net(const pinn_net_t* net) {
  if (!net) return PINN_E_ARG;
  if (net->n_in != 8) return PINN_E_ARCH;
  const bool wide = net->hidden == 512 || net->hidden == 1024 || net->hidden == 2048;    // layer-by-layer kernels (pinn_wide.hip)
  if (net->hidden != 128 && net->hidden != 256 && !wide) return PINN_E_ARCH;
  if (net->n_hidden < 1 || net->n_hidden > 8) return PINN_E_ARCH;
  if (net->precision < PINN_PREC_FP32 || net->precision > PINN_PREC_F32X6_G6) return PINN_E_ARG;
  if (wide && net->precision == PINN_PREC_FP32) return PINN_E_ARCH;                      // split-operand or bf16 arithmetic only
  if (net->precision != PINN_PREC_FP32 && !net->d_packed) return PINN_E_ARG;
  return PINN_OK;
}

static int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

int launch_forward_bf16(const pinn_net_t* net, const FwdArgs& a, bool mc, void* stream);   // pinn_bf16.hip
int launch_forward_x6(const pinn_net_t* net, const FwdArgs& a, bool mc, void* stream);     // pinn_x6.hip
int launch_forward_wide(const pinn_net_t* net, const FwdArgs& a, bool mc, void* stream);   // pinn_wide.hip

template <bool MC>
static int launch(const pinn_net_t* net, const FwdArgs& a, void* stream) {
  const long long n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  if (n_tiles == 0) return PINN_OK;
  if (net->hidden > 256) return launch_forward_wide(net, a, MC, stream);
  if (net->precision == PINN_PREC_BF16) return launch_forward_bf16(net, a, MC, stream);
  if (net->precision >= PINN_PREC_F32X6) return launch_forward_x6(net, a, MC, stream);
  const int grid = (int)(n_tiles < 2 * num_cus() ? n_tiles : 2 * num_cus());
  (void)hipGetLastError();   // drop a stale error left by another HIP user of this thread
  const bool bits = a.drop.mode == PINN_DROP_BITS;
  if (net->hidden == 256) {
    if (bits) hipLaunchKernelGGL((mlp_kernel<256, MC, true>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_kernel<256, MC, false>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  } else {
    if (bits) hipLaunchKernelGGL((mlp_kernel<128, MC, true>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_kernel<128, MC, false>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

}  // namespace pinn

using namespace pinn;

extern "C" long long pinn_param_count(const pinn_net_t* net) {
  // the shape alone decides (precision / d_packed are not needed to size the parameter buffer)
  if (!net) return PINN_E_ARG;
  const bool wide = net->hidden == 512 || net->hidden == 1024 || net->hidden == 2048;
  if (net->n_in != 8 || (net->hidden != 128 && net->hidden != 256 && !wide) || net->n_hidden < 1 || net->n_hidden > 8) return PINN_E_ARCH;
  ParamLayout L{net->hidden, net->n_hidden};
  return L.total();
}

extern "C" int pinn_mlp_forward(const pinn_net_t* net, const float* d_params, const float* d_x, long long n_rows,
                                const pinn_dropout_t* drop, float* d_u, float* d_logvar, void* stream) {
  int rc = check_net(net);
  if (rc) return rc;
  if (n_rows < 0 || !d_params) return PINN_E_ARG;
  if (n_rows == 0) return PINN_OK;
  if (!d_x || !d_u || !d_logvar) return PINN_E_ARG;
  FwdArgs a{};
  a.params = d_params; a.x = d_x; a.n_rows = n_rows; a.H = net->hidden; a.nh = net->n_hidden;
  rc = convert_drop(net, drop, &a.drop);
  if (rc) return rc;
  a.n_passes = 1; a.o0 = d_u; a.o1 = d_logvar; a.o2 = nullptr;
  return launch<false>(net, a, stream);
}

extern "C" int pinn_mc_dropout(const pinn_net_t* net, const float* d_params, const float* d_x, long long n_rows,
                               const pinn_dropout_t* drop, int n_passes, float* d_pred_mean, float* d_a_u, float* d_e_u,
                               void* stream) {
  int rc = check_net(net);
  if (rc) return rc;
  if (n_rows < 0 || !d_params || n_passes < 1 || !drop) return PINN_E_ARG;
  if (n_rows == 0) return PINN_OK;
  if (!d_x || !d_pred_mean || !d_a_u || !d_e_u) return PINN_E_ARG;
  if (drop->mode == PINN_DROP_NONE) return PINN_E_ARG;
  FwdArgs a{};
  a.params = d_params; a.x = d_x; a.n_rows = n_rows; a.H = net->hidden; a.nh = net->n_hidden;
  rc = convert_drop(net, drop, &a.drop);
  if (rc) return rc;
  a.n_passes = n_passes; a.o0 = d_pred_mean; a.o1 = d_a_u; a.o2 = d_e_u;
  return launch<true>(net, a, stream);
}

extern "C" int pinn_abi_version(void) { return PINN_ABI_VERSION; }
