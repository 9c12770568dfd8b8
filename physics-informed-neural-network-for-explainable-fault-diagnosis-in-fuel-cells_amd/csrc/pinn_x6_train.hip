// pinn_x6_train.hip -- train_dnn's forward + aleatoric_loss + backward chain (01:949-953) with fp32-accurate
// matrix products on the matrix cores (pinn_x6_core.h).  PINN_PREC_F32X6: forward and backward in scheme X3 (two fp16
// parts, three MFMAs per product), activations and d pre-activations stashed PACKED -- as the fp16 fragments the chain's
// own MFMAs read -- for the backward kernel and the packed weight-gradient kernels (pinn_x6_wgrad.hip).
// PINN_PREC_F32X6_G6: forward X3, backward x6 (three bf16 parts, six MFMAs), fp32 stash in train_chain_kernel's layout
// (pinn_train.hip) and the split-in-registers weight-gradient kernels.  The finalize kernel is shared by all.
#include <cstdlib>
#include "pinn_x6_core.h"

namespace pinn {
namespace x6 {

constexpr int kLossTermsX = 8;   // nll, |logvar|, (y-u)^2, du, dz, spare... (= kLossTerms of pinn_train.hip)

struct TrainArgsX {
  const float* params;
  const float* x;
  const float* y;
  long long n_rows, n_global;
  int H, nh;
  DropDev drop;
  TrainBuffers b;
};

// Two kernels.  The forward half (activations stashed, aleatoric loss, d loss / d (u, z)) runs in scheme X3 -- two fp16
// parts, three MFMAs per product: activations and weights fit fp16's range; the backward half needs bf16's range for the
// gradients (1 / N of the loss in front, precisions up to 1e6) and runs in x6 on the transposed copies.  The seam carries
// nothing in registers: du, dz go through their [rows] buffers (the weight-gradient kernels read them anyway), the
// tanh'ed last variance blocks through the stash.
// WAVES: 8 = 128-row tiles, two waves per SIMD; 4 = 64-row tiles, one wave per SIMD (small row counts, see mlp_x6_kernel)
template <int H, bool kBits, int WAVES, bool kPack>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void train_fwd_x3_kernel(TrainArgsX a, const __bf16* packed) {
  using S = X3;
  constexpr int kThreadsX = WAVES * 64, kTileRowsX = WAVES * 16;
  constexpr int kSmallBytes = kMaxSmall * 4, kW0Bytes = 8 * kW0Stride * 4, kRedBytes = 8 * kLossTermsX * 8;
  constexpr int kSlabAt = (kSmallBytes + kW0Bytes + kRedBytes + 1023) & ~1023;
  __shared__ __attribute__((aligned(1024))) char smem[kSlabAt + 2 * S::Pipe::kSlab];
  float* small = reinterpret_cast<float*>(smem);
  float* w0t = reinterpret_cast<float*>(smem + kSmallBytes);
  double (*red)[kLossTermsX] = reinterpret_cast<double (*)[kLossTermsX]>(smem + kSmallBytes + kW0Bytes);
  char* lds_w = smem + kSlabAt;
  ParamLayout L{a.H, a.nh};
  PackLayout K{a.H, a.nh};
#ifdef PINN_SMALLN_WARM
  if constexpr (WAVES == 4) l2_warm<kThreadsX>(packed + 3 * K.total(), (unsigned)(K.total() * 4), lds_w + S::Pipe::kSlab);
#endif
  fill_small<S, kThreadsX>(small, w0t, a.params, L);
  S::Pipe pipe;
  pipe.lds = lds_w;
  pipe.init(packed + 3 * K.total(), (unsigned)(K.total() * 2), threadIdx.x);      // the fp16 copies, behind the three bf16 ones
  pipe.template prime<clog2(H), WAVES>(first_mat<H>(K));

  const int lane = threadIdx.x & 63, wave = pipe.wave, kq = lane >> 4;
  const float inv_n = (float)(1.0 / (double)a.n_global);
  float s_nll = 0.f, s_abs = 0.f, s_mse = 0.f, s_du = 0.f, s_dz = 0.f;
  float gmax = 0.f;        // kPack: max over this lane's rows of max(|du|, |dz|) -> TrainBuffers::emax (the row scales' reference)
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.b.amax = 0u;      // the backward kernel (next in the stream) takes its atomicMax from 0

  const long long n_tiles = (a.n_rows + 127) / 128 * (128 / kTileRowsX);      // whole 128-row stash tiles
  const unsigned pass0 = train_pass(a.drop);      // 0, or the device's step counter (replayed graphs)
  // ... which the reduction advances while its own workgroups still need the old value (pinn_mlp_train_step_dev): a copy
  if (blockIdx.x == 0 && threadIdx.x == 0) a.b.amax[2] = pass0;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t16 = tile * WAVES + wave;
    const long long lrow = t16 * 16 + (lane & 15);
    const bool valid = lrow < a.n_rows;
    const long long srow = valid ? lrow : a.n_rows - 1;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2];
    const f32x4 xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    const float yv = a.y[srow];
    const RowCtx c{lane, kq, a.drop.row_offset + lrow, srow, a.n_rows, pass0, a.drop.mode};
    const StashX sx{(float*)a.b.stash_h, (float*)a.b.stash_v1, (float*)a.b.stash_v2, (float*)a.b.dpre_h, (float*)a.b.dpre_v1, (float*)a.b.dpre_v2,
                    a.b.t16, t16};

    if constexpr (kPack) {
      // the input rows as a packed group of their own (features 0 .. 7 of 32, the value x / 16: any |x| < 256 survives the
      // weight-gradient kernel's row scale in fp16): layer 0's weight gradient reads it like any other activation
      Frag2 fx;
      static_for<4>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const float v = kq == 0 ? xa[r] : (kq == 1 ? xb[r] : 0.0f);
        S::template split<r>(v * 0.0625f, 0.0f, fx);
      });
      float* xp = packed_ptr((float*)a.b.stash_x, t16, 32, lane);
      PINN_STASH_ST(reinterpret_cast<u32x4*>(xp), fx.hi);
      PINN_STASH_ST(reinterpret_cast<u32x4*>(xp + 256), fx.lo);
    }
    // ------------------------------------------------------------------ forward (activations stashed)
    float u, z;
    forward_pass<S, H, kBits, true, WAVES, kPack>(w0t, small, L, pipe, a.drop, c, xa, xb, u, z, &sx);

    // ------------------------------------------------------------------ aleatoric_loss (01:916-927) and its gradient
    float du = 0.f, dz = 0.f;
    {
      const float sp = softplus_f32(z);
      const float var = sp + 1e-6f;
      const float s = logf(var);                 // logvar
      const float prec = expf(-s);               // precision = exp(-logvar), 01:919
      const float e = yv - u;
      if (valid) {
        du = -(prec * e) * inv_n;
        const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
        const float ds = (-0.5f * prec * e * e + 0.5f + 0.01f * sgn) * inv_n;
        // d logvar / dz = softplus'(z) / (softplus(z) + 1e-6); torch: softplus' = 1 above threshold 20
        const float sig = z > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-z));
        dz = ds * sig / var;
        if (kq == 0) {
          s_nll += 0.5f * prec * e * e + 0.5f * s;
          s_abs += fabsf(s);
          s_mse += e * e;
          s_du += du;
          s_dz += dz;
        }
      }
      if (lane < 16) { a.b.du[t16 * 16 + lane] = du; a.b.dz[t16 * 16 + lane] = dz; }
      gmax = fmaxf(gmax, fmaxf(fabsf(du), fabsf(dz)));          // (fmaxf drops a NaN: it reaches the gradients through du, dz themselves)
    }
  }
  if constexpr (kPack) {      // one atomicMax per wave on the float's bits: order-independent, so bitwise reproducible
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
    if (lane == 0) atomicMax(a.b.emax, __float_as_uint(gmax));
  }

  // ---------------------------------------------------------------------- loss partial sums of this workgroup
  float terms[5] = {s_nll, s_abs, s_mse, s_du, s_dz};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double v = (double)terms[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLossTermsX) {
    double t = 0.0;
    if (threadIdx.x < 5)
      for (int w = 0; w < WAVES; ++w) t += red[w][threadIdx.x];
    a.b.loss_part[(long long)blockIdx.x * kLossTermsX + threadIdx.x] = t;
  }
}

// backward chain of every row tile: d pre-activations of all layers to the stash, on the transposed copies.
// S = X3 (PINN_PREC_F32X6): two fp16 parts of per-row-normalised gradients, three MFMAs per product (backward_pass); it also
// records the call's largest |d pre-activation| (TrainBuffers::amax), the common scale of the fp16 weight-gradient kernels.
// S = X6 (PINN_PREC_F32X6_G6): three bf16 parts of every operand, six MFMAs.  Against a float64 autograd both leave the
// gradient tensors as close as torch's own fp32 autograd does (rms error 5e-8 .. 1.2e-7 of a tensor's rms); the
// golden 3-step Adam trajectory passes with either.
template <typename S, int H, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void train_bwd_kernel(TrainArgsX a, const __bf16* packed) {
  constexpr int kThreadsX = WAVES * 64, kTileRowsX = WAVES * 16;
  constexpr int kSmallBytes = kMaxSmall * 4, kW0Bytes = 8 * kW0Stride * 4, kRingBytes = 8 * 2 * 2048;
  constexpr int kSlabAt = (kSmallBytes + kW0Bytes + kRingBytes + 1023) & ~1023;
  __shared__ __attribute__((aligned(1024))) char smem[kSlabAt + 2 * S::Pipe::kSlab];
  float* small = reinterpret_cast<float*>(smem);
  float* w0t = reinterpret_cast<float*>(smem + kSmallBytes);                 // (filled, unused: one LDS fill routine)
  char* ring_lds = smem + kSmallBytes + kW0Bytes;
  char* lds_w = smem + kSlabAt;
  ParamLayout L{a.H, a.nh};
  PackLayout K{a.H, a.nh};
#ifdef PINN_SMALLN_WARM
  if constexpr (WAVES == 4 && S::kCopies == 2) l2_warm<kThreadsX>(packed + 3 * K.total(), (unsigned)(K.total() * 4), lds_w + S::Pipe::kSlab);
#endif
  fill_small<X6, kThreadsX>(small, w0t, a.params, L);                        // (the backward pass reads head vectors only: no scaled biases)
  typename S::Pipe pipe;
  pipe.lds = lds_w;
  pipe.init(packed + (S::kCopies == 2 ? 3 * K.total() : 0), (unsigned)(K.total() * 2), threadIdx.x);
  constexpr int KPT1 = clog2((H / 4 + 63) & ~63);
  pipe.template prime<KPT1, WAVES>(Mat{(unsigned)K.wv1t(), clog2(H / 32)});   // the backward sequence starts with Wv1^T

  const int lane = threadIdx.x & 63, wave = pipe.wave;
  const StashRing ring{ring_lds + wave * 4096, lane};
  const long long n_tiles = (a.n_rows + 127) / 128 * (128 / kTileRowsX);
  float amax = 0.0f;       // X3: max |d pre-activation| this lane has stashed (true units)
  // X3 (packed stash): the call's largest row exponent E, complete since the forward kernel ended (grad_exponent: 4 if no row has a gradient)
  int E = 4;
  if constexpr (S::kCopies == 2) E = grad_exponent(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*a.b.emax)), 4);
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t16 = tile * WAVES + wave;
    const StashX sx{(float*)a.b.stash_h, (float*)a.b.stash_v1, (float*)a.b.stash_v2, (float*)a.b.dpre_h, (float*)a.b.dpre_v1, (float*)a.b.dpre_v2,
                    a.b.t16, t16};
    const float du = a.b.du[t16 * 16 + (lane & 15)], dz = a.b.dz[t16 * 16 + (lane & 15)];
    const RowMeta meta{S::kCopies == 2 ? (_Float16*)a.b.rowmeta + t16 * 128 : nullptr, E, a.b.qboost};
    backward_pass<S, H, WAVES>(small, L, pipe, a.drop, a.drop.mode, sx, ring, lane, du, dz, amax, meta);
  }
  if constexpr (S::kActScale != 1.0f) {
    // the call's common scale for the fp16 weight-gradient kernels: one atomicMax per wave on the float's bits (non-negative
    // floats order like unsigned integers, so the result does not depend on the order of the waves; fmaxf drops NaNs -- a NaN
    // gradient still reaches the stash and, through the operands, the weight gradients)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) atomicMax(a.b.amax, __float_as_uint(amax));
  }
}

void launch_pack_x6(const pinn_net_t* net, const float* d_params, hipStream_t st, unsigned* zero_word = nullptr);   // pinn_x6.hip

}  // namespace x6

// chain phase of pinn_mlp_train_grads for PINN_PREC_F32X6; *grid_out = workgroups (= loss partials)
int launch_train_chain_x6(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y, long long n_rows,
                          long long n_global, const DropDev& drop, const TrainBuffers& b, unsigned which, int* grid_out, void* stream) {
  using namespace x6;
  hipStream_t st = (hipStream_t)stream;
  const bool fast_bwd = net->precision == PINN_PREC_F32X6;           // backward chain in scheme X3 + packed stash (PINN_PREC_F32X6_G6: x6, fp32 stash)
  const bool run_fwd = which & 1u, fwd_only = !(which & 2u);
  launch_pack_x6(net, d_params, st, fast_bwd && run_fwd ? b.emax : nullptr);      // (also zeroes the forward kernel's row-gradient maximum)
  TrainArgsX a{};
  a.params = d_params; a.x = d_x; a.y = d_y; a.n_rows = n_rows; a.n_global = n_global; a.H = net->hidden; a.nh = net->n_hidden;
  a.drop = drop; a.b = b;
  const int cus = cu_count_cached();
  // the stash is padded to whole 128-row tiles: the 64-row kernel covers them too (two tiles each)
  const long long t128 = (n_rows + 127) / 128;
#ifdef PINN_DEBUG_HOOKS
  static const bool force8 = getenv("PINN_X6_WAVES8") != nullptr;     // measurement builds only: always the 8-wave kernels
#else
  constexpr bool force8 = false;
#endif
  const bool small_n = !force8 && 2 * t128 <= cus;           // 64-row tiles still fit one per CU
  const long long n_tiles = small_n ? 2 * t128 : t128;
  const int grid = (int)(n_tiles < cus ? n_tiles : cus);
  *grid_out = grid;
  const __bf16* packed = (const __bf16*)net->d_packed;
  const bool bits = drop.mode == PINN_DROP_BITS;
  if (fast_bwd && !fwd_only && !run_fwd) {          // backward alone (per-kernel timing): no forward kernel has reset the maximum
    hipError_t em = hipMemsetAsync(b.amax, 0, sizeof(unsigned), st);
    if (em != hipSuccess) return (int)em;
  }
#define PINN_LAUNCH_T(HH, BB)                                                                                                   \
  do {                                                                                                                          \
    if (small_n) {                                                                                                              \
      if (run_fwd && fast_bwd) hipLaunchKernelGGL((train_fwd_x3_kernel<HH, BB, 4, true>), dim3(grid), dim3(256), 0, st, a, packed);   \
      else if (run_fwd) hipLaunchKernelGGL((train_fwd_x3_kernel<HH, BB, 4, false>), dim3(grid), dim3(256), 0, st, a, packed);   \
      if (!fwd_only && fast_bwd) hipLaunchKernelGGL((train_bwd_kernel<X3, HH, 4>), dim3(grid), dim3(256), 0, st, a, packed);    \
      else if (!fwd_only) hipLaunchKernelGGL((train_bwd_kernel<X6, HH, 4>), dim3(grid), dim3(256), 0, st, a, packed);           \
    } else {                                                                                                                    \
      if (run_fwd && fast_bwd) hipLaunchKernelGGL((train_fwd_x3_kernel<HH, BB, 8, true>), dim3(grid), dim3(512), 0, st, a, packed);   \
      else if (run_fwd) hipLaunchKernelGGL((train_fwd_x3_kernel<HH, BB, 8, false>), dim3(grid), dim3(512), 0, st, a, packed);   \
      if (!fwd_only && fast_bwd) hipLaunchKernelGGL((train_bwd_kernel<X3, HH, 8>), dim3(grid), dim3(512), 0, st, a, packed);    \
      else if (!fwd_only) hipLaunchKernelGGL((train_bwd_kernel<X6, HH, 8>), dim3(grid), dim3(512), 0, st, a, packed);           \
    }                                                                                                                           \
  } while (0)
  if (net->hidden == 256) { if (bits) PINN_LAUNCH_T(256, true); else PINN_LAUNCH_T(256, false); }
  else { if (bits) PINN_LAUNCH_T(128, true); else PINN_LAUNCH_T(128, false); }
#undef PINN_LAUNCH_T
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

}  // namespace pinn
