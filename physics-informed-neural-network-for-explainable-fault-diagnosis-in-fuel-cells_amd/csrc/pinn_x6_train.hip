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

// ---------------------------------------------------------------------------------------------------------------------
// Small row counts (the reference's own sizes: 1e3 .. 1.6e4 rows), H = 256, scheme X3 with the packed stash.
// One wave per 16 rows is the wrong shape there: fewer row tiles than CUs, and a lone wave issues a tile's whole instruction
// stream itself -- per 32-feature slab 48 MFMAs (768 cycles) but ~260 vector instructions of activation arithmetic on top
// (measured with the stamps of tools/x6_stamps.py at 4200 rows: 2257 cycles per slab step, 42 us per pass; without any
// weight DMA 36 us: not a memory wait).  Here a 16-row tile belongs to FOUR waves, each owning a quarter of every layer's
// output features: a quarter of the MFMAs and a quarter of the activation arithmetic per wave.  The next layer needs all
// features of a row as its B operand, so the quarters meet in LDS -- as the packed fragments the stash holds anyway (one
// 16-B store per part to LDS, one to HBM).  A workgroup is two row tiles x four feature quarters = 8 waves sharing one weight
// stream (the slab machinery of the big kernels: layer_x6 with four output blocks per wave and PipeT::read_off).
// Same arithmetic per element as the big kernels (scales, Philox counters, kept-zero rule, stash layout): the backward
// kernels and the weight-gradient kernels cannot tell which forward kernel ran.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef PINN_ABL_Q_NTSTORE
#define PINN_Q_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define PINN_Q_ST(p, v) (*(p) = (v))
#endif
struct Xch {                 // a row tile's 32-feature groups as packed fragments in LDS: [group][part hi / lo][row][kq][16 B]
  char* base;
  int off;                   // (lane & 15) * 64 + (lane >> 4) * 16
  __device__ __forceinline__ u32x4 read(int g, int part) const { return *reinterpret_cast<const u32x4*>(base + g * 2048 + part * 1024 + off); }
  __device__ __forceinline__ void write(int g, const Frag2& f) const {
    *reinterpret_cast<u32x4*>(base + g * 2048 + off) = f.hi;
    *reinterpret_cast<u32x4*>(base + g * 2048 + 1024 + off) = f.lo;
  }
};
__device__ __forceinline__ void lds_barrier() {       // every wave's LDS writes visible; global stores stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// One 32-feature group of this lane's row: raw pre-activations v0, v1 (two 16-feature blocks, times the accumulator scale in
// `pre`) -> dropout(tanh) as the two fp16 parts of 8 x the activation: prep_micro's arithmetic, its six micro-steps at once.
// fp = the group's index inside its layer (the Philox call index).
template <bool kDot>
__device__ __forceinline__ void prep_group(Frag2& out, const f32x4& v0, const f32x4& v1, const DropDev& d, const RowCtx& c, const LayerDrop ld,
                                           float pre, int layer, int fp, const float* wp32, float& up) {
  unsigned keep = 0u;
  if (c.mode == PINN_DROP_BITS) {
    const unsigned word = d.bits[((long long)c.pass * c.n_rows + c.lrow) * d.words + layer * d.nb + fp];
    const unsigned lo = (word >> (4 * c.kq)) & 0xFu, hi = (word >> (16 + 4 * c.kq)) & 0xFu;
    keep = ld.thr == 0 ? 0xFFu : (lo | (hi << 4));
  } else {
    PrepBase s;
    s.w0 = (unsigned)c.grow; s.w1 = (unsigned)((unsigned long long)c.grow >> 32);
    s.w2 = ((unsigned)layer << 16) | ((unsigned)fp << 2) | (unsigned)c.kq; s.w3 = d.stream + c.pass;
    philox_rounds5<0>(s, d.seed_lo, d.seed_hi);
    philox_rounds5<5>(s, d.seed_lo, d.seed_hi);
    static_for<4>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      keep |= (keep_draw<0, r>(s, ld.thr) ? 1u : 0u) << r;
      keep |= (keep_draw<1, r>(s, ld.thr) ? 1u : 0u) << (4 + r);
    });
  }
  const float m2s = -2.0f * ld.scale;
  static_for<4>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    float a0 = fmaf(m2s, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v0[r] * pre) + 1.0f), ld.scale);
    float a1 = fmaf(m2s, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v1[r] * pre) + 1.0f), ld.scale);
    a0 = a0 == 0.0f ? 0x1p-24f : a0;          // the kept-zero rule of the packed stash (prep_micro)
    a1 = a1 == 0.0f ? 0x1p-24f : a1;
    const float hs0 = ((keep >> r) & 1u) ? a0 : 0.0f, hs1 = ((keep >> (4 + r)) & 1u) ? a1 : 0.0f;
    X3::split<r>(hs0, hs1, out);
    if constexpr (kDot) {
      const float h0 = hs0 * (1.0f / X3::kActScale), h1 = hs1 * (1.0f / X3::kActScale);
      up += fmaf(wp32[4 * c.kq + r], h0, wp32[16 + 4 * c.kq + r] * h1);
    }
  });
}

#ifdef PINN_Q_STAMP      // diagnostic build only (tools/q_stamps.py): cycle counts of the phases of workgroup 0, per wave
__device__ unsigned long long g_q_stamps[8 * 16];
__device__ __forceinline__ unsigned long long q_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PINN_QS(k) do { const unsigned long long t_ = q_now(); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_q_stamps[(threadIdx.x >> 6) * 16 + (k)] += t_ - q_last; q_last = t_; } while (0)
#define PINN_QS2(k, t0) do { const unsigned long long t_ = q_now(); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_q_stamps[(threadIdx.x >> 6) * 16 + (k)] += t_ - (t0); (t0) = t_; } while (0)
#else
#define PINN_QS(k) do { } while (0)
#define PINN_QS2(k, t0) do { } while (0)
#endif
// The weight stream of these kernels: THREE slabs in LDS, requested two steps ahead.  A step multiplies 12 MFMAs per wave
// (the big kernels: 48 plus the activation arithmetic), far less than the round trip of its LDS-DMA: with the big kernels'
// one-slab lookahead every step waited out that latency (first version of this kernel: 31 us per pass at 4200 rows against 36
// for one wave per tile).  The stream is a flat, cyclic list of slabs, so the lookahead runs across layers and tiles.
constexpr int kSmallThreads = 512, kQDepth = 3, kQPieces = 4;      // pieces per wave and slab: 32 / 8 waves (smaller matrices: fetched again)
template <int H, bool kBackward>
struct SlabSeq {
  static constexpr int NP = H / 32;
  PackLayout K;
  int nh, n;
  __device__ __forceinline__ SlabSeq(int H_, int nh_) : K{H_, nh_}, nh(nh_), n((nh_ - 1) * NP + (kBackward ? NP / 4 + NP / 2 : NP + NP / 2)) {}
  // slab i of a pass: its matrix (run-time row stride) and K-group
  __device__ __forceinline__ void get(int i, Mat& m, int& g) const {
    constexpr int KPW = clog2(H), KP2 = clog2((H / 2 + 63) & ~63), KP4 = clog2((H / 4 + 63) & ~63);
    if constexpr (!kBackward) {      // W_1 .. W_{nh-1} (NP groups each), Wv0 (NP), Wv1 (NP / 2)
      const int nhid = (nh - 1) * NP;
      if (i < nhid) { m = Mat{(unsigned)K.w(1 + i / NP), clog2(H / 16), KPW}; g = i % NP; }
      else if (i < nhid + NP) { m = Mat{(unsigned)K.wv0(), clog2(H / 32), KPW}; g = i - nhid; }
      else { m = Mat{(unsigned)K.wv1(), clog2(H / 64), KP2}; g = i - nhid - NP; }
    } else {                         // Wv1^T (K = H / 4: NP / 4 groups), Wv0^T (K = H / 2: NP / 2), W_{nh-1}^T .. W_1^T (NP each)
      if (i < NP / 4) { m = Mat{(unsigned)K.wv1t(), clog2(H / 32), KP4}; g = i; }
      else if (i < NP / 4 + NP / 2) { m = Mat{(unsigned)K.wv0t(), clog2(H / 16), KP2}; g = i - NP / 4; }
      else { const int j = i - NP / 4 - NP / 2; m = Mat{(unsigned)K.wt(nh - 1 - j / NP), clog2(H / 16), KPW}; g = j % NP; }
    }
  }
};
struct QPipe {
  X3::Pipe pipe;      // rsrc, copy_bytes, lds (3 slabs), wave, lane terms
  int par, si;        // buffer of the current slab; its index in the sequence
  __device__ __forceinline__ void piece(const Mat& m, int g, int j, int buf) {
    const int n = 2 << m.nrb_log;
    const int p = (pipe.wave + 8 * j) & (n - 1);
    const unsigned voff = (pipe.lane_row2 << 5) + pipe.lane_kq8;
    const int copy = p >> m.nrb_log, rb = p & ((1 << m.nrb_log) - 1);
    const unsigned soff = (unsigned)copy * pipe.copy_bytes + 2u * (m.off + (((unsigned)g << (m.nrb_log + 4)) << 5) + (unsigned)rb * 512u);
    char* dst = pipe.lds + buf * X3::Pipe::kSlab + copy * kCopyLds + rb * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(pipe.rsrc, (lptr_t)dst, 16, voff, soff, 0, 0);
  }
  template <typename Seq>
  __device__ __forceinline__ void prime(const Seq& seq) {
    Mat m; int g;
#pragma unroll
    for (int s = 0; s < kQDepth - 1; ++s) {
      seq.get(s, m, g);
#pragma unroll
      for (int j = 0; j < kQPieces; ++j) piece(m, g, j, s);
    }
    __syncthreads();
    par = 0; si = 0;
  }
  // One step: acc[t] += A_t x B for this wave's NTOUT row blocks (from byte read_off of the current slab); the slab two steps
  // ahead is requested between the MFMA groups; past the barrier the next slab is complete and the current one free.
  template <int NTOUT, typename Seq>
  __device__ __forceinline__ void step(f32x4 (&acc)[NTOUT], const Frag2& b, const Seq& seq, int read_off, int lane) {
    const int kq = lane >> 4, i = lane & 15;
    const char* base = pipe.lds + par * X3::Pipe::kSlab + read_off + i * 64 + ((kq ^ swz(i)) << 4);
    const unsigned addr = (unsigned)(unsigned long long)(lptr_t)base;
#ifdef PINN_Q_STAMP
    unsigned long long tq = q_now();
#endif
    X3::AFrag af[NTOUT];
    static_for<NTOUT>([&](auto tc) { X3::load<decltype(tc)::value>(af[decltype(tc)::value], addr); });
    int st = si + 2; st = st >= seq.n ? st - seq.n : st;
    Mat mt; int gt;
    seq.get(st, mt, gt);
    const int buf = par + 2 >= kQDepth ? par + 2 - kQDepth : par + 2;
    static_for<NTOUT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      X3::wait<2 * (NTOUT - 1 - t)>(af[t]);                 // LDS returns in order: the younger blocks' reads may still be in flight
      X3::mma(acc[t], af[t], b);
      __builtin_amdgcn_sched_barrier(0);
      static_for<(kQPieces + NTOUT - 1) / NTOUT>([&](auto jc) {
        constexpr int j = t * ((kQPieces + NTOUT - 1) / NTOUT) + decltype(jc)::value;
        if constexpr (j < kQPieces) piece(mt, gt, j, buf);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    PINN_QS2(13, tq);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kQPieces) : "memory");      // everything older than this step's pieces: the next slab
    PINN_QS2(14, tq);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PINN_QS2(15, tq);
    par = par + 1 == kQDepth ? 0 : par + 1;
    si = si + 1 == seq.n ? 0 : si + 1;
  }
};

template <int H>
__global__ __launch_bounds__(kSmallThreads, 1) void train_fwd_small_kernel(TrainArgsX a, const __bf16* packed) {
  using S = X3;
  using Frag = Frag2;
  static_assert(H == 256, "four feature quarters of 64");
  constexpr int NP = H / 32;
  constexpr int kSmallBytes = kMaxSmall * 4, kW0Bytes = 8 * kW0Stride * 4, kRedBytes = 8 * kLossTermsX * 8;
  constexpr int kXchBytes = 2 * NP * 2048, kHeadBytes = 2 * 4 * 16 * 2 * 4;
  constexpr int kSlabAt = (kSmallBytes + kW0Bytes + kRedBytes + kXchBytes + kHeadBytes + 1023) & ~1023;
  __shared__ __attribute__((aligned(1024))) char smem[kSlabAt + kQDepth * S::Pipe::kSlab];
  float* small = reinterpret_cast<float*>(smem);
  float* w0t = reinterpret_cast<float*>(smem + kSmallBytes);
  double (*red)[kLossTermsX] = reinterpret_cast<double (*)[kLossTermsX]>(smem + kSmallBytes + kW0Bytes);
  char* xch_lds = smem + kSmallBytes + kW0Bytes + kRedBytes;
  float* head = reinterpret_cast<float*>(xch_lds + kXchBytes);       // [row tile][quarter][row][u, z]
  ParamLayout L{a.H, a.nh};
  PackLayout K{a.H, a.nh};
  const SmallLayout SL{L.H, L.nh};
#ifdef PINN_Q_STAMP
  unsigned long long q_last = q_now();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int k = 0; k < 16; ++k) g_q_stamps[(threadIdx.x >> 6) * 16 + k] = 0;
#endif
  fill_small<S, kSmallThreads>(small, w0t, a.params, L);
  PINN_QS(0);
  const SlabSeq<H, false> seq(a.H, a.nh);
  QPipe q;
  q.pipe.lds = smem + kSlabAt;
  q.pipe.init(packed + 3 * K.total(), (unsigned)(K.total() * 2), threadIdx.x);
  q.prime(seq);
  PINN_QS(1);

  const int lane = threadIdx.x & 63, wave = q.pipe.wave, kq = lane >> 4;
  const int fq = wave & 3, rt = wave >> 2;                           // feature quarter, row tile of the workgroup
  const Xch xch{xch_lds + rt * (NP * 2048), (lane & 15) * 64 + kq * 16};
  const float inv_n = (float)(1.0 / (double)a.n_global);
  float s_nll = 0.f, s_abs = 0.f, s_mse = 0.f, s_du = 0.f, s_dz = 0.f, gmax = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.b.amax = 0u;
  const unsigned pass0 = train_pass(a.drop);
  if (blockIdx.x == 0 && threadIdx.x == 0) a.b.amax[2] = pass0;
  const float* wp = small + SL.wp();
  const int ll = L.nh - 1;
  constexpr float kPre0 = kTanhPre, kPreS = kTanhPre / S::kAccScale;
  auto drop_of = [&](int layer) { LayerDrop ld = layer_drop(a.drop, a.drop.mode, layer); ld.scale *= S::kActScale; return ld; };

  const long long n_tiles = a.b.t16 / 2;                             // 32-row tiles of the (128-row padded) stash
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t16 = tile * 2 + rt;
    const long long lrow = t16 * 16 + (lane & 15);
    const bool valid = lrow < a.n_rows;
    const long long srow = valid ? lrow : a.n_rows - 1;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2];
    const f32x4 xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    const RowCtx c{lane, kq, a.drop.row_offset + lrow, srow, a.n_rows, pass0, a.drop.mode};
    const StashX sx{(float*)a.b.stash_h, (float*)a.b.stash_v1, (float*)a.b.stash_v2, (float*)a.b.dpre_h, (float*)a.b.dpre_v1, (float*)a.b.dpre_v2,
                    a.b.t16, t16};
    if (fq == 0) {      // the input rows as a packed group of their own (train_fwd_x3_kernel)
      Frag fx;
      static_for<4>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const float v = kq == 0 ? xa[r] : (kq == 1 ? xb[r] : 0.0f);
        S::template split<r>(v * 0.0625f, 0.0f, fx);
      });
      float* xp = packed_ptr((float*)a.b.stash_x, t16, 32, lane);
      PINN_Q_ST(reinterpret_cast<u32x4*>(xp), fx.hi);
      PINN_Q_ST(reinterpret_cast<u32x4*>(xp + 256), fx.lo);
    }
    float up = 0.0f;
    // this wave's two groups (2 fq, 2 fq + 1) of a layer's output: activation -> LDS (the next layer's operand) and stash
    auto publish = [&](const f32x4 (&v)[4], int layer, float pre, bool dot, float* stash) {
      const LayerDrop ld = drop_of(layer);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int g = 2 * fq + j;
        Frag out;
        float dotv = 0.0f;
        prep_group<true>(out, v[2 * j], v[2 * j + 1], a.drop, c, ld, pre, layer, g, wp + 32 * g, dotv);
        up += dot ? dotv : 0.0f;
        xch.write(g, out);
        PINN_Q_ST(reinterpret_cast<u32x4*>(stash + 512 * g), out.hi);
        PINN_Q_ST(reinterpret_cast<u32x4*>(stash + 512 * g + 256), out.lo);
      }
    };
    // one matrix layer: NG K-groups, the B fragments from the exchange buffer (the next group's read one step ahead)
    auto layer = [&](auto ngc, auto& acc, int read_off) {
      constexpr int NG = decltype(ngc)::value;
      Frag b{xch.read(0, 0), xch.read(0, 1)};
      static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        Frag bn = b;
        if constexpr (g + 1 < NG) { bn.hi = xch.read(g + 1, 0); bn.lo = xch.read(g + 1, 1); }
        q.step(acc, b, seq, read_off, lane);
        b = bn;
      });
    };
    {
      f32x4 h0[4];
      layer_input_lds<4>(h0, w0t + 64 * fq, small + SL.b(0) + 64 * fq, xa, xb, lane);
      PINN_QS(2);
      publish(h0, 0, kPre0, ll == 0, sx.actp(0, H, lane));
    }
    PINN_QS(3);
    lds_barrier();
    PINN_QS(4);
#pragma unroll 1
    for (int l = 1; l < L.nh; ++l) {
      f32x4 acc[4];
      bias_blocks<4>(acc, small + SL.b(l) + 64 * fq, kq);
      layer(IC<NP>{}, acc, fq * 4 * 1024);
      PINN_QS(5);
      publish(acc, l, kPreS, l == ll, sx.actp(l, H, lane));
      PINN_QS(6);
      lds_barrier();
      PINN_QS(7);
    }
    // variance head, first layer: [H/2][H], two output blocks per wave = group fq of its activation (dropout module nh)
    {
      f32x4 v1[2];
      bias_blocks<2>(v1, small + SL.bv0() + 32 * fq, kq);
      layer(IC<NP>{}, v1, fq * 2 * 1024);
      PINN_QS(8);
      Frag out;
      float none = 0.0f;
      prep_group<false>(out, v1[0], v1[1], a.drop, c, drop_of(L.nh), kPreS, L.nh, fq, wp, none);
      xch.write(fq, out);
      float* sp = packed_ptr(sx.v1, sx.t16, H / 2, lane);
      PINN_Q_ST(reinterpret_cast<u32x4*>(sp + 512 * fq), out.hi);
      PINN_Q_ST(reinterpret_cast<u32x4*>(sp + 512 * fq + 256), out.lo);
    }
    lds_barrier();
    // second layer: [H/4][H/2], four output blocks: quarters 0 and 1 own two each (2 and 3 run the same steps for the weight
    // stream's sake and drop the result)
    float zp = 0.0f;
    {
      const int half = fq & 1;
      f32x4 v2[2];
      bias_blocks<2>(v2, small + SL.bv1() + 32 * half, kq);
      PINN_QS(9);
      layer(IC<NP / 2>{}, v2, half * 2 * 1024);
      PINN_QS(10);
      float* sp2 = tiled_ptr(sx.v2, sx.t16, H / 4, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v2[t][r] = tanh_pre(v2[t][r], kPreS);
        zp = block_dot(v2[t], small + SL.wv2() + (2 * half + t) * 16, kq, zp);
        if (fq < 2) store_block(sp2, 2 * half + t, v2[t]);
      }
    }
    {
      const float us = sum_kq(up), zs = sum_kq(zp);
      if (lane < 16) { head[((rt * 4 + fq) * 16 + lane) * 2] = us; head[((rt * 4 + fq) * 16 + lane) * 2 + 1] = zs; }
    }
    lds_barrier();
    if (fq == 0) {
      const float* hp = head + (rt * 4 * 16 + (lane & 15)) * 2;
      const float u = ((hp[0] + hp[32]) + hp[64]) + hp[96] + small[SL.bp()];
      const float z = (hp[1] + hp[33]) + small[SL.bv2()];
      const float yv = a.y[srow];
      // ---------------------------------------------------------------- aleatoric_loss (01:916-927) and its gradient
      float du = 0.f, dz = 0.f;
      const float sp = softplus_f32(z);
      const float var = sp + 1e-6f;
      const float sl = logf(var);                // logvar
      const float prec = expf(-sl);              // precision = exp(-logvar), 01:919
      const float e = yv - u;
      if (valid) {
        du = -(prec * e) * inv_n;
        const float sgn = (sl > 0.f) ? 1.f : ((sl < 0.f) ? -1.f : 0.f);
        const float ds = (-0.5f * prec * e * e + 0.5f + 0.01f * sgn) * inv_n;
        const float sig = z > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-z));
        dz = ds * sig / var;
        if (kq == 0) {
          s_nll += 0.5f * prec * e * e + 0.5f * sl;
          s_abs += fabsf(sl);
          s_mse += e * e;
          s_du += du;
          s_dz += dz;
        }
      }
      if (lane < 16) { a.b.du[t16 * 16 + lane] = du; a.b.dz[t16 * 16 + lane] = dz; }
      gmax = fmaxf(gmax, fmaxf(fabsf(du), fabsf(dz)));
    }
    PINN_QS(11);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slabs requested past the last step: landed before the LDS is released
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
  if (lane == 0 && fq == 0) atomicMax(a.b.emax, __float_as_uint(gmax));

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
      for (int w = 0; w < 8; ++w) t += red[w][threadIdx.x];
    a.b.loss_part[(long long)blockIdx.x * kLossTermsX + threadIdx.x] = t;
  }
  PINN_QS(12);
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
  // small row counts (at most one 32-row tile per CU, H = 256, packed stash): a row tile over four waves
  // (train_fwd_small_kernel), 32 rows per workgroup
#ifdef PINN_DEBUG_HOOKS
  static const bool no_small = getenv("PINN_X6_NOSMALL") != nullptr;  // measurement builds only: the one-wave-per-tile kernels at every size
#else
  constexpr bool no_small = false;
#endif
  // (one round of workgroups only: at 1e4 rows the 313 32-row tiles need two rounds and lose to the 157 64-row ones, 57 against 46 us)
  const bool quarters = !no_small && small_n && fast_bwd && net->hidden == 256 && 4 * t128 <= cus;
  if (quarters) {
    const long long t32 = 4 * t128;
    const int grid32 = (int)(t32 < cus ? t32 : cus);
    *grid_out = grid32;                                                // loss partials: one per forward workgroup
    if (run_fwd) hipLaunchKernelGGL((train_fwd_small_kernel<256>), dim3(grid32), dim3(kSmallThreads), 0, st, a, packed);
    if (!fwd_only) {
      if (!run_fwd) { hipError_t em = hipMemsetAsync(b.amax, 0, sizeof(unsigned), st); if (em != hipSuccess) return (int)em; }
      hipLaunchKernelGGL((train_bwd_kernel<X3, 256, 4>), dim3(grid), dim3(256), 0, st, a, packed);
    }
    hipError_t eq = hipGetLastError();
    return eq == hipSuccess ? PINN_OK : (int)eq;
  }
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

#ifdef PINN_Q_STAMP
extern "C" int pinn_q_debug_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pinn::x6::g_q_stamps), sizeof(unsigned long long) * 8 * 16);
}
#endif
