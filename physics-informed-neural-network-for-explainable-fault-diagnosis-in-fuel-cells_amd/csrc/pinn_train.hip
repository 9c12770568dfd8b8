// pinn_train.hip -- train_dnn's forward + aleatoric_loss + backward (01:949-953) for gfx950.
//
// Three kinds of kernel per step, all exact fp32 on the matrix cores:
//   K4/K5 train_chain_kernel : per 16-row wave tile, forward chain (activations in registers,
//        post-dropout activations + keep bits stashed to HBM as [tile16][feature][16]), NLL loss
//        and its gradient, backward (dgrad) chain; writes every layer's d(pre-activation) in the
//        same tiled layout.  v_mfma_f32_16x16x4_f32, two workgroups per CU.
//   K6 wgrad_kernel : dW = dpre^T . h_prev as a split-K GEMM over row tiles (K = rows), operands
//        read straight from the tiled stash as v_mfma_f32_32x32x2_f32 fragments (8 contiguous
//        floats per lane and tile), fp32 partial slabs per K-slice; bias / vector-head gradients
//        ride along as per-lane sums.
//   grad_finalize_kernel : fixed-order sum of the slabs -> flat gradient (bitwise reproducible,
//        no float atomics), already divided by the GLOBAL row count (data-parallel shards
//        all-reduce it with SUM).
#include <cstdlib>
#include "pinn_mlp_core.h"
#include "pinn_wgrad_args.h"
#include "pinn_adam_update.h"

namespace pinn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxSlices = 256;
constexpr int kLossTerms = 8;   // nll, |logvar|, (y-u)^2, du, dz, spare...

struct TrainArgs {
  const float* params;
  const float* x;
  const float* y;
  long long n_rows, n_global;
  int H, nh;
  DropDev drop;
  float* stash_h;    // [nh][T16][H][16]    post-dropout activations of hidden layers
  float* stash_v1;   // [T16][H/2][16]
  float* stash_v2;   // [T16][H/4][16]
  float* dpre_h;     // [nh][T16][H][16]    d loss / d pre-activation
  float* dpre_v1;
  float* dpre_v2;
  unsigned char* keep;   // [T16][nh*H/32 + H/64][64]  keep bits of every dropout module
  float* du;         // [T16*16]  d loss / d u      (already / n_global)
  float* dz;         // [T16*16]  d loss / d z
  double* loss_part; // [grid][kLossTerms]
  long long t16;     // 16-row tiles (padded to whole 64-row workgroup tiles)
};

// dpre = dh * scale * keep * (1 - a^2),  a = h / scale  (h = post-dropout activation), for blocks 2fp, 2fp+1
__device__ __forceinline__ void tanh_drop_backward_pair(f32x4& d0, f32x4& d1, const f32x4& h0, const f32x4& h1, unsigned keep,
                                                        float scale, float inv_scale) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a0 = h0[r] * inv_scale, a1 = h1[r] * inv_scale;
    const float g0 = d0[r] * (scale * (1.0f - a0 * a0));
    const float g1 = d1[r] * (scale * (1.0f - a1 * a1));
    d0[r] = ((keep >> r) & 1u) ? g0 : 0.0f;
    d1[r] = ((keep >> (4 + r)) & 1u) ? g1 : 0.0f;
  }
}

template <int H, bool kBits>
__global__ __launch_bounds__(kThreads, 2) void train_chain_kernel(TrainArgs a) {
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32;
  __shared__ __attribute__((aligned(16))) char lds_w[2 * kChunkBytes];
  __shared__ ChunkDesc tab[kMaxChunks];
  __shared__ double red[4][kLossTerms];
  __shared__ __attribute__((aligned(16))) float small[kMaxSmall];
  ParamLayout L{a.H, a.nh};
  const SmallLayout S{a.H, a.nh};
  const int n_fwd = (a.nh - 1) * NP + NP + NP / 2;
  const int n_bwd = H / 128 + H / 64 + (a.nh - 1) * NP;
  if (threadIdx.x == 0) {
    int k = build_forward_chunks(tab, L, 0);
    build_backward_chunks(tab, L, k);
  }
  load_small_params(small, a.params, L);
  Pipe pipe;
  pipe.params = a.params; pipe.tab = tab; pipe.lds = lds_w; pipe.n = n_fwd + n_bwd;
  pipe.prime();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4;
  const float* __restrict__ P = a.params;
  const bool drop = a.drop.mode != PINN_DROP_NONE;
  const float inv_n = (float)(1.0 / (double)a.n_global);
  const int n_groups = L.nh * NP + NP / 2;
  float s_nll = 0.f, s_abs = 0.f, s_mse = 0.f, s_du = 0.f, s_dz = 0.f;

  const long long n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  const unsigned pass0 = train_pass(a.drop);      // 0, or the device's step counter (replayed graphs)
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t16 = tile * 4 + wave;
    const long long lrow = t16 * 16 + (lane & 15);
    const bool valid = lrow < a.n_rows;
    const long long srow = valid ? lrow : a.n_rows - 1;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2];
    const f32x4 xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    const float yv = a.y[srow];
    const RowCtx c{lane, kq, a.drop.row_offset + lrow, srow, a.n_rows, pass0, a.drop.mode};
    const StashPtrs st{a.stash_h, a.stash_v1, a.stash_v2, a.keep, a.t16, t16};
    const unsigned char* keep = a.keep + (t16 * n_groups) * 64 + lane;

    // ------------------------------------------------------------------ forward (stash + keep bits written)
    float u, z;
    f32x4 v2[NT4];
    forward_pass<H, true, kBits>(P, small, L, pipe, a.drop, c, xa, xb, st, u, z, v2);

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
      if (lane < 16) { a.du[t16 * 16 + lane] = du; a.dz[t16 * 16 + lane] = dz; }
    }

    // ------------------------------------------------------------------ backward: variance head
    f32x4 dh[NT];
    {
      f32x4 dpv1[NT2];
      {
        // d pre_v2 = wv2[f] * dz * (1 - v2^2)
        float* sp = tiled_ptr(a.dpre_v2, t16, H / 4, lane);
#pragma unroll
        for (int t = 0; t < NT4; ++t) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(small + S.wv2() + t * 16 + 4 * kq);
#pragma unroll
          for (int r = 0; r < 4; ++r) v2[t][r] = w[r] * dz * (1.0f - v2[t][r] * v2[t][r]);
          store_block(sp, t, v2[t]);
        }
        zero_blocks<NT2>(dpv1);
        layer_backward<NT4, NT2>(dpv1, v2, pipe, lane, H / 2);
      }
      {
        const float scale = drop ? a.drop.scale[L.nh] : 1.0f, inv_scale = 1.0f / scale;
        const float* hp = tiled_ptr(a.stash_v1, t16, H / 2, lane);
        float* sp = tiled_ptr(a.dpre_v1, t16, H / 2, lane);
        // all stash loads first (independent addresses, one exposed latency), then the arithmetic
        f32x4 hl[NT2];
        unsigned kb[NP / 2];
#pragma unroll
        for (int t = 0; t < NT2; ++t) load_block(hp, t, hl[t]);
#pragma unroll
        for (int fp = 0; fp < NP / 2; ++fp) kb[fp] = keep[(L.nh * NP + fp) * 64];
#pragma unroll
        for (int fp = 0; fp < NP / 2; ++fp) {
          tanh_drop_backward_pair(dpv1[2 * fp], dpv1[2 * fp + 1], hl[2 * fp], hl[2 * fp + 1], kb[fp], scale, inv_scale);
          store_block(sp, 2 * fp, dpv1[2 * fp]);
          store_block(sp, 2 * fp + 1, dpv1[2 * fp + 1]);
        }
      }
      // d h_last = w_p * du + Wv0^T d pre_v1
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(small + S.wp() + t * 16 + 4 * kq);
        dh[t] = w * du;
      }
      layer_backward<NT2, NT>(dh, dpv1, pipe, lane, H);
    }

    // ------------------------------------------------------------------ backward: hidden layers nh-1 .. 0
#pragma unroll 1
    for (int l = L.nh - 1; l >= 0; --l) {
      {
        const float scale = drop ? a.drop.scale[l] : 1.0f, inv_scale = 1.0f / scale;
        const float* hp = tiled_ptr(a.stash_h + (long long)l * a.t16 * H * 16, t16, H, lane);
        float* sp = tiled_ptr(a.dpre_h + (long long)l * a.t16 * H * 16, t16, H, lane);
        f32x4 hl[NT];
        unsigned kb[NP];
#pragma unroll
        for (int t = 0; t < NT; ++t) load_block(hp, t, hl[t]);
#pragma unroll
        for (int fp = 0; fp < NP; ++fp) kb[fp] = keep[(l * NP + fp) * 64];
#pragma unroll
        for (int fp = 0; fp < NP; ++fp) {
          tanh_drop_backward_pair(dh[2 * fp], dh[2 * fp + 1], hl[2 * fp], hl[2 * fp + 1], kb[fp], scale, inv_scale);
          store_block(sp, 2 * fp, dh[2 * fp]);
          store_block(sp, 2 * fp + 1, dh[2 * fp + 1]);
        }
      }
      if (l > 0) {
        f32x4 acc[NT];
        zero_blocks<NT>(acc);
        layer_backward<NT, NT>(acc, dh, pipe, lane, H);
#pragma unroll
        for (int t = 0; t < NT; ++t) dh[t] = acc[t];
      }
    }
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
  if (threadIdx.x < kLossTerms) {
    double t = 0.0;
    if (threadIdx.x < 5)
      for (int w = 0; w < 4; ++w) t += red[w][threadIdx.x];
    a.loss_part[(long long)blockIdx.x * kLossTerms + threadIdx.x] = t;
  }
}

// ---------------------------------------------------------------------------------------
// K6: weight gradients.  dW[i][j] = sum_rows P[i][row] Q[j][row]  (+ bias / vector sums)
// MFMA 32x32x2: k-step s of a 16-row tile pairs rows (s, s + 8): lane half hh supplies row 8*hh + s,
// so a lane's operand for the 8 k-steps of a tile is 8 contiguous floats of one feature row.
// ---------------------------------------------------------------------------------------
#define PINN_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int TI, int TJ>
struct WFrag {
  f32x4 a[TI][2], b[TJ][2];
};

template <int TI, int TJ, bool QX>
__device__ __forceinline__ void wgrad_load(WFrag<TI, TJ>& f, const WgradArgs& a, long long t, int i0, int j0, int hh, int i) {
  const float* pP = a.P + ((t * a.OUT + i0 + i) * 16 + 8 * hh);
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    f.a[ti][0] = *reinterpret_cast<const f32x4*>(pP + ti * 512);
    f.a[ti][1] = *reinterpret_cast<const f32x4*>(pP + ti * 512 + 4);
  }
  if (!QX) {
    const float* pQ = a.Q + ((t * a.IN + j0 + i) * 16 + 8 * hh);
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      f.b[tj][0] = *reinterpret_cast<const f32x4*>(pQ + tj * 512);
      f.b[tj][1] = *reinterpret_cast<const f32x4*>(pQ + tj * 512 + 4);
    }
  } else {
    // Q = x^T: feature j = lane & 31 (< 8 valid), rows t*16 + 8*hh + s read from row-major x
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        long long row = t * 16 + 8 * hh + 4 * sg + j;
        if (row >= a.n_rows) row = a.n_rows - 1;     // dpre of such rows is exactly 0
        f.b[0][sg][j] = (i < 8) ? a.x[row * 8 + i] : 0.0f;
      }
  }
}

template <int TI, int TJ, int WI, int WJ, bool QX>
__global__ __launch_bounds__(kThreads, 1) void wgrad_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= WI * WJ) return;
  const int wi = wave / WJ, wj = wave % WJ;
  const int hh = lane >> 5, i = lane & 31;
  const int i0 = blockIdx.y * (TI * 32 * WI) + wi * TI * 32, j0 = wj * TJ * 32;     // blockIdx.y: output block (wide nets' layer 0)

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
  // (bias sums: a slice adds up to a few thousand rows per lane one after the other, where torch sums pairwise -- fp32 over
  //  eight tiles, those partial sums in float64; float64 adds in every tile cost the layer-0 kernel a quarter of its time)
  double bsum[TI];
  float bpart[TI], vq[TJ], vr[TI];
  int n_part = 0;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) { bsum[ti] = 0.0; bpart[ti] = 0.f; vr[ti] = 0.f; }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) vq[tj] = 0.f;

  const int slice = blockIdx.x;
  const long long per = (a.t16 + a.n_slices - 1) / a.n_slices;
  const long long t_begin = slice * per;
  long long t_end = t_begin + per;
  if (t_end > a.t16) t_end = a.t16;

  // software pipeline, two tiles deep where the registers allow (< 256 accumulator registers): the fragments and the
  // vector-head operands of tiles t+1 and t+2 are in flight while tile t multiplies (one tile ahead is about one HBM
  // latency of work; the layer-0 kernel streamed at 2.9 TB/s with it)
  constexpr bool kDeep = TI * TJ < 16;
  struct Side { f32x4 r[kDeep ? TI : 1][2], s2[2], s1[2]; };      // operands of the vector-head sums (prefetched iff kDeep)
  const bool want_r = wj == 0 && a.dvr, want_q = wi == 0 && a.dvq;
  auto load_r = [&](long long t, int ti, int sg) { return *reinterpret_cast<const f32x4*>(a.R + ((t * a.OUT + i0 + i) * 16 + 8 * hh) + ti * 512 + sg * 4); };
  auto load_s = [&](const float* sv, long long t, int sg) { return *reinterpret_cast<const f32x4*>(sv + t * 16 + 8 * hh + sg * 4); };
  auto fetch = [&](WFrag<TI, TJ>& f, Side& sd, long long t) {
    if (t >= t_end) t = t_end - 1;
    wgrad_load<TI, TJ, QX>(f, a, t, i0, j0, hh, i);
    if constexpr (kDeep) {
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        if (want_r) {
          sd.s2[sg] = load_s(a.s2, t, sg);
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) sd.r[ti][sg] = load_r(t, ti, sg);
        }
        if (want_q) sd.s1[sg] = load_s(a.s1, t, sg);
      }
    }
  };
  auto tile = [&](const WFrag<TI, TJ>& cur, const Side& sd, long long t) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = PINN_MFMA32(cur.a[ti][sg][j], cur.b[tj][sg][j], acc[ti][tj]);
    if (wj == 0) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) bpart[ti] += (cur.a[ti][sg][0] + cur.a[ti][sg][1]) + (cur.a[ti][sg][2] + cur.a[ti][sg][3]);
      if ((++n_part & 7) == 0) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) { bsum[ti] += (double)bpart[ti]; bpart[ti] = 0.f; }
      }
      if (a.dvr) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          f32x4 sv;
          if constexpr (kDeep) sv = sd.s2[sg]; else sv = load_s(a.s2, t, sg);
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) {
            f32x4 rv;
            if constexpr (kDeep) rv = sd.r[ti][sg]; else rv = load_r(t, ti, sg);
            vr[ti] += (sv[0] * rv[0] + sv[1] * rv[1]) + (sv[2] * rv[2] + sv[3] * rv[3]);
          }
        }
      }
    }
    if (wi == 0 && a.dvq) {
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        f32x4 sv;
        if constexpr (kDeep) sv = sd.s1[sg]; else sv = load_s(a.s1, t, sg);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          vq[tj] += (sv[0] * cur.b[tj][sg][0] + sv[1] * cur.b[tj][sg][1]) + (sv[2] * cur.b[tj][sg][2] + sv[3] * cur.b[tj][sg][3]);
      }
    }
  };
  if constexpr (kDeep) {
    WFrag<TI, TJ> f0, f1, f2;
    Side d0, d1, d2;
    if (t_begin < t_end) { fetch(f0, d0, t_begin); fetch(f1, d1, t_begin + 1); }
    for (long long t = t_begin; t < t_end; t += 3) {
      fetch(f2, d2, t + 2);
      tile(f0, d0, t);
      if (t + 1 < t_end) { fetch(f0, d0, t + 3); tile(f1, d1, t + 1); }
      if (t + 2 < t_end) { fetch(f1, d1, t + 4); tile(f2, d2, t + 2); }
    }
  } else {
    WFrag<TI, TJ> cur, nxt;
    Side none;
    if (t_begin < t_end) fetch(cur, none, t_begin);
    for (long long t = t_begin; t < t_end; ++t) {
      fetch(nxt, none, t + 1);
      tile(cur, none, t);
      cur = nxt;
    }
  }

  // ---- write this slice's slab
  const long long so = (long long)slice * a.slab_stride;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int col = j0 + tj * 32 + i;          // input feature (C/D layout: column on the lane)
      if (col < a.IN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          a.dW[so + (long long)row * a.IN + col] = acc[ti][tj][r];
        }
      }
    }
  if (wj == 0) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const double bl = bsum[ti] + (double)bpart[ti];
      const float b = (float)(bl + __shfl_xor(bl, 32, 64));
      if (hh == 0) a.db[so + i0 + ti * 32 + i] = b;
      if (a.dvr) {
        const float v = vr[ti] + __shfl_xor(vr[ti], 32, 64);
        if (hh == 0) a.dvr[so + i0 + ti * 32 + i] = v;
      }
    }
  }
  if (wi == 0 && a.dvq) {
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const float v = vq[tj] + __shfl_xor(vq[tj], 32, 64);
      if (hh == 0) a.dvq[so + j0 + tj * 32 + i] = v;
    }
  }
}

// grads[e] = sum_s slab[s][e] in a fixed order; scalar head biases and the loss sums from the chain partials.
// A workgroup owns 32 groups of four consecutive parameters; its 256 threads are 8 slice lanes x 32 groups: lane q adds
// the slices q, q + 8, ... of its group (8 independent 16-B loads in flight), the eight partial sums meet in LDS and are
// added in lane order -- the same tree on every run (bitwise reproducible; no float atomics).  (One thread per group
// walking all 256 slices was 32 dependent rounds of loads: 64 us whatever the row count, a fifth of a 65 536-row step.)
// The slabs are added in float64 (the launch is bound by its 180 MB of slab reads, not by the adds) and rounded to fp32 once:
// the reduction over slices adds no rounding of its own to what the slice kernels accumulated.
constexpr int kFinLanes = 8, kFinGroups = 32;
// pinn_mlp_train_step_dev: the optimizer step in the reduction's launch (p == nullptr: gradients only).  `snap` = the step
// counter as the forward kernel of this call read it (TrainBuffers::amax + 2): the counter itself moves during this launch.
struct FinAdam {
  float* p; float* m; float* v;
  const float* coeffs; const unsigned* snap;      // device table + step snapshot (pinn_mlp_train_step_dev), or nullptr:
  float step_size, bc2_sqrt;                      // the step's two scalars by value (pinn_mlp_train_step)
};
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 widen4(const f32x4& v) { return f64x4{(double)v[0], (double)v[1], (double)v[2], (double)v[3]}; }
__global__ __launch_bounds__(256) void grad_finalize_kernel(const float* __restrict__ slabs, int n_slices, long long total,
                                                            const double* __restrict__ loss_part, int n_parts,
                                                            long long off_bp, long long off_bv2, float* __restrict__ grads,
                                                            double* __restrict__ loss_out, const unsigned* __restrict__ amax,
                                                            unsigned* __restrict__ range_word, unsigned* __restrict__ step_counter,
                                                            long long e_lo, long long e_hi, int with_loss, FinAdam ad) {
  // [e_lo, e_hi): the part of the flat gradient this launch reduces (multiples of 4; the whole vector, or the head / the tail
  // of pinn_grad_split); with_loss: the launch that also owns the loss sums, the range record and the step counter (the tail)
  __shared__ f64x4 part[kFinLanes][kFinGroups];
  // pinn_net_range_status: the X3 backward chain's largest |d pre-activation| of this call is inf once a row-normalised
  // gradient has left fp16's range (a NaN alone does not move the maximum, but then the gradients are NaN: visible)
  if (with_loss && range_word && blockIdx.x == 0 && threadIdx.x == 0 && (*amax & 0x7FFFFFFFu) >= 0x7F800000u) *range_word = 1u;
  // pinn_dropout_t.d_step_counter: the gradients of this step are final with this launch; every kernel that read the counter
  // (the chain's dropout pass) ran before it, the optimizer step that reads it next runs behind it
  if (with_loss && step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1u;
  float step_size = 0.0f, bc2_sqrt = 1.0f;
  if (ad.p) {
    if (ad.coeffs) {
      const unsigned k = __builtin_amdgcn_readfirstlane(*ad.snap);
      step_size = ad.coeffs[2 * k]; bc2_sqrt = ad.coeffs[2 * k + 1];
    } else {
      step_size = ad.step_size; bc2_sqrt = ad.bc2_sqrt;
    }
  }
  const int g = threadIdx.x & (kFinGroups - 1), q = threadIdx.x / kFinGroups;
  // every tensor starts on a multiple of 4 floats, so the two scalar head biases sit at the start of a group whose other
  // three floats are padding
  const long long e = e_lo + 4 * ((long long)blockIdx.x * kFinGroups + g);
  f64x4 s = {0.0, 0.0, 0.0, 0.0};
  if (e < e_hi) {
    int k = q;
    for (; k + 7 * kFinLanes < n_slices; k += 8 * kFinLanes) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(slabs + (long long)(k + j * kFinLanes) * total + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += widen4(v[j]);
    }
    for (; k < n_slices; k += kFinLanes) s += widen4(*reinterpret_cast<const f32x4*>(slabs + (long long)k * total + e));
  }
  part[q][g] = s;
  __syncthreads();
  if (q == 0 && e < e_hi) {
    if (e == off_bp || e == off_bv2) {
      grads[e + 1] = 0.0f; grads[e + 2] = 0.0f; grads[e + 3] = 0.0f;       // [e] itself: the loss-sum block below
    } else {
#pragma unroll
      for (int j = 1; j < kFinLanes; ++j) s += part[j][g];
      const f32x4 gv = f32x4{(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
      *reinterpret_cast<f32x4*>(grads + e) = gv;
      if (ad.p) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(ad.p + e), mv = *reinterpret_cast<const f32x4*>(ad.m + e), vv = *reinterpret_cast<const f32x4*>(ad.v + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) { float pj = pv[j], mj = mv[j], vj = vv[j]; adam_update(pj, gv[j], mj, vj, step_size, bc2_sqrt); pv[j] = pj; mv[j] = mj; vv[j] = vj; }
        *reinterpret_cast<f32x4*>(ad.p + e) = pv; *reinterpret_cast<f32x4*>(ad.m + e) = mv; *reinterpret_cast<f32x4*>(ad.v + e) = vv;
      }
    }
  }
  if (blockIdx.x == 0 && with_loss) {
    // the chain's per-workgroup loss partials (<= 1024 x kLossTerms doubles): strided over the 256 threads, then a fixed
    // shuffle / LDS tree (one thread walking them serially cost ~50 us of dependent L2 latency)
    __shared__ double lred[4][kLossTerms];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kLossTerms; ++k) {
      double v = 0.0;
      for (int b = threadIdx.x; b < n_parts; b += 256) v += loss_part[(long long)b * kLossTerms + k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) lred[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kLossTerms) {
      const double t = ((lred[0][threadIdx.x] + lred[1][threadIdx.x]) + lred[2][threadIdx.x]) + lred[3][threadIdx.x];
      if (threadIdx.x < 4) loss_out[threadIdx.x] = t;    // nll sum, |logvar| sum, (y-u)^2 sum, (sum du)
      if (threadIdx.x == 3) grads[off_bp] = (float)t;    // d loss / d b_p  = sum du
      if (threadIdx.x == 4) grads[off_bv2] = (float)t;   // d loss / d bv_2 = sum dz
      if (ad.p && (threadIdx.x == 3 || threadIdx.x == 4)) {      // (their groups' padding floats have zero gradient: no update)
        const long long o = threadIdx.x == 3 ? off_bp : off_bv2;
        float pj = ad.p[o], mj = ad.m[o], vj = ad.v[o];
        adam_update(pj, (float)t, mj, vj, step_size, bc2_sqrt);
        ad.p[o] = pj; ad.m[o] = mj; ad.v[o] = vj;
      }
    }
  }
}

constexpr long long kFanOutT16 = 2048;       // row tiles below which the weight-gradient launches fan out over side streams
// Row tiles below which PINN_PREC_F32X6 at H = 256 runs every layer's weight gradient as ONE launch of 128 x 128 tiles over 32
// slices (wgrad_p_multi_kernel).  Built for the reference's 1e3 .. 1e4 rows, it wins far beyond: five launches of 256 slices
// write and read back 180 MB of slabs whatever the row count -- at BASELINE config 4's 65 536-row minibatch that was as much
// traffic as the operands themselves (weight gradients 164 -> 123 us, reduction 31 -> 6 us, step 441 -> 372 us; 131 072 rows:
// 770 -> 740; 262 144: 1356 vs 1420, the per-layer kernels' 256 x 256 tiles win from there).
#ifdef PINN_ABL_MULTI_T16
constexpr long long kMultiT16 = PINN_ABL_MULTI_T16;
#else
constexpr long long kMultiT16 = 10240;
#endif

// Side streams for the independent weight-gradient launches of one step at small row counts.  Fork / join with events on the
// caller's stream, so the pattern is legal inside a stream capture (model.train_dnn replays the step as a hipGraph, where the
// branches cost nothing on the host); created at the first small call.  One process drives one GPU (pinn_hip.h).
struct FanOut {
  hipStream_t side[3];
  hipEvent_t fork, join[3];
  bool ok = false;
  bool init() {
    if (ok) return true;
    for (int i = 0; i < 3; ++i) {
      if (hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess) return false;
    }
    if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
    return ok = true;
  }
};
static FanOut g_fan;

struct Workspace {
  long long t16;
  size_t off_stash_h, off_stash_v1, off_stash_v2, off_dpre_h, off_dpre_v1, off_dpre_v2, off_keep, off_du, off_dz, off_loss,
      off_amax, off_rowmeta, off_stash_x, off_slabs;
  int n_slices;
  size_t total;
};

static Workspace plan_workspace(const pinn_net_t* net, long long n_rows) {
  Workspace w;
  const long long H = net->hidden, nh = net->n_hidden;
  // whole workgroup tiles: 64 rows (4 wave tiles), 128 (8) for the x6 chain
  w.t16 = (net->precision >= PINN_PREC_F32X6 || net->hidden > 256) ? (n_rows + 127) / 128 * 8 : (n_rows + kTileRows - 1) / kTileRows * 4;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  w.off_stash_h = take((size_t)nh * w.t16 * H * 16 * 4);
  w.off_stash_v1 = take((size_t)w.t16 * (H / 2) * 16 * 4);
  w.off_stash_v2 = take((size_t)w.t16 * (H / 4) * 16 * 4);
  w.off_dpre_h = take((size_t)nh * w.t16 * H * 16 * 4);
  w.off_dpre_v1 = take((size_t)w.t16 * (H / 2) * 16 * 4);
  w.off_dpre_v2 = take((size_t)w.t16 * (H / 4) * 16 * 4);
  w.off_keep = take((size_t)w.t16 * (nh * (H / 32) + H / 64) * 64);
  w.off_du = take((size_t)w.t16 * 16 * 4);
  w.off_dz = take((size_t)w.t16 * 16 * 4);
  w.off_loss = take((size_t)1024 * kLossTerms * 8);
  w.off_amax = take(256);                                   // [0] TrainBuffers::amax, [1] ::emax, [2] the step counter as the forward kernel read it
  w.off_rowmeta = take((size_t)w.t16 * 256);                // struct RowMeta records
  w.off_stash_x = take((size_t)w.t16 * 2048);               // packed input rows (PINN_PREC_F32X6, fused nets)
  const long long t32 = (w.t16 + 1) / 2;
  // slices = workgroups per weight-gradient launch = slabs the reduction adds.  Large row counts: one per CU.  The reference's
  // own sizes (< 32 768 rows) are bound by the slabs instead -- at 1e4 rows 256 slices wrote 67 MB per 256 x 256 layer and the
  // reduction read 180 MB, 100 of the step's 200 us --: 64 slices there, and the five layers' launches run side by side
  // (fan_out below), 320 workgroups in all
  // (fan_out below), 320 workgroups in all; PINN_PREC_F32X6 at H = 256 runs them as ONE launch of 128 x 128 tiles
  // (wgrad_p_multi_kernel): 32 slices, ~450 workgroups
  const bool one_launch = net->precision == PINN_PREC_F32X6 && H == 256 && nh - 1 + 3 <= kMaxWgradProblems;
#ifdef PINN_ABL_MULTI_NS
  constexpr long long kMultiSlices = PINN_ABL_MULTI_NS;
#else
  constexpr long long kMultiSlices = 32;
#endif
  // wide nets: a gradient is (H / 256)^2 blocks of 256 x 256, each with its own slices -- two rounds of workgroups fill the chip, and
  // every further slice is a slab of the whole gradient written and read back (H = 1024 at 262 144 rows: 256 slices wrote 1.07 GB per
  // layer, profiles/r03/pmc_summary_wide.json; 32 slices: training step 21.05 -> 19.6 ms; 16: 20.4; 8: 25.6)
#ifdef PINN_ABL_WIDE_NS
  const long long wide_slices = PINN_ABL_WIDE_NS;
#else
  const long long hb = H / 256, wide_slices = hb > 0 && 512 / (hb * hb) > 8 ? 512 / (hb * hb) : 8;
#endif
  const long long cap = (one_launch && w.t16 < kMultiT16) ? kMultiSlices : (w.t16 < kFanOutT16 ? 64 : (H > 256 ? wide_slices : (long long)kMaxSlices));
  w.n_slices = (int)(t32 < cap ? (t32 < 1 ? 1 : t32) : cap);
  ParamLayout L{(int)H, (int)nh};
  w.off_slabs = take((size_t)w.n_slices * L.total() * 4);
  w.total = o;
  return w;
}

// c of the packed weight gradients' row scale t_r = 2^(e_r - E + c) (pinn_x6_core.h): 8 |h| t_r must stay a finite fp16 with
// |h| <= the largest dropout scale of the call
static int row_scale_boost(const DropDev& d, int nh) {
  float smax = 1.0f;
  if (d.mode != PINN_DROP_NONE)
    for (int l = 0; l <= nh; ++l) smax = d.scale[l] > smax ? d.scale[l] : smax;
  int c = (int)floor(log2(65504.0 / (8.0 * (double)smax)));
  return c < 0 ? 0 : (c > 12 ? 12 : c);
}

static int check_net_t(const pinn_net_t* net) {
  if (!net) return PINN_E_ARG;
  if (net->n_in != 8) return PINN_E_ARCH;
  const bool wide = net->hidden == 512 || net->hidden == 1024 || net->hidden == 2048;    // layer-by-layer kernels (pinn_wide.hip)
  if (net->hidden != 128 && net->hidden != 256 && !wide) return PINN_E_ARCH;
  if (net->n_hidden < 1 || net->n_hidden > 8) return PINN_E_ARCH;
  if (net->precision < PINN_PREC_FP32 || net->precision > PINN_PREC_F32X6_G6) return PINN_E_ARG;
  if (wide && net->precision == PINN_PREC_FP32) return PINN_E_ARCH;
  if (net->precision != PINN_PREC_FP32 && !net->d_packed) return PINN_E_ARG;
  return PINN_OK;
}

static int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int TI, int TJ, int WI, int WJ, bool QX>
static void launch_wgrad(const WgradArgs& a, hipStream_t st, int out_blocks = 1) {
  hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WI, WJ, QX>), dim3(a.n_slices, out_blocks), dim3(kThreads), 0, st, a);
}

// pick the wave tiling for an [OUT x IN] gradient (tiles of 32x32; IN = 8 is padded to one tile)
static int dispatch_wgrad(const WgradArgs& a, hipStream_t st) {
  const int to = a.OUT / 32, ti = (a.IN + 31) / 32;
  if (a.Q == nullptr) {
    if (to % 8 == 0) launch_wgrad<2, 1, 4, 1, true>(a, st, to / 8);       // 256 outputs per workgroup
    else if (to == 4) launch_wgrad<1, 1, 4, 1, true>(a, st);
    else return PINN_E_ARCH;
    return PINN_OK;
  }
  if (to == 8 && ti == 8) launch_wgrad<4, 4, 2, 2, false>(a, st);
  else if (to == 4 && ti == 8) launch_wgrad<2, 4, 2, 2, false>(a, st);
  else if (to == 2 && ti == 4) launch_wgrad<1, 2, 2, 2, false>(a, st);
  else if (to == 4 && ti == 4) launch_wgrad<2, 2, 2, 2, false>(a, st);
  else if (to == 1 && ti == 2) launch_wgrad<1, 1, 1, 2, false>(a, st);
  else return PINN_E_ARCH;
  return PINN_OK;
}

}  // namespace pinn

namespace pinn {
int launch_train_chain_x6(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y, long long n_rows,
                          long long n_global, const DropDev& drop, const TrainBuffers& b, unsigned which, int* grid_out,
                          void* stream);   // pinn_x6_train.hip; which: 1 = forward kernel, 2 = backward kernel, 3 = both
int launch_train_chain_wide(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y, long long n_rows,
                            long long n_global, const DropDev& drop, const TrainBuffers& b, int* grid_out, void* stream);  // pinn_wide.hip
int launch_train_bf16(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y, long long n_rows,
                      long long n_global, const DropDev& drop, const TrainBuffers& b, unsigned phases, int* grid_out, void* stream);
int dispatch_wgrad_x6(const WgradArgs& a, int ns, void* stream);   // pinn_x6_wgrad.hip
int dispatch_wgrad_p(const WgradPArgs& a, void* stream);            // pinn_x6_wgrad.hip: packed operands (PINN_PREC_F32X6, fused nets)
int dispatch_wgrad_p_multi(const WgradPMulti& m, void* stream);     //   several of them in one launch (small row counts, H = 256)
}

using namespace pinn;

extern "C" long long pinn_grad_split(const pinn_net_t* net) {
  if (check_net_t(net) != PINN_OK) return -1;
  if (net->precision == PINN_PREC_BF16 && net->hidden <= 256) return 0;      // the fused bf16 family launches its weight gradients as one block
  ParamLayout L{net->hidden, net->n_hidden};
  return net->n_hidden >= 2 ? L.w(net->n_hidden - 1) : L.wp();
}

extern "C" size_t pinn_train_workspace_bytes(const pinn_net_t* net, long long n_rows) {
  if (check_net_t(net) != PINN_OK || n_rows < 0) return 0;
  return plan_workspace(net, n_rows).total;
}

// fa: the optimizer step in the reduction's launch (pinn_mlp_train_step_dev), or nullptr
static int train_grads_impl(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                            long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                            double* d_loss, void* d_work, size_t work_bytes, void* stream, unsigned phases, const FinAdam* fa) {
  int rc = check_net_t(net);
  if (rc) return rc;
  if (!d_params || !d_x || !d_y || !d_grads || !d_loss || !d_work || n_rows <= 0 || n_global < n_rows) return PINN_E_ARG;
  if (((unsigned long long)d_grads | (unsigned long long)d_work) & 15) return PINN_E_ARG;      // 16-B vector accesses
  // the forward / backward halves are separate kernels only in the fused x6 path; elsewhere either bit means the chain
  if (!(net->precision >= PINN_PREC_F32X6 && net->hidden <= 256) && (phases & (PINN_PHASE_CHAIN_FWD | PINN_PHASE_CHAIN_BWD)))
    phases |= PINN_PHASE_CHAIN;
  // two-part weight gradients / reduction (pinn_grad_split): the single-part bits mean both parts; a precision whose kernels do
  // not split keeps everything in the "tail"
  if (phases & PINN_PHASE_WGRAD) phases |= PINN_PHASE_WGRAD_TAIL | PINN_PHASE_WGRAD_HEAD;
  if (phases & PINN_PHASE_REDUCE) phases |= PINN_PHASE_REDUCE_TAIL | PINN_PHASE_REDUCE_HEAD;
  const long long split = pinn_grad_split(net);
  if (split == 0) {
    phases = (phases & ~(PINN_PHASE_WGRAD | PINN_PHASE_REDUCE)) | ((phases & PINN_PHASE_WGRAD_TAIL) ? PINN_PHASE_WGRAD : 0u) |
             ((phases & PINN_PHASE_REDUCE_TAIL) ? PINN_PHASE_REDUCE : 0u);
  }
  const Workspace w = plan_workspace(net, n_rows);
  if (work_bytes < w.total) return PINN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();   // drop a stale error left by another HIP user of this thread
  char* base = (char*)d_work;
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};
  FinAdam fin_adam{};
  if (fa) { fin_adam = *fa; fin_adam.snap = fa->coeffs ? (const unsigned*)(base + w.off_amax) + 2 : nullptr; }

  TrainArgs a{};
  a.params = d_params; a.x = d_x; a.y = d_y; a.n_rows = n_rows; a.n_global = n_global; a.H = H; a.nh = nh;
  // dropout conversion (same rules as the forward entry points)
  {
    DropDev& d = a.drop;
    d.mode = PINN_DROP_NONE; d.bits = nullptr; d.words = 0; d.nb = H / 32; d.seed_lo = d.seed_hi = 0; d.stream = 0; d.row_offset = 0;
    d.step_counter = nullptr;
    for (int l = 0; l < kMaxDrop; ++l) { d.thr[l] = 0; d.scale[l] = 1.0f; }
    if (drop) {
      if (drop->mode < PINN_DROP_NONE || drop->mode > PINN_DROP_BITS) return PINN_E_ARG;
      if (drop->d_step_counter && H > 256) return PINN_E_ARCH;      // the layer-by-layer kernels take their pass index by value
      d.mode = drop->mode; d.row_offset = drop->row_offset; d.step_counter = drop->d_step_counter;
      if (drop->mode != PINN_DROP_NONE) {
        for (int l = 0; l <= nh; ++l) {
          const float p = drop->p[l];
          if (!(p >= 0.0f && p < 1.0f)) return PINN_E_ARG;
          double t = floor((double)p * 65536.0 + 0.5);
    if (p > 0.0f && t < 1.0) t = 1.0;      // a positive p never rounds to "no dropout"
          d.thr[l] = (unsigned)(t < 0 ? 0 : (t > 65536.0 ? 65536.0 : t));
          d.scale[l] = 1.0f / (float)(1.0 - (double)p);
        }
        d.seed_lo = (unsigned)(drop->seed & 0xFFFFFFFFull); d.seed_hi = (unsigned)(drop->seed >> 32); d.stream = drop->stream;
        if (drop->mode == PINN_DROP_BITS) {
          if (!drop->d_bits) return PINN_E_ARG;
          d.bits = drop->d_bits; d.words = nh * (H / 32) + H / 64;
        }
      }
    }
  }
  a.stash_h = (float*)(base + w.off_stash_h); a.stash_v1 = (float*)(base + w.off_stash_v1); a.stash_v2 = (float*)(base + w.off_stash_v2);
  a.dpre_h = (float*)(base + w.off_dpre_h); a.dpre_v1 = (float*)(base + w.off_dpre_v1); a.dpre_v2 = (float*)(base + w.off_dpre_v2);
  a.keep = (unsigned char*)(base + w.off_keep);
  a.du = (float*)(base + w.off_du); a.dz = (float*)(base + w.off_dz);
  a.loss_part = (double*)(base + w.off_loss);
  a.t16 = w.t16;
  const long long n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  int grid = (int)(n_tiles < 2 * cu_count() ? n_tiles : 2 * cu_count());
  if (grid > 1024) grid = 1024;
  if (net->precision == PINN_PREC_BF16 && H <= 256) {
    TrainBuffers b{};
    b.stash_h = a.stash_h; b.stash_v1 = a.stash_v1; b.stash_v2 = a.stash_v2;
    b.dpre_h = a.dpre_h; b.dpre_v1 = a.dpre_v1; b.dpre_v2 = a.dpre_v2;
    b.keep = a.keep; b.du = a.du; b.dz = a.dz; b.loss_part = a.loss_part;
    b.slabs = (float*)(base + w.off_slabs); b.t16 = w.t16; b.n_slices = w.n_slices;
    if ((rc = launch_train_bf16(net, d_params, d_x, d_y, n_rows, n_global, a.drop, b, phases, &grid, stream))) return rc;
    if (phases & PINN_PHASE_REDUCE)
      hipLaunchKernelGGL(grad_finalize_kernel, dim3((unsigned)((L.total() / 4 + kFinGroups - 1) / kFinGroups)), dim3(256), 0, st, b.slabs, w.n_slices,
                         L.total(), a.loss_part, grid, L.bp(), L.bv2(), d_grads, d_loss, (const unsigned*)nullptr, (unsigned*)nullptr, a.drop.step_counter,
                         0LL, (long long)L.total(), 1, fin_adam);
    hipError_t eb = hipGetLastError();
    return eb == hipSuccess ? PINN_OK : (int)eb;
  }
  if (net->precision >= PINN_PREC_F32X6 || H > 256) {
    // split-operand chain on the 16-bit matrix cores (wide nets: also bf16-mixed); the weight-gradient and finalize kernels below are shared
    if (phases & (PINN_PHASE_CHAIN | PINN_PHASE_CHAIN_FWD | PINN_PHASE_CHAIN_BWD)) {
      const unsigned which = (phases & PINN_PHASE_CHAIN) ? 3u : (((phases & PINN_PHASE_CHAIN_FWD) ? 1u : 0u) | ((phases & PINN_PHASE_CHAIN_BWD) ? 2u : 0u));
      TrainBuffers b{};
      b.stash_h = a.stash_h; b.stash_v1 = a.stash_v1; b.stash_v2 = a.stash_v2;
      b.dpre_h = a.dpre_h; b.dpre_v1 = a.dpre_v1; b.dpre_v2 = a.dpre_v2;
      b.keep = a.keep; b.du = a.du; b.dz = a.dz; b.loss_part = a.loss_part;
      b.slabs = (float*)(base + w.off_slabs); b.t16 = w.t16; b.n_slices = w.n_slices;
      b.amax = (unsigned*)(base + w.off_amax);
      b.emax = b.amax + 1; b.rowmeta = base + w.off_rowmeta; b.qboost = row_scale_boost(a.drop, nh); b.stash_x = base + w.off_stash_x;
      rc = H > 256 ? launch_train_chain_wide(net, d_params, d_x, d_y, n_rows, n_global, a.drop, b, &grid, stream)
                   : launch_train_chain_x6(net, d_params, d_x, d_y, n_rows, n_global, a.drop, b, which, &grid, stream);
      if (rc) return rc;
    } else if (H > 256) {
      const long long t4 = w.t16 / 4;
      grid = (int)(t4 < 1024 ? (t4 < 1 ? 1 : t4) : 1024);
    } else {
      const long long nt = (n_rows + 127) / 128;
      grid = (int)(nt < cu_count() ? nt : cu_count());
    }
  } else if (phases & PINN_PHASE_CHAIN) {
    const bool bits = a.drop.mode == PINN_DROP_BITS;
    if (H == 256) {
      if (bits) hipLaunchKernelGGL((train_chain_kernel<256, true>), dim3(grid), dim3(kThreads), 0, st, a);
      else hipLaunchKernelGGL((train_chain_kernel<256, false>), dim3(grid), dim3(kThreads), 0, st, a);
    } else {
      if (bits) hipLaunchKernelGGL((train_chain_kernel<128, true>), dim3(grid), dim3(kThreads), 0, st, a);
      else hipLaunchKernelGGL((train_chain_kernel<128, false>), dim3(grid), dim3(kThreads), 0, st, a);
    }
  }

  float* slabs = (float*)(base + w.off_slabs);
  const long long tot = L.total();
  const long long hs = (long long)w.t16 * H * 16;   // floats per hidden-layer stash
  if (phases & (PINN_PHASE_WGRAD_TAIL | PINN_PHASE_WGRAD_HEAD)) {
    // tail = the layers whose d pre-activations the backward chain finishes first (last hidden layer, heads); head = the rest
    const bool do_tail = phases & PINN_PHASE_WGRAD_TAIL, do_head = phases & PINN_PHASE_WGRAD_HEAD;
    auto in_part = [&](int l) { return l == nh - 1 ? do_tail : do_head; };      // hidden layer l >= 1
    // small row counts: the launches of the layers are independent and short -- side by side on up to four streams
    const bool multi = w.t16 < kMultiT16 && net->precision == PINN_PREC_F32X6 && H == 256 && nh - 1 + 3 <= kMaxWgradProblems;
    const bool fan = !multi && w.t16 < kFanOutT16 && g_fan.init();
    int n_launch = 0;
    bool used[3] = {false, false, false};
    if (fan && hipEventRecord(g_fan.fork, st) != hipSuccess) return (int)hipGetLastError();
    auto pick = [&]() -> hipStream_t {
      if (!fan) return st;
      const int k = n_launch++ % 4;
      if (k == 0) return st;
      if (!used[k - 1]) { (void)hipStreamWaitEvent(g_fan.side[k - 1], g_fan.fork, 0); used[k - 1] = true; }
      return g_fan.side[k - 1];
    };
    WgradPMulti mp{};
    auto issue_p = [&](const WgradPArgs& pa, int kind) -> int {
      if (!multi) return dispatch_wgrad_p(pa, (void*)pick());
      mp.p[mp.n] = pa; mp.kind[mp.n] = kind; ++mp.n;
      return PINN_OK;
    };
    WgradArgs g{};
    g.x = d_x; g.n_rows = n_rows; g.t16 = w.t16; g.n_slices = w.n_slices; g.slab_stride = tot;
    g.amax = (const unsigned*)(base + w.off_amax);
    // layer 0: dW0 = dpre_0 x^T
    g.P = a.dpre_h; g.Q = nullptr; g.OUT = H; g.IN = 8; g.dW = slabs + L.w0(); g.db = slabs + L.b0();
    g.s1 = nullptr; g.dvq = nullptr; g.s2 = nullptr; g.R = nullptr; g.dvr = nullptr;
    const bool packed0 = net->precision == PINN_PREC_F32X6;                   // layer 0 from the packed operands too (below)
    if (do_head && !packed0) { if ((rc = dispatch_wgrad(g, pick()))) return rc; }
    // every layer but the input one: split-bf16 products on the matrix cores for PINN_PREC_F32X6
    // operand split of the weight-gradient kernels: 0 = exact fp32 kernels; 3 = three bf16 parts, six products (x6); 4 = two fp16
    // parts under the common scale the X3 backward kernels measured (PINN_PREC_F32X6); 1 = bf16-mixed (wide nets)
    if (net->precision == PINN_PREC_F32X6) {
      // packed stash: the chain kernels (fused nets) / layer kernels (wide nets) left every operand as fp16 fragments
      // (pinn_x6_core.h), the row scales in the meta records
      WgradPArgs p{};
      p.meta = base + w.off_rowmeta; p.emax = (const unsigned*)(base + w.off_amax) + 1; p.qboost = row_scale_boost(a.drop, nh);
      p.t16 = w.t16; p.n_slices = w.n_slices; p.slab_stride = tot; p.q_log2 = 3;
      const long long hb = hs * 4;      // bytes per hidden-layer stash
      if (do_head) {
        // layer 0: dW0 = d pre_0^T x with the rows as a packed group of their own (8 features of 32, stored as x / 16: any
        // |x| < 256 survives the row scale in fp16); [H][8] of the [H][32] product is written
        p.P = (const char*)a.dpre_h; p.Q = (const char*)(base + w.off_stash_x); p.OUT = H; p.IN = 32; p.dW = slabs + L.w0(); p.db = slabs + L.b0();
        p.ldW = 8; p.n_cols = 8; p.q_log2 = -4;
        if ((rc = issue_p(p, 0))) return rc;
        p.ldW = 0; p.n_cols = 0; p.q_log2 = 3;
      }
      for (int l = nh - 1; l >= 1; --l) {
        if (!in_part(l)) continue;
        p.P = (const char*)a.dpre_h + l * hb; p.Q = (const char*)a.stash_h + (l - 1) * hb; p.OUT = H; p.IN = H; p.dW = slabs + L.w(l); p.db = slabs + L.b(l);
        if ((rc = issue_p(p, 1))) return rc;
      }
      if (do_tail) {
        // variance head layer 0 (+ predict weight: dw_p[j] = sum du * h_last[j])
        p.P = (const char*)a.dpre_v1; p.Q = (const char*)a.stash_h + (nh - 1) * hb; p.OUT = H / 2; p.IN = H; p.dW = slabs + L.wv0(); p.db = slabs + L.bv0();
        p.dvq = slabs + L.wp();
        if ((rc = issue_p(p, 2))) return rc;
        // variance head layer 1 (+ final weight: dwv2[i] = sum dz * v2[i], fp32 operands)
        p.P = (const char*)a.dpre_v2; p.Q = (const char*)a.stash_v1; p.OUT = H / 4; p.IN = H / 2; p.dW = slabs + L.wv1(); p.db = slabs + L.bv1();
        p.dvq = nullptr; p.s2 = a.dz; p.R = a.stash_v2; p.dvr = slabs + L.wv2();
        if ((rc = issue_p(p, 3))) return rc;
      }
      if (multi && mp.n > 0 && (rc = dispatch_wgrad_p_multi(mp, (void*)st))) return rc;
    } else {
    const int ns = net->precision == PINN_PREC_F32X6 ? 4 : (net->precision == PINN_PREC_F32X6_G6 ? 3 : (net->precision == PINN_PREC_BF16 ? 1 : 0));
    auto wgrad = [&](const WgradArgs& wa) { hipStream_t s_ = pick(); return ns ? dispatch_wgrad_x6(wa, ns, (void*)s_) : dispatch_wgrad(wa, s_); };
    for (int l = nh - 1; l >= 1; --l) {
      if (!in_part(l)) continue;
      g.P = a.dpre_h + l * hs; g.Q = a.stash_h + (l - 1) * hs; g.OUT = H; g.IN = H; g.dW = slabs + L.w(l); g.db = slabs + L.b(l);
      if ((rc = wgrad(g))) return rc;
    }
    if (do_tail) {
      // variance head layer 0 (+ predict weight: dw_p[j] = sum du * h_last[j])
      g.P = a.dpre_v1; g.Q = a.stash_h + (nh - 1) * hs; g.OUT = H / 2; g.IN = H; g.dW = slabs + L.wv0(); g.db = slabs + L.bv0();
      g.s1 = a.du; g.dvq = slabs + L.wp();
      if ((rc = wgrad(g))) return rc;
      // variance head layer 1 (+ final weight: dwv2[i] = sum dz * v2[i])
      g.P = a.dpre_v2; g.Q = a.stash_v1; g.OUT = H / 4; g.IN = H / 2; g.dW = slabs + L.wv1(); g.db = slabs + L.bv1();
      g.s1 = nullptr; g.dvq = nullptr; g.s2 = a.dz; g.R = a.stash_v2; g.dvr = slabs + L.wv2();
      if ((rc = wgrad(g))) return rc;
    }
    }
    for (int k = 0; k < 3; ++k)
      if (used[k]) { (void)hipEventRecord(g_fan.join[k], g_fan.side[k]); (void)hipStreamWaitEvent(st, g_fan.join[k], 0); }
  }

  if (phases & (PINN_PHASE_REDUCE_TAIL | PINN_PHASE_REDUCE_HEAD)) {
    // (the gradient word of the range record sits behind the pack kernel's words: pack_x6_kernel cleared it for this call)
    unsigned* rw = net->precision == PINN_PREC_F32X6 ? range_status_words(net) : nullptr;
    if (rw) rw += kRangePackBlocks * (2 * (nh - 1) + 4);
    const bool both = (phases & PINN_PHASE_REDUCE_TAIL) && (phases & PINN_PHASE_REDUCE_HEAD);
    auto reduce = [&](long long lo, long long hi, int with_loss) {
      hipLaunchKernelGGL(grad_finalize_kernel, dim3((unsigned)(((hi - lo) / 4 + kFinGroups - 1) / kFinGroups)), dim3(256), 0, st, slabs, w.n_slices, tot,
                         a.loss_part, grid, L.bp(), L.bv2(), d_grads, d_loss, (const unsigned*)(base + w.off_amax), rw, a.drop.step_counter, lo, hi,
                         with_loss, fin_adam);
    };
    if (both) reduce(0, tot, 1);                               // one launch over the whole vector
    else if (phases & PINN_PHASE_REDUCE_TAIL) reduce(split, tot, 1);
    else if (split > 0) reduce(0, split, 0);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

extern "C" int pinn_mlp_train_grads_phases(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                                           long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                                           double* d_loss, void* d_work, size_t work_bytes, void* stream, unsigned phases) {
  return train_grads_impl(net, d_params, d_x, d_y, n_rows, n_global, drop, d_grads, d_loss, d_work, work_bytes, stream, phases, nullptr);
}

// One optimizer step of train_dnn (01:949-954) as ONE launch sequence that can be captured and replayed: pinn_mlp_train_grads
// with the Adam step applied by the slab reduction's own launch (at the reference's row counts a step is a chain of short
// dependent launches, and the separate optimizer launch was 5 of its ~130 us).  d_params is updated in place.
extern "C" int pinn_mlp_train_step_dev(const pinn_net_t* net, float* d_params, const float* d_x, const float* d_y,
                                       long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                                       double* d_loss, void* d_work, size_t work_bytes, float* d_m, float* d_v,
                                       const float* d_coeffs, void* stream) {
  if (!net || !drop || !drop->d_step_counter || !d_m || !d_v || !d_coeffs) return PINN_E_ARG;
  if (!(net->precision >= PINN_PREC_F32X6 && net->hidden <= 256)) return PINN_E_ARCH;      // (the kernels that leave the counter's snapshot)
  if (((unsigned long long)d_params | (unsigned long long)d_m | (unsigned long long)d_v) & 15) return PINN_E_ARG;
  const FinAdam fa{d_params, d_m, d_v, d_coeffs, nullptr, 0.0f, 1.0f};
  return train_grads_impl(net, d_params, d_x, d_y, n_rows, n_global, drop, d_grads, d_loss, d_work, work_bytes, stream, PINN_PHASE_ALL, &fa);
}

// The same with the step's scalars by value (lr from StepLR, step 1-based: pinn_adam_step's arguments): pinn_mlp_train_grads +
// pinn_adam_step in one launch sequence, for callers that launch step by step.  Every precision and width.
extern "C" void pinn_adam_coeffs(float lr, int step, float* step_size, float* bc2_sqrt);
extern "C" int pinn_mlp_train_step(const pinn_net_t* net, float* d_params, const float* d_x, const float* d_y,
                                   long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                                   double* d_loss, void* d_work, size_t work_bytes, float* d_m, float* d_v, float lr, int step,
                                   void* stream) {
  if (!net || !d_m || !d_v || step < 1) return PINN_E_ARG;
  if (((unsigned long long)d_params | (unsigned long long)d_m | (unsigned long long)d_v) & 15) return PINN_E_ARG;
  FinAdam fa{d_params, d_m, d_v, nullptr, nullptr, 0.0f, 1.0f};
  pinn_adam_coeffs(lr, step, &fa.step_size, &fa.bc2_sqrt);
  return train_grads_impl(net, d_params, d_x, d_y, n_rows, n_global, drop, d_grads, d_loss, d_work, work_bytes, stream, PINN_PHASE_ALL, &fa);
}

extern "C" int pinn_mlp_train_grads(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                                    long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                                    double* d_loss, void* d_work, size_t work_bytes, void* stream) {
  return pinn_mlp_train_grads_phases(net, d_params, d_x, d_y, n_rows, n_global, drop, d_grads, d_loss, d_work, work_bytes,
                                     stream, PINN_PHASE_ALL);
}
