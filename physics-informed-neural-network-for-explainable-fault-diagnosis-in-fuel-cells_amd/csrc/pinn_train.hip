// pinn_train.hip -- train_dnn's forward + aleatoric_loss + backward (01:949-953) for gfx950.
//
// Three kernels per step, all on v_mfma_f32_32x32x2_f32 (exact fp32):
//   K4/K5 train_chain_kernel : per 32-row wave tile, forward chain (activations in registers,
//        post-dropout activations stashed to HBM as [tile32][feature][32]), NLL loss and its
//        gradient, backward (dgrad) chain; writes every layer's d(pre-activation) in the same
//        tiled layout.  Keep-masks are drawn once (Philox / injected bits) and parked in LDS
//        between the forward and backward halves.
//   K6 wgrad_kernel : dW = dpre^T . h_prev as a split-K GEMM over row tiles (K = rows), operands
//        read straight from the tiled stash as MFMA fragments, fp32 partial slabs per K-slice;
//        bias / vector-head gradients ride along as per-lane sums.
//   grad_finalize_kernel : fixed-order sum of the slabs -> flat gradient (bitwise reproducible,
//        no float atomics), already divided by the GLOBAL row count (data-parallel shards
//        all-reduce it with SUM).
#include "pinn_mlp_core.h"

namespace pinn {

constexpr int kMaxSlices = 256;
constexpr int kLossTerms = 8;   // nll, |logvar|, (y-u)^2, du, dz, spare...

struct TrainArgs {
  const float* params;
  const float* x;
  const float* y;
  long long n_rows, n_global;
  int H, nh;
  DropDev drop;
  float* stash_h;    // [nh][T32][H][32]    post-dropout activations of hidden layers
  float* stash_v1;   // [T32][H/2][32]
  float* stash_v2;   // [T32][H/4][32]
  float* dpre_h;     // [nh][T32][H][32]    d loss / d pre-activation
  float* dpre_v1;
  float* dpre_v2;
  float* du;         // [T32*32]  d loss / d u      (already / n_global)
  float* dz;         // [T32*32]  d loss / d z
  double* loss_part; // [grid][kLossTerms]
  long long t32;     // 32-row tiles (padded to whole 128-row workgroup tiles)
};

template <int NBLK>
__device__ __forceinline__ void store_tiled(float* __restrict__ base, long long tile32, int F, const f32x16 (&v)[NBLK],
                                            int lane) {
  // element (feature f, row n) of 32-row tile `tile32` lives at ((tile32*F + f)*32 + n)
  const int hh = lane >> 5, n = lane & 31;
  float* p = base + (tile32 * F + 4 * hh) * 32 + n;
#pragma unroll
  for (int mt = 0; mt < NBLK; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[(mt * 32 + (r & 3) + 8 * (r >> 2)) * 32] = v[mt][r];
}

template <int NBLK>
__device__ __forceinline__ void load_tiled(const float* __restrict__ base, long long tile32, int F, f32x16 (&v)[NBLK],
                                           int lane) {
  const int hh = lane >> 5, n = lane & 31;
  const float* p = base + (tile32 * F + 4 * hh) * 32 + n;
#pragma unroll
  for (int mt = 0; mt < NBLK; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[mt][r] = p[(mt * 32 + (r & 3) + 8 * (r >> 2)) * 32];
}

// dpre = dh * scale * keep * (1 - a^2),  a = h / scale  (h = post-dropout activation)
template <int NBLK>
__device__ __forceinline__ void tanh_drop_backward(f32x16 (&dh)[NBLK], const f32x16 (&h)[NBLK], const unsigned short* keep,
                                                   float scale) {
#pragma unroll
  for (int mt = 0; mt < NBLK; ++mt) {
    const unsigned k = keep[mt * 64];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float a = h[mt][r] / scale;
      const float g = dh[mt][r] * scale * (1.0f - a * a);
      dh[mt][r] = ((k >> r) & 1u) ? g : 0.0f;
    }
  }
}

template <int H>
__global__ __launch_bounds__(kThreads, 1) void train_chain_kernel(TrainArgs a) {
  constexpr int NB = H / 32, NB2 = H / 64, NB4 = H / 128;
  __shared__ __attribute__((aligned(16))) char lds_w[2 * kChunkBytes];
  __shared__ ChunkDesc tab[kMaxChunks];
  __shared__ unsigned short keep_lds[4][8 * (H / 32) + H / 64][64];   // [wave][module*NB + block][lane]
  __shared__ double red[4][kLossTerms];
  ParamLayout L{a.H, a.nh};
  const int n_fwd = (a.nh - 1) * NB + NB + NB2;
  const int n_bwd = NB4 + NB2 + (a.nh - 1) * NB;
  if (threadIdx.x == 0) {
    int k = build_forward_chunks(tab, L, 0);
    build_backward_chunks(tab, L, k);
  }
  __syncthreads();
  Pipe pipe;
  pipe.params = a.params; pipe.tab = tab; pipe.lds = lds_w; pipe.n = n_fwd + n_bwd;
  pipe.prime();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5;
  const float* __restrict__ P = a.params;
  unsigned short* keep = &keep_lds[wave][0][lane];
  const int mode = a.drop.mode;
  const float inv_n = (float)(1.0 / (double)a.n_global);
  float s_nll = 0.f, s_abs = 0.f, s_mse = 0.f, s_du = 0.f, s_dz = 0.f;

  const long long n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t32 = tile * 4 + wave;
    const long long lrow = t32 * 32 + (lane & 31);
    const bool valid = lrow < a.n_rows;
    const long long srow = valid ? lrow : a.n_rows - 1;
    const long long grow = a.drop.row_offset + lrow;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2];
    const f32x4 xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    const float yv = a.y[srow];

    // ------------------------------------------------------------------ forward
    f32x16 h[NB];
    float u, z;
    {
      unsigned kb[NB];
      layer_input<NB>(h, P + L.w0(), P + L.b0(), xa, xb, lane);
      epilogue_tanh_drop<NB>(h, a.drop, mode, 0, hh, grow, srow, a.n_rows, 0u, kb);
#pragma unroll
      for (int mt = 0; mt < NB; ++mt) keep[mt * 64] = (unsigned short)kb[mt];
      store_tiled<NB>(a.stash_h, t32, H, h, lane);
#pragma unroll 1
      for (int l = 1; l < L.nh; ++l) {
        f32x16 acc[NB];
        load_bias<NB>(acc, P + L.b(l), hh);
        layer_forward<NB, NB>(acc, h, pipe, lane);
        epilogue_tanh_drop<NB>(acc, a.drop, mode, l, hh, grow, srow, a.n_rows, 0u, kb);
#pragma unroll
        for (int mt = 0; mt < NB; ++mt) { keep[(l * NB + mt) * 64] = (unsigned short)kb[mt]; h[mt] = acc[mt]; }
        store_tiled<NB>(a.stash_h + (long long)l * a.t32 * H * 32, t32, H, h, lane);
      }
    }
    u = head_dot<NB>(h, P + L.wp(), hh) + P[L.bp()];
    f32x16 v2[NB4];
    {
      f32x16 v1[NB2];
      unsigned kb[NB2];
      load_bias<NB2>(v1, P + L.bv0(), hh);
      layer_forward<NB, NB2>(v1, h, pipe, lane);
      epilogue_tanh_drop<NB2>(v1, a.drop, mode, L.nh, hh, grow, srow, a.n_rows, 0u, kb);
#pragma unroll
      for (int mt = 0; mt < NB2; ++mt) keep[(L.nh * NB + mt) * 64] = (unsigned short)kb[mt];
      store_tiled<NB2>(a.stash_v1, t32, H / 2, v1, lane);
      load_bias<NB4>(v2, P + L.bv1(), hh);
      layer_forward<NB2, NB4>(v2, v1, pipe, lane);
    }
#pragma unroll
    for (int mt = 0; mt < NB4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) v2[mt][r] = tanh_f32(v2[mt][r]);
    store_tiled<NB4>(a.stash_v2, t32, H / 4, v2, lane);
    z = head_dot<NB4>(v2, P + L.wv2(), hh) + P[L.bv2()];

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
        if (hh == 0) {
          s_nll += 0.5f * prec * e * e + 0.5f * s;
          s_abs += fabsf(s);
          s_mse += e * e;
          s_du += du;
          s_dz += dz;
        }
      }
      if (lane < 32) { a.du[t32 * 32 + lane] = du; a.dz[t32 * 32 + lane] = dz; }
    }

    // ------------------------------------------------------------------ backward: variance head
    f32x16 dh[NB];
    {
      f32x16 dpv1[NB2];
      {
        // d pre_v2 = wv2[f] * dz * (1 - v2^2)
#pragma unroll
        for (int mt = 0; mt < NB4; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(P + L.wv2() + mt * 32 + 8 * q + 4 * hh);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float t = v2[mt][4 * q + j];
              v2[mt][4 * q + j] = w[j] * dz * (1.0f - t * t);
            }
          }
        store_tiled<NB4>(a.dpre_v2, t32, H / 4, v2, lane);
#pragma unroll
        for (int mt = 0; mt < NB2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dpv1[mt][r] = 0.0f;
        layer_backward<NB4, NB2>(dpv1, v2, pipe, lane, H / 2);
      }
      {
        f32x16 v1[NB2];
        load_tiled<NB2>(a.stash_v1, t32, H / 2, v1, lane);
        tanh_drop_backward<NB2>(dpv1, v1, keep + (L.nh * NB) * 64, mode != PINN_DROP_NONE ? a.drop.scale[L.nh] : 1.0f);
      }
      store_tiled<NB2>(a.dpre_v1, t32, H / 2, dpv1, lane);
      // d h_last = w_p * du + Wv0^T d pre_v1
#pragma unroll
      for (int mt = 0; mt < NB; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(P + L.wp() + mt * 32 + 8 * q + 4 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) dh[mt][4 * q + j] = w[j] * du;
        }
      layer_backward<NB2, NB>(dh, dpv1, pipe, lane, H);
    }

    // ------------------------------------------------------------------ backward: hidden layers nh-1 .. 0
#pragma unroll 1
    for (int l = L.nh - 1; l >= 0; --l) {
      {
        f32x16 hl[NB];
        load_tiled<NB>(a.stash_h + (long long)l * a.t32 * H * 32, t32, H, hl, lane);
        tanh_drop_backward<NB>(dh, hl, keep + (l * NB) * 64, mode != PINN_DROP_NONE ? a.drop.scale[l] : 1.0f);
      }
      store_tiled<NB>(a.dpre_h + (long long)l * a.t32 * H * 32, t32, H, dh, lane);
      if (l > 0) {
        f32x16 acc[NB];
#pragma unroll
        for (int mt = 0; mt < NB; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        layer_backward<NB, NB>(acc, dh, pipe, lane, H);
#pragma unroll
        for (int mt = 0; mt < NB; ++mt) dh[mt] = acc[mt];
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
// ---------------------------------------------------------------------------------------
struct WgradArgs {
  const float* P;    // [T32][OUT][32]   d pre-activation of this layer
  const float* Q;    // [T32][IN][32]    its input activation (or nullptr: read x rows, IN = 8)
  const float* x;    // [n_rows][8] when Q == nullptr
  long long n_rows;
  int OUT, IN;
  long long t32;
  int n_slices;
  long long slab_stride;    // floats between consecutive slices' slabs (= padded param count)
  float* dW;                // slab of slice 0: [OUT][IN] row-major at the parameter's offset
  float* db;                // slab of slice 0: [OUT]
  const float* s1; float* dvq;                    // optional: dvq[j] = sum_rows s1[row] Q[j][row]
  const float* s2; const float* R; float* dvr;    // optional: dvr[i] = sum_rows s2[row] R[i][row]
};

template <int TI, int TJ, int WI, int WJ, bool QX>
__global__ __launch_bounds__(kThreads, 1) void wgrad_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= WI * WJ) return;
  const int wi = wave / WJ, wj = wave % WJ;
  const int hh = lane >> 5, i = lane & 31;
  const int i0 = wi * TI * 32, j0 = wj * TJ * 32;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
  float bsum[TI], vq[TJ], vr[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) { bsum[ti] = 0.f; vr[ti] = 0.f; }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) vq[tj] = 0.f;

  const int slice = blockIdx.x;
  const long long per = (a.t32 + a.n_slices - 1) / a.n_slices;
  const long long t_begin = slice * per;
  long long t_end = t_begin + per;
  if (t_end > a.t32) t_end = a.t32;

  for (long long t = t_begin; t < t_end; ++t) {
    // k-step s pairs rows (s, s + 16) of the tile: lane half hh supplies row 16*hh + s
    f32x4 af[TI][4], bf[TJ][4];
    const float* pP = a.P + ((t * a.OUT + i0 + i) * 32 + 16 * hh);
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) af[ti][sg] = *reinterpret_cast<const f32x4*>(pP + ti * 1024 + sg * 4);
    if (!QX) {
      const float* pQ = a.Q + ((t * a.IN + j0 + i) * 32 + 16 * hh);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) bf[tj][sg] = *reinterpret_cast<const f32x4*>(pQ + tj * 1024 + sg * 4);
    } else {
      // Q = x^T: feature j = lane & 31 (< 8 valid), rows t*32 + 16*hh + s read from row-major x
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          long long row = t * 32 + 16 * hh + 4 * sg + j;
          if (row >= a.n_rows) row = a.n_rows - 1;     // dpre of such rows is exactly 0
          bf[0][sg][j] = (i < 8) ? a.x[row * 8 + i] : 0.0f;
        }
    }
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = PINN_MFMA(af[ti][sg][j], bf[tj][sg][j], acc[ti][tj]);
    if (wj == 0) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) bsum[ti] += (af[ti][sg][0] + af[ti][sg][1]) + (af[ti][sg][2] + af[ti][sg][3]);
      if (a.dvr) {
        const float* pR = a.R + ((t * a.OUT + i0 + i) * 32 + 16 * hh);
        const float* ps = a.s2 + t * 32 + 16 * hh;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(ps + sg * 4);
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(pR + ti * 1024 + sg * 4);
            vr[ti] += (sv[0] * rv[0] + sv[1] * rv[1]) + (sv[2] * rv[2] + sv[3] * rv[3]);
          }
        }
      }
    }
    if (wi == 0 && a.dvq) {
      const float* ps = a.s1 + t * 32 + 16 * hh;
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(ps + sg * 4);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          vq[tj] += (sv[0] * bf[tj][sg][0] + sv[1] * bf[tj][sg][1]) + (sv[2] * bf[tj][sg][2] + sv[3] * bf[tj][sg][3]);
      }
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
      const float b = bsum[ti] + __shfl_xor(bsum[ti], 32, 64);
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

// grads[e] = sum_s slab[s][e] in fixed order; scalar head biases and the loss sums from the chain partials
__global__ __launch_bounds__(256) void grad_finalize_kernel(const float* __restrict__ slabs, int n_slices, long long total,
                                                            const double* __restrict__ loss_part, int n_parts,
                                                            long long off_bp, long long off_bv2, float* __restrict__ grads,
                                                            double* __restrict__ loss_out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    const bool pad = (e > off_bp && e < off_bp + 4) || (e > off_bv2 && e < off_bv2 + 4);
    if (e != off_bp && e != off_bv2) {
      float s = 0.0f;
      if (!pad)
        for (int k = 0; k < n_slices; ++k) s += slabs[(long long)k * total + e];
      grads[e] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < kLossTerms) {
    double t = 0.0;
    for (int b = 0; b < n_parts; ++b) t += loss_part[(long long)b * kLossTerms + threadIdx.x];
    if (threadIdx.x < 4) loss_out[threadIdx.x] = t;    // nll sum, |logvar| sum, (y-u)^2 sum, (sum du)
    if (threadIdx.x == 3) grads[off_bp] = (float)t;    // d loss / d b_p  = sum du
    if (threadIdx.x == 4) grads[off_bv2] = (float)t;   // d loss / d bv_2 = sum dz
  }
}

struct Workspace {
  long long t32;
  size_t off_stash_h, off_stash_v1, off_stash_v2, off_dpre_h, off_dpre_v1, off_dpre_v2, off_du, off_dz, off_loss, off_slabs;
  int n_slices;
  size_t total;
};

static Workspace plan_workspace(const pinn_net_t* net, long long n_rows) {
  Workspace w;
  const long long H = net->hidden, nh = net->n_hidden;
  w.t32 = (n_rows + kTileRows - 1) / kTileRows * 4;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  w.off_stash_h = take((size_t)nh * w.t32 * H * 32 * 4);
  w.off_stash_v1 = take((size_t)w.t32 * (H / 2) * 32 * 4);
  w.off_stash_v2 = take((size_t)w.t32 * (H / 4) * 32 * 4);
  w.off_dpre_h = take((size_t)nh * w.t32 * H * 32 * 4);
  w.off_dpre_v1 = take((size_t)w.t32 * (H / 2) * 32 * 4);
  w.off_dpre_v2 = take((size_t)w.t32 * (H / 4) * 32 * 4);
  w.off_du = take((size_t)w.t32 * 32 * 4);
  w.off_dz = take((size_t)w.t32 * 32 * 4);
  w.off_loss = take((size_t)1024 * kLossTerms * 8);
  w.n_slices = (int)(w.t32 < kMaxSlices ? (w.t32 < 1 ? 1 : w.t32) : kMaxSlices);
  ParamLayout L{(int)H, (int)nh};
  w.off_slabs = take((size_t)w.n_slices * L.total() * 4);
  w.total = o;
  return w;
}

static int check_net_t(const pinn_net_t* net) {
  if (!net) return PINN_E_ARG;
  if (net->n_in != 8) return PINN_E_ARCH;
  if (net->hidden != 128 && net->hidden != 256) return PINN_E_ARCH;
  if (net->n_hidden < 1 || net->n_hidden > 8) return PINN_E_ARCH;
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
static void launch_wgrad(const WgradArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WI, WJ, QX>), dim3(a.n_slices), dim3(kThreads), 0, st, a);
}

// pick the wave tiling for an [OUT x IN] gradient (tiles of 32x32; IN = 8 is padded to one tile)
static int dispatch_wgrad(const WgradArgs& a, hipStream_t st) {
  const int to = a.OUT / 32, ti = (a.IN + 31) / 32;
  if (a.Q == nullptr) {
    if (to == 8) launch_wgrad<2, 1, 4, 1, true>(a, st);
    else if (to == 4) launch_wgrad<1, 1, 4, 1, true>(a, st);
    else return PINN_E_ARCH;
    return PINN_OK;
  }
  if (to == 8 && ti == 8) launch_wgrad<4, 4, 2, 2, false>(a, st);
  else if (to == 4 && ti == 8) launch_wgrad<2, 4, 2, 2, false>(a, st);
  else if (to == 2 && ti == 4) launch_wgrad<1, 2, 2, 2, false>(a, st);
  else if (to == 4 && ti == 4) launch_wgrad<2, 2, 2, 2, false>(a, st);
  else if (to == 2 && ti == 4) launch_wgrad<1, 2, 2, 2, false>(a, st);
  else if (to == 1 && ti == 2) launch_wgrad<1, 1, 1, 2, false>(a, st);
  else return PINN_E_ARCH;
  return PINN_OK;
}

}  // namespace pinn

using namespace pinn;

extern "C" size_t pinn_train_workspace_bytes(const pinn_net_t* net, long long n_rows) {
  if (check_net_t(net) != PINN_OK || n_rows < 0) return 0;
  return plan_workspace(net, n_rows).total;
}

extern "C" int pinn_mlp_train_grads_phases(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                                           long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                                           double* d_loss, void* d_work, size_t work_bytes, void* stream, unsigned phases) {
  int rc = check_net_t(net);
  if (rc) return rc;
  if (!d_params || !d_x || !d_y || !d_grads || !d_loss || !d_work || n_rows <= 0 || n_global < n_rows) return PINN_E_ARG;
  const Workspace w = plan_workspace(net, n_rows);
  if (work_bytes < w.total) return PINN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();   // drop a stale error left by another HIP user of this thread
  char* base = (char*)d_work;
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};

  TrainArgs a{};
  a.params = d_params; a.x = d_x; a.y = d_y; a.n_rows = n_rows; a.n_global = n_global; a.H = H; a.nh = nh;
  // dropout conversion (same rules as the forward entry points)
  {
    DropDev& d = a.drop;
    d.mode = PINN_DROP_NONE; d.bits = nullptr; d.words = 0; d.nb = H / 32; d.seed_lo = d.seed_hi = 0; d.stream = 0; d.row_offset = 0;
    for (int l = 0; l < kMaxDrop; ++l) { d.thr[l] = 0; d.scale[l] = 1.0f; }
    if (drop) {
      if (drop->mode < PINN_DROP_NONE || drop->mode > PINN_DROP_BITS) return PINN_E_ARG;
      d.mode = drop->mode; d.row_offset = drop->row_offset;
      if (drop->mode != PINN_DROP_NONE) {
        for (int l = 0; l <= nh; ++l) {
          const float p = drop->p[l];
          if (!(p >= 0.0f && p < 1.0f)) return PINN_E_ARG;
          double t = floor((double)p * 65536.0 + 0.5);
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
  a.du = (float*)(base + w.off_du); a.dz = (float*)(base + w.off_dz);
  a.loss_part = (double*)(base + w.off_loss);
  a.t32 = w.t32;
  const long long n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  int grid = (int)(n_tiles < cu_count() ? n_tiles : cu_count());
  if (grid > 1024) grid = 1024;
  if (phases & PINN_PHASE_CHAIN) {
    if (H == 256) hipLaunchKernelGGL((train_chain_kernel<256>), dim3(grid), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((train_chain_kernel<128>), dim3(grid), dim3(kThreads), 0, st, a);
  }

  float* slabs = (float*)(base + w.off_slabs);
  const long long tot = L.total();
  const long long hs = (long long)w.t32 * H * 32;   // floats per hidden-layer stash
  WgradArgs g{};
  g.x = d_x; g.n_rows = n_rows; g.t32 = w.t32; g.n_slices = w.n_slices; g.slab_stride = tot;
  if (phases & PINN_PHASE_WGRAD) {
  // layer 0: dW0 = dpre_0 x^T
  g.P = a.dpre_h; g.Q = nullptr; g.OUT = H; g.IN = 8; g.dW = slabs + L.w0(); g.db = slabs + L.b0();
  g.s1 = nullptr; g.dvq = nullptr; g.s2 = nullptr; g.R = nullptr; g.dvr = nullptr;
  if ((rc = dispatch_wgrad(g, st))) return rc;
  for (int l = 1; l < nh; ++l) {
    g.P = a.dpre_h + l * hs; g.Q = a.stash_h + (l - 1) * hs; g.OUT = H; g.IN = H; g.dW = slabs + L.w(l); g.db = slabs + L.b(l);
    if ((rc = dispatch_wgrad(g, st))) return rc;
  }
  // variance head layer 0 (+ predict weight: dw_p[j] = sum du * h_last[j])
  g.P = a.dpre_v1; g.Q = a.stash_h + (nh - 1) * hs; g.OUT = H / 2; g.IN = H; g.dW = slabs + L.wv0(); g.db = slabs + L.bv0();
  g.s1 = a.du; g.dvq = slabs + L.wp();
  if ((rc = dispatch_wgrad(g, st))) return rc;
  // variance head layer 1 (+ final weight: dwv2[i] = sum dz * v2[i])
  g.P = a.dpre_v2; g.Q = a.stash_v1; g.OUT = H / 4; g.IN = H / 2; g.dW = slabs + L.wv1(); g.db = slabs + L.bv1();
  g.s1 = nullptr; g.dvq = nullptr; g.s2 = a.dz; g.R = a.stash_v2; g.dvr = slabs + L.wv2();
  if ((rc = dispatch_wgrad(g, st))) return rc;
  }

  if (phases & PINN_PHASE_REDUCE)
  hipLaunchKernelGGL(grad_finalize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, slabs, w.n_slices, tot,
                     a.loss_part, grid, L.bp(), L.bv2(), d_grads, d_loss);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

extern "C" int pinn_mlp_train_grads(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y,
                                    long long n_rows, long long n_global, const pinn_dropout_t* drop, float* d_grads,
                                    double* d_loss, void* d_work, size_t work_bytes, void* stream) {
  return pinn_mlp_train_grads_phases(net, d_params, d_x, d_y, n_rows, n_global, drop, d_grads, d_loss, d_work, work_bytes,
                                     stream, PINN_PHASE_ALL);
}
