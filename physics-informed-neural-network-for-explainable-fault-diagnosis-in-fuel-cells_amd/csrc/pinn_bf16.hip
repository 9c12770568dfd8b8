// pinn_bf16.hip -- bf16/fp32-mixed variants of the forward, MC-dropout and training kernels
// (pinn_net_t.precision = PINN_PREC_BF16).  See pinn_bf16_core.h for the layout and policy.
#include "pinn_bf16_core.h"

namespace pinn {

struct PackJobs {
  PackJob j[18];
  int n;
};

static int cu_count_b() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

__global__ __launch_bounds__(256) void pack_bf16_all(const float* __restrict__ params, __bf16* __restrict__ packed, PackJobs jobs) {
  const PackJob j = jobs.j[blockIdx.y];
  const long long n = (long long)j.rows * j.Kp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(e / j.Kp), q = (int)(e % j.Kp);
    const int k = (q & ~31) + pack_col(q & 31);
    float v = 0.0f;
    if (k < j.K) v = j.transposed ? params[j.src + (long long)k * j.src_ld + row] : params[j.src + (long long)row * j.src_ld + k];
    packed[j.dst + e] = (__bf16)v;
  }
}

// every bf16 call starts by re-packing the current fp32 weights (0.35 MB of bf16 per copy: a few microseconds)
void launch_pack(const pinn_net_t* net, const float* d_params, hipStream_t st) {
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};
  PackLayout K{H, nh};
  PackJobs jobs;
  int n = 0;
  auto add = [&](long long dst, long long src, int rows, int Kdim, int src_ld, int tr) {
    jobs.j[n++] = PackJob{dst, src, rows, Kdim, round_up64(Kdim), src_ld, tr};
  };
  for (int l = 1; l < nh; ++l) {
    add(K.w(l), L.w(l), H, H, H, 0);
    add(K.wt(l), L.w(l), H, H, H, 1);
  }
  add(K.wv0(), L.wv0(), H / 2, H, H, 0);
  add(K.wv0t(), L.wv0(), H, H / 2, H, 1);
  add(K.wv1(), L.wv1(), H / 4, H / 2, H / 2, 0);
  add(K.wv1t(), L.wv1(), H / 2, H / 4, H / 2, 1);
  jobs.n = n;
  hipLaunchKernelGGL(pack_bf16_all, dim3(64, n), dim3(256), 0, st, d_params, (__bf16*)net->d_packed, jobs);
}

template <int H, bool MC, bool kBits>
__global__ __launch_bounds__(kThreads, 2) void mlp_bf16_kernel(FwdArgs a, const float* packed) {
  __shared__ __attribute__((aligned(16))) char lds_w[2 * kChunkBytes];
  __shared__ ChunkDesc tab[kMaxChunks];
  __shared__ __attribute__((aligned(16))) float small[kMaxSmall];
  ParamLayout L{a.H, a.nh};
  PackLayout K{a.H, a.nh};
  if (threadIdx.x == 0) build_forward_chunks_bf16(tab, K, 0);
  load_small_params(small, a.params, L);
  Pipe pipe;
  pipe.params = packed; pipe.tab = tab; pipe.lds = lds_w; pipe.n = n_forward_slabs_bf16(a.H, a.nh);
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
    const StashPtrsBf16 st{};
    f32x4 v2[H / 64];
    if (!MC) {
      float u, z;
      forward_pass_bf16<H, false, kBits>(a.params, small, L, pipe, a.drop, c, xa, xb, st, u, z, v2);
      if (valid && lane < 16) {
        a.o0[lrow] = u;
        a.o1[lrow] = logf(softplus_f32(z) + 1e-6f);
      }
    } else {
      float u_eval = 0.f, mean = 0.f, m2 = 0.f, sl = 0.f;
#pragma unroll 1
      for (int t = -1; t < a.n_passes; ++t) {
        c.mode = (t < 0) ? PINN_DROP_NONE : a.drop.mode;
        c.pass = (unsigned)(t < 0 ? 0 : t);
        float u, z;
        forward_pass_bf16<H, false, kBits>(a.params, small, L, pipe, a.drop, c, xa, xb, st, u, z, v2);
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
        a.o1[lrow] = expf(0.5f * (sl * inv_t));
        a.o2[lrow] = sqrtf(var);
      }
    }
  }
}

int launch_forward_bf16(const pinn_net_t* net, const FwdArgs& a, bool mc, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();
  launch_pack(net, a.params, st);
  const long long n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  const int grid = (int)(n_tiles < 2 * cu_count_b() ? n_tiles : 2 * cu_count_b());
  const float* packed = (const float*)net->d_packed;
  const bool bits = a.drop.mode == PINN_DROP_BITS;
#define PINN_LAUNCH_B(HH, MCC, BB) hipLaunchKernelGGL((mlp_bf16_kernel<HH, MCC, BB>), dim3(grid), dim3(kThreads), 0, st, a, packed)
  if (net->hidden == 256) {
    if (mc) { if (bits) PINN_LAUNCH_B(256, true, true); else PINN_LAUNCH_B(256, true, false); }
    else    { if (bits) PINN_LAUNCH_B(256, false, true); else PINN_LAUNCH_B(256, false, false); }
  } else {
    if (mc) { if (bits) PINN_LAUNCH_B(128, true, true); else PINN_LAUNCH_B(128, true, false); }
    else    { if (bits) PINN_LAUNCH_B(128, false, true); else PINN_LAUNCH_B(128, false, false); }
  }
#undef PINN_LAUNCH_B
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}


// =======================================================================================
// training: forward + NLL + backward chain (bf16 MFMA inputs, bf16 stash)
// =======================================================================================
constexpr int kLossTermsB = 8;

struct TrainArgsB {
  const float* params;
  const float* packed;
  const float* x;
  const float* y;
  long long n_rows, n_global;
  int H, nh;
  DropDev drop;
  TrainBuffers b;
};

// dpre = dh * scale * keep * (1 - a^2), a = h / scale; returns the bf16 fragment of the pair
__device__ __forceinline__ bf16x8 tanh_drop_backward_frag(f32x4& d0, f32x4& d1, const f32x4& h0, const f32x4& h1, unsigned keep,
                                                          float scale, float inv_scale) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a0 = h0[r] * inv_scale, a1 = h1[r] * inv_scale;
    const float g0 = d0[r] * (scale * (1.0f - a0 * a0));
    const float g1 = d1[r] * (scale * (1.0f - a1 * a1));
    d0[r] = ((keep >> r) & 1u) ? g0 : 0.0f;
    d1[r] = ((keep >> (4 + r)) & 1u) ? g1 : 0.0f;
  }
  return make_frag(d0, d1);
}

template <int H, bool kBits>
__global__ __launch_bounds__(kThreads, 2) void train_chain_bf16_kernel(TrainArgsB a) {
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32;
  constexpr int NG4 = (H / 4 + 31) / 32;     // 32-groups of the H/4-wide layer (H = 128: one, half filled by a second block? no: H/4 = 32 -> 1)
  __shared__ __attribute__((aligned(16))) char lds_w[2 * kChunkBytes];
  __shared__ ChunkDesc tab[kMaxChunks];
  __shared__ double red[4][kLossTermsB];
  __shared__ __attribute__((aligned(16))) float small[kMaxSmall];
  ParamLayout L{a.H, a.nh};
  PackLayout K{a.H, a.nh};
  const SmallLayout S{a.H, a.nh};
  if (threadIdx.x == 0) {
    int k = build_forward_chunks_bf16(tab, K, 0);
    build_backward_chunks_bf16(tab, K, k);
  }
  load_small_params(small, a.params, L);
  Pipe pipe;
  pipe.params = a.packed; pipe.tab = tab; pipe.lds = lds_w;
  pipe.n = n_forward_slabs_bf16(a.H, a.nh) + n_backward_slabs_bf16(a.H, a.nh);
  pipe.prime();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4;
  const float* __restrict__ P = a.params;
  const bool drop = a.drop.mode != PINN_DROP_NONE;
  const float inv_n = (float)(1.0 / (double)a.n_global);
  const int n_groups = L.nh * NP + NP / 2;
  float s_nll = 0.f, s_abs = 0.f, s_mse = 0.f, s_du = 0.f, s_dz = 0.f;
  __bf16* const stash_h = (__bf16*)a.b.stash_h;
  __bf16* const stash_v1 = (__bf16*)a.b.stash_v1;
  __bf16* const stash_v2 = (__bf16*)a.b.stash_v2;
  __bf16* const dpre_h = (__bf16*)a.b.dpre_h;
  __bf16* const dpre_v1 = (__bf16*)a.b.dpre_v1;
  __bf16* const dpre_v2 = (__bf16*)a.b.dpre_v2;

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
    const StashPtrsBf16 st{stash_h, stash_v1, stash_v2, a.b.keep, a.b.t16, t16};
    const unsigned char* keep = a.b.keep + (t16 * n_groups) * 64 + lane;

    float u, z;
    f32x4 v2[NT4];
    forward_pass_bf16<H, true, kBits>(P, small, L, pipe, a.drop, c, xa, xb, st, u, z, v2);

    // aleatoric_loss (01:916-927) and its gradient, fp32
    float du = 0.f, dz = 0.f;
    {
      const float sp = softplus_f32(z);
      const float var = sp + 1e-6f;
      const float s = logf(var);
      const float prec = expf(-s);
      const float e = yv - u;
      if (valid) {
        du = -(prec * e) * inv_n;
        const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
        const float ds = (-0.5f * prec * e * e + 0.5f + 0.01f * sgn) * inv_n;
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
    }

    // ---------------- backward: variance head
    f32x4 dh[NT];
    {
      f32x4 dpv1[NT2];
      {
        bf16x8 g2[NG4];
        __bf16* sp = tiled_ptr_bf16(dpre_v2, t16, H / 4, lane);
#pragma unroll
        for (int t = 0; t < NT4; ++t) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(small + S.wv2() + t * 16 + 4 * kq);
#pragma unroll
          for (int r = 0; r < 4; ++r) v2[t][r] = w[r] * dz * (1.0f - v2[t][r] * v2[t][r]);
        }
#pragma unroll
        for (int g = 0; g < NG4; ++g) {
          f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          g2[g] = make_frag(v2[2 * g], (2 * g + 1 < NT4) ? v2[(2 * g + 1 < NT4) ? 2 * g + 1 : 0] : zero4);
        }
#pragma unroll
        for (int t = 0; t < NT4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) sp[(t * 16 + r) * 16] = (__bf16)v2[t][r];
        zero_blocks<NT2>(dpv1);
        layer_bf16<NG4, NT2>(dpv1, g2, pipe, lane);
      }
      bf16x8 g1[NP / 2];
      {
        const float scale = drop ? a.drop.scale[L.nh] : 1.0f, inv_scale = 1.0f / scale;
        const __bf16* hp = tiled_ptr_bf16(stash_v1, t16, H / 2, lane);
        __bf16* sp = tiled_ptr_bf16(dpre_v1, t16, H / 2, lane);
        f32x4 hl[NT2];
        unsigned kb[NP / 2];
#pragma unroll
        for (int fp = 0; fp < NP / 2; ++fp) { load_pair_bf16(hp, fp, hl[2 * fp], hl[2 * fp + 1]); kb[fp] = keep[(L.nh * NP + fp) * 64]; }
#pragma unroll
        for (int fp = 0; fp < NP / 2; ++fp) {
          g1[fp] = tanh_drop_backward_frag(dpv1[2 * fp], dpv1[2 * fp + 1], hl[2 * fp], hl[2 * fp + 1], kb[fp], scale, inv_scale);
          store_frag_bf16(sp, fp, g1[fp]);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(small + S.wp() + t * 16 + 4 * kq);
        dh[t] = w * du;
      }
      layer_bf16<NP / 2, NT>(dh, g1, pipe, lane);
    }

    // ---------------- backward: hidden layers nh-1 .. 0
#pragma unroll 1
    for (int l = L.nh - 1; l >= 0; --l) {
      bf16x8 gh[NP];
      {
        const float scale = drop ? a.drop.scale[l] : 1.0f, inv_scale = 1.0f / scale;
        const __bf16* hp = tiled_ptr_bf16(stash_h + (long long)l * a.b.t16 * H * 16, t16, H, lane);
        __bf16* sp = tiled_ptr_bf16(dpre_h + (long long)l * a.b.t16 * H * 16, t16, H, lane);
        f32x4 hl[NT];
        unsigned kb[NP];
#pragma unroll
        for (int fp = 0; fp < NP; ++fp) { load_pair_bf16(hp, fp, hl[2 * fp], hl[2 * fp + 1]); kb[fp] = keep[(l * NP + fp) * 64]; }
#pragma unroll
        for (int fp = 0; fp < NP; ++fp) {
          gh[fp] = tanh_drop_backward_frag(dh[2 * fp], dh[2 * fp + 1], hl[2 * fp], hl[2 * fp + 1], kb[fp], scale, inv_scale);
          store_frag_bf16(sp, fp, gh[fp]);
        }
      }
      if (l > 0) {
        zero_blocks<NT>(dh);
        layer_bf16<NP, NT>(dh, gh, pipe, lane);
      }
    }
  }

  float terms[5] = {s_nll, s_abs, s_mse, s_du, s_dz};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double v = (double)terms[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLossTermsB) {
    double t = 0.0;
    if (threadIdx.x < 5)
      for (int w = 0; w < 4; ++w) t += red[w][threadIdx.x];
    a.b.loss_part[(long long)blockIdx.x * kLossTermsB + threadIdx.x] = t;
  }
}

// =======================================================================================
// weight gradients from the bf16 stash: v_mfma_f32_32x32x16_bf16, K = the 16 rows of a tile
// (lane half hh supplies rows 8 hh .. 8 hh + 7 of its feature = 16 contiguous bytes)
// =======================================================================================
typedef float f32x16b __attribute__((ext_vector_type(16)));
#define PINN_MFMA32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

struct WgradArgsB {
  const __bf16* P;   // [T16][OUT][16]
  const __bf16* Q;   // [T16][IN][16] or nullptr (x rows, IN = 8)
  const float* x;
  long long n_rows;
  int OUT, IN;
  long long t16;
  int n_slices;
  long long slab_stride;
  float* dW; float* db;
  const float* s1; float* dvq;
  const float* s2; const __bf16* R; float* dvr;
};

template <int TI, int TJ>
struct WFragB {
  bf16x8 a[TI], b[TJ];
};

template <int TI, int TJ, bool QX>
__device__ __forceinline__ void wgrad_load_b(WFragB<TI, TJ>& f, const WgradArgsB& a, long long t, int i0, int j0, int hh, int i) {
  const __bf16* pP = a.P + ((t * a.OUT + i0 + i) * 16 + 8 * hh);
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) f.a[ti] = *reinterpret_cast<const bf16x8*>(pP + ti * 512);
  if (!QX) {
    const __bf16* pQ = a.Q + ((t * a.IN + j0 + i) * 16 + 8 * hh);
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) f.b[tj] = *reinterpret_cast<const bf16x8*>(pQ + tj * 512);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long long row = t * 16 + 8 * hh + j;
      if (row >= a.n_rows) row = a.n_rows - 1;
      f.b[0][j] = (__bf16)((i < 8) ? a.x[row * 8 + i] : 0.0f);
    }
  }
}

template <int TI, int TJ, int WI, int WJ, bool QX>
__global__ __launch_bounds__(kThreads, 1) void wgrad_bf16_kernel(WgradArgsB a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= WI * WJ) return;
  const int wi = wave / WJ, wj = wave % WJ;
  const int hh = lane >> 5, i = lane & 31;
  const int i0 = wi * TI * 32, j0 = wj * TJ * 32;
  f32x16b acc[TI][TJ];
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
  const long long per = (a.t16 + a.n_slices - 1) / a.n_slices;
  const long long t_begin = slice * per;
  long long t_end = t_begin + per;
  if (t_end > a.t16) t_end = a.t16;

  WFragB<TI, TJ> cur, nxt;
  if (t_begin < t_end) wgrad_load_b<TI, TJ, QX>(cur, a, t_begin, i0, j0, hh, i);
  for (long long t = t_begin; t < t_end; ++t) {
    const long long tn = (t + 1 < t_end) ? t + 1 : t;
    wgrad_load_b<TI, TJ, QX>(nxt, a, tn, i0, j0, hh, i);
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = PINN_MFMA32_BF16(cur.a[ti], cur.b[tj], acc[ti][tj]);
    if (wj == 0) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)cur.a[ti][j];
        bsum[ti] += s;
      }
      if (a.dvr) {
        const __bf16* pR = a.R + ((t * a.OUT + i0 + i) * 16 + 8 * hh);
        const float* ps = a.s2 + t * 16 + 8 * hh;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(ps), s1v = *reinterpret_cast<const f32x4*>(ps + 4);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          const bf16x8 rv = *reinterpret_cast<const bf16x8*>(pR + ti * 512);
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) s += s0[j] * (float)rv[j] + s1v[j] * (float)rv[4 + j];
          vr[ti] += s;
        }
      }
    }
    if (wi == 0 && a.dvq) {
      const float* ps = a.s1 + t * 16 + 8 * hh;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(ps), s1v = *reinterpret_cast<const f32x4*>(ps + 4);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += s0[j] * (float)cur.b[tj][j] + s1v[j] * (float)cur.b[tj][4 + j];
        vq[tj] += s;
      }
    }
    cur = nxt;
  }

  const long long so = (long long)slice * a.slab_stride;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int col = j0 + tj * 32 + i;
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

template <int TI, int TJ, int WI, int WJ, bool QX>
static void launch_wgrad_b(const WgradArgsB& a, hipStream_t st) {
  hipLaunchKernelGGL((wgrad_bf16_kernel<TI, TJ, WI, WJ, QX>), dim3(a.n_slices), dim3(kThreads), 0, st, a);
}
static int dispatch_wgrad_b(const WgradArgsB& a, hipStream_t st) {
  const int to = a.OUT / 32, ti = (a.IN + 31) / 32;
  if (a.Q == nullptr) {
    if (to == 8) launch_wgrad_b<2, 1, 4, 1, true>(a, st);
    else if (to == 4) launch_wgrad_b<1, 1, 4, 1, true>(a, st);
    else return PINN_E_ARCH;
    return PINN_OK;
  }
  if (to == 8 && ti == 8) launch_wgrad_b<4, 4, 2, 2, false>(a, st);
  else if (to == 4 && ti == 8) launch_wgrad_b<2, 4, 2, 2, false>(a, st);
  else if (to == 2 && ti == 4) launch_wgrad_b<1, 2, 2, 2, false>(a, st);
  else if (to == 4 && ti == 4) launch_wgrad_b<2, 2, 2, 2, false>(a, st);
  else if (to == 1 && ti == 2) launch_wgrad_b<1, 1, 1, 2, false>(a, st);
  else return PINN_E_ARCH;
  return PINN_OK;
}

// chain + weight-gradient launches of one bf16 training step (the fp32 finalize kernel is shared)
int launch_train_bf16(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y, long long n_rows,
                      long long n_global, const DropDev& drop, const TrainBuffers& b, unsigned phases, int* grid_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};
  const long long n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  int grid = (int)(n_tiles < 2 * cu_count_b() ? n_tiles : 2 * cu_count_b());
  if (grid > 1024) grid = 1024;
  *grid_out = grid;
  if (phases & PINN_PHASE_CHAIN) {
    launch_pack(net, d_params, st);
    TrainArgsB a{};
    a.params = d_params; a.packed = (const float*)net->d_packed; a.x = d_x; a.y = d_y; a.n_rows = n_rows; a.n_global = n_global;
    a.H = H; a.nh = nh; a.drop = drop; a.b = b;
    const bool bits = drop.mode == PINN_DROP_BITS;
    if (H == 256) {
      if (bits) hipLaunchKernelGGL((train_chain_bf16_kernel<256, true>), dim3(grid), dim3(kThreads), 0, st, a);
      else hipLaunchKernelGGL((train_chain_bf16_kernel<256, false>), dim3(grid), dim3(kThreads), 0, st, a);
    } else {
      if (bits) hipLaunchKernelGGL((train_chain_bf16_kernel<128, true>), dim3(grid), dim3(kThreads), 0, st, a);
      else hipLaunchKernelGGL((train_chain_bf16_kernel<128, false>), dim3(grid), dim3(kThreads), 0, st, a);
    }
  }
  if (phases & PINN_PHASE_WGRAD) {
    const long long tot = L.total();
    const long long hs = (long long)b.t16 * H * 16;
    const __bf16* sh = (const __bf16*)b.stash_h;
    const __bf16* dh = (const __bf16*)b.dpre_h;
    WgradArgsB g{};
    g.x = d_x; g.n_rows = n_rows; g.t16 = b.t16; g.n_slices = b.n_slices; g.slab_stride = tot;
    g.P = dh; g.Q = nullptr; g.OUT = H; g.IN = 8; g.dW = b.slabs + L.w0(); g.db = b.slabs + L.b0();
    int rc;
    if ((rc = dispatch_wgrad_b(g, st))) return rc;
    for (int l = 1; l < nh; ++l) {
      g.P = dh + l * hs; g.Q = sh + (l - 1) * hs; g.OUT = H; g.IN = H; g.dW = b.slabs + L.w(l); g.db = b.slabs + L.b(l);
      if ((rc = dispatch_wgrad_b(g, st))) return rc;
    }
    g.P = (const __bf16*)b.dpre_v1; g.Q = sh + (nh - 1) * hs; g.OUT = H / 2; g.IN = H; g.dW = b.slabs + L.wv0(); g.db = b.slabs + L.bv0();
    g.s1 = b.du; g.dvq = b.slabs + L.wp();
    if ((rc = dispatch_wgrad_b(g, st))) return rc;
    g.P = (const __bf16*)b.dpre_v2; g.Q = (const __bf16*)b.stash_v1; g.OUT = H / 4; g.IN = H / 2; g.dW = b.slabs + L.wv1(); g.db = b.slabs + L.bv1();
    g.s1 = nullptr; g.dvq = nullptr; g.s2 = b.dz; g.R = (const __bf16*)b.stash_v2; g.dvr = b.slabs + L.wv2();
    if ((rc = dispatch_wgrad_b(g, st))) return rc;
  }
  return PINN_OK;
}

}  // namespace pinn

using namespace pinn;

namespace pinn { size_t wide_scratch_floats(int H); }   // pinn_wide.hip

extern "C" size_t pinn_packed_bytes(const pinn_net_t* net) {
  if (!net || net->n_in != 8 || net->n_hidden < 1 || net->n_hidden > 8) return 0;
  const bool wide = net->hidden == 512 || net->hidden == 1024 || net->hidden == 2048;
  if (net->hidden != 128 && net->hidden != 256 && !wide) return 0;
  PackLayout K{net->hidden, net->n_hidden};
  // wide nets: the three copies + the activation scratch of one row chunk (layer-by-layer kernels)
  // (+ kRangeStatusBytes at the very end of every buffer the x6 pack kernel fills: pinn_net_range_status's record)
  if (wide) return net->precision != PINN_PREC_FP32 ? (size_t)K.total() * 2 * 5 + pinn::wide_scratch_floats(net->hidden) * 4 + pinn::kRangeStatusBytes : 0;
  if (net->precision == PINN_PREC_BF16) return (size_t)K.total() * 2;
  // three bf16 copies (hi, mid, lo: backward pass, weight gradients) + two fp16 copies of the forward matrices (scheme X3)
  if (net->precision == PINN_PREC_F32X6 || net->precision == PINN_PREC_F32X6_G6) return (size_t)K.total() * 2 * 5 + pinn::kRangeStatusBytes;
  return 0;
}

namespace pinn {
unsigned* range_status_words(const pinn_net_t* net) {
  const size_t n = pinn_packed_bytes(net);
  return (n >= kRangeStatusBytes && net->d_packed) ? (unsigned*)((char*)net->d_packed + n - kRangeStatusBytes) : nullptr;
}
}  // namespace pinn

extern "C" int pinn_net_range_status(const pinn_net_t* net, void* stream) {
  if (!net) return PINN_E_ARG;
  const bool x6 = net->precision == PINN_PREC_F32X6 || net->precision == PINN_PREC_F32X6_G6;
  if (!x6) return (net->precision == PINN_PREC_FP32 || net->precision == PINN_PREC_BF16) ? PINN_OK : PINN_E_ARG;
  unsigned* d = pinn::range_status_words(net);
  if (!d) return PINN_E_ARG;
  static unsigned host[pinn::kRangeStatusBytes / 4];          // one host thread per process drives the library (pinn_hip.h)
  const int n_words = pinn::kRangePackBlocks * (2 * (net->n_hidden - 1) + 4) + 1;      // pack jobs x blocks, then the gradient word
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(host, d, (size_t)n_words * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return (int)e;
  unsigned any = 0;
  for (int i = 0; i < n_words; ++i) any |= host[i];
  return any ? PINN_E_RANGE : PINN_OK;
}
