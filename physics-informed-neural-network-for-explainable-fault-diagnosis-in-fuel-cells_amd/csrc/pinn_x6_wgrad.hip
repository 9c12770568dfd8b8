// pinn_x6_wgrad.hip -- weight gradients on the 16-bit matrix cores (every precision but exact fp32).
//
// dW = dpre^T . h over K = rows, both operands read straight from the fp32 stash ([T16][F][16]: 8 contiguous rows
// of one feature per lane = one 32x32x16 MFMA fragment), split in registers, fp32 accumulation:
//     NS = 4 (kF16S): two fp16 parts under a common power-of-two scale, three MFMAs per product -- see split8_p
//                     (PINN_PREC_F32X6 on the fused nets)
//     NS = 3:  three bf16 parts x = hi + mid + lo (exact), six MFMAs: hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid
//                     (PINN_PREC_F32X6_G6; PINN_PREC_F32X6 on the wide nets)
//     NS = 1:  hi.hi  (bf16-mixed: PINN_PREC_BF16 on the wide nets)
// Same slices / slabs / bias and vector-head sums as wgrad_kernel (pinn_train.hip); layer 0 (IN = 8) stays there.
#include <cstdlib>
#include "pinn_x6_core.h"
#ifdef PINN_ABL_NTLOAD
#define PINN_WG_LD4(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))
#define PINN_WG_AUX 2
#else
#define PINN_WG_LD4(p) (*reinterpret_cast<const f32x4*>(p))
#define PINN_WG_AUX 0
#endif
#include "pinn_wgrad_args.h"

namespace pinn {
namespace x6 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PINN_MFMA32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// 8 fp32 -> NS bf16x8 parts
template <int NS>
struct Parts {
  u32x4 p[NS];
};
// One packed conversion per pair and stage, the rounded values recovered from the packed word by a shift / a mask (left to
// itself hipcc converts every element a second time on its own), the residual of both elements in one v_pk_add_f32:
// 9 VALU instructions per pair for three parts (as the casts were written before: 15).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float x0, float x1) {
  const bf16x2 h = {(__bf16)x0, (__bf16)x1};
  unsigned p = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(p));
  return p;
}
template <int NS>
__device__ __forceinline__ Parts<NS> split8(const f32x4& v0, const f32x4& v1) {
  Parts<NS> o;
  f32x2 x[4] = {{v0[0], v0[1]}, {v0[2], v0[3]}, {v1[0], v1[1]}, {v1[2], v1[3]}};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned p = pack_bf16(x[q][0], x[q][1]);
      o.p[s][q] = p;
      if (s + 1 < NS) {
        const f32x2 f = {__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
        x[q] = x[q] - f;
      }
    }
  }
  return o;
}

// ---- NS = kF16S: two fp16 parts, three products (v_mfma_f32_32x32x16_f16), 22 significant bits per operand.
// d pre-activations span many orders of magnitude (1 / N of the loss in front, per-row precisions): G = a power of two that
// puts the call's largest |d pre| (TrainBuffers::amax, measured by the X3 backward chain) in [2^14, 2^15), and the low part is
// stored SCALED, lo' = fp16((G x - hi) 2^11), so that it is a normal fp16 wherever hi is (down to 2^-29 of the largest
// element; below that the error is absolute, 2^-36 of it).  Its product needs the other operand's high part times 2^-11:
//     G x = hi + lo' / 2048,   8 h = hh + hl,   hs = fp16(hh / 2048)   ->   8 G x h ~ hi hh + hi hl + lo' hs
// (dropped: lo' hl / 2048, 2^-22 of the product).  Activations (|h| <= 1.67 / (1 - p)) carry the x 8 of scheme X3.
constexpr int kF16S = 4;
#define PINN_MFMA32_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
struct PartsP { u32x4 hi, lo; };
struct PartsQ { u32x4 hi, lo, hs; };
__device__ __forceinline__ unsigned pack_f16(float x0, float x1) {
  const f16x2 h = {(_Float16)x0, (_Float16)x1};
  unsigned p = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(p));
  return p;
}
__device__ __forceinline__ PartsP split8_p(const f32x4& v0, const f32x4& v1, float g) {
  PartsP o;
  const f32x2 x[4] = {{v0[0], v0[1]}, {v0[2], v0[3]}, {v1[0], v1[1]}, {v1[2], v1[3]}};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 xg = x[q] * g;
    const unsigned hp = pack_f16(xg[0], xg[1]);
    const f16x2 h = __builtin_bit_cast(f16x2, hp);
    const f32x2 hf = {(float)h[0], (float)h[1]};
    const f32x2 r = (xg - hf) * 2048.0f;               // exact: the residual of a rounding, times a power of two
    o.hi[q] = hp;
    o.lo[q] = pack_f16(r[0], r[1]);
  }
  return o;
}
__device__ __forceinline__ PartsQ split8_q(const f32x4& v0, const f32x4& v1) {
  PartsQ o;
  const f32x2 x[4] = {{v0[0], v0[1]}, {v0[2], v0[3]}, {v1[0], v1[1]}, {v1[2], v1[3]}};
  const f16x2 k = {(_Float16)0x1p-11f, (_Float16)0x1p-11f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x8 = x[q] * 8.0f;
    const unsigned hp = pack_f16(x8[0], x8[1]);
    const f16x2 h = __builtin_bit_cast(f16x2, hp);
    const f32x2 hf = {(float)h[0], (float)h[1]};
    const f32x2 r = x8 - hf;
    o.hi[q] = hp;
    o.lo[q] = pack_f16(r[0], r[1]);
    o.hs[q] = __builtin_bit_cast(unsigned, h * k);    // v_pk_mul_f16
  }
  return o;
}
// G and 1 / (8 G) from the call's largest |d pre-activation| (bits in *amax); 1 where it is zero or not finite
struct GScale { float g, inv; };
__device__ __forceinline__ GScale g_scale(const unsigned* amax) {
  const float am = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*amax));
  int e = 0;
  (void)frexpf(am, &e);
  const bool ok = am > 0.0f && am < __builtin_inff();
  e = ok ? (e < -100 ? -100 : e) : 15;
  return GScale{ldexpf(1.0f, 15 - e), ldexpf(1.0f, e - 18)};
}
// the three products of one block, smallest first
__device__ __forceinline__ void mma_f16s(f32x16& acc, const PartsP& pa, const PartsQ& pb) {
  acc = PINN_MFMA32_F16(__builtin_bit_cast(f16x8, pa.lo), __builtin_bit_cast(f16x8, pb.hs), acc);
  acc = PINN_MFMA32_F16(__builtin_bit_cast(f16x8, pa.hi), __builtin_bit_cast(f16x8, pb.lo), acc);
  acc = PINN_MFMA32_F16(__builtin_bit_cast(f16x8, pa.hi), __builtin_bit_cast(f16x8, pb.hi), acc);
}

template <int TI, int TJ>
struct Raw {
  f32x4 a[TI][2], b[TJ][2];
};
template <int TI, int TJ>
__device__ __forceinline__ void wgrad_load(Raw<TI, TJ>& f, const WgradArgs& a, long long t, int i0, int j0, int hh, int i) {
  const float* pP = a.P + ((t * a.OUT + i0 + i) * 16 + 8 * hh);
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    f.a[ti][0] = PINN_WG_LD4(pP + ti * 512);
    f.a[ti][1] = PINN_WG_LD4(pP + ti * 512 + 4);
  }
  const float* pQ = a.Q + ((t * a.IN + j0 + i) * 16 + 8 * hh);
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    f.b[tj][0] = PINN_WG_LD4(pQ + tj * 512);
    f.b[tj][1] = PINN_WG_LD4(pQ + tj * 512 + 4);
  }
}

template <int TI, int TJ, int WI, int WJ, int NS>
__global__ __launch_bounds__(256, 1) void wgrad_x_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= WI * WJ) return;
  const int wi = wave / WJ, wj = wave % WJ;
  const int hh = lane >> 5, i = lane & 31;
  // blockIdx.y / z: which [32 TI WI] x [32 TJ WJ] block of a larger gradient this workgroup computes
  const int i0 = blockIdx.y * (TI * 32 * WI) + wi * TI * 32, j0 = blockIdx.z * (TJ * 32 * WJ) + wj * TJ * 32;
  const bool row_sums = wj == 0 && blockIdx.z == 0, col_sums = wi == 0 && blockIdx.y == 0;
  GScale gs{1.0f, 1.0f};
  if constexpr (NS == kF16S) gs = g_scale(a.amax);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
  double bsum[TI];        // (float64: a slice adds thousands of rows one after the other, where torch sums pairwise)
  float vq[TJ], vr[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) { bsum[ti] = 0.0; vr[ti] = 0.f; }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) vq[tj] = 0.f;

  const int slice = blockIdx.x;
  const long long per = (a.t16 + a.n_slices - 1) / a.n_slices;
  const long long t_begin = slice * per;
  long long t_end = t_begin + per;
  if (t_end > a.t16) t_end = a.t16;

  // software pipeline, two tiles deep where the registers allow (< 256 accumulator registers): the fp32 fragments and the
  // vector-head operands of tiles t+1 and t+2 are in flight while tile t is split and multiplied -- one tile ahead is
  // about one HBM latency of work and stalls on most loads
  constexpr bool kDeep = TI * TJ < 16;
  struct Side { f32x4 r[kDeep ? TI : 1][2], s2[2], s1[2]; };      // operands of the vector-head sums (prefetched iff kDeep)
  const bool want_r = row_sums && a.dvr, want_q = col_sums && a.dvq;
  auto load_r = [&](long long t, int ti, int sg) { return PINN_WG_LD4(a.R + ((t * a.OUT + i0 + i) * 16 + 8 * hh) + ti * 512 + sg * 4); };
  auto load_s = [&](const float* sv, long long t, int sg) { return PINN_WG_LD4(sv + t * 16 + 8 * hh + sg * 4); };
  auto fetch = [&](Raw<TI, TJ>& f, Side& sd, long long t) {
    if (t >= t_end) t = t_end - 1;
    wgrad_load<TI, TJ>(f, a, t, i0, j0, hh, i);
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
  auto tile = [&](const Raw<TI, TJ>& cur, const Side& sd, long long t) {
    if constexpr (NS == kF16S) {
      PartsP pa[TI];
      PartsQ pb[TJ];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) pa[ti] = split8_p(cur.a[ti][0], cur.a[ti][1], gs.g);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) pb[tj] = split8_q(cur.b[tj][0], cur.b[tj][1]);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) mma_f16s(acc[ti][tj], pa[ti], pb[tj]);
    } else {
      constexpr int NP = NS == kF16S ? 1 : NS;
      Parts<NP> pa[TI], pb[TJ];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) pa[ti] = split8<NP>(cur.a[ti][0], cur.a[ti][1]);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) pb[tj] = split8<NP>(cur.b[tj][0], cur.b[tj][1]);
      // products (sa, sb) with sa + sb < NS, smallest first
#pragma unroll
      for (int tot = NP - 1; tot >= 0; --tot)
#pragma unroll
        for (int sa = 0; sa <= tot; ++sa)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
              acc[ti][tj] = PINN_MFMA32_BF16(__builtin_bit_cast(bf16x8, pa[ti].p[sa]), __builtin_bit_cast(bf16x8, pb[tj].p[tot - sa]), acc[ti][tj]);
    }
    if (row_sums) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) bsum[ti] += (double)((cur.a[ti][sg][0] + cur.a[ti][sg][1]) + (cur.a[ti][sg][2] + cur.a[ti][sg][3]));
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
    if (col_sums && a.dvq) {
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
    Raw<TI, TJ> f0, f1, f2;
    Side d0, d1, d2;
    if (t_begin < t_end) { fetch(f0, d0, t_begin); fetch(f1, d1, t_begin + 1); }
    for (long long t = t_begin; t < t_end; t += 3) {
      fetch(f2, d2, t + 2);
      tile(f0, d0, t);
      if (t + 1 < t_end) { fetch(f0, d0, t + 3); tile(f1, d1, t + 1); }
      if (t + 2 < t_end) { fetch(f1, d1, t + 4); tile(f2, d2, t + 2); }
    }
  } else {
    Raw<TI, TJ> cur, nxt;
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
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        a.dW[so + (long long)row * a.IN + col] = NS == kF16S ? acc[ti][tj][r] * gs.inv : acc[ti][tj][r];
      }
    }
  if (row_sums) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const float b = (float)(bsum[ti] + __shfl_xor(bsum[ti], 32, 64));
      if (hh == 0) a.db[so + i0 + ti * 32 + i] = b;
      if (a.dvr) {
        const float v = vr[ti] + __shfl_xor(vr[ti], 32, 64);
        if (hh == 0) a.dvr[so + i0 + ti * 32 + i] = v;
      }
    }
  }
  if (col_sums && a.dvq) {
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const float v = vq[tj] + __shfl_xor(vq[tj], 32, 64);
      if (hh == 0) a.dvq[so + j0 + tj * 32 + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Deep-prefetch version for the plain layers (no vector-head sums): the fragments of tile t + 2 stream global -> LDS by
// LDS-DMA into a PRIVATE two-stage ring per wave (no sharing, so no barrier: only a counted vmcnt), tile t + 1 moves
// LDS -> registers while tile t is split and multiplied.  The register-staged kernel above has one tile (1.9 us of
// work) between a load and its use -- about one HBM latency -- and stalls on most of them.
// ---------------------------------------------------------------------------------------
template <int TI, int TJ, int WI, int WJ, int NS>
__global__ __launch_bounds__(256, 1) void wgrad_d_kernel(WgradArgs a) {
  constexpr int kFrags = TI + TJ, kStage = kFrags * 2048;            // bytes per wave and tile: 2 KB per fragment
  extern __shared__ __attribute__((aligned(1024))) char ring_all[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= WI * WJ) return;
  char* ring = ring_all + wave * 2 * kStage;
  const int wi = wave / WJ, wj = wave % WJ;
  const int hh = lane >> 5, i = lane & 31;
  const int i0 = blockIdx.y * (TI * 32 * WI) + wi * TI * 32, j0 = blockIdx.z * (TJ * 32 * WJ) + wj * TJ * 32;
  const bool row_sums = wj == 0 && blockIdx.z == 0;
  GScale gs{1.0f, 1.0f};
  if constexpr (NS == kF16S) gs = g_scale(a.amax);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
  double bsum[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) bsum[ti] = 0.0;

  const int slice = blockIdx.x;
  const long long per = (a.t16 + a.n_slices - 1) / a.n_slices;
  const long long t_begin = slice * per;
  long long t_end = t_begin + per;
  if (t_end > a.t16) t_end = a.t16;

  // 2 x kFrags DMA instructions per tile: half q (rows 8 hh + 4 q ..) of fragment f, lane-linear 16 B per lane
  auto fetch = [&](long long t, int stage) {
    const float* pP = a.P + ((t * a.OUT + i0 + i) * 16 + 8 * hh);
    const float* pQ = a.Q + ((t * a.IN + j0 + i) * 16 + 8 * hh);
    char* st = ring + stage * kStage;
#pragma unroll
    for (int f = 0; f < kFrags; ++f) {
      const float* src = f < TI ? pP + f * 512 : pQ + (f - TI) * 512;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + f * 2048), 16, 0, PINN_WG_AUX);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + 4), (lptr_t)(st + f * 2048 + 1024), 16, 0, PINN_WG_AUX);
    }
  };
  auto read = [&](Raw<TI, TJ>& r, int stage) {
    const char* st = ring + stage * kStage + lane * 16;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      r.a[ti][0] = *reinterpret_cast<const f32x4*>(st + ti * 2048);
      r.a[ti][1] = *reinterpret_cast<const f32x4*>(st + ti * 2048 + 1024);
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      r.b[tj][0] = *reinterpret_cast<const f32x4*>(st + (TI + tj) * 2048);
      r.b[tj][1] = *reinterpret_cast<const f32x4*>(st + (TI + tj) * 2048 + 1024);
    }
  };
  if (t_begin < t_end) {
    auto clampt = [&](long long t) { return t < t_end ? t : t_end - 1; };
    Raw<TI, TJ> cur, nxt;
    fetch(t_begin, 0);
    fetch(clampt(t_begin + 1), 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kFrags) : "memory");       // tile t_begin has landed
    read(cur, 0);
    for (long long t = t_begin; t < t_end; ++t) {
      const int s0 = (int)((t - t_begin) & 1);
      // stage s0 (tile t) is in registers: it takes tile t + 2; then tile t + 1 (the other stage) must have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // ... the LDS reads of `cur` are done
      fetch(clampt(t + 2), s0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kFrags) : "memory");
      read(nxt, s0 ^ 1);
      if (row_sums) {          // (before the split: the fp32 fragments are dead once their parts exist)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int sg = 0; sg < 2; ++sg) bsum[ti] += (double)((cur.a[ti][sg][0] + cur.a[ti][sg][1]) + (cur.a[ti][sg][2] + cur.a[ti][sg][3]));
      }
      if constexpr (NS == kF16S) {
        PartsP pa[TI];
        PartsQ pb[TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) pa[ti] = split8_p(cur.a[ti][0], cur.a[ti][1], gs.g);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) pb[tj] = split8_q(cur.b[tj][0], cur.b[tj][1]);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) mma_f16s(acc[ti][tj], pa[ti], pb[tj]);
      } else {
        constexpr int NP = NS == kF16S ? 1 : NS;
        Parts<NP> pa[TI], pb[TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) pa[ti] = split8<NP>(cur.a[ti][0], cur.a[ti][1]);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) pb[tj] = split8<NP>(cur.b[tj][0], cur.b[tj][1]);
#pragma unroll
        for (int tot = NP - 1; tot >= 0; --tot)
#pragma unroll
          for (int sa = 0; sa <= tot; ++sa)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
              for (int tj = 0; tj < TJ; ++tj)
                acc[ti][tj] = PINN_MFMA32_BF16(__builtin_bit_cast(bf16x8, pa[ti].p[sa]), __builtin_bit_cast(bf16x8, pb[tj].p[tot - sa]), acc[ti][tj]);
      }
      cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing re-fetches must land before the LDS is released
  }

  const long long so = (long long)slice * a.slab_stride;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int col = j0 + tj * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        a.dW[so + (long long)row * a.IN + col] = NS == kF16S ? acc[ti][tj][r] * gs.inv : acc[ti][tj][r];
      }
    }
  if (row_sums) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const float b = (float)(bsum[ti] + __shfl_xor(bsum[ti], 32, 64));
      if (hh == 0) a.db[so + i0 + ti * 32 + i] = b;
    }
  }
}

template <int TI, int TJ, int WI, int WJ>
static int launch_d(const WgradArgs& a, int ns, hipStream_t st) {
  const dim3 grid(a.n_slices, a.OUT / (TI * 32 * WI), a.IN / (TJ * 32 * WJ));
  const size_t lds = (size_t)WI * WJ * 2 * (TI + TJ) * 2048;
  auto k3 = wgrad_d_kernel<TI, TJ, WI, WJ, 3>;
  auto k1 = wgrad_d_kernel<TI, TJ, WI, WJ, 1>;
  auto k4 = wgrad_d_kernel<TI, TJ, WI, WJ, kF16S>;
  auto k = ns == 3 ? k3 : (ns == kF16S ? k4 : k1);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
  return PINN_OK;
}

template <int TI, int TJ, int WI, int WJ>
static void launch(const WgradArgs& a, int ns, hipStream_t st) {
  const dim3 grid(a.n_slices, a.OUT / (TI * 32 * WI), a.IN / (TJ * 32 * WJ));
  if (ns == 3) hipLaunchKernelGGL((wgrad_x_kernel<TI, TJ, WI, WJ, 3>), grid, dim3(256), 0, st, a);
  else if (ns == kF16S) hipLaunchKernelGGL((wgrad_x_kernel<TI, TJ, WI, WJ, kF16S>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad_x_kernel<TI, TJ, WI, WJ, 1>), grid, dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// Packed operands (PINN_PREC_F32X6 on the fused nets): dW = dpre^T . h from the fragments the chain kernels stashed.
//
// A stash group (32 features x 16 rows) is a 2-KB block [part hi / lo][row][64 B]; a row's 64 B hold its 32 features as four
// 16-B lane chunks of the chain's B fragment (packed_ptr, pinn_x6_core.h).  The weight gradient contracts over ROWS, so
// it needs the transpose -- 8 rows of one feature per lane -- and gfx950 reads exactly that out of LDS:
// ds_read_b64_tr_b16 hands each lane of a 16-lane group one 16-bit column of a 4-row block (tools/tr16_probe.hip pins
// the lane map).  So: the blocks stream global -> LDS by LDS-DMA as they lie in memory (1 KB per instruction, fully
// coalesced), four transposing reads per fragment give (hi, lo) x rows 8 hh .. 8 hh + 7 of feature tr_feature(lane & 31),
// and no operand is split or shuffled in registers.  d pre-activations come in the rows' normalised units: the row scale
// t_r = 2^(e_r - E + c) goes onto the OTHER operand, one v_pk_mul_f16 per dword (struct RowMeta / grad_exponent); bias and
// predict-head sums are v_dot2_f32_f16 with the same row vectors.  Per 48 MFMAs: 32 + 32 (bias waves) VALU instructions,
// where the split-in-registers kernel above needs 365.
// ---------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// lane i (0 .. 31) of a 32-feature tile holds feature tr_feature(i) of its group: 16-lane group x = i >> 4 reads the lane
// chunks kq = 2 x, 2 x + 1; lane 4 p' + e of the group gets element e of 8-byte piece p' = 2 (kq & 1) + s, i.e. fragment
// element 4 s + e = 2 r + b of chunk kq: feature 16 b + 4 kq + r
__host__ __device__ inline int tr_feature(int i) {
  const int x = i >> 4, p = (i >> 2) & 3, e = i & 3;
  return 16 * (e & 1) + 4 * (2 * x + (p >> 1)) + 2 * (p & 1) + (e >> 1);
}
struct PFrag { u32x4 hi, lo; };            // 8 rows x (hi, lo) of one feature: dword k = rows 8 hh + 2 k, + 2 k + 1
__device__ __forceinline__ u32x4 pk_mul4(const u32x4& v, const u32x4& t) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (a vector ELEMENT handed to __builtin_bit_cast reads element 0 whatever k is, hipcc 7.2: go through scalars)
    const unsigned vk = v[k], tk = t[k];
    o[k] = __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, vk) * __builtin_bit_cast(f16x2, tk));
  }
  return o;
}
__device__ __forceinline__ float dot8(const u32x4& v, const u32x4& w, float acc) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned vk = v[k], wk = w[k];
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, vk), __builtin_bit_cast(f16x2, wk), acc, false);
  }
  return acc;
}

// Schedule.  One wave per SIMD (256 accumulator registers), so a wave issues everything itself, and an LDS-DMA instruction holds
// its issue for 60-180 cycles: the first packed version fetched into private rings in a burst at the top of a tile and took
// 0.49 ms per 256 x 256 layer at 1e6 rows against 0.54 split in registers -- the VALU work was gone, the burst was not.
// Ablations of the next version (pieces spread under the MFMAs; whole weight-gradient phase 1.75 ms): without the MFMAs
// 1.72 -- the matrix work is hidden entirely --, without the DMA 1.11, non-temporal loads 2.0: the kernel is bound by its
// memory path, and private rings fetch every P block for both waves of a wave row (every Q block for both of a column):
// twice the DMA instructions and the second copy from L2 at best.  So the ring is SHARED: the workgroup's P, Q (and R) blocks
// of a tile form one stage, every wave fetches a quarter of its 1-KB pieces, one barrier per tile publishes it.  The tile is
// cut into its TI x TJ blocks of three MFMAs; every block carries its share of tile t + S's pieces and, in the second half,
// of tile t + 1's transposing reads, pinned with sched_barrier.
template <int TI, int TJ, int WI, int WJ, bool kVQ, bool kVR, int S>
__device__ __forceinline__ void wgrad_p_body(const WgradPArgs& a, const int slice, const int by, const int bz, char* ring) {
  constexpr int nW = WI * WJ, GP = TI * WI, GQ = TJ * WJ, GR = kVR ? GP : 0;      // the workgroup's blocks of a tile: P, Q, R groups
  constexpr int kBlocks = GP + GQ + GR, kData = 2 * kBlocks;                      // 2-KB blocks, 1-KB data pieces per tile
  constexpr int kMetaAt = kBlocks * 2048, kStage = kMetaAt + 256;                 // + the tile's row record
  constexpr int kMine = (kData + nW - 1) / nW, kW = kMine + 1;                    // DMA instructions per wave and tile (+ the record, by every wave)
  constexpr int kFrags = TI + TJ;                                                 // fragments a wave reads
  constexpr int NB = TI * TJ, kHalf = NB / 2;
#ifdef PINN_ABL_WGP_BURST      // (ablation: all of a tile's pieces behind its first block)
  constexpr int kPer = kW;
#else
  constexpr int kPer = (kW + NB - 1) / NB;
#endif
  constexpr int kRd = (4 * kFrags + (NB - kHalf) - 1) / (NB - kHalf);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (ring: S stages)
  const int wi = wave / WJ, wj = wave % WJ;
  const int hh = lane >> 5, i = lane & 31;
  const int gi0 = by * GP, gj0 = bz * GQ;                                         // the workgroup's first 32-feature group of P / of Q
  const int i0 = 32 * (gi0 + wi * TI), j0 = 32 * (gj0 + wj * TJ);                 // this wave's first output row / column
  const bool row_sums = wj == 0 && bz == 0, col_sums = wi == 0 && by == 0;
  const bool want_q = kVQ && col_sums && a.dvq;
  const int gP = a.OUT / 32, gQ = a.IN / 32;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
  double bsum[TI];
  float vq[TJ], vr[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) { bsum[ti] = 0.0; vr[ti] = 0.f; }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) vq[tj] = 0.f;

  // slice s takes the row tiles s, s + n_slices, ...: at any moment the workgroups of a launch read one contiguous window of the
  // stash (n_slices x 16-32 KB), spread over every HBM channel; contiguous per-slice ranges put 256 streams 3.9 MB apart
  const long long n_mine = slice < a.t16 ? (a.t16 - slice + a.n_slices - 1) / a.n_slices : 0;
  auto tile_of = [&](long long k) { return slice + (k < n_mine ? k : n_mine - 1) * (long long)a.n_slices; };      // (clamped: trailing re-fetches)

  // this wave's j-th piece of tile t: piece p = wave + nW j of the tile's kData (wrapping: a wave with one piece fewer fetches
  // an earlier one again -- same bytes to the same place); the last instruction of every wave is the tile's 256-B row record.
  // Source (tile 0), bytes per tile and LDS offset of every piece are settled here, once: scalar registers, no branch in the loop.
  const char* p_src[kMine];
  long long p_step[kMine];
  int p_dst[kMine];
#pragma unroll
  for (int j = 0; j < kMine; ++j) {
    int p = wave + nW * j;
    p = p < kData ? p : p - kData;
    const int f = p >> 1, half = p & 1;
    const bool isP = f < GP, isQ = !isP && f < GP + GQ;
    p_src[j] = (isP ? a.P + (long long)(gi0 + f) * 2048 : (isQ ? a.Q + (long long)(gj0 + f - GP) * 2048 : (const char*)a.R + (long long)(gi0 + f - GP - GQ) * 2048)) + half * 1024;
    p_step[j] = isP ? (long long)gP * 2048 : (isQ ? (long long)gQ * 2048 : (long long)a.OUT * 64);
    p_dst[j] = f * 2048 + half * 1024;
  }
  auto piece = [&](long long t, char* st, auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < kMine) {
      __builtin_amdgcn_global_load_lds((gptr_t)(p_src[j] + t * p_step[j] + lane * 16), (lptr_t)(st + p_dst[j]), 16, 0, PINN_WG_AUX);
    } else {
      __builtin_amdgcn_global_load_lds((gptr_t)((const char*)a.meta + t * 256 + lane * 4), (lptr_t)(st + kMetaAt), 4, 0, 0);
    }
  };
  struct Regs {
    u32x2 w[kFrags][4];                              // fragment f: hi rows 0-3, hi rows 4-7, lo rows 0-3, lo rows 4-7 (of 8 hh ..)
    u32x4 tq, dh[kVQ ? 1 : 0], dl[kVQ ? 1 : 0];     // rows 8 hh .. 8 hh + 7: scales t_r; (kVQ) du_r norm_r as two fp16 parts
    f32x4 rr[kVR ? TI : 0][2], s2[kVR ? 2 : 0];     // (kVR) fp32 operands of the vector-head sum: R rows, dz
  };
  // transposing reads: lane 4 q' + p' of a 16-lane group addresses (row q' of the 4-row block, 8-byte piece p'); this wave's P
  // fragments start at block wi TI of a stage, its Q fragments at block GP + wj TJ: two base registers, the rest immediates
  const int L = lane & 15;
  const unsigned lane_off = (unsigned)((8 * hh + (L >> 2)) * 64 + (2 * ((lane >> 4) & 1) + ((L & 3) >> 1)) * 16 + (L & 1) * 8);
  const unsigned off_p = lane_off + (unsigned)(wi * TI) * 2048u, off_q = lane_off + (unsigned)(GP + wj * TJ) * 2048u;
  auto read_one = [&](Regs& r, unsigned stage_addr, auto jc) {
    constexpr int j = decltype(jc)::value, f = j / 4, sub = j % 4;
    constexpr int fl = f < TI ? f : f - TI;
    r.w[f][sub] = lds_read_tr16<fl * 2048 + (sub >> 1) * 1024 + (sub & 1) * 256>(stage_addr + (f < TI ? off_p : off_q));
  };
  auto read_side = [&](Regs& r, const char* st) {
    const char* m = st + kMetaAt + 16 * hh;
    r.tq = *reinterpret_cast<const u32x4*>(m);
    if constexpr (kVQ) {
      r.dh[0] = *reinterpret_cast<const u32x4*>(m + 32);
      r.dl[0] = *reinterpret_cast<const u32x4*>(m + 64);
    }
    if constexpr (kVR) {
      r.s2[0] = *reinterpret_cast<const f32x4*>(st + kMetaAt + 128 + 32 * hh);
      r.s2[1] = *reinterpret_cast<const f32x4*>(st + kMetaAt + 128 + 32 * hh + 16);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        r.rr[ti][0] = *reinterpret_cast<const f32x4*>(st + (GP + GQ + wi * TI + ti) * 2048 + i * 64 + 32 * hh);
        r.rr[ti][1] = *reinterpret_cast<const f32x4*>(st + (GP + GQ + wi * TI + ti) * 2048 + i * 64 + 32 * hh + 16);
      }
    }
  };
  auto frag_hi = [](const Regs& r, int f) { return u32x4{r.w[f][0][0], r.w[f][0][1], r.w[f][1][0], r.w[f][1][1]}; };
  auto frag_lo = [](const Regs& r, int f) { return u32x4{r.w[f][2][0], r.w[f][2][1], r.w[f][3][0], r.w[f][3][1]}; };
  auto stage_addr = [&](const char* st) { return (unsigned)(unsigned long long)(lptr_t)st; };

  if (n_mine > 0) {
    Regs cur, nxt;
    // S tiles in flight; the first one into registers
#pragma unroll
    for (int s = 0; s < S; ++s) static_for<kW>([&](auto jc) { piece(tile_of(s), ring + s * kStage, jc); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * kW) : "memory");       // this wave's pieces of tile t_begin have landed ...
    __builtin_amdgcn_s_barrier();                                              // ... and so have the other waves'
    asm volatile("" ::: "memory");
    static_for<4 * kFrags>([&](auto jc) { read_one(cur, stage_addr(ring), jc); });
    read_side(cur, ring);
    int s_cur = 0;                     // stage of tile t
    for (long long kt = 0; kt < n_mine; ++kt) {
      const int s_next = s_cur + 1 == S ? 0 : s_cur + 1;
      // `cur` (tile t, read from stage s_cur) is complete; this wave's pieces of tile t + 1 have landed.  Past the barrier that
      // holds for every wave: stage s_next can be read, stage s_cur refilled with tile t + S.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * kW) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const long long t_fill = tile_of(kt + S);
      char* st_fill = ring + s_cur * kStage;
      const char* st_next = ring + s_next * kStage;
      const unsigned a_next = stage_addr(st_next);
      u32x4 qh[TJ], ql[TJ];
      static_for<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value, ti = b / TJ, tj = b % TJ;
        if constexpr (ti == 0) {      // the row scale onto this Q fragment, at its first use
          qh[tj] = pk_mul4(frag_hi(cur, TI + tj), cur.tq);
          ql[tj] = pk_mul4(frag_lo(cur, TI + tj), cur.tq);
        }
        const u32x4 ph = frag_hi(cur, ti), pl = frag_lo(cur, ti);
#ifdef PINN_ABL_WGP_NOMFMA      // (ablation builds, tools/build_variant.py: what the tile costs without its matrix work)
        asm volatile("" ::"v"(ph), "v"(pl), "v"(qh[tj]), "v"(ql[tj]));
#else
        acc[ti][tj] = PINN_MFMA32_F16(__builtin_bit_cast(f16x8, pl), __builtin_bit_cast(f16x8, qh[tj]), acc[ti][tj]);      // smallest first
        acc[ti][tj] = PINN_MFMA32_F16(__builtin_bit_cast(f16x8, ph), __builtin_bit_cast(f16x8, ql[tj]), acc[ti][tj]);
        acc[ti][tj] = PINN_MFMA32_F16(__builtin_bit_cast(f16x8, ph), __builtin_bit_cast(f16x8, qh[tj]), acc[ti][tj]);
#endif
#ifndef PINN_ABL_WGP_NODMA
        static_for<kPer>([&](auto ic) {
          constexpr int k = b * kPer + decltype(ic)::value;
          if constexpr (k < kW) piece(t_fill, st_fill, IC<k>{});
        });
#endif
#ifndef PINN_ABL_WGP_NOREAD
        if constexpr (b == kHalf) read_side(nxt, st_next);
        if constexpr (b >= kHalf) {
          static_for<kRd>([&](auto ic) {
            constexpr int j = (b - kHalf) * kRd + decltype(ic)::value;
            if constexpr (j < 4 * kFrags) read_one(nxt, a_next, IC<j>{});
          });
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
      if (row_sums) {          // bias: sum_r d pre_r = 2^(E - c - 4) sum_r p_r t_r
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) bsum[ti] += (double)dot8(frag_hi(cur, ti), cur.tq, dot8(frag_lo(cur, ti), cur.tq, 0.0f));
        if constexpr (kVR) {
          if (a.dvr) {
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
#pragma unroll
              for (int ti = 0; ti < TI; ++ti) {
                const f32x4 sv = cur.s2[sg], rv = cur.rr[ti][sg];
                vr[ti] += (sv[0] * rv[0] + sv[1] * rv[1]) + (sv[2] * rv[2] + sv[3] * rv[3]);
              }
          }
        }
      }
      if constexpr (kVQ) {
        if (want_q) {          // predict head: sum_r du_r h_r = 2^(E - c - 7) sum_r (du_r norm_r) (8 h_r t_r)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) vq[tj] += dot8(qh[tj], cur.dh[0], dot8(ql[tj], cur.dh[0], dot8(qh[tj], cur.dl[0], 0.0f)));
        }
      }
#ifndef PINN_ABL_WGP_NOREAD
      cur = nxt;
#endif
      s_cur = s_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing re-fetches must land before the LDS is released
  }

  // ---- write this slice's slab; the powers of two of the row scale leave here
  const int E = grad_exponent(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*a.emax)), 4);
  const float sw = ldexpf(1.0f, E - a.qboost - 4 - a.q_log2), sb = ldexpf(1.0f, E - a.qboost - 4);
  const int ldw = a.ldW ? a.ldW : a.IN, ncol = a.n_cols ? a.n_cols : a.IN;
  const long long so = (long long)slice * a.slab_stride;
  const int fi = tr_feature(i);
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int col = j0 + tj * 32 + fi;           // input feature (C/D layout: B's lane = column)
      if (col < ncol) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + ti * 32 + tr_feature((r & 3) + 8 * (r >> 2) + 4 * hh);      // A's lane = row of the tile
          a.dW[so + (long long)row * ldw + col] = acc[ti][tj][r] * sw;
        }
      }
    }
  if (row_sums) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const double bb = bsum[ti] + __shfl_xor(bsum[ti], 32, 64);
      if (hh == 0) a.db[so + i0 + ti * 32 + fi] = (float)(bb * (double)sb);
      if constexpr (kVR) {
        if (a.dvr) {
          const float v = vr[ti] + __shfl_xor(vr[ti], 32, 64);
          if (hh == 0) a.dvr[so + i0 + ti * 32 + i] = v;       // (R is fp32 in the natural feature order)
        }
      }
    }
  }
  if constexpr (kVQ) {
    if (want_q) {
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const float v = vq[tj] + __shfl_xor(vq[tj], 32, 64);
        if (hh == 0) a.dvq[so + j0 + tj * 32 + fi] = v * sw;
      }
    }
  }
}

template <int TI, int TJ, int WI, int WJ, bool kVQ, bool kVR, int S>
__global__ __launch_bounds__(64 * WI * WJ, 1) void wgrad_p_kernel(WgradPArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  int slice = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gridDim.y * gridDim.z > 1 && (gridDim.x & 7) == 0) {
    // A gradient of several blocks (the wide nets: 4 x 4 blocks of 256 x 256 at H = 1024): every block of a block ROW reads the
    // same P tiles, and in launch order (x fastest) the workgroups resident at one time were the slices of ONE block -- each
    // operand came from HBM once per block, four times in all (29 GB, 8.5 ms at 262 144 rows).  Workgroups are handed to the
    // XCDs round-robin by their linear index: re-number them so that an XCD holds, side by side, the blocks bz = 0 .. nz - 1 of
    // the same (slice, block row) -- they stream the same P tiles at the same pace, three of four reads hit that XCD's L2.
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = id & 7u, k = id >> 3, per_xcd = gridDim.x >> 3;
    bz = (int)(k % gridDim.z);
    const unsigned r = k / gridDim.z;
    slice = (int)((r % per_xcd) * 8u + xcd);
    by = (int)(r / per_xcd);
  }
  wgrad_p_body<TI, TJ, WI, WJ, kVQ, kVR, S>(a, slice, by, bz, ring);
}

// Small row counts (the reference's own: 1e3 .. 3e4 rows), H = 256: every layer's gradient in ONE launch.  Five launches
// side by side on forked streams cost ~8 us at the fork and ~10 us at the join inside a replayed graph, and a workgroup that
// owns a whole 256 x 256 gradient spends more time zeroing and writing its 256 accumulator registers than multiplying.  Here a
// workgroup is (problem, slice, 128 x 128 or smaller output tile): four waves, <= 64 accumulator registers, ~450 workgroups
// resident at once.  Same arithmetic per output element as the per-layer launches (the K order of a slice is the same).
template <int S>
__global__ __launch_bounds__(256, 2) void wgrad_p_multi_kernel(WgradPMulti m) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int b = blockIdx.x;
  int k = 0;
  while (k + 1 < m.n && b >= m.first[k + 1]) ++k;
  k = __builtin_amdgcn_readfirstlane(k);
  const WgradPArgs& a = m.p[k];
  const int local = b - m.first[k];
  const int slice = local % a.n_slices, t = local / a.n_slices;
  const int kind = m.kind[k];
  if (kind == 0) wgrad_p_body<2, 1, 4, 1, false, false, S>(a, slice, 0, 0, ring);                 // layer 0: [256][32 -> 8]
  else if (kind == 1) wgrad_p_body<2, 2, 2, 2, false, false, S>(a, slice, t & 1, t >> 1, ring);    // hidden: four 128 x 128 tiles
  else if (kind == 2) wgrad_p_body<1, 2, 2, 2, true, false, S>(a, slice, t & 1, t >> 1, ring);     // variance head 0 (+ predict weight): four 64 x 128 tiles
  else wgrad_p_body<1, 2, 2, 2, false, true, S>(a, slice, 0, 0, ring);                             // variance head 1 (+ last weight): [64][128]
}

int launch_wgrad_p_multi(const WgradPMulti& m_in, hipStream_t st) {
#ifdef PINN_ABL_MULTI_S
  constexpr int S = PINN_ABL_MULTI_S;
#else
  constexpr int S = 3;
#endif
  WgradPMulti m = m_in;
  int nb = 0;
  size_t lds = 0;
  for (int k = 0; k < m.n; ++k) {
    const WgradPArgs& a = m.p[k];
    if (!a.P || !a.Q || !a.meta || !a.emax) return PINN_E_ARG;
    int tiles = 1, blocks = 0;
    const int to = a.OUT / 32, ti = a.IN / 32;
    if (m.kind[k] == 0) { if (to != 8 || ti != 1) return PINN_E_ARCH; blocks = 8 + 1; }
    else if (m.kind[k] == 1) { if (to != 8 || ti != 8) return PINN_E_ARCH; tiles = 4; blocks = 4 + 4; }
    else if (m.kind[k] == 2) { if (to != 4 || ti != 8 || !a.dvq) return PINN_E_ARCH; tiles = 4; blocks = 2 + 4; }
    else { if (to != 2 || ti != 4 || !a.dvr || !a.R || !a.s2) return PINN_E_ARCH; blocks = 2 + 4 + 2; }
    m.first[k] = nb;
    nb += tiles * a.n_slices;
    const size_t need = (size_t)S * (blocks * 2048 + 256);
    lds = need > lds ? need : lds;
  }
  m.first[m.n] = nb;
  auto kfn = wgrad_p_multi_kernel<S>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(S * (9 * 2048 + 256)));
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(kfn, dim3(nb), dim3(256), lds, st, m);
  return PINN_OK;
}

template <int TI, int TJ, int WI, int WJ, bool kVQ = false, bool kVR = false>
static int launch_p(const WgradPArgs& a, hipStream_t st) {
  constexpr int kBlocks = TI * WI + TJ * WJ + (kVR ? TI * WI : 0), kStage = kBlocks * 2048 + 256;
#ifdef PINN_ABL_WGP_S
  constexpr int S = PINN_ABL_WGP_S;
#else
  constexpr int S = 3;      // stages: two tiles ahead (measured at 1e6 rows, whole phase: S = 2 1.57 ms, 3 1.50, 4 1.53)
#endif
  const dim3 grid(a.n_slices, a.OUT / (TI * 32 * WI), a.IN / (TJ * 32 * WJ));
  const size_t lds = (size_t)S * kStage;
  auto k = wgrad_p_kernel<TI, TJ, WI, WJ, kVQ, kVR, S>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, grid, dim3(64 * WI * WJ), lds, st, a);
  return PINN_OK;
}

}  // namespace x6

// [OUT x IN] gradient with IN a multiple of 32 (every layer but the input one); ns = the operand split (4: two fp16 parts,
// kF16S; 3: three bf16 parts, x6; 1: bf16-mixed, one product -- wide nets under PINN_PREC_BF16)
int dispatch_wgrad_x6(const WgradArgs& a, int ns, void* stream) {
  using namespace x6;
  hipStream_t st = (hipStream_t)stream;
  const int to = a.OUT / 32, ti = a.IN / 32;
  if (a.Q == nullptr || a.IN % 32 || a.OUT % 32) return PINN_E_ARCH;
  if (ns != 1 && ns != 3 && ns != x6::kF16S) return PINN_E_ARG;
  if (ns == x6::kF16S && !a.amax) return PINN_E_ARG;
#ifdef PINN_DEBUG_HOOKS
  static const bool direct = getenv("PINN_WGRAD_DIRECT") != nullptr;     // measurement builds only: the register-staged kernel everywhere
#else
  constexpr bool direct = false;
#endif
  if (!direct && !a.dvq && !a.dvr && to % 8 == 0 && ti % 8 == 0) return launch_d<4, 4, 2, 2>(a, ns, st);   // plain layers: deep prefetch
  if (to == 8 && ti == 8) launch<4, 4, 2, 2>(a, ns, st);
  else if (to == 4 && ti == 8) launch<2, 4, 2, 2>(a, ns, st);
  else if (to == 2 && ti == 4) launch<1, 2, 2, 2>(a, ns, st);
  else if (to == 4 && ti == 4) launch<2, 2, 2, 2>(a, ns, st);
  else if (to == 1 && ti == 2) launch<1, 1, 1, 2>(a, ns, st);
  else if (to % 8 == 0 && ti % 8 == 0) launch<4, 4, 2, 2>(a, ns, st);        // wide nets: blocks of 256 x 256
  else if (to % 4 == 0 && ti % 8 == 0) launch<2, 4, 2, 2>(a, ns, st);        //            blocks of 128 x 256
  else return PINN_E_ARCH;
  return PINN_OK;
}

// the packed form of the same gradients (PINN_PREC_F32X6, fused nets)
int dispatch_wgrad_p(const WgradPArgs& a, void* stream) {
  using namespace x6;
  hipStream_t st = (hipStream_t)stream;
  const int to = a.OUT / 32, ti = a.IN / 32;
  if (!a.P || !a.Q || !a.meta || !a.emax || a.IN % 32 || a.OUT % 32) return PINN_E_ARG;
  // (the wide nets, H = 512 / 1024 / 2048: the same kernels over blocks of the gradient, blockIdx.y / z)
  if (a.dvq) {                     // variance head layer 0 (+ the predict head's weight)
    if (to == 2 && ti == 4) return launch_p<1, 2, 2, 2, true>(a, st);                   // H = 128
    if (to % 8 == 0 && ti % 8 == 0) return launch_p<4, 4, 2, 2, true>(a, st);           // wide nets: blocks of 256 x 256 (7.2 -> 7.0 ms at H = 1024)
    if (to % 4 == 0 && ti % 8 == 0) return launch_p<2, 4, 2, 2, true>(a, st);           // H = 256: one block of 128 x 256; wide: several
    return PINN_E_ARCH;
  }
  if (a.dvr) {                     // variance head layer 1 (+ the head's last weight, fp32 operands)
    if (!a.R || !a.s2) return PINN_E_ARG;
    if (to == 1 && ti == 2) return launch_p<1, 1, 1, 2, false, true>(a, st);           // H = 128
    if (to % 2 == 0 && ti % 4 == 0) return launch_p<1, 2, 2, 2, false, true>(a, st);    // blocks of 64 x 128
    return PINN_E_ARCH;
  }
  if (to % 8 == 0 && ti % 8 == 0) return launch_p<4, 4, 2, 2>(a, st);                    // blocks of 256 x 256
  if (to == 4 && ti == 4) return launch_p<2, 2, 2, 2>(a, st);                           // H = 128
  if (to % 8 == 0 && ti == 1) return launch_p<2, 1, 4, 1>(a, st);                       // layer 0 (Q = the packed input rows)
  if (to == 4 && ti == 1) return launch_p<1, 1, 4, 1>(a, st);
  return PINN_E_ARCH;
}

int dispatch_wgrad_p_multi(const WgradPMulti& m, void* stream) { return x6::launch_wgrad_p_multi(m, (hipStream_t)stream); }

}  // namespace pinn
