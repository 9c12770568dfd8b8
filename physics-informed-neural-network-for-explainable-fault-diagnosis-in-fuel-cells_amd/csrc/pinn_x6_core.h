// pinn_x6_core.h -- fp32-accurate chain on the bf16 matrix cores ("x6": 3-way bf16 split, 6 products).
//
// Why: on gfx950 v_mfma_f32_*_f32 runs at the VALU rate on the vector datapath and blocks every
// other VALU instruction of the SIMD (tools/mfma_valu_share.hip), so the exact-fp32 chain can never
// hide its tanh / Philox work.  The bf16 matrix cores are 16x faster and independent of the VALU.
// An fp32 number is exactly hi + mid + lo with three bf16 parts (8 + 8 + 8 mantissa bits), and
//     a * w  ~=  a_hi w_hi + (a_hi w_mid + a_mid w_hi) + (a_hi w_lo + a_lo w_hi + a_mid w_mid)
// drops only terms of relative size 2^-24: six v_mfma_f32_16x16x32_bf16 with fp32 accumulation give
// the same accuracy as an fp32 matmul (measured: 7.6e-7 max error vs float64 against 1.2e-6 for
// torch's fp32 matmul on the same 256-long dot products) at 6/16 of the f32-MFMA time -- and the
// VALU work of the co-resident wave overlaps with them.
//
// Structure: one 512-thread workgroup per CU = 8 waves x 16 rows (two waves per SIMD, <= 256
// registers each).  Weights are pre-split and pre-permuted into three bf16 copies (hi, mid, lo;
// pack kernel, once per call); one LDS slab = one 32-feature K-group of a layer for all output rows
// = 3 x [rows][64 B], two slabs in flight (<= 96 KB), one barrier per slab.  Activation is LAZY: while
// slab kb multiplies, the raw accumulators of K-group kb+1 are turned into their three bf16
// fragments (bias is already in the accumulator; tanh, dropout, split).  Waves 4-7 run "MFMAs, then
// prepare" and waves 0-3 "prepare, then MFMAs", so on every SIMD one wave feeds the matrix core while
// its partner uses the VALU, although both follow the same barrier-synchronised slab sequence.
#pragma once
#include "pinn_bf16_core.h"

namespace pinn {
namespace x6 {

constexpr int kThreadsX = 512;       // 8 waves
constexpr int kTileRowsX = 128;      // rows per workgroup tile
constexpr int kSlabBytes = 49152;    // 3 copies x 256 rows x 64 B
constexpr int kMaxSlabs = 200;

// packed buffer: three copies (hi, mid, lo) of the bf16 layout of pinn_bf16_core.h (PackLayout), back to back
struct Slab {
  unsigned off;            // bf16-element offset of (row 0, this 32-group) inside one copy
  unsigned char kp_log;    // log2 of the row stride of the packed matrix (bf16 elements)
  unsigned char nrb_log;   // log2 of (output rows of the layer / 16): 16-row blocks per copy
  unsigned short pad;
};

// swizzle of the four 16-B kq chunks of a 64-B row so that ds_read_b128 is conflict-free for the hardware's
// lane groups {0-3,12-15,20-27} ... : chunk' = kq ^ g[(row >> 2) & 3], g = {0, 2, 3, 1}
__device__ __forceinline__ int swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // 2-bit table {0, 2, 3, 1}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Weight stream, global (L2) -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass,
// and nothing for the compute phase's waits to trip over).  One wave-instruction moves one 1-KB piece = 16 rows x
// 64 B of one copy; the LDS image of a piece is lane-linear, so the kq-chunk swizzle is applied to the SOURCE
// address: LDS chunk c of row r holds kq = c ^ swz(r) (an involution, the reader applies the same one).
struct Pipe6 {
  const char* packed;        // copy 0 (bytes); copies 1, 2 follow at +copy_bytes
  unsigned copy_bytes;
  const Slab* tab;           // in LDS
  char* lds;                 // 2 x kSlabBytes
  int n, ci, wave;
  unsigned lane_row, lane_kq8;   // (lane >> 2), 16 B * ((lane & 3) ^ swz(lane >> 2))

  __device__ __forceinline__ void init(int tid) {
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    lane_row = lane >> 2;
    lane_kq8 = (unsigned)(((lane & 3) ^ swz(lane >> 2)) << 4);
  }
  __device__ __forceinline__ void issue(int idx_slab, int buf) {
    const unsigned off = __builtin_amdgcn_readfirstlane(tab[idx_slab].off);
    const int kp_log = __builtin_amdgcn_readfirstlane(tab[idx_slab].kp_log);
    const int nrb_log = __builtin_amdgcn_readfirstlane(tab[idx_slab].nrb_log);
    const unsigned voff = ((lane_row << kp_log) << 1) + lane_kq8;          // bytes, per lane
    const int n_pieces = 3 << nrb_log;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int piece = wave + 8 * j;
      if (piece < n_pieces) {                                              // wave-uniform
        const int copy = piece >> nrb_log, rb = piece & ((1 << nrb_log) - 1);
        const unsigned long long goff = (unsigned long long)copy * copy_bytes + 2ull * (off + ((unsigned)(rb * 16) << kp_log));
        char* dst = lds + buf * kSlabBytes + copy * (kSlabBytes / 3) + rb * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(packed + goff + voff), (lptr_t)dst, 16, 0, 0);
      }
    }
  }
  __device__ __forceinline__ void prime() {
    ci = 0;
    issue(0, 0);
    __syncthreads();           // (drains the DMA: vmcnt(0) + barrier)
    issue(n > 1 ? 1 : 0, 1);
  }
  __device__ __forceinline__ const char* cur() const { return lds + (ci & 1) * kSlabBytes; }
  // end of a slab step: the DMA of slab ci + 1 has had the whole step to land; after the barrier every wave is done
  // reading slab ci, whose buffer the DMA of slab ci + 2 may overwrite
  __device__ __forceinline__ void advance() {
    __syncthreads();
    ++ci;
    issue((ci + 1) % n, (ci + 1) & 1);
  }
};

// slab cycle of one forward pass: every 32-group of every matrix, in consumption order
__device__ __forceinline__ int add_groups(Slab* tab, int k, long long off_bf16, int rows, int K, int Kp) {
  const unsigned char kp_log = (unsigned char)(31 - __builtin_clz((unsigned)Kp)), nrb_log = (unsigned char)(31 - __builtin_clz((unsigned)(rows / 16)));
  for (int g = 0; g < K / 32; ++g) tab[k++] = Slab{(unsigned)(off_bf16 + g * 32), kp_log, nrb_log, 0};
  return k;
}
__device__ __forceinline__ int build_forward_slabs(Slab* tab, const PackLayout& L, int at) {
  int k = at;
  const int H = L.H;
  for (int l = 1; l < L.nh; ++l) k = add_groups(tab, k, L.w(l), H, H, H);
  k = add_groups(tab, k, L.wv0(), H / 2, H, H);
  k = add_groups(tab, k, L.wv1(), H / 4, H / 2, round_up64(H / 2));
  return k;
}
__device__ __forceinline__ int build_backward_slabs(Slab* tab, const PackLayout& L, int at) {
  int k = at;
  const int H = L.H;
  k = add_groups(tab, k, L.wv1t(), H / 2, H / 4, round_up64(H / 4));
  k = add_groups(tab, k, L.wv0t(), H, H / 2, round_up64(H / 2));
  for (int l = L.nh - 1; l >= 1; --l) k = add_groups(tab, k, L.wt(l), H, H, H);
  return k;
}
__host__ __device__ inline int n_forward_slabs(int H, int nh) { return (nh - 1) * (H / 32) + H / 32 + H / 64; }
__host__ __device__ inline int n_backward_slabs(int H, int nh) { return H / 128 + H / 64 + (nh - 1) * (H / 32); }

// three bf16 fragments of the 8 fp32 values a lane holds in one 32-group: v = hi + mid + lo (exact)
struct Frag3 {
  bf16x8 hi, mid, lo;
};
__device__ __forceinline__ Frag3 split3(const f32x4& v0, const f32x4& v1) {
  Frag3 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = j < 4 ? v0[j] : v1[j - 4];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    f.hi[j] = h; f.mid[j] = m; f.lo[j] = (__bf16)r2;
  }
  return f;
}

// the 6 x NTOUT MFMAs of one slab: acc[mt] += A_mt (hi, mid, lo) x B (hi, mid, lo), terms of order >= 2^-24 dropped.
// The three A fragments of tile mt + 1 are requested before the MFMAs of tile mt are issued (hipcc reuses one
// register set and waits lgkmcnt(0) per tile otherwise: the whole LDS latency exposed 16 times per slab).
struct AFrag3 {
  bf16x8 h, m, l;
};
__device__ __forceinline__ AFrag3 load_a3(const char* base, int mt) {
  constexpr int kCopy = kSlabBytes / 3;
  AFrag3 a;
  a.h = *reinterpret_cast<const bf16x8*>(base + mt * 1024);
  a.m = *reinterpret_cast<const bf16x8*>(base + kCopy + mt * 1024);
  a.l = *reinterpret_cast<const bf16x8*>(base + 2 * kCopy + mt * 1024);
  return a;
}
__device__ __forceinline__ void mfma6(f32x4& acc, const AFrag3& a, const Frag3& b) {
  f32x4 c = acc;
  c = PINN_MFMA_BF16(a.l, b.hi, c);
  c = PINN_MFMA_BF16(a.h, b.lo, c);
  c = PINN_MFMA_BF16(a.m, b.mid, c);
  c = PINN_MFMA_BF16(a.m, b.hi, c);
  c = PINN_MFMA_BF16(a.h, b.mid, c);
  c = PINN_MFMA_BF16(a.h, b.hi, c);
  acc = c;
}
template <int NTOUT>
__device__ __forceinline__ void slab_mfma(f32x4 (&acc)[NTOUT], const Frag3& b, const char* slab, int lane) {
  const int kq = lane >> 4, i = lane & 15;
  const char* base = slab + i * 64 + ((kq ^ swz(i)) << 4);      // rows mt*16 + i: (row >> 2) & 3 == (i >> 2) & 3
  AFrag3 a0 = load_a3(base, 0), a1 = a0;
#pragma unroll
  for (int mt = 0; mt < NTOUT; mt += 2) {
    a1 = load_a3(base, mt + 1);
    mfma6(acc[mt], a0, b);
    if (mt + 2 < NTOUT) a0 = load_a3(base, mt + 2);
    mfma6(acc[mt + 1], a1, b);
  }
}

// input layer from LDS: acc = b0 + W0 x^T in exact fp32 (K = 8); w0t is [8][kW0Stride] (k-major, padded: conflict-free)
constexpr int kW0Stride = 272;
template <int NTOUT>
__device__ __forceinline__ void layer_input_lds(f32x4 (&acc)[NTOUT], const float* w0t, const float* b0, const f32x4& xa, const f32x4& xb,
                                                int lane) {
  const int kq = lane >> 4, i = lane & 15;
  const float x0 = kq == 0 ? xa[0] : (kq == 1 ? xa[1] : (kq == 2 ? xa[2] : xa[3]));
  const float x1 = kq == 0 ? xb[0] : (kq == 1 ? xb[1] : (kq == 2 ? xb[2] : xb[3]));
#pragma unroll
  for (int mt = 0; mt < NTOUT; ++mt) {
    const float w0 = w0t[kq * kW0Stride + mt * 16 + i];
    const float w1 = w0t[(4 + kq) * kW0Stride + mt * 16 + i];
    f32x4 c = *reinterpret_cast<const f32x4*>(b0 + mt * 16 + 4 * kq);
    c = PINN_MFMA16(w0, x0, c);
    c = PINN_MFMA16(w1, x1, c);
    acc[mt] = c;
  }
}

// One layer: NG K-groups.  prep(g) returns the Frag3 of group g (lazy activation of the previous layer's raw
// output, or the backward chain's dpre).  Early waves (0-3) run "prepare g, multiply g", late waves (4-7) "multiply g,
// prepare g + 1": on every SIMD one wave feeds the matrix core while its partner is on the VALU, and neither holds
// more than one fragment set.
template <int NG, int NTOUT, typename F>
__device__ __forceinline__ void layer_x6(f32x4 (&acc)[NTOUT], Pipe6& pipe, int lane, bool late, F&& prep) {
  Frag3 cur;
  if (late) cur = prep(0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (late) {
      slab_mfma<NTOUT>(acc, cur, pipe.cur(), lane);
      if (g + 1 < NG) cur = prep(g + 1);
    } else {
      cur = prep(g);
      slab_mfma<NTOUT>(acc, cur, pipe.cur(), lane);
    }
    pipe.advance();
  }
}

// The layer loops are fully unrolled straight-line code, and hipcc schedules every group's Philox counter setup
// (and first-round products) to the top of it -- tens of long-lived registers, i.e. spills.  Passing the lane's kq
// through an empty volatile asm ties each group's generator to its own slab step (volatile asm is not moved across
// the step's barrier).
__device__ __forceinline__ RowCtx pinned(const RowCtx& c) {
  RowCtx r = c;
  asm volatile("" : "+v"(r.kq));
  return r;
}

// One forward pass for this wave's 16 rows (x6 matrix math).  Returns (u, z); v2 = tanh'ed last hidden block(s).
template <int H, bool kBits>
__device__ __forceinline__ void forward_pass_x6(const float* w0t, const float* smallp, const ParamLayout& L, Pipe6& pipe,
                                                const DropDev& d, const RowCtx& c, const f32x4& xa, const f32x4& xb, bool late,
                                                float& u, float& z) {
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32;
  const int lane = c.lane, kq = c.kq;
  const SmallLayout S{L.H, L.nh};
  f32x4 h[NT];
  layer_input_lds<NT>(h, w0t, smallp + S.b(0), xa, xb, lane);      // 8 -> H in exact fp32 (K = 8)
#pragma unroll 1
  for (int l = 1; l < L.nh; ++l) {
    f32x4 acc[NT];
    bias_blocks<NT>(acc, smallp + S.b(l), kq);
    const LayerDrop ldr = layer_drop(d, c.mode, l - 1);
    layer_x6<NP, NT>(acc, pipe, lane, late, [&](int g) {
      const RowCtx cc = pinned(c);
      activate_pair<kBits>(h[2 * g], h[2 * g + 1], d, cc, ldr, l - 1, g);
      return split3(h[2 * g], h[2 * g + 1]);
    });
#pragma unroll
    for (int t = 0; t < NT; ++t) h[t] = acc[t];
  }
  // last hidden layer: lazily activated while the variance head's first layer multiplies; predict head on the fly
  f32x4 v1[NT2];
  bias_blocks<NT2>(v1, smallp + S.bv0(), kq);
  float up = 0.0f;
  {
    const int ll = L.nh - 1;
    const LayerDrop ldr = layer_drop(d, c.mode, ll);
    layer_x6<NP, NT2>(v1, pipe, lane, late, [&](int g) {
      const RowCtx cc = pinned(c);
      activate_pair<kBits>(h[2 * g], h[2 * g + 1], d, cc, ldr, ll, g);
      up = block_dot(h[2 * g], smallp + S.wp() + (2 * g) * 16, kq, up);
      up = block_dot(h[2 * g + 1], smallp + S.wp() + (2 * g + 1) * 16, kq, up);
      return split3(h[2 * g], h[2 * g + 1]);
    });
  }
  u = sum_kq(up) + smallp[S.bp()];
  f32x4 v2[NT4];
  bias_blocks<NT4>(v2, smallp + S.bv1(), kq);
  {
    const LayerDrop ldr = layer_drop(d, c.mode, L.nh);
    layer_x6<NP / 2, NT4>(v2, pipe, lane, late, [&](int g) {
      const RowCtx cc = pinned(c);
      activate_pair<kBits>(v1[2 * g], v1[2 * g + 1], d, cc, ldr, L.nh, g);
      return split3(v1[2 * g], v1[2 * g + 1]);
    });
  }
  float zp = 0.0f;
#pragma unroll
  for (int t = 0; t < NT4; ++t) {
    activate_tanh(v2[t]);
    zp = block_dot(v2[t], smallp + S.wv2() + t * 16, kq, zp);
  }
  z = sum_kq(zp) + smallp[S.bv2()];
}

}  // namespace x6
}  // namespace pinn
