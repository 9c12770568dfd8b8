// pinn_bf16_core.h -- bf16-input / fp32-accumulate ("bf16/fp32-mixed") variant of the chain.
//
// Why a second variant: on gfx950 the exact-fp32 MFMA occupies the vector datapath (every tanh /
// Philox cycle is a lost matrix cycle), while v_mfma_f32_16x16x32_bf16 runs on the matrix cores,
// overlaps with VALU work of the co-resident workgroup and does 8x the MACs in half the cycles.
// With bf16 matrix inputs the chain becomes VALU/HBM-bound instead of MFMA-bound.
//
// Precision policy: weights and activations are rounded to bf16 only as MFMA INPUTS; accumulators,
// biases, tanh, dropout, the heads, the loss, master weights and Adam stay fp32.  The training stash
// (activations, d pre-activations) is stored in bf16.  Opt-in (pinn_net_t.precision = PINN_PREC_BF16);
// parity is checked against an oracle that applies the same roundings, and against the fp32
// reference at rtol 2e-2.
//
// Layout: the accumulator / activation layout is the same as the fp32 chain (lane (kq, n) holds
// features 16 ib + 4 kq + r of row n).  The B operand of v_mfma_f32_16x16x32_bf16 wants, per lane,
// k = 8 kq + j (j = 0..7): the 8 features a lane holds in the 32-feature group fp are assigned
//     k = 8 kq + 4 b + r   <->   feature 32 fp + 16 b + 4 kq + r,
// and the weights are PRE-PACKED (pack_bf16_kernel, once per call) with the matching column
// permutation, so an A fragment (8 bf16 of one output row) is one conflict-free ds_read_b128.
// The dgrad uses a second packed copy, W^T, so forward and backward share one code path.
#pragma once
#include "pinn_mlp_core.h"

namespace pinn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define PINN_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__host__ __device__ inline int round_up64(int k) { return (k + 63) / 64 * 64; }

// packed bf16 buffer: element offsets of each matrix copy.  Every matrix is [rows][Kp] with
// Kp = K rounded up to 64 (zero padded), K permuted inside every 32-group.
struct PackLayout {
  int H, nh;
  __host__ __device__ long long hidden_sz() const { return (long long)H * H; }
  __host__ __device__ long long w(int l) const { return (long long)(l - 1) * 2 * hidden_sz(); }            // W_l   [H][H], l >= 1
  __host__ __device__ long long wt(int l) const { return w(l) + hidden_sz(); }                             // W_l^T [H][H]
  __host__ __device__ long long wv0() const { return (long long)(nh - 1) * 2 * hidden_sz(); }              // [H/2][H]
  __host__ __device__ long long wv0t() const { return wv0() + (long long)(H / 2) * H; }                    // [H][round64(H/2)]
  __host__ __device__ long long wv1() const { return wv0t() + (long long)H * round_up64(H / 2); }          // [H/4][round64(H/2)]
  __host__ __device__ long long wv1t() const { return wv1() + (long long)(H / 4) * round_up64(H / 2); }    // [H/2][round64(H/4)]
  __host__ __device__ long long total() const { return wv1t() + (long long)(H / 2) * round_up64(H / 4); }
};

// position q (0..31) inside a packed 32-group  <->  original column c of the group
__host__ __device__ inline int pack_col(int q) { return 16 * ((q & 7) >> 2) + 4 * (q >> 3) + (q & 3); }

// one matrix copy: dst [rows][Kp] bf16; src element (row, k) = transposed ? W[k][row] : W[row][k], W is [out][in] fp32
struct PackJob {
  long long dst;       // bf16 element offset in the packed buffer
  long long src;       // float offset of W in the flat parameter buffer
  int rows, K, Kp, src_ld, transposed;
};

// chunk cycles on the packed buffer (offsets / ld in FLOAT units = bf16 units / 2; all slabs are "forward" kind:
// [rows][128 B] = 64 packed bf16 columns = two 32-groups)
__device__ __forceinline__ int add_slabs(ChunkDesc* tab, int k, long long off_bf16, int rows, int Kp) {
  for (int s = 0; s < Kp / 64; ++s)
    tab[k++] = ChunkDesc{(unsigned)(off_bf16 / 2 + s * 32), (unsigned short)(Kp / 2), 0, (unsigned char)(rows / 32)};
  return k;
}
__device__ __forceinline__ int build_forward_chunks_bf16(ChunkDesc* tab, const PackLayout& L, int at) {
  int k = at;
  const int H = L.H;
  for (int l = 1; l < L.nh; ++l) k = add_slabs(tab, k, L.w(l), H, H);
  k = add_slabs(tab, k, L.wv0(), H / 2, H);
  k = add_slabs(tab, k, L.wv1(), H / 4, round_up64(H / 2));
  return k;
}
__device__ __forceinline__ int build_backward_chunks_bf16(ChunkDesc* tab, const PackLayout& L, int at) {
  int k = at;
  const int H = L.H;
  k = add_slabs(tab, k, L.wv1t(), H / 2, round_up64(H / 4));
  k = add_slabs(tab, k, L.wv0t(), H, round_up64(H / 2));
  for (int l = L.nh - 1; l >= 1; --l) k = add_slabs(tab, k, L.wt(l), H, H);
  return k;
}
__host__ __device__ inline int n_forward_slabs_bf16(int H, int nh) { return (nh - 1) * (H / 64) + H / 64 + round_up64(H / 2) / 64; }
__host__ __device__ inline int n_backward_slabs_bf16(int H, int nh) {
  return round_up64(H / 4) / 64 + round_up64(H / 2) / 64 + (nh - 1) * (H / 64);
}

// out^T[f2][n] += sum_k A[f2][k] B[k][n] over NG 32-groups (NGP = NG rounded up to even: one slab = 2 groups)
template <int NG, int NTOUT>
__device__ __forceinline__ void layer_bf16(f32x4 (&acc)[NTOUT], const bf16x8 (&b)[NG], Pipe& pipe, int lane) {
  const int kq = lane >> 4, i = lane & 15;
  const int sw = (i >> 1) & 7;
  const int base = i * 128;
  constexpr int NS = (NG + 1) / 2;
#pragma unroll
  for (int kb = 0; kb < NS; ++kb) {
    const char* buf = pipe.cur() + base;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (2 * kb + g < NG) {     // compile-time after unrolling; a padded (zero) group is skipped
        const int off = ((4 * g + kq) ^ sw) << 4;
#pragma unroll
        for (int mt = 0; mt < NTOUT; ++mt) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(buf + off + mt * 2048);
          acc[mt] = PINN_MFMA_BF16(a, b[2 * kb + g], acc[mt]);
        }
      }
    }
    pipe.advance();
  }
}

// the 8 features a lane holds in group fp (blocks 2fp, 2fp+1) -> one bf16 B fragment, k = 4 b + r
__device__ __forceinline__ bf16x8 make_frag(const f32x4& v0, const f32x4& v1) {
  bf16x8 f;
  f[0] = (__bf16)v0[0]; f[1] = (__bf16)v0[1]; f[2] = (__bf16)v0[2]; f[3] = (__bf16)v0[3];
  f[4] = (__bf16)v1[0]; f[5] = (__bf16)v1[1]; f[6] = (__bf16)v1[2]; f[7] = (__bf16)v1[3];
  return f;
}

// bf16 tiled stash: element (feature f, row n) of tile t16 at ((t16*F + f)*16 + n) [bf16 units]
__device__ __forceinline__ __bf16* tiled_ptr_bf16(__bf16* base, long long t16, int F, int lane) {
  return base + (t16 * F + 4 * (lane >> 4)) * 16 + (lane & 15);
}
__device__ __forceinline__ void store_frag_bf16(__bf16* __restrict__ p, int fp, const bf16x8& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) p[(fp * 32 + 16 * (j >> 2) + (j & 3)) * 16] = f[j];
}
__device__ __forceinline__ void load_pair_bf16(const __bf16* __restrict__ p, int fp, f32x4& v0, f32x4& v1) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v0[r] = (float)p[(fp * 32 + r) * 16];
    v1[r] = (float)p[(fp * 32 + 16 + r) * 16];
  }
}

struct StashPtrsBf16 {
  __bf16* h;               // [nh][T16][H][16]
  __bf16* v1;              // [T16][H/2][16]
  __bf16* v2;              // [T16][H/4][16]
  unsigned char* keep;     // [T16][nh*H/32 + H/64][64]
  long long t16_total;
  long long t16;
};

// One forward pass, bf16 matrix inputs.  Returns (u, z) and, for the training kernel, the fp32 v2 blocks.
template <int H, bool TRAIN, bool kBits>
__device__ __forceinline__ void forward_pass_bf16(const float* __restrict__ P, const float* smallp, const ParamLayout& L,
                                                  Pipe& pipe, const DropDev& d, const RowCtx& c, const f32x4& xa, const f32x4& xb,
                                                  const StashPtrsBf16& st, float& u, float& z, f32x4 (&v2)[H / 64]) {
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32;
  const int lane = c.lane, kq = c.kq;
  const SmallLayout S{L.H, L.nh};
  const int n_groups = L.nh * NP + NP / 2;
  unsigned char* keep = TRAIN ? st.keep + (st.t16 * n_groups) * 64 + lane : nullptr;
  f32x4 h[NT];
  bf16x8 hb[NP];
  layer_input<NT>(h, P + L.w0(), smallp + S.b(0), xa, xb, lane);      // 8 -> H stays exact fp32 (K = 8)
  float up = 0.0f;
#pragma unroll 1
  for (int l = 0; l < L.nh; ++l) {
    const LayerDrop ldr = layer_drop(d, c.mode, l);
    __bf16* sp = TRAIN ? tiled_ptr_bf16(st.h + (long long)l * st.t16_total * H * 16, st.t16, H, lane) : nullptr;
    const bool last = l + 1 == L.nh;
#pragma unroll
    for (int fp = 0; fp < NP; ++fp) {
      const unsigned k8 = activate_pair<kBits>(h[2 * fp], h[2 * fp + 1], d, c, ldr, l, fp);
      hb[fp] = make_frag(h[2 * fp], h[2 * fp + 1]);
      if (last) {     // predict head on the fp32 activations of the last hidden layer
        up = block_dot(h[2 * fp], smallp + S.wp() + (2 * fp) * 16, kq, up);
        up = block_dot(h[2 * fp + 1], smallp + S.wp() + (2 * fp + 1) * 16, kq, up);
      }
      if (TRAIN) {
        keep[(l * NP + fp) * 64] = (unsigned char)k8;
        store_frag_bf16(sp, fp, hb[fp]);
      }
    }
    if (!last) {
      bias_blocks<NT>(h, smallp + S.b(l + 1), kq);
      layer_bf16<NP, NT>(h, hb, pipe, lane);
    }
  }
  u = sum_kq(up) + smallp[S.bp()];
  f32x4 v1[NT2];
  bias_blocks<NT2>(v1, smallp + S.bv0(), kq);
  layer_bf16<NP, NT2>(v1, hb, pipe, lane);
  bf16x8 vb[NP / 2];
  {
    const LayerDrop ldr = layer_drop(d, c.mode, L.nh);
    __bf16* sp = TRAIN ? tiled_ptr_bf16(st.v1, st.t16, H / 2, lane) : nullptr;
#pragma unroll
    for (int fp = 0; fp < NP / 2; ++fp) {
      const unsigned k8 = activate_pair<kBits>(v1[2 * fp], v1[2 * fp + 1], d, c, ldr, L.nh, fp);
      vb[fp] = make_frag(v1[2 * fp], v1[2 * fp + 1]);
      if (TRAIN) {
        keep[(L.nh * NP + fp) * 64] = (unsigned char)k8;
        store_frag_bf16(sp, fp, vb[fp]);
      }
    }
  }
  bias_blocks<NT4>(v2, smallp + S.bv1(), kq);
  layer_bf16<NP / 2, NT4>(v2, vb, pipe, lane);
  float zp = 0.0f;
#pragma unroll
  for (int t = 0; t < NT4; ++t) {
    activate_tanh(v2[t]);
    zp = block_dot(v2[t], smallp + S.wv2() + t * 16, kq, zp);
  }
  if (TRAIN) {
    __bf16* sp = tiled_ptr_bf16(st.v2, st.t16, H / 4, lane);
#pragma unroll
    for (int t = 0; t < NT4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[(t * 16 + r) * 16] = (__bf16)v2[t][r];
  }
  z = sum_kq(zp) + smallp[S.bv2()];
}

}  // namespace pinn
